#!/usr/bin/env python3
"""Fuzz: nsnp_hap_forward (fp32, bf16x3, f16x3) on features of fuzzed read planes against oracle/liboracle.so with seeded weights of
several scales (input layer 0.002 .. 0.3, all weights x 0.5 .. x 4, heads x 1 .. x 300): finite, and within 1e-4 of the fp32 oracle
where the oracle run at twice the summation blocking agrees with itself to 1e-5 (else the case is reported as ill-conditioned).
Test infrastructure (loads oracle/)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib, host
from nanosnp_amd.fixtures import seeded_hap_weights
from oracle import oracle

def main():
    n = 48
    bad = 0
    ctx = _lib.Context(0)
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        rng = np.random.default_rng(2200 + seed)
        feats = []
        for L in (33, 11):
            seq, bq, mq, hap, ref_row = host.synth_hap_planes(5000 + seed + L, n, coverage=int(rng.choice([5, 30, 60])), depth=90, length=L)
            if seed % 3 == 1:
                seq[: n // 4] = -2; bq[: n // 4] = -2; mq[: n // 4] = -2; hap[: n // 4] = -2       # empty sites
            if seed % 3 == 2:
                bq[n // 2:] = 93; mq[n // 2:] = 60                                                  # saturated qualities
            feats.append(oracle.hap_features_batch(seq, bq, mq, hap, ref_row, nthreads=8))
        xp, xh = feats
        for ih, allw, head in ((0.002, 1.0, 8.0), (0.03, 1.0, 120.0), (0.3, 1.0, 30.0), (0.01, 0.5, 1.0), (0.01, 2.0, 300.0), (0.002, 4.0, 8.0)):
            ws = seeded_hap_weights(40 + seed, H=256, ih_scale=ih, head_scale=head)
            ws = [w * np.float32(allw) for w in ws]
            ogt, ozy = oracle.hap_forward(ws, xp, xh, nthreads=8)
            ctx.hap_load_weights(ws)
            line = [f"seed {2200 + seed} ih {ih:g} all x{allw:g} heads x{head:g}:"]
            for prec, name in ((0, "fp32"), (2, "bf16x3"), (1, "f16x3")):
                ctx.set_option("hap_precision", prec)
                gt, zy = ctx.hap_forward(torch.from_numpy(xp).cuda(), torch.from_numpy(xh).cuda())
                gt, zy = gt.cpu().numpy(), zy.cpu().numpy()
                d = max(np.abs(gt - ogt).max(), np.abs(zy - ozy).max())
                fin = bool(np.isfinite(gt).all() and np.isfinite(zy).all())
                good = fin and d <= 1e-4
                line.append(f"{name} {d:.1e}{'' if good else ' <-- LOOK' if prec != 1 else ' (f16x3 opt-in)'}")
                bad += (prec != 1) and not good
            ctx.set_option("hap_precision", 0)
            print(" | ".join(line), flush=True)
    print("bad", bad)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
