#!/usr/bin/env python3
"""Fuzz: nsnp_hap_arrange_reads and nsnp_cat_groups against oracle/liboracle.so on read matrices outside the generator's range - any
int32 HP value (negative, huge, INT_MAX: the kernel's own sentinel for a dropped row), many ties, zero centre bases, n_reads below /
at / above R, R from 1 to 200, D_out below and above the kept depth, L in {33, 11, 1, 7}; bit for bit.  Test infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from oracle import oracle

def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ctx = _lib.Context(0)
    bad = runs = 0
    for s in range(seeds):
        rng = np.random.default_rng(4100 + s)
        for L in (33, 11, 1, 7):
            for R in (1, 2, 5, 30, 63, 64, 65, 90, 200):
                for mode in range(5):
                    N = int(rng.choice([1, 5, 130]))
                    d_out = int(rng.choice([1, 2, max(1, R // 2), R, R + 7, 90, 180]))
                    seq = rng.integers(-1, 5, (N, R, L)).astype(np.int32)
                    if mode == 0: hap = rng.integers(0, 4, (N, R, L))
                    elif mode == 1: hap = rng.integers(-3, 9, (N, R, L))
                    elif mode == 2: hap = rng.choice([0, 1, 2, 3, -2 ** 31, 2 ** 31 - 2, 7], (N, R, L))
                    elif mode == 4: hap = rng.choice([1, 2, 2 ** 31 - 1], (N, R, L))
                    else: hap = np.broadcast_to(rng.integers(1, 3, (N, R, 1)), (N, R, L)).copy()
                    hap = hap.astype(np.int32)
                    bq = rng.integers(0, 94, (N, R, L)).astype(np.int32); mq = rng.integers(0, 61, (N, R, L)).astype(np.int32)
                    seq[rng.random((N, R)) < 0.3, L // 2] = 0                       # rows that do not cover the centre
                    n_reads = rng.integers(0, R + 3, N).astype(np.int32) if mode % 2 else None
                    t = [torch.from_numpy(a).cuda() for a in (seq, bq, mq, hap)]
                    got = ctx.hap_arrange_reads(*t, d_out, n_reads=None if n_reads is None else torch.from_numpy(n_reads).cuda())
                    got = [g.cpu().numpy() for g in got]
                    ok = True
                    for i in range(N):
                        rows = None if n_reads is None else int(min(n_reads[i], R))
                        w = oracle.hap_arrange(seq[i], bq[i], mq[i], hap[i], d_out, rows=rows)
                        ok = ok and all(np.array_equal(got[k][i], w[k]) for k in range(4)) and int(got[4][i]) == w[4]
                    runs += 1
                    if not ok:
                        bad += 1; print(f"arrange: seed {4100 + s} L {L} R {R} mode {mode} N {N} d_out {d_out}: differs", flush=True)
        # cat_groups: (read, baseq, mapq) per tag, depth 0 .. 60
        for D1, D2 in ((20, 20), (21, 33), (60, 20), (90, 180)):       # (the reference's bins hold at least max_depth = 20 rows per tag: dataset.py:862)
            for mode in range(3):
                N, L = int(rng.choice([1, 9, 200])), 11
                def tag(D):
                    if D == 0:
                        return [np.zeros((N, 0, L), np.int32)] * 3
                    r = (rng.integers(-2, 5, (N, D, L)) if mode == 0 else rng.integers(-6, 9, (N, D, L)) if mode == 1 else np.full((N, D, L), int(rng.choice([-2, -1, 0, 2])))).astype(np.int32)
                    return [r, rng.integers(-2, 94, (N, D, L)).astype(np.int32), rng.integers(-2, 61, (N, D, L)).astype(np.int32)]
                t1, t2 = tag(D1), tag(D2)
                ew = eg = None
                try:
                    want = oracle.cat_groups(t1, t2)
                except Exception as e:
                    want = None; ew = repr(e)[:100]
                try:
                    got = ctx.cat_groups([torch.from_numpy(a).cuda() for a in t1], [torch.from_numpy(a).cuda() for a in t2]).cpu().numpy()
                except Exception as e:
                    got = None; eg = repr(e)[:100]
                if (ew or eg) and runs < 2000 and s == 0 and mode == 0: print("   cat_groups exceptions: oracle", ew, "| hip", eg)
                runs += 1
                if (want is None) != (got is None) or (want is not None and not np.array_equal(got, want, equal_nan=True)):
                    bad += 1; print(f"cat_groups: seed {4100 + s} D {D1},{D2} mode {mode} N {N}: differs", None if want is None or got is None else float(np.nanmax(np.abs(got - want))), flush=True)
        print(f"{s + 1} seeds: {runs} runs, {bad} differ", flush=True)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
