#!/usr/bin/env python3
"""Fuzz: nsnp_hap_features / nsnp_hap_features_i8 against oracle/liboracle.so (itself identical to the reference's
dataset_dev.get_frequency_feature on the same kinds of planes: docs/rounds/r05.md) on planes outside the generator's range - codes the
reference ignores, negative and huge qualities (the packed 8- / 16-bit running sums must fall back), all padding / all deletions, a
read set present through ONE element, D from 1 to 200, L in {33, 11, 1, 5, 40}; bit for bit.  Test infrastructure (loads oracle/)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from oracle import oracle

def planes(rng, mode, N, D, L):
    if mode == 0:
        seq = rng.integers(-2, 5, (N, D, L)); hap = rng.integers(-2, 4, (N, D, L)); bq = rng.integers(-2, 61, (N, D, L)); mq = rng.integers(-2, 61, (N, D, L))
    elif mode == 1:
        seq = rng.integers(-5, 9, (N, D, L)); hap = rng.integers(-4, 7, (N, D, L)); bq = rng.integers(-50, 127, (N, D, L)); mq = rng.integers(-50, 127, (N, D, L))
    elif mode == 2:
        seq = rng.integers(1, 5, (N, D, L)); hap = rng.integers(0, 4, (N, D, L)); bq = rng.integers(0, 2 ** 20, (N, D, L)); mq = rng.integers(-2 ** 20, 2 ** 20, (N, D, L))
    elif mode == 3:
        v = rng.choice([-2, -1, 0, 3], (N, 1, 1)); seq = np.broadcast_to(v, (N, D, L)).copy(); hap = rng.integers(0, 4, (N, D, L)); bq = rng.integers(0, 61, (N, D, L)); mq = rng.integers(0, 61, (N, D, L))
    elif mode == 4:
        seq = rng.integers(-1, 5, (N, D, L)); h = rng.integers(0, 4, (N, D, 1)); hap = np.where(seq != 0, np.broadcast_to(h, (N, D, L)), 0); bq = rng.integers(0, 61, (N, D, L)); mq = rng.integers(0, 61, (N, D, L))
        pad = rng.integers(0, D + 1, N)
        for i in range(N):
            seq[i, pad[i]:] = -2; hap[i, pad[i]:] = -2; bq[i, pad[i]:] = -2; mq[i, pad[i]:] = -2
    else:
        seq = rng.integers(1, 5, (N, D, L)); hap = np.zeros((N, D, L), int)
        hap[np.arange(N), rng.integers(0, D, N), rng.integers(0, L, N)] = rng.integers(1, 4, N); bq = rng.integers(0, 61, (N, D, L)); mq = rng.integers(0, 61, (N, D, L))
    return [a.astype(np.int32) for a in (seq, bq, mq, hap)]

def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ctx = _lib.Context(0)
    bad = runs = 0
    for s in range(seeds):
        rng = np.random.default_rng(700 + s)
        for L in (33, 11, 1, 5, 40):
            for D in (1, 2, 7, 30, 90, 180, 200):
                for mode in range(6):
                    N = int(rng.choice([1, 3, 64, 257]))
                    seq, bq, mq, hap = planes(rng, mode, N, D, L)
                    ref_row = rng.integers(0, 5, (N, L)).astype(np.int32)
                    want = oracle.hap_features_batch(seq, bq, mq, hap, ref_row, nthreads=8)
                    t = [torch.from_numpy(a).cuda() for a in (seq, bq, mq, hap, ref_row)]
                    got = ctx.hap_features(*t).cpu().numpy()
                    ok = np.array_equal(got, want, equal_nan=True)
                    runs += 1
                    if all(a.min() >= -128 and a.max() <= 127 for a in (seq, bq, mq, hap)):
                        g8 = ctx.hap_features(*[x.to(torch.int8) for x in t[:4]], t[4]).cpu().numpy()
                        ok = ok and np.array_equal(g8, want, equal_nan=True)
                    if not ok:
                        bad += 1
                        d = np.argwhere(got != want)
                        print(f"seed {700 + s} L {L} D {D} mode {mode} N {N}: {len(d)} values differ, first {d[:3].tolist()}", [(got[tuple(i)], want[tuple(i)]) for i in d[:3]], flush=True)
        print(f"{s + 1} seeds: {runs} launches, {bad} differ", flush=True)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
