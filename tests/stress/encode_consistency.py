#!/usr/bin/env python3
"""Stress: a column's counts / depth / flags must not depend on which launch, wave or staging sub-batch it falls into - the encode of a
block of columns against the same columns encoded in ragged pieces (cuts at arbitrary columns), at 5x / 30x / 60x / 200x coverage."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib, host
ctx = _lib.Context(0)
rng = np.random.default_rng(7)
bad = 0
for cov, m in ((5.0, 300000), (30.0, 300000), (60.0, 200000), (200.0, 60000)):
    cols = host.synth_columns(4000 + int(cov), m, coverage=cov, max_depth=int(cov * 3))
    b = torch.from_numpy(cols.bases).cuda(); off = torch.from_numpy(cols.col_off).cuda(); ref = torch.from_numpy(cols.ref).cuda()
    c0, d0, f0 = ctx.pileup_encode_columns(b, off, ref)
    for trial in range(6):
        cuts = [0] + np.sort(rng.choice(np.arange(1, m), size=int(rng.integers(1, 12)), replace=False)).tolist() + [m]
        for a, e in zip(cuts, cuts[1:]):
            o = off[a:e + 1] - off[a]
            c, d, f = ctx.pileup_encode_columns(b[int(cols.col_off[a]):int(cols.col_off[e])].contiguous() if cols.col_off[e] > cols.col_off[a] else b[:1], o.contiguous(), ref[a:e].contiguous())
            if not (torch.equal(c, c0[a:e]) and torch.equal(d, d0[a:e]) and torch.equal(f, f0[a:e])):
                bad += 1; print("encode differs", cov, a, e)
    print(f"{cov:g}x, {m} columns: pieces", "identical" if not bad else "DIFFER")
sys.exit(1 if bad else 0)
