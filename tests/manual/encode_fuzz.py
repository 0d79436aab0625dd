#!/usr/bin/env python3
"""One-off fuzz on the GPU box: adversarial mpileup columns (tests/golden/make_golden.py generator, fresh seeds) through the HIP
encode kernel against the oracle, bit for bit, plus completely random byte strings over the mpileup alphabet and opener-dense ones."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from nanosnp_amd import _lib
from oracle import oracle
import make_golden as mg

ctx = _lib.Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
alphabet = np.frombuffer(b"ACGTNacgtn*#+-^$0123456789.,<>!IiDd~", np.uint8)
tot = 0; t0 = time.time()
for r in range(rounds):
    rng = np.random.default_rng(9000 + r)
    ref = rng.choice(np.frombuffer(b"ACGTacgtNn", np.uint8), n_cols).astype(np.uint8)
    if r % 3 == 0:
        cols = [c.encode() for c in mg.adversarial_columns(rng, n_cols, ref)]
    elif r % 3 == 2:   # opener-dense: mostly + - ^ and digits, lengths around the 253-byte end of the fast path, among ordinary columns
        dense = np.frombuffer(b"+-^+-^+-^0123ACGTacgt*#", np.uint8)
        lens = rng.integers(0, 300, n_cols)
        cols = [bytes(rng.choice(dense if rng.random() < 0.3 else alphabet, int(l))) for l in lens]
    else:       # unstructured: any byte sequence of the alphabet, lengths 0..400 (grammar errors included)
        lens = rng.integers(0, 400, n_cols)
        cols = [bytes(rng.choice(alphabet, int(l))) for l in lens]
    off = np.zeros(n_cols + 1, np.int64); np.cumsum([len(c) for c in cols], out=off[1:])
    bases = np.frombuffer(b"".join(cols), np.uint8)
    for min_af, min_cov in ((0.12, 6), (0.3, 2)):
        oc, od, of_ = oracle.encode_columns(bases, off, ref, min_af, min_cov)
        b = torch.from_numpy(bases if bases.size else np.zeros(1, np.uint8)).cuda()
        gc, gd, gf = ctx.pileup_encode_columns(b, torch.from_numpy(off).cuda(), torch.from_numpy(ref).cuda(), min_af=min_af, min_coverage=min_cov)
        torch.cuda.synchronize()
        bad = np.nonzero((gc.cpu().numpy() != oc).any(1) | (gd.cpu().numpy() != od) | (gf.cpu().numpy() != of_))[0]
        assert bad.size == 0, (r, min_af, bad[:5], [cols[i][:80] for i in bad[:2]])
    tot += n_cols
print(f"{tot} fuzz columns x 2 threshold sets: counts, depth and flags bit-identical to the oracle ({time.time()-t0:.0f} s)")
