#!/usr/bin/env python3
"""Fuzz: columns built from random BYTES under a loose grammar (every symbol class of tensor_maker.cpp:83-114 - bases in both cases,
* #, ^x, $, +n / -n with n up to 3 digits and allele text that may be shorter than n or run into the next column's boundary, digits and
punctuation the scanner ignores) - the HIP column encode against oracle/liboracle.so, bit for bit (counts, depth, flags), seed after
seed.  Test infrastructure (loads oracle/): `python tests/manual/encode_fuzz_bytes.py [seeds] [columns]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from oracle import oracle

def make(seed, m):
    rng = np.random.default_rng(seed)
    alpha_sets = [b"ACGTNacgtn", b"*#", b"^", b"$", b"+-", b"0123456789", b".,<>!?@~ ;:\t=/\\|%&()[]{}'\"`_", bytes(range(1, 256))]
    w = rng.dirichlet(np.ones(len(alpha_sets)) * rng.uniform(0.2, 2.0))
    cols, ref = [], []
    for c in range(m):
        kind = rng.integers(0, 10)
        depth = int(rng.integers(0, 4)) if kind == 0 else int(rng.integers(0, 200)) if kind < 8 else int(rng.integers(200, 1500))
        out = bytearray()
        for _ in range(depth):
            k = rng.choice(len(alpha_sets), p=w) if kind != 9 else rng.choice(len(alpha_sets))
            a = alpha_sets[k]
            ch = a[rng.integers(0, len(a))]
            out.append(ch)
            if ch in b"+-" and rng.random() < 0.9:
                n = int(rng.choice([0, 1, 2, 3, 5, 9, 10, 30, 59, 60, 61, 99, 100, 250]))
                out += str(n).encode() if rng.random() < 0.95 else b""
                ln = n if rng.random() < 0.8 else int(rng.integers(0, n + 3))
                al = bytes(rng.choice(np.frombuffer(b"ACGTNacgtn*#", np.uint8), size=ln)) if rng.random() < 0.85 else bytes(rng.integers(1, 256, size=ln, dtype=np.uint8))
                out += al
        cols.append(bytes(out).replace(b"\n", b"x").replace(b"\0", b"y"))
        ref.append(rng.choice(np.frombuffer(b"ACGTNacgtnRYKM*", np.uint8)))
    off = np.zeros(m + 1, np.int64); np.cumsum([len(c) for c in cols], out=off[1:])
    return np.frombuffer(b"".join(cols) or b"\0", np.uint8).copy(), off, np.asarray(ref, np.uint8)

def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    ctx = _lib.Context(0)
    bad = 0
    for s in range(seeds):
        bases, off, ref = make(9000 + s + int(os.environ.get("NSNP_STRESS_SEED", "0")), m)
        for af, mc in ((0.12, 6), (0.0, 0), (0.5, 20)):
            oc, od, of = oracle.encode_columns(bases, off, ref, af, mc)
            c, d, f = ctx.pileup_encode_columns(torch.from_numpy(bases).cuda(), torch.from_numpy(off).cuda(), torch.from_numpy(ref).cuda(), af, mc)
            ok = np.array_equal(c.cpu().numpy(), oc) and np.array_equal(d.cpu().numpy(), od) and np.array_equal(f.cpu().numpy(), of)
            if not ok:
                bad += 1
                diff = np.nonzero((c.cpu().numpy() != oc).any(1) | (d.cpu().numpy() != od) | (f.cpu().numpy() != of))[0]
                print(f"seed {9000 + s} af {af} cov {mc}: {diff.size} columns differ, first {diff[:5]}", flush=True)
                i = int(diff[0]); print("   column bytes:", bytes(bases[off[i]:off[i + 1]])[:200], "ref", chr(ref[i]))
                print("   hip   ", c[i].tolist(), int(d[i]), int(f[i])); print("   oracle", oc[i].tolist(), int(od[i]), int(of[i]))
        if s % 10 == 9:
            print(f"{s + 1} seeds x {m} columns x 3 thresholds: {'identical' if not bad else str(bad) + ' DIFFER'}", flush=True)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
