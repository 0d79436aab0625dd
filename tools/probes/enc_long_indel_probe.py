#!/usr/bin/env python3
"""Development probe: column encode on pileups whose indel lengths follow a geometric law (mean 3, capped at 80: ONT-like; the G2
generator of the bench draws 1..3), checked against the oracle and timed beside G2 columns of the same size.
    python tools/enc_long_indel_probe.py [windows] [coverage] [mean_len]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib, host

n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
mean_len = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
M = 33 * n_win
rng = np.random.default_rng(7)
UP, LO = b"ACGT", b"acgt"
cols, ref = [], np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, M)]
depth = rng.poisson(cov, M)
for c in range(M):
    d = int(depth[c]); out = bytearray()
    r = rng.random((d, 4)); k = rng.integers(0, 4, (d, 2)); ln = np.minimum(rng.geometric(1.0 / mean_len, d), 80)
    for i in range(d):
        fwd = r[i, 0] < 0.5
        if r[i, 1] < 0.0097: out += b"^" + bytes([33 + int(k[i, 1]) * 9])
        out.append((UP if fwd else LO)[k[i, 0]] if r[i, 2] > 0.01 else (42 if fwd else 35))
        if r[i, 3] < 0.03:
            L = int(ln[i]); seq = bytes((UP if fwd else LO)[j] for j in rng.integers(0, 4, L))
            out += (b"+" if r[i, 3] < 0.015 else b"-") + str(L).encode() + seq
        elif r[i, 3] > 0.99: out += b"$"
    cols.append(bytes(out))
REP = 16                                                  # the generated block repeated: enough columns to fill the chip
off1 = np.zeros(M + 1, np.int64); off1[1:] = np.cumsum([len(x) for x in cols])
b1 = np.frombuffer(b"".join(cols), np.uint8)
bases = np.tile(b1, REP)
off = np.concatenate([off1[:-1] + r * off1[-1] for r in range(REP)] + [np.array([REP * off1[-1]])]).astype(np.int64)
ref = np.tile(ref, REP); M *= REP
print(f"{M} columns, {bases.size / M:.1f} bytes per column, indel lengths: mean {mean_len}, share >= 5: {np.mean(np.minimum(rng.geometric(1.0 / mean_len, 100000), 80) >= 5):.3f}")

dev = torch.device("cuda:0")
ctx = _lib.Context(0)
def run(bases, off, ref, tag):
    b = torch.from_numpy(bases).to(dev); co = torch.from_numpy(off).to(dev); rf = torch.from_numpy(np.ascontiguousarray(ref)).to(dev)
    res = ctx.pileup_encode_columns(b, co, rf); torch.cuda.synchronize()
    ctx.enable_timing(True); ctx.read_timing()
    for _ in range(20): ctx.pileup_encode_columns(b, co, rf)
    torch.cuda.synchronize()
    ev = ctx.read_timing()["encode_columns"]; us = ev[0] / ev[1] * 1e3
    print(f"{tag:34s} {us:8.1f} us per launch  {(bases.size + 81.0 * (off.size - 1)) / us / 1e3:7.0f} GB/s")
    return res
res = run(bases, off, ref, "geometric indel lengths")
g2 = host.synth_columns(20260001, M, coverage=cov, window=33)
run(g2.bases, g2.col_off, g2.ref, "G2 (lengths 1..3), same columns")
from oracle import oracle
oc, od, of = oracle.encode_columns(bases, off, ref)
cnt, dep, flg = (t.cpu().numpy() for t in res[:3])
print("vs oracle:", int((cnt.reshape(M, 18) != oc.reshape(M, 18)).any(1).sum() + (dep != od).sum() + (flg != of).sum()), "differences")
