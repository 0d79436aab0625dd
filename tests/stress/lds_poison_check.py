#!/usr/bin/env python3
"""Stress: does any kernel read LDS it never wrote?  Every forward / encode runs once on a fresh chip state, then again after every CU's
LDS was filled with NaN patterns (fp32 quiet NaN, which is also a pair of bf16 NaNs; then 0xffffffff) - the results must be the same bits.
(The bf16x3 layer-0 bug of round 4 - stale split planes - is of this class.)  build: __graft_entry__.build() (hipcc -shared tests/helpers_native/lds_poison.hip ->
tests/helpers_native/liblds_poison.so)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib, host
from nanosnp_amd.fixtures import load_pileup_weights, seeded_hap_weights, seeded_cat_weights, synth_cat_groups
poison = C.CDLL(os.path.join(ROOT, "tests", "helpers_native", "liblds_poison.so")).lds_poison
poison.argtypes = [C.c_void_p, C.c_uint]
rng = np.random.default_rng(11)
ctx = _lib.Context(0)
ctx.pileup_load_weights(load_pileup_weights()); ctx.hap_load_weights(seeded_hap_weights(12, H=256)); ctx.cat_load_weights(seeded_cat_weights(21))
x = torch.from_numpy((rng.integers(0, 60, (20000, 33, 18)) - 12).astype(np.int32)).cuda()
x[7, 4, 1] = 3000; x[4100, 8] *= 500
xp = torch.from_numpy((rng.standard_normal((4096, 105, 33)) * 30).astype(np.float32)).cuda()
xh = torch.from_numpy((rng.standard_normal((4096, 105, 11)) * 30).astype(np.float32)).cuda()
g0, g1 = synth_cat_groups(5, 512); g0 = torch.from_numpy(g0).cuda(); g1 = torch.from_numpy(g1).cuda()
cols = host.synth_columns(99, 200000, coverage=40.0)
b = torch.from_numpy(cols.bases).cuda(); off = torch.from_numpy(cols.col_off).cuda(); ref = torch.from_numpy(cols.ref).cuda()
D, L = 90, 33
planes = [torch.from_numpy(rng.integers(-2, 5, (2048, D, L)).astype(np.int32)).cuda(), torch.from_numpy(rng.integers(0, 60, (2048, D, L)).astype(np.int32)).cuda(),
          torch.from_numpy(rng.integers(0, 61, (2048, D, L)).astype(np.int32)).cuda(), torch.from_numpy(rng.integers(-2, 4, (2048, D, L)).astype(np.int32)).cuda()]
refrow = torch.from_numpy(rng.integers(0, 5, (2048, L)).astype(np.int32)).cuda()

def everything():
    out = {}
    for prec in (0, 2, 1):
        ctx.set_option("pileup_precision", prec); ctx.set_option("hap_precision", prec); ctx.set_option("cat_precision", prec)
        for n in (20000, 3000, 40):
            out[("pileup", prec, n)] = ctx.pileup_forward(x[:n])
        out[("hap", prec)] = ctx.hap_forward(xp, xh)
        out[("hap small", prec)] = ctx.hap_forward(xp[:300], xh[:300])
        out[("cat", prec)] = (ctx.cat_forward(g0, g1),)
    out[("encode",)] = ctx.pileup_encode_columns(b, off, ref)
    out[("features",)] = (ctx.hap_features(*planes, refrow),)
    torch.cuda.synchronize()
    return out

base = everything()
bad = 0
for pattern in (0x7fc07fc0, 0xffffffff, 0x7f800000):
    assert poison(None, pattern) == 0
    torch.cuda.synchronize()
    got = everything()
    for k, v in got.items():
        same = all(torch.equal(a, c) or (torch.isnan(a) & torch.isnan(c)).all() for a, c in zip(v, base[k]) if a is not None)
        if not same:
            bad += 1; print("DIFFERS after LDS poison %08x:" % pattern, k)
print("results after LDS poisoning:", "identical" if not bad else f"{bad} DIFFER")
# the same for the contexts' WORKSPACES in global memory: a pass over NaN / huge inputs leaves NaNs in every intermediate buffer (also
# in the rows a later, smaller call pads its tiles with); the later call must not see them
nan = float("nan")
for prec in (0, 2, 1):
    ctx.set_option("pileup_precision", prec); ctx.set_option("hap_precision", prec); ctx.set_option("cat_precision", prec)
    ctx.hap_forward(torch.full_like(xp, nan), torch.full_like(xh, nan))
    ctx.cat_forward(torch.full_like(g0, nan), torch.full_like(g1, nan))
    ctx.pileup_forward(torch.full_like(x, 2 ** 30))
torch.cuda.synchronize()
got = everything()
badw = 0
for k, v in got.items():
    if not all(torch.equal(a, c) for a, c in zip(v, base[k]) if a is not None):
        badw += 1; print("DIFFERS after workspace poisoning:", k)
print("results after workspace poisoning:", "identical" if not badw else f"{badw} DIFFER")
bad += badw
sys.exit(1 if bad else 0)
