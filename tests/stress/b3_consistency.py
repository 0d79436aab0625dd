#!/usr/bin/env python3
"""Stress: the bf16x3 kernels choose their workgroup shapes by launch size (layer 0 plain / skewed, layer 1 site groups 1 / 2 / 4, the tile
GEMM's 128 x 128 / 256 x 256 LSTM tiles) - every site's result must not depend on the choice.  PileupModel: forward of N sites against the
same sites in ragged sub-batches; HaplotypeModel: pass sizes that fall on either side of the 256 x 256 threshold, hap_b3x on / off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from nanosnp_amd.fixtures import load_pileup_weights, seeded_hap_weights

rng = np.random.default_rng(20261004)
ctx = _lib.Context(0)
ctx.pileup_load_weights(load_pileup_weights())
N = 40000
x = torch.from_numpy((rng.integers(0, 60, (N, 33, 18)) - 12).astype(np.int32)).cuda()
x[5, 3, 2] = 70000; x[17] *= 300                       # the two- and three-term input levels
x[1000:1016, 5, 7] = 131008; x[1016, 9, 1] = -200000; x[20000, 16, :] = 3000
bad_total = 0
for prec in (int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("2", "0", "1"))):
  ctx.set_option("pileup_precision", prec)
  gt, zy = ctx.pileup_forward(x); torch.cuda.synchronize()
  bad = 0
  for trial in range(12):
      cuts = np.sort(rng.choice(np.arange(1, N), size=int(rng.integers(1, 9)), replace=False)).tolist()
      edges = [0] + cuts + [N]
      for a, b in zip(edges, edges[1:]):
          g, z = ctx.pileup_forward(x[a:b])
          if not (torch.equal(g, gt[a:b]) and torch.equal(z, zy[a:b])):
              bad += 1; print("pileup precision", prec, "differs on", a, b, float((g - gt[a:b]).abs().max()))
  for n in (1, 15, 16, 17, 31, 33, 4095, 4096, 4097, 8191, 8192, 8193):
      g, z = ctx.pileup_forward(x[:n])
      if not (torch.equal(g, gt[:n]) and torch.equal(z, zy[:n])):
          bad += 1; print("pileup precision", prec, "differs at n =", n)
  print(f"PileupModel precision {prec}: sub-batch results", "identical" if not bad else f"DIFFER ({bad})")
  bad_total += bad
bad = bad_total

# windows addressed through centre indices into a counts array against the same windows gathered: same bits, every precision
M = 60000
counts = torch.from_numpy((rng.integers(0, 60, (M, 18)) - 12).astype(np.int32)).cuda()
counts[1000:1040, 3] = 900; counts[30000, 0] = 1 << 21
badw = 0
for prec in (2, 0, 1):
    ctx.set_option("pileup_precision", prec)
    for n in (1, 17, 4095, 4096, 9000):
        centers = torch.from_numpy(np.sort(rng.choice(np.arange(16, M - 16), size=n, replace=False))).cuda()
        gw, zw = ctx.pileup_forward_windows(counts, centers)
        xg = ctx.pileup_gather_windows(counts, centers)
        gx, zx = ctx.pileup_forward(xg)
        if not (torch.equal(gw, gx) and torch.equal(zw, zx)):
            badw += 1; print("windows / gathered differ: precision", prec, "n", n)
print("forward_windows against gathered windows:", "identical" if not badw else f"DIFFER ({badw})")
bad += badw

h = _lib.Context(0)
h.hap_load_weights(seeded_hap_weights(12, H=256))
h.set_option("hap_precision", 2)
Nh = 9000
xp = torch.from_numpy((rng.standard_normal((Nh, 105, 33)) * 30).astype(np.float32)).cuda()
xh = torch.from_numpy((rng.standard_normal((Nh, 105, 11)) * 30).astype(np.float32)).cuda()
h.set_option("hap_b3x", 0); h.set_option("hap_pass_sites", 16384)
g0, z0 = h.hap_forward(xp, xh); torch.cuda.synchronize()
badh = 0
for b3x in (1, 0):
    h.set_option("hap_b3x", b3x)
    for ps in (16384, 8192, 4096, 2048, 1024, 768, 256, 128):
        h.set_option("hap_pass_sites", ps)
        g, z = h.hap_forward(xp, xh)
        if not (torch.equal(g, g0) and torch.equal(z, z0)):
            badh += 1; print("hap bf16x3 differs: b3x", b3x, "pass", ps, float((g - g0).abs().max()))
    for n in (1, 127, 128, 129, 255, 256, 257, 4095, 4096, 4097, 8191, 8192):
        h.set_option("hap_pass_sites", 16384)
        g, z = h.hap_forward(xp[:n], xh[:n])
        if not (torch.equal(g, g0[:n]) and torch.equal(z, z0[:n])):
            badh += 1; print("hap bf16x3 differs: b3x", b3x, "n", n)
print("HaplotypeModel bf16x3: pass sizes / tile shapes", "identical" if not badh else f"DIFFER ({badh})")
sys.exit(1 if (bad or badh) else 0)
