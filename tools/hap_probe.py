#!/usr/bin/env python3
"""Development probe: HaplotypeModel forward alone (sites/s): hap_probe.py [N] [precisions, e.g. 0 or 0,1]."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from tests.helpers import seeded_hap_weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.Context(0)
ctx.hap_load_weights(seeded_hap_weights(12, H=256))
rng = np.random.default_rng(0)
xp = torch.from_numpy((rng.standard_normal((N, 105, 33)) * 30).astype(np.float32)).cuda()
xh = torch.from_numpy((rng.standard_normal((N, 105, 11)) * 30).astype(np.float32)).cuda()
precs = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1]
for prec in precs:
    ctx.set_option("hap_precision", prec)
    ctx.hap_forward(xp, xh); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3): ctx.hap_forward(xp, xh)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 3
    print(f"hap_forward precision={prec} N={N}: {dt*1e3:.2f} ms  {N/dt/1e3:.1f} k sites/s  ({N*353.7e6/dt/1e12:.0f} TFLOP/s algorithmic)")
