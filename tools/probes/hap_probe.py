#!/usr/bin/env python3
"""Development probe: HaplotypeModel forward alone (sites/s): hap_probe.py [N] [precisions, e.g. 0 or 0,1] [pass sizes, e.g. 4096,16384]."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from nanosnp_amd.fixtures import seeded_hap_weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = _lib.Context(0)
ctx.hap_load_weights(seeded_hap_weights(12, H=256))
rng = np.random.default_rng(0)
xp = torch.from_numpy((rng.standard_normal((N, 105, 33)) * 30).astype(np.float32)).cuda()
xh = torch.from_numpy((rng.standard_normal((N, 105, 11)) * 30).astype(np.float32)).cuda()
precs = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1]
passes = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [16384]
for o in sys.argv[4:]:                       # further nsnp_ctx_set_option pairs: name=value
    ctx.set_option(o.split('=')[0], int(o.split('=')[1]))
EXEC = 276.0e6          # executed flop per site (last layer: 17 of 33 / 6 of 11 steps), DESIGN.md section 4
ref = {}
for prec in precs:
    ctx.set_option("hap_precision", prec)
    for ps in passes:
        ctx.set_option("hap_pass_sites", ps)
        gt, zy = ctx.hap_forward(xp, xh); torch.cuda.synchronize()
        if prec in ref:
            same = bool((gt == ref[prec][0]).all() and (zy == ref[prec][1]).all())
        else:
            ref[prec] = (gt.clone(), zy.clone()); same = True
        reps = int(os.environ.get('HAP_PROBE_REPS', '3'))
        t = time.time()
        for _ in range(reps): ctx.hap_forward(xp, xh)
        torch.cuda.synchronize()
        dt = (time.time() - t) / reps
        print(f"hap_forward precision={prec} N={N} pass={ps}: {dt*1e3:.2f} ms  {N/dt/1e3:.1f} k sites/s  "
              f"executed {N*EXEC/dt/1e12:.1f} TFLOP/s = {N*EXEC/dt/1e12/157.3:.3f} of fp32 MFMA peak ({N*353.7e6/dt/1e12:.0f} algorithmic)  "
              f"bit-identical to first pass size: {same}  crc {__import__('zlib').crc32(gt.cpu().numpy().tobytes()) & 0xffffffff:08x}", flush=True)
