#!/usr/bin/env python3
"""Development probe: legacy CatModel forward alone at one batch size (sites/s)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from nanosnp_amd.fixtures import seeded_cat_weights, synth_cat_groups
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _lib.Context(0)
ctx.cat_load_weights(seeded_cat_weights(21))
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx.set_option("cat_precision", prec)
ctx.set_option("cat_conv_lds", int(os.environ.get("CONVLDS", "1")))
ctx.set_option("cat_conv_pix2", int(os.environ.get("PIX2", "1")))
ctx.enable_timing(True)
g0, g1 = synth_cat_groups(5, 256)
reps = (N + 255) // 256
g0 = torch.from_numpy(np.tile(g0, (reps, 1, 1, 1))[:N]).cuda(); g1 = torch.from_numpy(np.tile(g1, (reps, 1, 1, 1))[:N]).cuda()
ctx.cat_forward(g0, g1); torch.cuda.synchronize()
t = time.time()
for _ in range(iters): ctx.cat_forward(g0, g1)
torch.cuda.synchronize()
dt = (time.time() - t) / iters
tm = ctx.read_timing()
conv = tm["cat_conv_chain"]; allp = tm["cat_forward_pass"]
crc = __import__("zlib").crc32(ctx.cat_forward(g0, g1).cpu().numpy().tobytes()) & 0xffffffff
print(f"cat_forward precision={prec} conv_lds={os.environ.get('CONVLDS', '1')} N={N}: {dt*1e3:.2f} ms  {N/dt/1e3:.1f} k sites/s   conv chain {conv[0]/max(conv[1],1):.3f} ms per pass of 4096, whole pass {allp[0]/max(allp[1],1):.3f} ms  crc {crc:08x}")
