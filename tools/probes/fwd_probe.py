#!/usr/bin/env python3
"""Development probe: PileupModel forward alone at one batch size, per-kernel HIP-event times."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from nanosnp_amd.fixtures import load_pileup_weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
wpb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
ctx = _lib.Context(0, chunk_sites=min(N, 131072))
ctx.pileup_load_weights(load_pileup_weights())
ctx.set_option("pileup_precision", prec)
ctx.set_option("recurrence_waves", wpb)
ctx.set_option("fused_l1", int(os.environ.get("FUSED", "1")))
ctx.set_option("fused_waves", int(os.environ.get("FWAVES", "0")))
ctx.set_option("l0_register_stationary", int(os.environ.get("L0RS", "1")))
ctx.set_option("l0_site_groups", int(os.environ.get("L0SG", "0")))
ctx.set_option("l1_register_stationary", int(os.environ.get("L1RS", "1")))
ctx.set_option("l1_site_groups", int(os.environ.get("L1SG", "0")))
ctx.set_option("l1_stagger", int(os.environ.get("STAG", "0")))
ctx.set_option("head_split", int(os.environ.get("HEADS", "1")))
ctx.set_option("static_priority", int(os.environ.get("PRIO", "0")))
ctx.set_option("l0_input_weights_in_lds", int(os.environ.get("WXL", "0")))
torch.manual_seed(1)
x = torch.randint(-20, 40, (N, 33, 18), dtype=torch.int32, device=dev)
gt = torch.empty((N, 21), device=dev); zy = torch.empty((N, 3), device=dev)
ctx.pileup_forward(x, gt, zy); torch.cuda.synchronize()
ctx.enable_timing(True)
t = time.time()
for _ in range(iters): ctx.pileup_forward(x, gt, zy)
torch.cuda.synchronize()
dt = (time.time() - t) / iters
tm = ctx.read_timing()
import zlib
crc = zlib.crc32(gt.cpu().numpy().tobytes()) & 0xffffffff
print(f"crc {crc:08x} N={N} precision={prec} wpb={wpb} L0RS={os.environ.get('L0RS','1')} L1RS={os.environ.get('L1RS','1')} L0SG={os.environ.get('L0SG','0')} L1SG={os.environ.get('L1SG','0')} STAG={os.environ.get('STAG','0')} PRIO={os.environ.get('PRIO','0')} WXL={os.environ.get('WXL','0')}: {dt*1e3:.3f} ms  {N/dt/1e6:.2f} M sites/s ", {k: round(v[0]/max(v[1],1), 4) for k, v in tm.items() if v[1]})
