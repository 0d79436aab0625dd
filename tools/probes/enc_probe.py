#!/usr/bin/env python3
"""Development probe: column encode alone (for rocprofv3 counter passes)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib, host
n_win = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
ctx = _lib.Context(0)
cols = host.synth_columns(20260001, 33 * n_win, coverage=cov, window=33)
b = torch.from_numpy(cols.bases).to(dev); co = torch.from_numpy(cols.col_off).to(dev); rf = torch.from_numpy(cols.ref).to(dev)
ctx.pileup_encode_columns(b, co, rf); torch.cuda.synchronize()
ctx.enable_timing(True)
t = time.time()
for _ in range(iters): ctx.pileup_encode_columns(b, co, rf)
torch.cuda.synchronize()
dt = (time.time() - t) / iters
M = 33 * n_win
ev = ctx.read_timing()["encode_columns"]
print(f"encode {M} columns cov {cov}: {dt*1e3:.3f} ms  {M/dt/1e9:.2f} G cols/s  {(cols.bases.size + 73*M)/dt/1e9:.0f} GB/s algorithmic;  HIP events: {ev[0]/max(ev[1],1)*1e3:.1f} us per launch = {(cols.bases.size + 73*M)/(ev[0]/max(ev[1],1)*1e-3)/1e9:.0f} GB/s")
