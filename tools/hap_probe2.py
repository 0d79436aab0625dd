#!/usr/bin/env python3
"""Development probe: HaplotypeModel forward of N sites as K independent site ranges on K streams / contexts (K = 1, 2, 3, 4):
do the tails of one chain's dependent step launches overlap with the other chains' bodies?  hap_probe2.py [N] [pass]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib
from nanosnp_amd.fixtures import seeded_hap_weights
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
w = seeded_hap_weights(12, H=256)
rng = np.random.default_rng(0)
xp = torch.from_numpy((rng.standard_normal((N, 105, 33)) * 30).astype(np.float32)).cuda()
xh = torch.from_numpy((rng.standard_normal((N, 105, 11)) * 30).astype(np.float32)).cuda()
EXEC = 277.1e6
for K in (1, 2, 3, 4, 1, 2):
    ctxs = []
    for k in range(K):
        c = _lib.Context(0); c.set_option("hap_pass_sites", 8192 if K > 2 else 16384); c.hap_load_weights(w); ctxs.append(c)
    streams = [torch.cuda.Stream() for _ in range(K)]
    per = -(-N // K // 128) * 128
    def run():
        outs = []
        for k in range(K):
            lo, hi = k * per, min(N, (k + 1) * per)
            if lo < hi:
                outs.append(ctxs[k].hap_forward(xp[lo:hi], xh[lo:hi], stream=streams[k]))
        return outs
    run(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3): run()
    torch.cuda.synchronize()
    dt = (time.time() - t) / 3
    print(f"K={K} chains: {dt*1e3:.2f} ms  {N/dt/1e3:.1f} k sites/s  executed {N*EXEC/dt/1e12:.1f} TFLOP/s = {N*EXEC/dt/1e12/157.3:.3f}", flush=True)
    for c in ctxs: c.close()
