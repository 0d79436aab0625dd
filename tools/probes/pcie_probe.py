#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench workload: the column bytes / offsets / reference bases of every batch start in pinned
host memory and are copied on the batch's stream before its kernels (never the bench `value`)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from nanosnp_amd import _lib, host
from nanosnp_amd.fixtures import load_pileup_weights

batch = 4096; S = 32; K = int(sys.argv[1]) if len(sys.argv) > 1 else 256; W = 16
dev = torch.device("cuda", 0)
n_windows = (K + W) * batch if (K + W) * batch < (1 << 20) else 1 << 20
n_batches = n_windows // batch
cols = host.synth_columns(20260000, n_windows * 33, coverage=30.0, window=33)
h_bases = torch.from_numpy(cols.bases).pin_memory()
h_off = torch.from_numpy(cols.col_off).pin_memory()
h_ref = torch.from_numpy(cols.ref).pin_memory()
mcols = batch * 33
max_bytes = int(max(cols.col_off[(b + 1) * mcols] - cols.col_off[b * mcols] for b in range(n_batches)))
centers = (torch.arange(batch, dtype=torch.int64, device=dev) * 33 + 16).contiguous()
lib = _lib.load(); P = C.c_void_p
ctxs, streams, bufs = [], [], []
for s in range(S):
    ctx = _lib.Context(0, chunk_sites=batch); ctx.pileup_load_weights(load_pileup_weights()); ctx.set_option("l0_site_groups", 4); ctx.set_option("fused_waves", 8)
    ctxs.append(ctx); streams.append(torch.cuda.Stream(device=dev))
    bufs.append(dict(bases=torch.empty(max_bytes + 64, dtype=torch.uint8, device=dev), off=torch.empty(mcols + 1, dtype=torch.int64, device=dev),
                     ref=torch.empty(mcols, dtype=torch.uint8, device=dev), counts=torch.empty((mcols, 18), dtype=torch.int32, device=dev),
                     depth=torch.empty(mcols, dtype=torch.int32, device=dev), flags=torch.empty(mcols, dtype=torch.uint8, device=dev),
                     gt=torch.empty((batch, 21), device=dev), zy=torch.empty((batch, 3), device=dev)))
# per-batch offsets rebased to the batch's own byte range (host side, once)
h_off_local = torch.empty((n_batches, mcols + 1), dtype=torch.int64).pin_memory()
for b in range(n_batches):
    h_off_local[b] = h_off[b * mcols:(b + 1) * mcols + 1] - h_off[b * mcols]

def step(i):
    b = i % n_batches; s = i % S; bf = bufs[s]; st = streams[s]
    b0, b1 = int(cols.col_off[b * mcols]), int(cols.col_off[(b + 1) * mcols])
    with torch.cuda.stream(st):
        bf["bases"][:b1 - b0].copy_(h_bases[b0:b1], non_blocking=True)
        bf["off"].copy_(h_off_local[b], non_blocking=True)
        bf["ref"].copy_(h_ref[b * mcols:(b + 1) * mcols], non_blocking=True)
    sp = P(st.cuda_stream); h = ctxs[s].handle
    rc = lib.nsnp_pileup_encode_columns(h, P(bf["bases"].data_ptr()), P(bf["off"].data_ptr()), P(bf["ref"].data_ptr()), mcols,
                                        C.c_double(0.12), 6, P(bf["counts"].data_ptr()), P(bf["depth"].data_ptr()), P(bf["flags"].data_ptr()), sp)
    rc = rc or lib.nsnp_pileup_forward_windows(h, P(bf["counts"].data_ptr()), P(centers.data_ptr()), batch, P(bf["gt"].data_ptr()), P(bf["zy"].data_ptr()), sp)
    assert rc == 0

for i in range(W): step(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(W, W + K): step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t
nbytes = sum(int(cols.col_off[((i % n_batches) + 1) * mcols] - cols.col_off[(i % n_batches) * mcols]) + mcols * 9 + 8 for i in range(W, W + K))
print(f"PCIe-inclusive: {K * batch / dt / 1e6:.2f} M sites/s, {dt / K * 1e3:.3f} ms/step, H2D {nbytes / dt / 1e9:.1f} GB/s ({nbytes / (K * batch):.0f} B/site)")
