#!/usr/bin/env python3
"""Development probe: device time of every operation stream_contig issues for ONE chunk of mpileup text (750 k columns at 30x), each
bracketed by events on an otherwise idle device."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import host
from nanosnp_amd.fixtures import load_pileup_weights
from nanosnp_amd.pileup_model import LSTMNetwork
from nanosnp_amd.predict import COV_CHANNELS

M = int(sys.argv[1]) if len(sys.argv) > 1 else 750000
dev = torch.device("cuda", 0)
model = LSTMNetwork(device=0).load_weight_list(load_pileup_weights())
ctx = model.ctx
cols = host.synth_columns(20260900, M, coverage=30.0, het_rate=0.03)
d_seq = torch.from_numpy(cols.ref).to(dev)
h_pos = torch.from_numpy(cols.pos).pin_memory(); h_off = torch.from_numpy(cols.col_off).pin_memory(); h_b = torch.from_numpy(cols.bases).pin_memory()
cov_idx = torch.tensor(list(COV_CHANNELS), dtype=torch.int64, device=dev)

def timed(name, fn, reps=5):
    out = fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:<46s} {e0.elapsed_time(e1) / reps:8.3f} ms")
    return out

d_pos = timed("H2D pos", lambda: h_pos.to(dev, non_blocking=True))
d_off = timed("H2D col_off", lambda: h_off.to(dev, non_blocking=True))
d_b = timed("H2D bases", lambda: h_b.to(dev, non_blocking=True))
d_ref = timed("reference bases d_seq[pos - 1]", lambda: d_seq[d_pos - 1])
counts, depth, flags = timed("encode_columns (incl. output allocation)", lambda: ctx.pileup_encode_columns(d_b, d_off, d_ref, 0.12, 6))
center, n_sel = timed("select_sites_async (incl. torch.full)", lambda: ctx.pileup_select_sites_async(d_pos, flags))
meta = timed("halo counts + stack", lambda: torch.stack([n_sel[0], (center < 16).sum(), (center < M - 16).sum(), n_sel[0]]))
n, lo, hi, _ = meta.tolist()
centers = center[lo:hi]
print("sites", hi - lo)
outs = timed("forward_windows_calls", lambda: ctx.pileup_forward_windows_calls(counts, centers))
gt, zy, ga, za, gm, zm = outs
def rows():
    cov = counts.index_select(0, centers).index_select(1, cov_idx).to(torch.float64)
    f64 = lambda t: t.to(torch.float64)[:, None]
    return torch.cat([f64(d_pos.index_select(0, centers)), f64(ga), f64(za), f64(gm), f64(zm), cov], dim=1)
timed("call rows [n, 13] float64", rows)
