#!/usr/bin/env python3
"""Development probe: nsnp_mpileup_tokenise alone on one chunk of synthetic 30x mpileup text (default 64 MB), HIP-event time per call and
the HBM rate of its algorithmic bytes (text read once + column-5 bytes + 17 B per line written).  Under rocprofv3 --kernel-trace --stats
the five launches are listed one by one (tools/prof_cmd.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import _lib, host
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n_cols = mb * (1 << 20) // 88
cols = host.synth_columns(20260900, n_cols, coverage=30.0, het_rate=0.03)
text = torch.from_numpy(np.frombuffer(memoryview(cols.mpileup_text_native("chr20s")), np.uint8).copy()).cuda()
seq = torch.from_numpy(cols.ref.copy()).cuda()
ctx = _lib.Context(0)
ctx.set_option("tok_fused", int(os.environ.get("NSNP_TOK_FUSED", "0")))
T = text.numel()
cap = T // 10 + 2
pos = torch.empty(cap, dtype=torch.int64, device="cuda"); off = torch.empty(cap + 1, dtype=torch.int64, device="cuda")
ref = torch.empty(cap, dtype=torch.uint8, device="cuda"); bases = torch.empty(T, dtype=torch.uint8, device="cuda")
meta = torch.zeros(4, dtype=torch.int64, pin_memory=True)
for _ in range(3):
    ctx.mpileup_tokenise_into(text, seq, pos, off, bases, ref, meta)
torch.cuda.synchronize()
m, nb, status, _ = meta.tolist()
assert status == 0 and m == n_cols and nb == int(cols.col_off[-1])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ctx.mpileup_tokenise_into(text, seq, pos, off, bases, ref, meta)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
alg = T + nb + 17 * m
print(f"{T / 1e6:.1f} MB of text, {m} lines: {ms * 1e3:.1f} us per call = {T / ms / 1e6:.0f} GB/s of text, {alg / ms / 1e6:.0f} GB/s algorithmic ({alg / ms / 1e6 / 8000:.3f} of HBM)")
