#!/usr/bin/env python3
"""Development probe: the timeline of one pass of pipeline.call_contig over the e2e bench's contig - parser thread, copy stream, compute
stream and the issuing thread on one clock (ms since the start of the pass)."""
import mmap, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nanosnp_amd import host
from nanosnp_amd.fixtures import load_pileup_weights
from nanosnp_amd.pileup_model import LSTMNetwork
from nanosnp_amd.pipeline import call_contig
n_cols = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
chunk = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 20
model = LSTMNetwork(device=0).load_weight_list(load_pileup_weights())
cols = host.synth_columns(20260900, n_cols, coverage=30.0, het_rate=0.03)
path = os.path.join(tempfile.gettempdir(), "nsnp_timeline.mpileup")
open(path, "wb").write(memoryview(cols.mpileup_text_native("chr20s")))
f = open(path, "rb"); text = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
for _ in range(2):
    call_contig(model, text, "chr20s", cols.ref, chunk_bytes=chunk)
st = {"trace": []}
t0 = time.perf_counter()
call_contig(model, text, "chr20s", cols.ref, chunk_bytes=chunk, stats=st)
t1 = time.perf_counter()
base = min(x[2] for x in st["trace"])
print(f"pass {1e3 * (t1 - t0):.2f} ms; first traced event at {1e3 * (base - t0):.2f} ms")
for what in ("parse", "stage", "h2d", "tokenise", "encode+select", "forward+rows", "main: wait parse", "main: wait stage", "main: issue"):
    rows = sorted((x for x in st["trace"] if x[0] == what), key=lambda x: x[1])
    if not rows:
        continue
    print(f"{what:<18s}" + "  ".join(f"{k}:{1e3 * (a - base):6.2f}-{1e3 * (b - base):6.2f}" for _, k, a, b in rows))
print({k: round(v, 4) for k, v in st.items() if k != "trace" and isinstance(v, float)})
os.remove(path)
