# the mpileup parser on the GPU box's host CPU: per-stage cost on one thread (tools/probes/parse_probe.c), then the whole call
# (nsnp_mpileup_parse_into, 131 MB of text, pinned-style caller buffers) at several thread counts
python - <<'PY'
from nanosnp_amd import host
cols = host.synth_columns(5, 1500000, coverage=30)
open('/tmp/nsnp_parse_probe.mpileup','wb').write(memoryview(cols.mpileup_text_native("chr20s")))
PY
gcc -O3 -std=gnu11 -fopenmp -Iinclude -o /tmp/parse_probe tools/probes/parse_probe.c -lm && /tmp/parse_probe
for t in 1 4 8 16 32 64; do
NSNP_HOST_THREADS=$t python - <<'PY'
import os, time, numpy as np
from nanosnp_amd import host
big=np.fromfile('/tmp/nsnp_parse_probe.mpileup',np.uint8)
n=len(big)
out=(np.empty(n//8+2,np.int64),np.empty(n//8+3,np.int64),np.empty(n,np.uint8))
for o in out: o[:]=0
for g in "140":
    os.environ["NSNP_PARSE_GENERIC"]=g
    ts=[]
    for r in range(9):
        t=time.perf_counter(); p,_,_=host.mpileup_parse_range(big,0,n,out=out); ts.append(time.perf_counter()-t)
    ts.sort()
    print(os.environ["NSNP_HOST_THREADS"], "threads", {"1": "portable", "4": "block-oriented", "0": "line-oriented"}[g], p.size, f"{n/1e6:.0f} MB  min {ts[0]*1e3:.2f} ms  median {ts[4]*1e3:.2f} ms  {n/ts[4]/1e9:.1f} GB/s")
PY
done
