# per-stage cost of the mpileup parser on the GPU box's host CPU, one thread, then the whole call at 1..16 threads
python - <<'PY'
from nanosnp_amd import host
cols = host.synth_columns(5, 1500000, coverage=30)
open('/tmp/nsnp_parse_probe.mpileup','wb').write(memoryview(cols.mpileup_text_native("chr20s")))
PY
gcc -O3 -std=gnu11 -fopenmp -Iinclude -o /tmp/parse_probe tools/probes/parse_probe.c -lm && /tmp/parse_probe && /tmp/parse_probe
python - <<'PY'
import os, time, numpy as np, mmap
from nanosnp_amd import host
f=open('/tmp/nsnp_parse_probe.mpileup','rb'); big=np.frombuffer(mmap.mmap(f.fileno(),0,access=mmap.ACCESS_READ),np.uint8)
n=len(big)
out=(np.empty(n//8+2,np.int64),np.empty(n//8+3,np.int64),np.empty(n,np.uint8))
for o in out: o[:]=0
for g in "1010":
    os.environ["NSNP_PARSE_GENERIC"]=g
    best=9
    for r in range(7):
        t=time.perf_counter(); host.mpileup_parse_range(big,0,n,out=out); best=min(best,time.perf_counter()-t)
    print("generic" if g=="1" else "avx2", f"{n/1e6:.0f} MB {best*1e3:.1f} ms {n/best/1e9:.2f} GB/s")
PY
