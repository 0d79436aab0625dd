# bf16x3 layer 0 at small launches: the plain kernel (16 sites per workgroup, L0SG=1) against the skewed two-group kernel (L0SG=2)
for n in 4096 8192 16384; do for sg in 0 1 2; do L0SG=$sg python tools/fwd_probe.py $n 2 20 2>&1 | tail -1 | sed 's/wpb=0 L0RS=1 L1RS=1//; s/ L1SG=0 STAG=0 PRIO=0 WXL=0//'; done; done
