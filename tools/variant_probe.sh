# tools/variant_probe.sh NAME...: forward probes (fp32 and bf16x3, 131072 sites) of the shipped library and of the A/B builds build_tmp/libs/libnanosnp_hip_NAME.so
for p in 0 2; do python tools/fwd_probe.py 131072 $p 5 2>&1 | tail -1 | sed "s/^/base  /"; done
for v in "$@"; do
  for p in 0 2; do NANOSNP_DEV_LIB_OVERRIDE=1 NANOSNP_HIP_LIB=build_tmp/libs/libnanosnp_hip_$v.so python tools/fwd_probe.py 131072 $p 5 2>&1 | tail -1 | sed "s/^/$v  /"; done
done
