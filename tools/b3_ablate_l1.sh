for v in "" b3l1_NOCELL b3l1_NOSTAGE b3l1_NOBAR b3l1_ALL; do
  for sg in 2 4; do
    if [ -z "$v" ]; then L0SG=2 L1SG=$sg python tools/fwd_probe.py 131072 2 5 2>&1 | tail -1 | sed "s/^/base     /";
    else NANOSNP_DEV_LIB_OVERRIDE=1 NANOSNP_HIP_LIB=build_tmp/libs/libnanosnp_hip_$v.so L0SG=2 L1SG=$sg python tools/fwd_probe.py 131072 2 5 2>&1 | tail -1 | sed "s/^/$v /"; fi
  done
done
