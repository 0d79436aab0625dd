for v in "" b3_NOREC b3_NOCELL b3_NOSPLIT b3_NOH0 b3_NOX b3_NOBAR b3_ALLVALU; do
  for sg in 1 2 4; do
    if [ -z "$v" ]; then L0SG=$sg L1SG=4 python tools/probes/fwd_probe.py 131072 2 5 2>&1 | tail -1 | sed "s/^/base     /";
    else NANOSNP_DEV_LIB_OVERRIDE=1 NANOSNP_HIP_LIB=variants/libnanosnp_hip_$v.so L0SG=$sg L1SG=4 python tools/probes/fwd_probe.py 131072 2 5 2>&1 | tail -1 | sed "s/^/$v /"; fi
  done
done
