# removal timing of k_hap_features (L = 33, D = 90, 16384 sites): variants built with tools/build_variant.sh hf_X -DNSNP_HF_X
python tools/probes/feat_warm_probe.py 2>&1 | grep "warm   32" | sed "s/^/base     /"
for v in NOOUT NOFLUSH NOLOAD; do NANOSNP_DEV_LIB_OVERRIDE=1 NANOSNP_HIP_LIB=variants/libnanosnp_hip_hf_$v.so python tools/probes/feat_warm_probe.py 2>&1 | grep "warm   32" | sed "s/^/$v  /"; done
