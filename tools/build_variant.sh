#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> variants/libnanosnp_hip_NAME.so  (A/B builds; select with NANOSNP_DEV_LIB_OVERRIDE=1 NANOSNP_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics -ffp-contract=off \
  -Iinclude "$@" -o variants/libnanosnp_hip_$NAME.so nanosnp_amd/csrc/*.hip 2>&1 | grep -E "error|spill|Unknown" || true
ls -la variants/libnanosnp_hip_$NAME.so
