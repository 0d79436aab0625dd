#!/bin/bash
# build_enc_lab.sh [candidate.hip]  -> build_tmp/enc_lab
set -e
cd "$(dirname "$0")/../.."
mkdir -p build_tmp
gcc -O2 -c nanosnp_amd/csrc/nsnp_synth.c -Iinclude -o build_tmp/nsnp_synth.o
gcc -O2 -c oracle/pileup_encode_oracle.c -Ioracle -o build_tmp/enc_oracle.o
CAND=""
if [ -n "$1" ]; then CAND="-DNSNP_ENC_CANDIDATE=\"$(realpath $1)\""; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inanosnp_amd/csrc $CAND ${ENC_LAB_DEFS} -c tests/manual/enc_lab.hip -o build_tmp/enc_lab.o 2>&1 | grep -E "error|warning: v|spill" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 build_tmp/enc_lab.o build_tmp/nsnp_synth.o build_tmp/enc_oracle.o -lm -o build_tmp/${ENC_LAB_OUT:-enc_lab}
