#!/bin/bash
# tools/probes/prof_enc.sh TAG <enc_probe.py args...> -- SQ / GRBM / traffic counter passes of the column encode alone
set -u
TAG=$1; shift
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/prof_${TAG}_$name -o run -- python3 tools/probes/enc_probe.py $ARGS > $OUT/prof_${TAG}_$name.log 2> $OUT/prof_${TAG}_$name.err; echo "$name rc=$?"; }
ARGS="$*"
run sqa --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD
run sqb --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
run clk --kernel-trace --pmc GRBM_GUI_ACTIVE
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
