#!/bin/bash
# tools/prof_probe.sh TAG <probe.py args...> -- SQ / GRBM counter passes of tools/fwd_probe.py (forward alone, one stream)
set -u
TAG=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/prof_${TAG}_$name -o run -- python3 tools/fwd_probe.py $ARGS > $OUT/prof_${TAG}_$name.log 2> $OUT/prof_${TAG}_$name.err; echo "$name rc=$?"; }
ARGS="$*"
run sqa --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES
run sqb --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
run clk --kernel-trace --pmc GRBM_GUI_ACTIVE
