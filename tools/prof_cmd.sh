#!/bin/bash
# tools/prof_cmd.sh TAG PASSES SCRIPT [args...] -- on the GPU box: rocprofv3 passes of `python3 SCRIPT args` into gpurun_out/prof_TAG_<pass>:
#   PASSES = comma list of: stats (--kernel-trace --stats), fetch, write (--pmc FETCH_SIZE | WRITE_SIZE, separate passes as
#   MI355X_MICROARCH.md prescribes), sqa, sqb (SQ counters), clk (GRBM_GUI_ACTIVE), tcc (L2 hits / misses / requests)
# The program itself follows `--` (no env / bash -c hop); counter passes carry no trace domain other than the kernel trace.
set -u
TAG=$1; PASSES=$2; shift 2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
run() { # name, rocprof args...
  local name=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/prof_${TAG}_$name -o run -- python3 "${CMD[@]}" > $OUT/prof_${TAG}_$name.out 2> $OUT/prof_${TAG}_$name.err
  echo "$name rc=$?"
}
CMD=("$@")
for p in ${PASSES//,/ }; do
  case $p in
    stats) run stats --kernel-trace --stats ;;
    fetch) run fetch --kernel-trace --pmc FETCH_SIZE ;;
    write) run write --kernel-trace --pmc WRITE_SIZE ;;
    sqa) run sqa --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES ;;
    sqb) run sqb --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS ;;
    clk) run clk --kernel-trace --pmc GRBM_GUI_ACTIVE ;;
    tcc) run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ;;
  esac
done
