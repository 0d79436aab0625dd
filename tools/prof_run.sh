#!/bin/bash
# tools/prof_run.sh TAG WORKLOAD [bench.py args...] -- on the GPU box: rocprofv3 passes of one bench.py command into gpurun_out/prof_TAG_*:
#   stats  --kernel-trace --stats                         (per-kernel durations; compared with the live HIP-event numbers)
#   fetch / write  --pmc FETCH_SIZE | WRITE_SIZE          (HBM traffic, separate passes as MI355X_MICROARCH.md prescribes)
#   sqa / sqb / clk   SQ and GRBM counters                (MFMA busy, VALU/MFMA co-execution, waits, LDS conflicts, clock)
# WORKLOAD = pileup | haplotype | two-stage | deep60.  The program itself follows `--` (no env / bash -c hop); counter passes never
# carry trace domains other than the kernel trace.  Summaries: tools/summarize_prof.py, tools/summarize_sq.py.
set -u
TAG=$1; WL=$2; shift 2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
COMMON="--workload $WL --no-cpu-baseline --no-second-precision --no-parity-sample --repeat 1 --workloads none"
run() { # name, rocprof args..., -- bench args
  local name=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/prof_${TAG}_$name -o run -- python3 bench.py $COMMON $BENCH_ARGS > $OUT/prof_${TAG}_$name.json 2> $OUT/prof_${TAG}_$name.err
  echo "$name rc=$?"
}
BENCH_ARGS="$*"
run stats --kernel-trace --stats
case $WL in
  pileup)    SHORT="--no-kernel-timing --steps 2 --warmup 1 --windows 65536" ;;
  haplotype) SHORT="--steps 2 --warmup 1 --hap-sites 32768" ;;
  two-stage) SHORT="--steps 1 --warmup 1" ;;
  deep60)    SHORT="--steps 1 --warmup 1 --hap-sites 16384" ;;
esac
BENCH_ARGS="$* $SHORT"
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
if [ "${PROF_SQ:-1}" = "1" ]; then
run sqa --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES
run sqb --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
run clk --kernel-trace --pmc GRBM_GUI_ACTIVE
fi
