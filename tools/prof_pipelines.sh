#!/bin/bash
# tools/prof_pipelines.sh TAG -- on the GPU box: rocprofv3 kernel + memory-copy trace of the three host-fed pipeline benches themselves (the program directly
# behind `--`), cut to their timed regions (NSNP_TRACE_MARK) and summarised into profiles/TAG_{e2e,hap_e2e,pd_e2e}_{overlap.json,kernel_stats.csv,line.json}
set -u
TAG=$1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT profiles
for wl in e2e hap-e2e pd-e2e; do
  name=${wl//-/_}
  rm -f $OUT/mark_$name.json
  export NSNP_TRACE_MARK=$PWD/$OUT/mark_$name.json
  # (hap-e2e: rocprofv3's memory-copy tracing crashes in its own teardown at the exit of that run - SIGSEGV under __cxa_finalize, no output files;
  #  measured twice in round 6 - so that pipeline is traced with the kernel trace alone and its copy time stays the line's HIP-event figure)
  COPYTRACE=--memory-copy-trace; [ $wl = hap-e2e ] && COPYTRACE=
  rocprofv3 --kernel-trace $COPYTRACE --stats --output-format csv -d $OUT/prof_${TAG}_$name -o run -- \
      python3 bench.py --workload $wl --steps 8 --warmup 2 --no-second-precision --no-cpu-baseline > $OUT/prof_${TAG}_$name.out 2> $OUT/prof_${TAG}_$name.err
  echo "$wl rc=$?"
  unset NSNP_TRACE_MARK
  cp bench_details_$name.json $OUT/prof_${TAG}_${name}_details.json
  python3 tools/summarize_pipeline_trace.py $OUT/prof_${TAG}_$name $OUT/mark_$name.json $OUT/prof_${TAG}_${name}_details.json \
      profiles/${TAG}_${name}_overlap.json profiles/${TAG}_${name}_kernel_stats.csv
  cp $OUT/prof_${TAG}_${name}_details.json profiles/${TAG}_${name}_traced_line.json      # (the traced run: no second values; the full line is ${TAG}_${name}_line.json of tools/round_profiles.sh)
done
mkdir -p $OUT/${TAG}_profiles      # (gpurun brings gpurun_out/ back, not profiles/: this script's own outputs travel through it)
for name in e2e hap_e2e pd_e2e; do cp profiles/${TAG}_${name}_overlap.json profiles/${TAG}_${name}_kernel_stats.csv profiles/${TAG}_${name}_traced_line.json $OUT/${TAG}_profiles/; done
