#!/bin/bash
# tools/host_scaling.sh -- on a ONE-GPU box: the three host-fed paths (mpileup text -> VCF, haplotype site file -> csv, window file -> VCF) at 1 / 2 / 4 / 8 ranks that
# SHARE GPU 0 (collectives over gloo): what the host side of N ranks costs under the box's CPU quota.  Device time is serialised across the
# ranks here, so `value` is NOT a scaling number; the per-rank host times (parse / staging / formatting and the waits for them) are the result.
# Lines land in gpurun_out/host_scaling/<workload>_<N>.json; tools/host_scaling_table.py turns them into the table of DESIGN.md.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/host_scaling
for wl in e2e hap-e2e pd-e2e; do
  for n in 1 2 4 8; do
    extra=""; [ $n -gt 1 ] && extra="--share-gpu --dist-backend gloo"
    timeout 900 python bench.py --workload $wl --gpus $n $extra --steps 4 --warmup 1 --no-cpu-baseline --no-second-precision \
      > gpurun_out/host_scaling/${wl}_$n.line 2> gpurun_out/host_scaling/${wl}_$n.err
    echo "$wl x$n rc=$?"
    cp bench_details_${wl//-/_}.json gpurun_out/host_scaling/${wl}_$n.json      # (the run's full result object; stdout carries the driver's compact line)
  done
done
