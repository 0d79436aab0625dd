#!/bin/bash
# R=r06 tools/round_profiles.sh -- on the GPU box: a round's evidence under profiles/ (copied to gpurun_out/${R}_profiles/ for retrieval):
#   rocprofv3 stats / FETCH / WRITE / SQ passes of the four BASELINE workloads (tools/prof_run.sh), of the bf16x3 forwards and of the
#   column encode (30x and 60x: the instruction counts behind the valu-issue bound); then the lines themselves: the DEFAULT line (with every
#   configuration as a sub-line under "workloads"), the full-pool lines of each workload, the two host-fed paths.
set -u
R=${R:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/${R}_profiles
W=1; K=4
ONLY="${ONLY:-pileup haplotype two-stage deep60 bf16x3 hapb3 encode lines}"
has() { case " $ONLY " in *" $1 "*) return 0;; *) return 1;; esac; }
if has pileup; then
bash tools/prof_run.sh ${R}p pileup --steps $K --warmup $W
python3 tools/summarize_prof.py ${R}_pileup gpurun_out/prof_${R}p_stats gpurun_out/prof_${R}p_fetch gpurun_out/prof_${R}p_write --workload pileup --batch 4096 --enc-group 32 --timed $((W*256)) $((K*256)) > /dev/null
python3 tools/summarize_sq.py ${R}_pileup gpurun_out/prof_${R}p_sqa gpurun_out/prof_${R}p_sqb gpurun_out/prof_${R}p_clk > /dev/null
fi
if has haplotype; then
PROF_SQ=1 bash tools/prof_run.sh ${R}h haplotype --steps 4 --warmup 1
python3 tools/summarize_prof.py ${R}_haplotype gpurun_out/prof_${R}h_stats gpurun_out/prof_${R}h_fetch gpurun_out/prof_${R}h_write --workload haplotype --D 90 > /dev/null
python3 tools/summarize_sq.py ${R}_haplotype gpurun_out/prof_${R}h_sqa gpurun_out/prof_${R}h_sqb gpurun_out/prof_${R}h_clk > /dev/null
fi
if has two-stage; then
PROF_SQ=0 bash tools/prof_run.sh ${R}t two-stage --steps 1 --warmup 1
python3 tools/summarize_prof.py ${R}_two_stage gpurun_out/prof_${R}t_stats gpurun_out/prof_${R}t_fetch gpurun_out/prof_${R}t_write --workload two-stage --D 90 --enc-group 32 > /dev/null
fi
if has deep60; then
PROF_SQ=0 bash tools/prof_run.sh ${R}d deep60 --steps 2 --warmup 1
python3 tools/summarize_prof.py ${R}_deep60 gpurun_out/prof_${R}d_stats gpurun_out/prof_${R}d_fetch gpurun_out/prof_${R}d_write --workload deep60 --D 180 --enc-group 32 > /dev/null
fi
if has bf16x3; then
bash tools/prof_cmd.sh ${R}b3 stats,sqa,sqb,clk tools/probes/fwd_probe.py 131072 2 3 > /dev/null
python3 tools/summarize_sq.py ${R}_pileup_bf16x3 gpurun_out/prof_${R}b3_sqa gpurun_out/prof_${R}b3_sqb gpurun_out/prof_${R}b3_clk > /dev/null
python3 tools/summarize_prof.py ${R}_pileup_bf16x3 gpurun_out/prof_${R}b3_stats > /dev/null
fi
if has hapb3; then
HAP_PROBE_REPS=1 bash tools/prof_cmd.sh ${R}hb3 sqa,sqb,clk tools/probes/hap_probe.py 16384 2 > /dev/null
python3 tools/summarize_sq.py ${R}_hap_forward_bf16x3 gpurun_out/prof_${R}hb3_sqa gpurun_out/prof_${R}hb3_sqb gpurun_out/prof_${R}hb3_clk > /dev/null
fi
if has encode; then
# the column encode alone: 4.3 M columns at 30x (= one launch of the bench: 32 batches of 4096 windows) and 2.1 M at 60x
bash tools/probes/prof_enc.sh ${R}e30 131072 30 4 > /dev/null
python3 tools/summarize_sq.py ${R}_encode_4M_30x gpurun_out/prof_${R}e30_sqa gpurun_out/prof_${R}e30_sqb gpurun_out/prof_${R}e30_clk > /dev/null
bash tools/probes/prof_enc.sh ${R}e60 65536 60 4 > /dev/null
python3 tools/summarize_sq.py ${R}_encode_2M_60x gpurun_out/prof_${R}e60_sqa gpurun_out/prof_${R}e60_sqb gpurun_out/prof_${R}e60_clk > /dev/null
fi
cp profiles/${R}_* profiles/roofline_traffic.json gpurun_out/${R}_profiles/ 2>/dev/null
if has lines; then
# every workload's line: the run's FULL result object (bench_details*.json; stdout carries the driver's compact line, kept beside it as *_stdout.txt)
line() { # name details-file bench args...
  local name=$1 det=$2; shift 2
  python bench.py "$@" > gpurun_out/${R}_profiles/${R}_${name}_stdout.txt 2> gpurun_out/${R}_${name}_line.err; echo "$name rc=$?"
  cp $det gpurun_out/${R}_profiles/${R}_${name}_line.json
}
line default bench_details.json --steps 20 --warmup 5
line haplotype bench_details_haplotype.json --workload haplotype --steps 20 --warmup 5
line two_stage bench_details_two_stage.json --workload two-stage --steps 3 --warmup 1
line deep60 bench_details_deep60.json --workload deep60 --steps 8 --warmup 2
line e2e bench_details_e2e.json --workload e2e --steps 8 --warmup 2
line hap_e2e bench_details_hap_e2e.json --workload hap-e2e --steps 8 --warmup 1
line pd_e2e bench_details_pd_e2e.json --workload pd-e2e --steps 16 --warmup 1
fi
rm -rf gpurun_out/prof_${R}*           # raw rocprof output stays on the box (the summaries travel)
ls gpurun_out/${R}_profiles
