#!/bin/bash
# tools/r4_profiles.sh -- on the GPU box: the round-4 evidence under profiles/ (copied to gpurun_out/r04_profiles/ for retrieval):
#   rocprofv3 stats / FETCH / WRITE / SQ passes of the four BASELINE workloads (tools/prof_run.sh), of the bf16x3 PileupModel forward and of
#   the column encode; then the five bench lines themselves.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_profiles
W=1; K=4
ONLY="${ONLY:-pileup haplotype two-stage deep60 bf16x3 hapb3 lines}"
has() { case " $ONLY " in *" $1 "*) return 0;; *) return 1;; esac; }
if has pileup; then
bash tools/prof_run.sh r04p pileup --steps $K --warmup $W
python3 tools/summarize_prof.py r04_pileup gpurun_out/prof_r04p_stats gpurun_out/prof_r04p_fetch gpurun_out/prof_r04p_write --workload pileup --batch 4096 --enc-group 32 --timed $((W*256)) $((K*256)) > /dev/null
python3 tools/summarize_sq.py r04_pileup gpurun_out/prof_r04p_sqa gpurun_out/prof_r04p_sqb gpurun_out/prof_r04p_clk > /dev/null
fi
if has haplotype; then
PROF_SQ=1 bash tools/prof_run.sh r04h haplotype --steps 4 --warmup 1
python3 tools/summarize_prof.py r04_haplotype gpurun_out/prof_r04h_stats gpurun_out/prof_r04h_fetch gpurun_out/prof_r04h_write --workload haplotype --D 90 > /dev/null
python3 tools/summarize_sq.py r04_haplotype gpurun_out/prof_r04h_sqa gpurun_out/prof_r04h_sqb gpurun_out/prof_r04h_clk > /dev/null
fi
if has two-stage; then
PROF_SQ=0 bash tools/prof_run.sh r04t two-stage --steps 1 --warmup 1
python3 tools/summarize_prof.py r04_two_stage gpurun_out/prof_r04t_stats gpurun_out/prof_r04t_fetch gpurun_out/prof_r04t_write --workload two-stage --D 90 --enc-group 32 > /dev/null
fi
if has deep60; then
PROF_SQ=0 bash tools/prof_run.sh r04d deep60 --steps 2 --warmup 1
python3 tools/summarize_prof.py r04_deep60 gpurun_out/prof_r04d_stats gpurun_out/prof_r04d_fetch gpurun_out/prof_r04d_write --workload deep60 --D 180 --enc-group 32 > /dev/null
fi
if has bf16x3; then
# the bf16x3 PileupModel forward alone (N = 131072): kernel stats + SQ counters + clock
bash tools/prof_cmd.sh r04b3 stats,sqa,sqb,clk tools/fwd_probe.py 131072 2 3 > /dev/null
python3 tools/summarize_sq.py r04_pileup_bf16x3 gpurun_out/prof_r04b3_sqa gpurun_out/prof_r04b3_sqb gpurun_out/prof_r04b3_clk > /dev/null
python3 tools/summarize_prof.py r04_pileup_bf16x3 gpurun_out/prof_r04b3_stats > /dev/null
fi
if has hapb3; then
# the bf16x3 HaplotypeModel forward alone (one pass of 16384 sites): SQ counters + clock of its kernels
HAP_PROBE_REPS=1 bash tools/prof_cmd.sh r04hb3 sqa,sqb,clk tools/hap_probe.py 16384 2 > /dev/null
python3 tools/summarize_sq.py r04_hap_forward_bf16x3 gpurun_out/prof_r04hb3_sqa gpurun_out/prof_r04hb3_sqb gpurun_out/prof_r04hb3_clk > /dev/null
fi
cp profiles/r04_* profiles/roofline_traffic.json gpurun_out/r04_profiles/
if has lines; then
# the lines
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_profiles/r04_pileup_line.json 2> gpurun_out/r04_pileup_line.err; echo pileup rc=$?
python bench.py --workload haplotype --steps 20 --warmup 5 > gpurun_out/r04_profiles/r04_haplotype_line.json 2> gpurun_out/r04_haplotype_line.err; echo hap rc=$?
python bench.py --workload two-stage --steps 3 --warmup 1 > gpurun_out/r04_profiles/r04_two_stage_line.json 2> gpurun_out/r04_two_stage_line.err; echo two rc=$?
python bench.py --workload deep60 --steps 8 --warmup 2 > gpurun_out/r04_profiles/r04_deep60_line.json 2> gpurun_out/r04_deep60_line.err; echo deep rc=$?
python bench.py --workload e2e --steps 5 --warmup 1 > gpurun_out/r04_profiles/r04_e2e_line.json 2> gpurun_out/r04_e2e_line.err; echo e2e rc=$?
fi
rm -rf gpurun_out/prof_r04*           # raw rocprof output stays on the box (the summaries travel)
ls gpurun_out/r04_profiles
