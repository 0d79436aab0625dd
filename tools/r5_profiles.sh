#!/bin/bash
# tools/r5_profiles.sh -- on the GPU box: the round-5 evidence under profiles/ (copied to gpurun_out/r05_profiles/ for retrieval):
#   rocprofv3 stats / FETCH / WRITE / SQ passes of the four BASELINE workloads (tools/prof_run.sh), of the bf16x3 forwards and of the
#   column encode (30x and 60x: the instruction counts behind the valu-issue bound); then the lines themselves: the DEFAULT line (with every
#   configuration as a sub-line under "workloads"), the full-pool lines of each workload, the two host-fed paths.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05_profiles
W=1; K=4
ONLY="${ONLY:-pileup haplotype two-stage deep60 bf16x3 hapb3 encode lines}"
has() { case " $ONLY " in *" $1 "*) return 0;; *) return 1;; esac; }
if has pileup; then
bash tools/prof_run.sh r05p pileup --steps $K --warmup $W
python3 tools/summarize_prof.py r05_pileup gpurun_out/prof_r05p_stats gpurun_out/prof_r05p_fetch gpurun_out/prof_r05p_write --workload pileup --batch 4096 --enc-group 32 --timed $((W*256)) $((K*256)) > /dev/null
python3 tools/summarize_sq.py r05_pileup gpurun_out/prof_r05p_sqa gpurun_out/prof_r05p_sqb gpurun_out/prof_r05p_clk > /dev/null
fi
if has haplotype; then
PROF_SQ=1 bash tools/prof_run.sh r05h haplotype --steps 4 --warmup 1
python3 tools/summarize_prof.py r05_haplotype gpurun_out/prof_r05h_stats gpurun_out/prof_r05h_fetch gpurun_out/prof_r05h_write --workload haplotype --D 90 > /dev/null
python3 tools/summarize_sq.py r05_haplotype gpurun_out/prof_r05h_sqa gpurun_out/prof_r05h_sqb gpurun_out/prof_r05h_clk > /dev/null
fi
if has two-stage; then
PROF_SQ=0 bash tools/prof_run.sh r05t two-stage --steps 1 --warmup 1
python3 tools/summarize_prof.py r05_two_stage gpurun_out/prof_r05t_stats gpurun_out/prof_r05t_fetch gpurun_out/prof_r05t_write --workload two-stage --D 90 --enc-group 32 > /dev/null
fi
if has deep60; then
PROF_SQ=0 bash tools/prof_run.sh r05d deep60 --steps 2 --warmup 1
python3 tools/summarize_prof.py r05_deep60 gpurun_out/prof_r05d_stats gpurun_out/prof_r05d_fetch gpurun_out/prof_r05d_write --workload deep60 --D 180 --enc-group 32 > /dev/null
fi
if has bf16x3; then
bash tools/prof_cmd.sh r05b3 stats,sqa,sqb,clk tools/fwd_probe.py 131072 2 3 > /dev/null
python3 tools/summarize_sq.py r05_pileup_bf16x3 gpurun_out/prof_r05b3_sqa gpurun_out/prof_r05b3_sqb gpurun_out/prof_r05b3_clk > /dev/null
python3 tools/summarize_prof.py r05_pileup_bf16x3 gpurun_out/prof_r05b3_stats > /dev/null
fi
if has hapb3; then
HAP_PROBE_REPS=1 bash tools/prof_cmd.sh r05hb3 sqa,sqb,clk tools/hap_probe.py 16384 2 > /dev/null
python3 tools/summarize_sq.py r05_hap_forward_bf16x3 gpurun_out/prof_r05hb3_sqa gpurun_out/prof_r05hb3_sqb gpurun_out/prof_r05hb3_clk > /dev/null
fi
if has encode; then
# the column encode alone: 4.3 M columns at 30x (= one launch of the bench: 32 batches of 4096 windows) and 2.1 M at 60x
bash tools/prof_enc.sh r05e30 131072 30 4 > /dev/null
python3 tools/summarize_sq.py r05_encode_4M_30x gpurun_out/prof_r05e30_sqa gpurun_out/prof_r05e30_sqb gpurun_out/prof_r05e30_clk > /dev/null
bash tools/prof_enc.sh r05e60 65536 60 4 > /dev/null
python3 tools/summarize_sq.py r05_encode_2M_60x gpurun_out/prof_r05e60_sqa gpurun_out/prof_r05e60_sqb gpurun_out/prof_r05e60_clk > /dev/null
fi
cp profiles/r05_* profiles/roofline_traffic.json gpurun_out/r05_profiles/ 2>/dev/null
if has lines; then
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_profiles/r05_default_line.json 2> gpurun_out/r05_default_line.err; echo default rc=$?
python bench.py --workload haplotype --steps 20 --warmup 5 > gpurun_out/r05_profiles/r05_haplotype_line.json 2> gpurun_out/r05_haplotype_line.err; echo hap rc=$?
python bench.py --workload two-stage --steps 3 --warmup 1 > gpurun_out/r05_profiles/r05_two_stage_line.json 2> gpurun_out/r05_two_stage_line.err; echo two rc=$?
python bench.py --workload deep60 --steps 8 --warmup 2 > gpurun_out/r05_profiles/r05_deep60_line.json 2> gpurun_out/r05_deep60_line.err; echo deep rc=$?
python bench.py --workload e2e --steps 8 --warmup 2 > gpurun_out/r05_profiles/r05_e2e_line.json 2> gpurun_out/r05_e2e_line.err; echo e2e rc=$?
python bench.py --workload hap-e2e --steps 8 --warmup 1 > gpurun_out/r05_profiles/r05_hap_e2e_line.json 2> gpurun_out/r05_hap_e2e_line.err; echo hape2e rc=$?
python bench.py --workload pd-e2e --steps 16 --warmup 1 > gpurun_out/r05_profiles/r05_pd_e2e_line.json 2> gpurun_out/r05_pd_e2e_line.err; echo pde2e rc=$?
fi
rm -rf gpurun_out/prof_r05*           # raw rocprof output stays on the box (the summaries travel)
ls gpurun_out/r05_profiles
