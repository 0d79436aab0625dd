# all four bench workloads (reduced sizes unless FULL=1) -> gpurun_out/r04_<wl>_line.json
mkdir -p gpurun_out
python tools/hap_probe.py 16384 0,2,1 2>&1 | tail -3
if [ "${FULL:-0}" = "1" ]; then
  python bench.py --steps 20 --warmup 5 > gpurun_out/r04_pileup_line.json 2> gpurun_out/r04_pileup_line.err; echo pileup rc=$?
  python bench.py --workload haplotype --steps 20 --warmup 5 > gpurun_out/r04_haplotype_line.json 2> gpurun_out/r04_haplotype_line.err; echo hap rc=$?
  python bench.py --workload two-stage --steps 3 --warmup 1 > gpurun_out/r04_two_stage_line.json 2> gpurun_out/r04_two_stage_line.err; echo two rc=$?
  python bench.py --workload deep60 --steps 8 --warmup 2 > gpurun_out/r04_deep60_line.json 2> gpurun_out/r04_deep60_line.err; echo deep rc=$?
else
  python -m pytest tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -8
fi
