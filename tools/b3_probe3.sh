# after a change of the shared LSTM cell / the bf16x3 layer-0 kernel: parity tests of the three forwards, then the timing probes
python -m pytest tests/test_gpu_pileup_forward.py tests/test_gpu_hap.py tests/test_gpu_cat.py tests/test_gpu_stage_parity.py -x -q 2>&1 | tail -4
for p in 0 2; do python tools/fwd_probe.py 131072 $p 5 2>&1 | tail -1; done
python tools/hap_probe.py 16384 0,2 2>&1 | tail -3
python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fp32', d['value'], d['roofline']['frac'], d.get('kernel_exclusive_ms'), 'f16x3', d['f16x3']['value'])
b=d['bf16x3']; print('bf16x3', b['value'], b['max_abs_dp_vs_fp32_on_the_pool'], b['parity_sample']['ok'], b['roofline']['frac'], b['kernel_exclusive_ms'])
"
