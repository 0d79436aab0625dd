python -m pytest tests/test_gpu_pileup_forward.py -x -q -k "bf16x3" 2>&1 | tail -5
for n in 131072 4096; do
 for p in 0 2; do python tools/fwd_probe.py $n $p 5 2>&1 | tail -1; done
 for sg in "1 1" "2 2" "4 4" "1 4" "2 4"; do set -- $sg; L0SG=$1 L1SG=$2 python tools/fwd_probe.py $n 2 5 2>&1 | tail -1; done
done
python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fp32', d['value'], 'f16x3', d['f16x3']['value'], d['f16x3']['max_abs_dp_vs_fp32_on_the_pool'])
b=d['bf16x3']; print('bf16x3', b['value'], b['max_abs_dp_vs_fp32_on_the_pool'], b['parity_sample']['ok'], b['parity_sample']['max_abs_dp'], b['roofline']['frac'], b['roofline']['kernel'], b['kernel_exclusive_ms'])
"
