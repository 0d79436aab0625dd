python -m pytest tests/test_gpu_pileup_forward.py -x -q -k "bf16x3" 2>&1 | tail -3
for n in 131072 4096; do
 for sg in "2 4" "2 2" "1 2"; do set -- $sg; L0SG=$1 L1SG=$2 python tools/fwd_probe.py $n 2 5 2>&1 | tail -1; done
 L0RS=0 L0SG=2 L1SG=4 python tools/fwd_probe.py $n 2 5 2>&1 | tail -1
done
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity-sample 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fp32', d['value'], 'f16x3', d['f16x3']['value'])
b=d['bf16x3']; print('bf16x3', b['value'], b['max_abs_dp_vs_fp32_on_the_pool'], b['roofline']['frac'], b['roofline']['kernel'], b['kernel_exclusive_ms'])
"
