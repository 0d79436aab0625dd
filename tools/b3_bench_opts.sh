for o in "" "--opt l1_site_groups=4" "--opt l1_site_groups=1" "--opt l0_site_groups=1" "--opt l0_site_groups=1 --opt l1_site_groups=4" "--streams 16" "--streams 64" "--batch 8192" "--batch 16384"; do
python bench.py --precision 2 --steps 4 --warmup 1 --repeat 1 --no-cpu-baseline --no-parity-sample --no-second-precision $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$o', round(d['value']/1e6,2), 'M sites/s', d['kernel_exclusive_ms'])
"
done
