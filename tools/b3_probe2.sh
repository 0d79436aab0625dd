python -m pytest tests/test_gpu_pileup_forward.py -x -q -k "bf16x3" 2>&1 | tail -3
for n in 131072 4096; do
 for sg in "1 4" "2 4" "4 4" "2 2"; do set -- $sg; L0SG=$1 L1SG=$2 python tools/fwd_probe.py $n 2 5 2>&1 | tail -1; done
done
