# HaplotypeModel forward probe: fp32, bf16x3 (256 x 256 LSTM tiles) and bf16x3 with hap_b3x=0 (128 x 128 tiles); crc must agree between the last two
python tools/hap_probe.py 16384 0,2 2>&1 | grep hap_forward | cut -c1-200
python tools/hap_probe.py 16384 2 16384 hap_b3x=0 2>&1 | grep hap_forward | cut -c1-200
python tools/hap_probe.py 32768 2 16384,32768 2>&1 | grep hap_forward | cut -c1-200
python tools/hap_probe.py 3000 2 16384 2>&1 | grep hap_forward | cut -c1-200
python tools/hap_probe.py 3000 2 16384 hap_b3x=0 2>&1 | grep hap_forward | cut -c1-200
