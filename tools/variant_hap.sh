# HaplotypeModel forward probe: the wide-tile LSTM step kernels (fp32 128 x 256, bf16x3 256 x 256) against the 128 x 128 kernel; crc must agree per precision
python tools/hap_probe.py 16384 0,2 2>&1 | grep hap_forward | cut -c1-200
python tools/hap_probe.py 16384 0,2 16384 hap_b3x=0 2>&1 | grep hap_forward | cut -c1-200
python tools/hap_probe.py 32768 0 16384,32768 2>&1 | grep hap_forward | cut -c1-200
python tools/hap_probe.py 3000 0 16384 2>&1 | grep hap_forward | cut -c1-200
python tools/hap_probe.py 3000 0 16384 2>&1 | grep hap_forward | cut -c1-200
