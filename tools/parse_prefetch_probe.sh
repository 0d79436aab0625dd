python - <<'PY'
from nanosnp_amd import host
cols = host.synth_columns(5, 1500000, coverage=30)
open('/tmp/nsnp_parse_probe.mpileup','wb').write(memoryview(cols.mpileup_text_native("chr20s")))
PY
for pf in 0 256 1024 4096; do gcc -O3 -std=gnu11 -fopenmp -Iinclude -DNSNP_TOK_PREFETCH=$pf -o /tmp/parse_probe tools/probes/parse_probe.c -lm && echo "prefetch $pf: $(/tmp/parse_probe)"; done
