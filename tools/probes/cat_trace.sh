#!/bin/bash
# per-launch durations of the CatModel convolution launches (kernel trace of tools/probes/cat_probe.py)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_cattrace -o run -- python3 tools/probes/cat_probe.py 4096 2 ${1:-0} > gpurun_out/cattrace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_cattrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_cat_conv" in r["Kernel_Name"] or ("k_hap_gemm" in r["Kernel_Name"] and "true" in r["Kernel_Name"].split("(")[0][-8:])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.OrderedDict()
for r in rows:
    key = (int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]))
    by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in by.items():
    print("grid %6d x %d: %3d launches, avg %8.1f us" % (k[0], k[1], len(v), sum(v) / len(v))); tot += sum(v)
print("total conv us per pass:", tot / (len(rows) / 12))
PY
