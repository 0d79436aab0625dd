#!/bin/bash
# tools/probes/fetch_calib.sh -- on the GPU box: FETCH_SIZE calibration for the access shapes of the HBM-bound kernels -> profiles/r04_fetch_calibration.json
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out build_tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o build_tmp/fetch_calib || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch_calib -o run -- ./build_tmp/fetch_calib ${1:-2} > gpurun_out/fetch_calib.log 2> gpurun_out/fetch_calib.err
echo "rc=$?"
python3 - <<'PY'
import csv, glob, json, collections, re
log = json.loads([l for l in open("gpurun_out/fetch_calib.log") if l.startswith("{")][-1])
n = log["bytes_streamed_per_launch"]
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_fetch_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            m = re.search(r"k_calib_\w+(<[^>]*>)?", r["Kernel_Name"])
            if m: acc[m.group(0)].append(float(r["Counter_Value"]))
out = {"source": "tools/probes/fetch_calib.sh: rocprofv3 --pmc FETCH_SIZE around tools/probes/fetch_calib.hip; every kernel streams the same buffer once",
       "bytes_streamed_per_launch": n, "shapes": {}}
for k, v in sorted(acc.items()):
    v = v[1:] if len(v) > 1 else v                      # (the first launch of a shape may find parts of the memset's lines on-die)
    kib = sum(v) / len(v)
    out["shapes"][k] = {"fetch_size_kib": kib, "launches": len(v), "factor_bytes_per_counted_byte": n / (kib * 1024.0)}
json.dump(out, open("profiles/r04_fetch_calibration.json", "w"), indent=1)
json.dump(out, open("gpurun_out/r04_fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
