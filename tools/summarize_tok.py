#!/usr/bin/env python3
"""profiles/<tag>_tokenise.json from rocprofv3 passes of tools/probes/tok_probe.py (stats, FETCH_SIZE, WRITE_SIZE: tools/prof_cmd.sh):
per kernel of nsnp_mpileup_tokenise the average duration and the HBM bytes per launch (counters in KiB, FETCH x2 on gfx950:
tools/summarize_prof.py), and the call's roofline against its ALGORITHMIC bytes (the text read once + column-5 bytes + 17 B per line).

    python3 tools/summarize_tok.py r06 gpurun_out/prof_r06tok_stats gpurun_out/prof_r06tok_fetch gpurun_out/prof_r06tok_write TEXT_BYTES LINES BASES_BYTES"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def short(n):
    m = re.search(r"(k_tok_[a-z0-9_]+)", n)
    return m.group(1) if m else None


tag, d_stats, d_fetch, d_write = sys.argv[1:5]
text_bytes, lines, bases_bytes = (int(v) for v in sys.argv[5:8])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(one(d_stats, "*kernel_trace.csv"))):
    k = short(r["Kernel_Name"])
    if k:
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
for d, name in ((d_fetch, "FETCH_SIZE"), (d_write, "WRITE_SIZE")):
    f = one(d, "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == name:
            cnt[name][k].append(float(r["Counter_Value"]))
kern = {}
tot_ns = tot_hbm = 0.0
for k, v in dur.items():
    v = v[len(v) // 4:]                                   # (the first calls warm the caches and the clocks)
    avg = sum(v) / len(v)
    fe = cnt["FETCH_SIZE"].get(k); wr = cnt["WRITE_SIZE"].get(k)
    fetch = 2.0 * 1024 * sum(fe) / len(fe) if fe else None
    write = 1024 * sum(wr) / len(wr) if wr else None
    kern[k] = {"avg_us": round(avg / 1e3, 2), "launches": len(v), "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write}
    tot_ns += avg
    tot_hbm += (fetch or 0) + (write or 0)
alg = text_bytes + bases_bytes + 17 * lines
out = {"what": "nsnp_mpileup_tokenise alone on one chunk of synthetic 30x mpileup text (tools/probes/tok_probe.py), rocprofv3 kernel trace + separate --pmc FETCH_SIZE / "
               "WRITE_SIZE passes (KiB units, FETCH x2: MI355X_MICROARCH.md)",
       "text_bytes": text_bytes, "lines": lines, "column5_bytes": bases_bytes, "algorithmic_bytes_per_call": alg,
       "kernels": kern, "sum_of_kernel_us": round(tot_ns / 1e3, 2), "hbm_bytes_per_call": tot_hbm or None,
       "roofline": {"bound": "hbm", "achieved": alg / tot_ns, "peak": 8000.0, "unit": "GB/s", "frac": alg / tot_ns / 8000.0,
                    "traffic": tot_hbm or None, "traffic_over_algorithmic": round(tot_hbm / alg, 3) if tot_hbm else None}}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_tokenise.json"), "w"), indent=1)
print(json.dumps(out["roofline"]), json.dumps({k: v["avg_us"] for k, v in kern.items()}))
