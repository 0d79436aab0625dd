#!/usr/bin/env python3
"""gpurun_out/host_scaling/*.json (tools/host_scaling.sh) -> a markdown table + profiles/<tag>_host_scaling.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
d = os.path.join(ROOT, "gpurun_out", "host_scaling")
out = {"what": "ranks SHARE one GPU (gloo collectives): device time is serialised, value is not a scaling number; host seconds per step and rank are",
       "rows": []}
print("| workload | ranks | value (sites/s, TEST configuration) | wall ms/step | host work per rank: max (s/step) | wait for host work: max | device busy per rank: max | formatting (rank 0) |")
print("|---|---|---|---|---|---|---|---|")
for wl, hostk, waitk, fmtk in (("e2e", "parse_s", "wait_parse_s", "vcf_s"), ("hap-e2e", "stage_s", "wait_stage_s", "csv_s"), ("pd-e2e", "stage_s", "wait_stage_s", "vcf_s")):
    for n in (1, 2, 4, 8):
        f = os.path.join(d, f"{wl}_{n}.json")
        try:
            line = json.load(open(f))
        except Exception:
            continue
        pr = line.get("per_rank_s_per_step")
        if pr is None:
            sb = line["stage_busy_s_per_step"]
            vals = list(sb.values())
            mt = line["main_thread_s_per_step"]
            pr = [{hostk: vals[0], waitk: mt.get(waitk, 0.0), "gpu_s": vals[2], fmtk: vals[3], "rank": 0}]
        row = {"workload": wl, "ranks": n, "value": line["value"], "ms_per_step": line["ms_per_step"], "usable_cores": line.get("usable_cores"),
               "host_s_max": max(r[hostk] for r in pr), "wait_s_max": max(r[waitk] for r in pr), "gpu_s_max": max(r["gpu_s"] for r in pr),
               "format_s_rank0": pr[0][fmtk], "per_rank": pr}
        out["rows"].append(row)
        print(f"| {wl} | {n} | {row['value']:.3g} | {row['ms_per_step']:.1f} | {row['host_s_max']:.4f} | {row['wait_s_max']:.4f} | {row['gpu_s_max']:.4f} | {row['format_s_rank0']:.4f} |")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_host_scaling.json"), "w"), indent=1)
