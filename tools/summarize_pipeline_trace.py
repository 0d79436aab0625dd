#!/usr/bin/env python3
"""Copy-engine / compute overlap of a host-fed pipeline from a rocprofv3 trace of the bench run itself.

    rocprofv3 --kernel-trace --memory-copy-trace --stats -d DIR -o run -- python3 bench.py --workload e2e ...   (NSNP_TRACE_MARK=mark.json set)
    python3 tools/summarize_pipeline_trace.py DIR mark.json details.json OUT.json [kernel_stats_out.csv]

The bench tool wrote the host clocks at both ends of its timed region into mark.json (tools/bench_common.py::mark_region); the trace's
records are cut to that region (the clock rocprofv3 stamps with is found by looking which of the recorded clocks covers the traced
dispatches) and summed up per step: time with a kernel running (union of the dispatch intervals), time with a copy running (per
direction), both at once, neither; per-kernel totals.  `details.json` is the bench line's full result object: its stage_busy_s_per_step
(HIP-event arithmetic of the pipeline itself) is printed beside the trace's figures, with the relative difference - the numbers a bench
line claims for "device busy" and "H2D busy" are reproducible from profiles/ this way ("Next round" item 4)."""
import csv
import glob
import json
import os
import re
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def main():
    d, mark_p, details_p, out_p = sys.argv[1:5]
    mark = json.load(open(mark_p))
    K = int(mark["steps"])
    kt = list(csv.DictReader(open(find(d, "*kernel_trace.csv"))))
    mf = find(d, "*memory_copy_trace.csv")
    mt = list(csv.DictReader(open(mf))) if mf else []
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in kt]
    lo_all, hi_all = min(k[0] for k in ks), max(k[1] for k in ks)
    clock = None
    for name, b in mark["begin"].items():                      # the clock whose region lies inside the traced span
        e = mark["end"][name]
        if lo_all <= b and e <= hi_all + 5_000_000_000 and b < hi_all:
            clock = name
            break
    if clock is None:
        sys.exit(f"no recorded clock covers the trace ({lo_all}..{hi_all}; {mark['begin']})")
    t0, t1 = mark["begin"][clock], mark["end"][clock]
    ks = [k for k in ks if k[0] >= t0 and k[1] <= t1]
    cps = []
    for r in mt:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if a >= t0 and b <= t1:
            cps.append((a, b, r.get("Direction", r.get("Kind", ""))))
    kern_u = union([(a, b) for a, b, _ in ks])
    h2d_u = union([(a, b) for a, b, dr in cps if "HOST_TO_DEVICE" in dr.upper() or "H2D" in dr.upper()])
    d2h_u = union([(a, b) for a, b, dr in cps if "DEVICE_TO_HOST" in dr.upper() or "D2H" in dr.upper()])
    copy_u = union([(a, b) for a, b, _ in cps])
    any_u = union([(a, b) for a, b, _ in ks] + [(a, b) for a, b, _ in cps])
    wall = t1 - t0
    short = lambda n: (re.search(r"(k_[A-Za-z0-9_]+)", n) or re.search(r"([A-Za-z0-9_]+)\(", n) or re.search(r"(\S+)", n)).group(1)
    per_k = {}
    for a, b, n in ks:
        e = per_k.setdefault(short(n), [0, 0])
        e[0] += 1; e[1] += b - a
    gaps = sorted((any_u[i + 1][0] - any_u[i][1] for i in range(len(any_u) - 1)), reverse=True)
    details = json.load(open(details_p)) if os.path.exists(details_p) else {}
    ms = lambda ns: round(ns / K / 1e6, 4)
    out = {
        "what": "rocprofv3 --kernel-trace --memory-copy-trace of the bench run itself, cut to the timed region the bench tool marked (%s); per step = / %d steps" % (clock, K),
        "workload": mark.get("workload"), "steps": K, "region_ms": round(wall / 1e6, 3), "ms_per_step": ms(wall),
        "per_step_ms": {"kernel_busy": ms(total(kern_u)), "h2d_busy": ms(total(h2d_u)), "d2h_busy": ms(total(d2h_u)), "any_copy_busy": ms(total(copy_u)),
                        "kernel_and_copy_at_once": ms(total(intersect(kern_u, copy_u))), "nothing_running": ms(wall - total(any_u))},
        "fractions_of_the_region": {"kernel_busy": round(total(kern_u) / wall, 4), "copy_busy": round(total(copy_u) / wall, 4),
                                    "nothing_running": round(1 - total(any_u) / wall, 4)},
        "dispatches_in_region": len(ks), "copies_in_region": len(cps) if mf else "not traced (kernel trace only)",
        "largest_idle_gaps_ms": [round(g / 1e6, 3) for g in gaps[:5]],
        "kernels": {k: {"calls": v[0], "total_ms_per_step": ms(v[1]), "avg_us": round(v[1] / v[0] / 1e3, 2)} for k, v in sorted(per_k.items(), key=lambda kv: -kv[1][1])},
    }
    sb = details.get("stage_busy_s_per_step")
    if sb:
        dev_key = next((k for k in sb if k.startswith("device") or k == "gpu_s"), None)
        h2d_key = next((k for k in sb if k.startswith("H2D") or k == "h2d_s"), None)
        cmp_ = {}
        if dev_key:
            cmp_["device_busy_ms_per_step"] = {"line": round(sb[dev_key] * 1e3, 4), "trace_kernel_busy": out["per_step_ms"]["kernel_busy"],
                                               "relative_difference": round(sb[dev_key] * 1e3 / max(out["per_step_ms"]["kernel_busy"], 1e-9) - 1, 4)}
        if h2d_key and cps:
            cmp_["h2d_busy_ms_per_step"] = {"line": round(sb[h2d_key] * 1e3, 4), "trace_h2d_busy": out["per_step_ms"]["h2d_busy"],
                                            "relative_difference": round(sb[h2d_key] * 1e3 / max(out["per_step_ms"]["h2d_busy"], 1e-9) - 1, 4)}
        cmp_["note"] = ("the line's device figure is HIP-event time between the first and last launch of every pass on the compute stream (it includes the "
                        "gaps between a pass's kernels), the trace's is the union of the dispatch intervals")
        out["line_against_trace"] = cmp_
        out["line_value"] = {"value": details.get("value"), "ms_per_step": details.get("ms_per_step"), "bound_by": details.get("bound_by")}
    json.dump(out, open(out_p, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("workload", "ms_per_step", "per_step_ms", "fractions_of_the_region", "largest_idle_gaps_ms")}))
    if "line_against_trace" in out:
        print(json.dumps(out["line_against_trace"]))
    if len(sys.argv) > 5:
        st = find(d, "*kernel_stats.csv")
        if st:
            open(sys.argv[5], "w").write(open(st).read())


if __name__ == "__main__":
    main()
