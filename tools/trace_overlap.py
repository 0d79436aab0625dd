#!/usr/bin/env python3
"""Concurrency picture of a rocprofv3 kernel trace (bench.py, several streams): wall time of the busiest window, per-kernel
busy time, how much of the wall time 0 / 1 / 2 / ... kernels are in flight, and the gaps in front of each kernel type.
    python tools/trace_overlap.py gpurun_out/prof_<tag>_stats [skip_first_fraction]"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_" in r["Kernel_Name"]]
short = lambda n: re.search(r"(k_[a-z0-9_]+)", n).group(1)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
t0 = ev[int(len(ev) * float(sys.argv[2]) if len(sys.argv) > 2 else 0)][0]
ev = [e for e in ev if e[0] >= t0]
wall = max(e[1] for e in ev) - t0
busy = collections.Counter(); cnt = collections.Counter()
for s, e, k, q in ev: busy[k] += e - s; cnt[k] += 1
print(f"window {wall/1e6:.3f} ms, {len(ev)} dispatches, queues {sorted(set(e[3] for e in ev))}")
for k in busy: print(f"  {k:24s} n={cnt[k]:5d} avg {busy[k]/cnt[k]/1e3:8.1f} us  sum/wall {busy[k]/wall:5.2f}")
pts = sorted([(s, 1) for s, e, k, q in ev] + [(e, -1) for s, e, k, q in ev])
hist = collections.Counter(); cur = 0; last = t0
for t, dlt in pts:
    hist[cur] += t - last; last = t; cur += dlt
print("  kernels in flight -> share of wall:", {k: round(v / wall, 3) for k, v in sorted(hist.items())})
