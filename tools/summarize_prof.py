#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/prof_*) into the small tracked summaries under
profiles/: the --stats kernel table, and per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE
PMC passes with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): counters are in KiB,
FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x, WRITE_SIZE is exact.

    python tools/summarize_prof.py <round-tag> <stats_dir> [<fetch_dir> <write_dir>] [--workload pileup] [--batch 4096] [--precision 0]
                                   [--enc-group 8] [--D 90] [--timed SKIP TAKE]

Traffic of every workload is kept in profiles/roofline_traffic.json under "workloads" (tools/bench_common.py committed_traffic reads it).

--timed SKIP TAKE: bench.py launches every kernel W x batches_per_step times before the timed region, K x batches_per_step times
inside it and 36 more times alone afterwards (the `exclusive` figures); the average over launches SKIP .. SKIP+TAKE of each
kernel (by start time, from the kernel trace of the stats pass) is the figure that corresponds to `roofline.avg_launch_ms`.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"k_pileup_l0_b3": "pileup_l0_bf16x3", "k_pileup_l1_b3": "pileup_l1f_bf16x3", "k_pileup_head_b3": "pileup_head_bf16x3",
         "k_hap_features_L33": "hap_features", "k_hap_features_L11": "hap_features_L11",
         "k_hap_gemm<0,0,false>": "hap_gemm_lstm", "k_hap_gemm<0,1,false>": "hap_gemm_lstm_f16x3", "k_hap_gemm<0,2,false>": "hap_gemm_lstm_bf16x3",
         "k_hap_gemm<3,0,true>": "cat_conv", "k_hap_gemm<3,1,true>": "cat_conv_f16x3", "k_hap_gemm<3,2,true>": "cat_conv_bf16x3",
         "k_cat_conv<0": "cat_conv", "k_cat_conv<1": "cat_conv_f16x3", "k_cat_conv<2": "cat_conv_bf16x3",
         "k_pileup_l1_rs4": "pileup_l1f", "k_pileup_l1_rs": "pileup_l1f", "k_pileup_head_rs": "pileup_head", "k_pileup_l0_rs32": "pileup_l0", "k_pileup_l1f": "pileup_l1f", "k_pileup_l0": "pileup_l0", "k_pileup_proj1": "pileup_proj1", "k_pileup_l1": "pileup_l1",
         "k_pileup_head": "pileup_head", "k_encode_columns": "encode_columns", "k_hap_features": "hap_features",
         "k_pileup_post": "pileup_post", "k_select": "select_sites", "k_gather_windows": "gather_windows",
         "k_hap_gemm<0,false,false>": "hap_gemm_lstm", "k_hap_gemm<0,true,false>": "hap_gemm_lstm_f16x3",
         "k_hap_gemm<3,false,true>": "cat_conv", "k_hap_gemm<3,true,true>": "cat_conv_f16x3", "k_hap_gemm<1": "hap_gemm_linear",
         "k_hap_gemm<2": "hap_gemm_linear_tanh", "k_hap_pack_input": "hap_pack_input", "k_hap_heads": "hap_heads", "k_hap_": "hap_other"}


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>(]*>)?", name)
    if not m:
        return name[:60].replace(",", ";")
    t = (m.group(2) or "").replace(" ", "")
    return m.group(1) + (t if m.group(1) in ("k_hap_gemm", "k_cat_conv") else "")


# FETCH_SIZE correction (MI355X_MICROARCH.md, HBM: 16 B/lane streaming reads are counted at exactly 1/2; "other access widths are
# uncalibrated: calibrate on a known byte count in your own access pattern").  tools/probes/fetch_calib.sh measured this repository's shapes
# (profiles/r04_fetch_calibration.json): 16 B/lane and 4 B/lane contiguous streams both read 2.00; rows of 33 int32 read by a wave each
# (lanes 33..63 idle) 1.65 in the stand-alone probe.  k_hap_features itself - four planes, four loads in flight per lane, waves of one
# workgroup on neighbouring rows - moves 778.6 MB of read planes per 16384-site launch at L = 33 and is counted at 395.7 MB: 1.97, i.e.
# the x2 of the streaming case, not the probe's 1.65 (every line is fetched once while its neighbours are in flight).  So x2 is applied
# to every kernel, and each entry records the factor its algorithmic bytes imply where they are known.
def fetch_factor(kernel):
    return 2.0, "x2 (gfx950 counts streaming reads at 1/2; calibration: profiles/r04_fetch_calibration.json)"


def one(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def main():
    argv = list(sys.argv[1:])
    batch, precision, timed, workload, enc_group, Dd = 4096, 0, None, "pileup", 8, 90
    if "--workload" in argv:
        i = argv.index("--workload"); workload = argv[i + 1]; del argv[i:i + 2]
    if "--enc-group" in argv:
        i = argv.index("--enc-group"); enc_group = int(argv[i + 1]); del argv[i:i + 2]
    if "--D" in argv:
        i = argv.index("--D"); Dd = int(argv[i + 1]); del argv[i:i + 2]
    if "--batch" in argv:
        i = argv.index("--batch"); batch = int(argv[i + 1]); del argv[i:i + 2]
    if "--precision" in argv:
        i = argv.index("--precision"); precision = int(argv[i + 1]); del argv[i:i + 2]
    if "--timed" in argv:
        i = argv.index("--timed"); timed = (int(argv[i + 1]), int(argv[i + 2])); del argv[i:i + 3]
    args = argv
    tag, stats_dir = args[0], args[1]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(one(stats_dir, "*kernel_stats.csv"))))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
        for r in rows:
            f.write(f"{short(r['Name']).replace(',', ';')},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.1f},"
                    f"{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
    if timed:
        tr = collections.defaultdict(list)
        for r in csv.DictReader(open(one(stats_dir, "*kernel_trace.csv"))):
            tr[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "a") as f:
            f.write(f"# launches {timed[0]} .. {timed[0] + timed[1]} of each kernel by start time = bench.py's timed region (kernel,launches,avg_ns); "
                    f"the launches behind it run alone (bench.py `exclusive`)\n")
            n_max = max((len(v) for k, v in tr.items() if k.startswith("k_")), default=1)
            for k, v in sorted(tr.items()):
                if not k.startswith("k_"):
                    continue
                v.sort()
                # a kernel launched once per GROUP of batches (the column encode: 8 batches per launch) has proportionally fewer
                # launches in every phase of the run: scale the window by its launch count
                sc = len(v) / n_max if len(v) * 2 < n_max else 1.0
                t0, t1 = int(round(timed[0] * sc)), int(round((timed[0] + timed[1]) * sc))
                w = v[t0:t1]
                tail = v[t1:]
                if w:
                    f.write(f"timed_region,{k.replace(',', ';')},{len(w)},{sum(e - s for s, e in w) / len(w):.1f}\n")
                if tail:
                    f.write(f"alone_after,{k.replace(',', ';')},{len(tail)},{sum(e - s for s, e in tail) / len(tail):.1f}\n")
    # the tile GEMM is launched with different grids by different callers (HaplotypeModel passes, the CatModel's LSTMs, ragged last
    # batches): per-grid averages, so that a bench line's per-launch figure can be compared with launches of the same shape
    tp = one(stats_dir, "*kernel_trace.csv")
    if tp:
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(tp)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_hap_gemm"):
                g = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
                by[(k, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if by:
            with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "a") as f:
                f.write("# tile GEMM launches by grid (site tiles x row tiles x slices): by_grid,kernel,grid,launches,avg_ns,total_ns\n")
                for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                    f.write(f"by_grid,{k.replace(',', ';')},{g[0]}x{g[1]}x{g[2]},{len(v)},{sum(v) / len(v):.1f},{sum(v)}\n")
    print(open(os.path.join(out_dir, f"{tag}_kernel_stats.csv")).read())
    if len(args) >= 4:
        traffic = collections.defaultdict(dict)
        for cname, d in (("FETCH_SIZE", args[2]), ("WRITE_SIZE", args[3])):
            acc = collections.defaultdict(list)
            grid = collections.defaultdict(int)
            rows_c = list(csv.DictReader(open(one(d, "*counter_collection.csv"))))
            # k_hap_features is launched in two shapes (window length 33 and 11) that move 3x different bytes; same grid, same LDS:
            # told apart by the counter itself (two clusters a factor of three apart)
            fv = [float(r["Counter_Value"]) for r in rows_c if r["Counter_Name"] == cname and short(r["Kernel_Name"]) == "k_hap_features"]
            f_cut = (min(fv) + max(fv)) / 2 if fv and max(fv) > 1.8 * min(fv) else -1.0
            for r in rows_c:
                if r["Counter_Name"] == cname:
                    k = short(r["Kernel_Name"])
                    if k == "k_hap_features":
                        k += "_L33" if float(r["Counter_Value"]) > f_cut else "_L11"
                    acc[k].append(float(r["Counter_Value"]))
                    grid[k] += int(r["Grid_Size"])
            for k, v in acc.items():
                if k.startswith("k_"):
                    traffic[k][cname] = sum(v) / len(v)
                    traffic[k][cname + "_per_thread"] = sum(v) / max(grid[k], 1)      # (the column encode: one thread = one column)
                    traffic[k]["launches"] = len(v)
        summary = {"batch": batch, "precision": precision, "enc_group": enc_group, "D": Dd, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes",
                   "correction": "KiB units; FETCH_SIZE x2 (gfx950 wide-read under-count) except where a kernel's entry names a calibrated factor, WRITE_SIZE x1",
                   "kernels": {}}
        for k, v in traffic.items():
            ff, fhow = fetch_factor(k)
            rd = v.get("FETCH_SIZE", 0.0) * 1024 * ff
            wr = v.get("WRITE_SIZE", 0.0) * 1024
            name = next((n for p, n in NAMES.items() if k.startswith(p)), k)
            summary["kernels"][name] = {"fetch_kib_raw": v.get("FETCH_SIZE"), "write_kib_raw": v.get("WRITE_SIZE"), "fetch_factor": ff, "fetch_factor_source": fhow,
                                        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                                        "hbm_bytes_per_launch": rd + wr, "launches": v.get("launches")}
            if name == "encode_columns":      # launches of different sizes (groups of batches): bytes per column travel with it
                summary["kernels"][name]["hbm_bytes_per_column"] = v.get("FETCH_SIZE_per_thread", 0.0) * 1024 * ff + v.get("WRITE_SIZE_per_thread", 0.0) * 1024
        json.dump(summary, open(os.path.join(out_dir, f"{tag}_pmc_traffic.json"), "w"), indent=1)
        rt = os.path.join(out_dir, "roofline_traffic.json")
        allw = {"workloads": {}}
        if os.path.exists(rt):
            try:
                old = json.load(open(rt))
                allw = old if "workloads" in old else {"workloads": {}}
            except Exception:
                pass
        allw["workloads"][workload] = summary
        json.dump(allw, open(rt, "w"), indent=1)
        print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
