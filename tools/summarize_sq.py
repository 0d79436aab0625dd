#!/usr/bin/env python3
"""SQ / GRBM counter passes of tools/prof_run.sh -> profiles/<tag>_sq_counters.json: per kernel, the per-launch average of every
counter plus the derived figures the recurrence kernels are judged by (MI355X_MICROARCH.md, 'rocprofv3 PMC slots' and
'Two waves per SIMD' item 9):

    mfma_busy_frac      SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs ... see `units`)  -- matrix-pipe utilisation
    coexec_frac         SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES                  -- VALU issue beside the matrix pipe
    wait_any_frac       SQ_WAIT_ANY / SQ_WAVE_CYCLES          wait_inst_frac  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
    lds_conflict_frac   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    clock_ghz           GRBM_GUI_ACTIVE / 8 XCDs / kernel duration

    python tools/summarize_sq.py <tag> gpurun_out/prof_<tag>_sqa gpurun_out/prof_<tag>_sqb [gpurun_out/prof_<tag>_clk]
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    meta = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k:
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
                meta[k] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                           "lds_bytes": int(r["LDS_Block_Size"]), "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
    out = {"source": "rocprofv3 --pmc passes of tools/prof_run.sh (one counter set per pass, kernels serialised by the profiler)",
           "units": "SQ_*_CYCLES of waves / waits / active instructions are per-wave quad-cycles summed over all waves; "
                    "SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles (MI355X_MICROARCH.md cycle-constants table)",
           "kernels": {}}
    for k, cs in sorted(acc.items()):
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        e = {"launches": max(len(v) for v in cs.values()), **meta[k], "counters_per_launch": {c: round(v, 1) for c, v in sorted(avg.items())}}
        g = avg.get
        der = {}
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_VALU_MFMA_COEXEC_CYCLES") is not None:
            der["coexec_frac"] = g("SQ_VALU_MFMA_COEXEC_CYCLES") / g("SQ_VALU_MFMA_BUSY_CYCLES")
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES: cycles an SQ (one per CU... reported summed over SEs) has work; 4 SIMDs share it
            der["mfma_busy_per_sq_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")
        if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU") is not None:
            der["valu_active_frac"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
        if g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
            der["wave_quad_cycles_per_wave"] = g("SQ_WAVE_CYCLES") / g("SQ_WAVES")
        # waits are collected in the pass without SQ_WAVE_CYCLES: normalise by SQ_ACTIVE_INST_ANY + waits (disjoint buckets)
        if g("SQ_WAIT_ANY") is not None and g("SQ_WAIT_INST_ANY") is not None and g("SQ_ACTIVE_INST_ANY") is not None:
            tot = g("SQ_WAIT_ANY") + g("SQ_WAIT_INST_ANY") + g("SQ_ACTIVE_INST_ANY")
            if tot:
                der["wait_any_frac"] = g("SQ_WAIT_ANY") / tot
                der["wait_inst_frac"] = g("SQ_WAIT_INST_ANY") / tot
                der["active_inst_frac"] = g("SQ_ACTIVE_INST_ANY") / tot
        if g("SQ_LDS_IDX_ACTIVE"):
            der["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT", 0.0) / g("SQ_LDS_IDX_ACTIVE")
        if dur.get(k) and g("GRBM_GUI_ACTIVE"):
            d = sum(dur[k]) / len(dur[k])
            der["kernel_s_under_profiler"] = d
            der["clock_ghz"] = g("GRBM_GUI_ACTIVE") / 8 / d / 1e9
        e["derived"] = {n: (round(v, 4) if v is not None else None) for n, v in der.items() if v is not None}
        out["kernels"][k] = e
    p = os.path.join(ROOT, "profiles", f"{tag}_sq_counters.json")
    json.dump(out, open(p, "w"), indent=1)
    for k, e in out["kernels"].items():
        print(k, e["derived"])


if __name__ == "__main__":
    main()
