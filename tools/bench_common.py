"""Shared pieces of the bench workloads (bench.py, tools/hap_bench.py, tools/two_stage_bench.py): chip peaks, roofline objects,
the work-per-unit tables of DESIGN.md section 4, CPU-baseline helpers.

Every fraction printed by a bench line is PHYSICAL: executed flops (or algorithmic bytes, which the HBM kernels move one to
one) divided by a measured time and the chip peak, so it can never exceed 1.  The flop count of the reference's own schedule
(12.55 MFLOP/site for the PileupModel, 353.7 MFLOP/site for the HaplotypeModel), which the kernels
legitimately shorten (only the centre position is consumed: PileupModel/model.py:68, HaplotypeModel/model_dev.py:139-140),
rides along as the labelled `achieved_algorithmic` and is never divided by the peak."""
from __future__ import annotations

import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak at 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak (the f16x3 paths issue 3 fp16 MFMAs per fp32 product)
PEAK_HBM_GBS = 8000.0

# ---- PileupModel: work per site -----------------------------------------------------------------------------------
# reference schedule
PILEUP_ALG_FLOP = {
    "pileup_l0": 2 * 1_385_472,      # layer-0 BiLSTM, 33 steps x 2 directions (model.py:34-35)
    "pileup_proj1": 2 * 2_162_688,   # layer-1 input GEMMs, 33 steps x 2 directions
    "pileup_l1": 2 * 1_081_344,      # layer-1 recurrent GEMMs
    "pileup_head": 2 * 1_645_056,    # output_proj + dense on 33 positions + 4 heads (model.py:37,67-72)
}
PILEUP_ALG_FLOP["pileup_l1f"] = PILEUP_ALG_FLOP["pileup_proj1"] + PILEUP_ALG_FLOP["pileup_l1"]   # fused kernel
PILEUP_ALG_FLOP_FORWARD = 2 * 6_274_560                                                           # 12.55 MFLOP/site
assert sum(v for k, v in PILEUP_ALG_FLOP.items() if k != "pileup_l1f") == PILEUP_ALG_FLOP_FORWARD
# what the kernels execute (exact reduced schedule: layer 1 only on the 17 steps per direction that reach position 16,
# output_proj / dense / heads only at position 16 -- model.py:68; layer-0 K padded 18 -> 20 incl. the bias column)
PILEUP_EXEC_FLOP = {"pileup_l0": 2 * 33 * 256 * (20 + 64) * 2, "pileup_l1f": 2 * 17 * 256 * (128 + 64) * 2,
                    "pileup_proj1": 2 * 17 * 256 * 128 * 2, "pileup_l1": 2 * 17 * 256 * 64 * 2,
                    "pileup_head": (128 * 128 + 256 * 128 + 32 * 256) * 2}
PILEUP_EXEC_FLOP_FORWARD = PILEUP_EXEC_FLOP["pileup_l0"] + PILEUP_EXEC_FLOP["pileup_l1f"] + PILEUP_EXEC_FLOP["pileup_head"]
# bf16x3 (pileup_forward_bf16x3.hip): bf16 MFMA flops as EXECUTED - six v_mfma_f32_16x16x32_bf16 per 32-deep K block of a product of
# two fp32 operands, three for the layer-0 input block (integer counts up to 256 are one bf16 term; K padded 18 + 1 -> 32)
PILEUP_EXEC_FLOP_BF16X3 = {"pileup_l0": 2 * 33 * 256 * (32 * 3 + 64 * 6) * 2, "pileup_l1f": 2 * 17 * 256 * (128 + 64) * 6 * 2,
                           "pileup_head": (128 * 128 + 256 * 128 + 32 * 256) * 6 * 2}


def pileup_exec_flop(kernel, precision):
    """MFMA flops a forward kernel executes per site in the given arithmetic (0 fp32, 1 f16x3: three fp16 MFMAs per product,
    2 bf16x3: six bf16 MFMAs per product)"""
    if precision == 2:
        return PILEUP_EXEC_FLOP_BF16X3[kernel]
    return PILEUP_EXEC_FLOP[kernel] * (3 if precision == 1 else 1)

# ---- HaplotypeModel (model_dev.LSTMNetwork, H = 256, F = 105, 3 layers, L = 33 / 11) ---------------------------------
HAP_ALG_FLOP = 353.7e6             # 176.8 M MAC per site as the reference computes it


def hap_exec_flop(F=105, H=256, Lp=33, Lh=11):
    """MFMA flops the fused step launches execute per site: K padded to 16-wide chunks (F = 105 -> 112), all steps in
    layers 0 / 1, only the steps that reach the centre position in layer 2 (17 of 33, 6 of 11), two directions."""
    kin0 = 16 * math.ceil(F / 16)
    per_step = lambda kin: 2 * (4 * H) * (kin + H) * 2          # two directions, flop = 2 MAC
    chain = lambda L: L * per_step(kin0) + L * per_step(2 * H) + (L // 2 + 1) * per_step(2 * H)
    return chain(Lp) + chain(Lh)


def hap_lstm_launches(Lp=33, Lh=11):
    """fused step launches per pass (both encoders share a launch while the short one is still running)"""
    return Lp + Lp + (Lp // 2 + 1)


# ---- legacy CatModel (model.CatModel: ResCRNN + percentage RNN) ----------------------------------------------------
CAT_CH = (10, 32, 64, 128, 128, 256, 256)
CAT_POOL = (2, 2, 0, 3, 0, 2)          # max-pool height after block i (crnn.py:118-190), 0 = none


def cat_conv_exec_flop(rows=40, L=11):
    """flops of the 12 implicit-GEMM convolution launches per site as executed: K = 9 x C_in padded to 16-channel chunks
    (+ the 1x1 shortcut as extra K chunks of the second conv), output rows padded to the 32-row MFMA tile."""
    pad16 = lambda c: 16 * math.ceil(c / 16)
    pad32 = lambda c: 32 * math.ceil(c / 32)
    total, h = 0, rows
    for i in range(6):
        cin, cout = CAT_CH[i], CAT_CH[i + 1]
        pix = h * L
        total += 2 * pix * pad32(cout) * 9 * pad16(cin)                       # conv1
        total += 2 * pix * pad32(cout) * (9 * pad16(cout) + pad16(cin))       # conv2 + shortcut
        if CAT_POOL[i]:
            h = (h - CAT_POOL[i]) // CAT_POOL[i] + 1
    return total


def cat_conv_alg_flop(rows=40, L=11):
    """the same convolutions as the reference computes them (no padding of K or rows): crnn.py:92-117"""
    total, h = 0, rows
    for i in range(6):
        cin, cout = CAT_CH[i], CAT_CH[i + 1]
        pix = h * L
        total += 2 * pix * cout * (9 * cin + 9 * cout + cin)
        if CAT_POOL[i]:
            h = (h - CAT_POOL[i]) // CAT_POOL[i] + 1
    return total


def settle_collector():
    """Called right BEFORE a timed region starts (outside it): one full pass of the interpreter's cyclic collector over the garbage of
    the set-up and the warm-up runs.  The streamed pipelines pause the collector while they run (host.gc_paused); what was pending
    when they were entered runs at the first allocation after they return - measured at 26 + 35 ms right behind a 16-file window-file
    run, inside `torch.cuda.synchronize`'s Python wrapper, i.e. inside the clock - although none of it is the timed run's garbage.
    Collections that the timed run's own allocations cause stay inside the region."""
    import gc
    if os.environ.get("NSNP_NO_SETTLE") == "1":
        return
    gc.collect()


def cgroup_cpu_stat():
    """(throttled periods, throttled thread-microseconds, cpu microseconds used) of this cgroup so far, or None outside cgroup v2: the
    difference around a timed region says whether a CPU quota stalled it (a streamed pipeline that keeps 16 host threads spinning is
    frozen for the rest of a 100 ms period when the quota runs out: 10-20 ms holes in the GPU timeline)"""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0))
    except (OSError, ValueError):
        return None


def usable_cores():
    """cores this process may actually use: the affinity mask, cut by a cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(math.ceil(int(q) / int(per)))))
    except Exception:
        pass
    return n


def host_cpu_name():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return ""


# `traffic` of every roofline object is NOT measured by the run that prints the line: rocprofv3 cannot wrap a bench run from inside
TRAFFIC_SOURCE = ("profiles/roofline_traffic.json: HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the "
                  "same bench command (tools/prof_run.sh, committed), not from this run")


def roofline_mfma(kernel, exec_flop_per_launch, avg_launch_ms, launches, *, alg_flop_per_launch=None, peak=PEAK_F32_MFMA_TFLOPS,
                  traffic=None, how="", **extra):
    """`achieved` = flops the kernel EXECUTES per launch / its average launch duration; frac = achieved / peak <= 1."""
    ach = exec_flop_per_launch / (avg_launch_ms * 1e-3) / 1e12
    r = {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
         "avg_launch_ms": avg_launch_ms, "launches_timed": launches, "executed_flop_per_launch": exec_flop_per_launch, "traffic": traffic}
    if alg_flop_per_launch is not None:
        r["achieved_algorithmic"] = {"tflops": alg_flop_per_launch / (avg_launch_ms * 1e-3) / 1e12, "flop_per_launch": alg_flop_per_launch,
                                     "note": "flops of the reference's own schedule (SURVEY.md 8(d)) over the same time; the kernels execute the "
                                             "exact reduced schedule (only the centre position is consumed), so this figure is NOT a fraction "
                                             "of the chip peak and may exceed it"}
    r["traffic_source"] = TRAFFIC_SOURCE if traffic is not None else None
    if how:
        r["measured"] = how
    r.update(extra)
    return r


def roofline_hbm(kernel, bytes_per_launch, avg_launch_ms, launches, *, traffic=None, how="", **extra):
    ach = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
         "avg_launch_ms": avg_launch_ms, "launches_timed": launches, "algorithmic_bytes_per_launch": bytes_per_launch, "traffic": traffic,
         "traffic_source": TRAFFIC_SOURCE if traffic is not None else None}
    if how:
        r["measured"] = how
    r.update(extra)
    return r


def valu_issue_bound(kernel, coverage, columns_per_launch, avg_launch_ms, clock_mhz=2400.0):
    """Second, truthful bound of a kernel that is limited by vector-instruction issue rather than by HBM: the time the
    kernel's VALU wave-instructions need on the chip's 1,024 SIMDs at one wave64 instruction per 4 cycles per SIMD over the measured
    launch time.  (4 cycles: the same counter pass reads SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00 quad-cycle per instruction - the
    kernel's vector work is 32-bit integer / logic / address arithmetic, none of it the packed-fp32 forms that reach the 157 TFLOP/s
    vector peak at twice that rate.)  The instruction count is a property of kernel + data: SQ_INSTS_VALU / SQ_WAVES of the newest committed counter pass of
    this kernel at this coverage (profiles/r*_encode_*_<cov>x_sq_counters.json: counters are NOT from this run, the duration is)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_encode_*_%dx_sq_counters.json" % int(coverage))))
    for f in reversed(files):
        try:
            k = json.load(open(f))["kernels"][kernel]
            c = k["counters_per_launch"]
            per_wave = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
            cyc = 4.0 * c.get("SQ_ACTIVE_INST_VALU", c["SQ_INSTS_VALU"]) / c["SQ_INSTS_VALU"]      # cycles a wave's vector instruction holds the port
            salu_per_wave = c.get("SQ_INSTS_SALU", 0.0) / c["SQ_WAVES"]
        except Exception:
            continue
        waves = columns_per_launch / 64.0
        cycles = per_wave * waves * cyc / 1024.0
        t_min_ms = cycles / (clock_mhz * 1e3)
        return {"bound": "valu-issue", "frac": t_min_ms / avg_launch_ms, "valu_wave_instructions_per_64_columns": round(per_wave, 1),
                "salu_instructions_per_64_columns": round(salu_per_wave, 1), "cycles_per_wave_instruction": round(cyc, 2), "simds": 1024, "clock_mhz": clock_mhz,
                "min_launch_ms_at_full_issue": t_min_ms, "avg_launch_ms": avg_launch_ms,
                "counters_from": os.path.basename(f) + " (a committed rocprofv3 --pmc pass, not this run; the launch time is this run's)",
                "what": "time the kernel's vector instructions need if every SIMD's vector port were busy every cycle, over the measured time: how "
                        "close the kernel is to the bound it actually runs into (it moves 0.6 vector instructions per input byte; its HBM fraction "
                        "beside this one says how far that bound is from the memory system's)"}
    return None


def committed_traffic(workload, kernel, **match):
    """HBM bytes per launch from the committed PMC passes (profiles/roofline_traffic.json: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs, KiB units, FETCH x2 gfx950 correction), or None when no pass matches this configuration."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
        w = tj.get("workloads", {}).get(workload)
        if w is None and workload == "pileup" and "kernels" in tj:          # round-2 layout
            w = tj
        columns = match.pop("columns", None)
        if not w or any(w.get(k) != v for k, v in match.items()) or kernel not in w.get("kernels", {}):
            return None
        if columns is not None and "hbm_bytes_per_column" in w["kernels"][kernel]:
            return w["kernels"][kernel]["hbm_bytes_per_column"] * columns      # launches of the profiled run differ in size
        return w["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def reference_cpu(section):
    """the reference ITSELF timed on CPU in the development container (tests/manual/time_reference_cpu.py; it cannot travel
    to the GPU box): profiles/r03_reference_cpu.json, falling back to the round-2 file"""
    for name in ("r03_reference_cpu.json", "r02_reference_cpu.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            try:
                j = json.load(open(p))
                return j, j.get(section) if section in j else (j.get("haplotype", {}) or {}).get(section)
            except Exception:
                pass
    return None, None


# ---- the line the driver reads ------------------------------------------------------------------------------------------
# bench.py's LAST stdout line is one flat JSON object of a few KB: the contract's keys, one `roofline`, one `cpu_baseline`, numbers
# per sub-workload - no nested "metric" key, no prose.  Everything else a run knows (per-kernel tables, notes, every sub-workload's
# own full line) goes to bench_details.json beside bench.py.  Round 5's single 58 KB line with six nested full lines could not be
# read back by the driver (BENCH_r05.json: parsed null).
COMPACT_LINE_MAX_BYTES = 4096
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed")


def _num(v, digits=6):
    """numbers of the compact line: 6 significant digits are plenty and keep the line short"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{digits}g}")
    return v


def compact_roofline(r):
    if not isinstance(r, dict):
        return None
    out = {k: _num(r.get(k)) for k in _ROOF_KEYS}
    if isinstance(r.get("chip"), dict):
        out["chip_frac"] = _num(r["chip"].get("frac"))
    return out


def _parity_ok(line):
    p = line.get("parity_sample") if isinstance(line, dict) else None
    return p.get("ok") if isinstance(p, dict) else None


def compact_workload(line):
    """one sub-workload of the default run -> numbers only"""
    if not isinstance(line, dict) or "error" in line:
        return {"value": None, "ms_per_step": None, "frac": None, "parity_ok": False, "cpu": None,
                "error": str((line or {}).get("error", "no line"))[:80]}
    roof = line.get("roofline") or {}
    cb = line.get("cpu_baseline") or {}
    out = {"value": _num(line.get("value")), "ms_per_step": _num(line.get("ms_per_step")), "frac": _num(roof.get("frac")),
           "bound": roof.get("bound"), "parity_ok": _parity_ok(line), "cpu": _num(cb.get("value")),
           "wall_s": line.get("wall_s_of_this_sub_run")}
    if line.get("value") and line.get("ms_per_step"):
        out["sites_per_step"] = int(round(line["value"] * line["ms_per_step"] * 1e-3))      # units one step processed (value x step time)
    if line.get("bound_by"):
        out["bound_by"] = str(line["bound_by"]).split(":")[0].split(" (")[0][:16]
    if line.get("fraction_of_hbm_resident_rate") is not None:
        out["frac_of_resident_rate"] = _num(line["fraction_of_hbm_resident_rate"])
    b = line.get("bf16x3")
    if isinstance(b, dict) and b.get("value"):
        out["bf16x3"] = _num(b["value"])
    return out


def compact_line(full, details_path=None):
    """The driver's line from a run's full result object (`full` keeps every key it had; this only selects)."""
    cfg = full.get("config") or {}
    cb = full.get("cpu_baseline")
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _num(out["value"], 9), _num(out["ms_per_step"], 9)
    out["dtype"] = str(out["dtype"]).split(" ")[0]
    out["config"] = {k: (str(v)[:160] if isinstance(v, str) else v) for k, v in cfg.items()
                     if k in ("workload", "batch", "windows_resident_per_gpu", "batches_per_step", "sites_per_step", "streams", "coverage", "precision",
                              "parallelism", "world_size_observed", "gather", "TEST_CONFIGURATION")}
    out["roofline"] = compact_roofline(full.get("roofline"))
    for k in sorted(full):
        if k.startswith("roofline_") and isinstance(full[k], dict):
            out[k] = compact_roofline(full[k])
    if isinstance(cb, dict):
        out["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": str(cb.get("sample_short") or cb.get("sample") or "")[:100], "host_cpu": str(cb.get("host_cpu", ""))[:48]}
    else:
        out["cpu_baseline"] = None
    out["timed_region_s"] = _num(full.get("timed_region_s"))
    out["parity_ok"] = _parity_ok(full)
    p = full.get("parity_sample")
    if isinstance(p, dict):
        out["parity"] = {"sites": p.get("sites"), "max_abs_dp": _num(p.get("max_abs_dp")), "tolerance": p.get("tolerance"),
                         "encode_bit_exact": p.get("encode_bit_exact")}
    if isinstance(full.get("repeats"), dict):
        out["repeats"] = full["repeats"].get("values")
    if isinstance(full.get("shader_clock_mhz"), dict):
        out["shader_clock_mhz"] = _num(full["shader_clock_mhz"].get("value"))
    for label in ("bf16x3", "f16x3"):
        s = full.get(label)
        if isinstance(s, dict):
            out[label] = {"value": _num(s.get("value")), "ms_per_step": _num(s.get("ms_per_step")),
                          "frac": _num((s.get("roofline") or {}).get("frac")), "max_abs_dp_vs_fp32": _num(s.get("max_abs_dp_vs_fp32_on_the_pool")),
                          "parity_ok": _parity_ok(s)}
    if full.get("workloads"):
        out["workloads"] = {name: compact_workload(line) for name, line in full["workloads"].items()}
    if details_path:
        out["details"] = os.path.basename(details_path)
    return out


def dump_compact(line):
    """-> the text of the line; refuses to print something the driver could not read back"""
    s = json.dumps(line, separators=(",", ":"), allow_nan=False)
    if len(s) >= COMPACT_LINE_MAX_BYTES or s.count('"metric"') != 1 or "\n" in s:
        raise RuntimeError(f"bench line is {len(s)} bytes with {s.count(chr(34) + 'metric' + chr(34))} metric keys: the driver's reader needs one flat line under "
                           f"{COMPACT_LINE_MAX_BYTES} bytes")
    return s


def write_details(full, name="bench_details.json"):
    """the run's full result object -> bench_details.json beside bench.py (falls back to the working directory, then to the
    temporary directory); returns the path written, or None"""
    import tempfile
    for d in (ROOT, os.getcwd(), tempfile.gettempdir()):
        p = os.path.join(d, name)
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1)
                f.write("\n")
            return p
        except OSError:
            continue
    return None


def flush_native_stdout():
    """what native libraries hold in C stdio buffers -> out now (RCCL's version banner would otherwise appear at exit, behind the line)"""
    import sys
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def emit_line(full, tag=None, file=None):
    """print the driver's line for a run's full result object and keep the object itself in bench_details[_<tag>].json"""
    import sys
    path = write_details(full, "bench_details.json" if not tag else f"bench_details_{tag}.json")
    if path:
        print(f"bench.py: full result object -> {path}", file=sys.stderr, flush=True)
    # whatever native libraries still hold in C stdio buffers (RCCL prints a version banner to stdout at communicator init: it would be
    # flushed at exit, BEHIND this line) goes out first: the line must be the last thing on stdout
    flush_native_stdout()
    print(dump_compact(compact_line(full, path)), file=file or sys.stdout, flush=True)


# ---- marking the timed region for a rocprofv3 trace of the same run -------------------------------------------------------------------
def clocks_ns():
    """host clocks a tracer may stamp its records with (rocprofv3 writes nanoseconds of ONE of them; tools/summarize_pipeline_trace.py finds
    out which by looking where the traced kernels fall)"""
    import time
    out = {}
    for name in ("CLOCK_MONOTONIC", "CLOCK_BOOTTIME", "CLOCK_REALTIME", "CLOCK_MONOTONIC_RAW"):
        c = getattr(time, name, None)
        if c is not None:
            try:
                out[name] = time.clock_gettime_ns(c)
            except OSError:
                pass
    return out


def mark_region(begin, end, steps, extra=None):
    """NSNP_TRACE_MARK=<path>: the clocks at both ends of the (first) timed region of a host-fed pipeline run + its steps -> that JSON file, so
    that a kernel / memory-copy trace of the same process can be cut to the region the line's numbers come from"""
    path = os.environ.get("NSNP_TRACE_MARK")
    if not path or os.path.exists(path):
        return
    with open(path, "w") as f:
        json.dump({"begin": begin, "end": end, "steps": steps, **(extra or {})}, f)
