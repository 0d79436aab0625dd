"""bench.py prints one JSON line with the contract's keys (short run on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fractions_are_physical(d):
    """every fraction of every roofline object of a line is executed work / time / peak: 0 < frac <= 1"""
    n = 0
    for k, r in d.items():
        if not k.startswith("roofline") or r is None:
            continue
        assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9, k
        assert 0.0 < r["frac"] <= 1.0, (k, r["frac"])
        for sub in ("chip", "exclusive"):
            if sub in r:
                assert 0.0 < r[sub]["frac"] <= 1.0, (k, sub, r[sub]["frac"])
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed"):
            assert key in r, (k, key)
        n += 1
    return n


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
             "config", "roofline", "cpu_baseline", "timed_region_s", "parity_ok")


def _parse(stdout):
    """What the driver does with a run: the LAST stdout line is one flat JSON object under 4 KB with exactly one "metric" key (round 5's 58 KB
    line with six nested full lines could not be read back: BENCH_r05.json parsed null).  -> the run's full result object from the details
    file the line names, with the line itself under "_line"; every number of the line must be the details' number."""
    last = stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert len(last) < 4096 and last.count('"metric"') == 1, (len(last), last.count('"metric"'))
    assert [l for l in stdout.splitlines() if l.startswith("{")] == [last]
    for k in LINE_KEYS:
        assert k in line, k
    assert all(not isinstance(v, str) or len(v) <= 160 for k, x in line.items() if k != "metric" for v in _leaves(x)), "prose in the driver's line"
    d = json.load(open(os.path.join(ROOT, line["details"])))
    assert abs(line["value"] - d["value"]) <= 1e-6 * d["value"] and abs(line["ms_per_step"] - d["ms_per_step"]) <= 1e-6 * d["ms_per_step"]
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "data"):
        assert line[k] == d[k], k
    assert line["config"]["workload"] == d["config"]["workload"][:160]
    if d.get("roofline"):
        for k in ("bound", "kernel", "peak", "unit"):
            assert line["roofline"][k] == d["roofline"][k]
        for k in ("achieved", "frac", "avg_launch_ms"):
            assert abs(line["roofline"][k] - d["roofline"][k]) <= 1e-5 * abs(d["roofline"][k])
    else:
        assert line["roofline"] is None
    if d.get("cpu_baseline"):
        assert abs(line["cpu_baseline"]["value"] - d["cpu_baseline"]["value"]) <= 1e-5 * d["cpu_baseline"]["value"]
        assert line["cpu_baseline"]["cores"] == d["cpu_baseline"]["cores"] and line["cpu_baseline"]["kind"] == d["cpu_baseline"]["kind"] and line["cpu_baseline"]["sample"]
    if isinstance(d.get("parity_sample"), dict):
        assert line["parity_ok"] is d["parity_sample"]["ok"]
    d["_line"] = line
    return d


def _leaves(v):
    if isinstance(v, dict):
        for x in v.values():
            yield from _leaves(x)
    elif isinstance(v, list):
        for x in v:
            yield from _leaves(x)
    else:
        yield v


def _run(*argv, env=None, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return _parse(out.stdout)


SMALL_POOLS = dict(NSNP_TWO_STAGE_N2="40960", NSNP_TWO_STAGE_N5="4096", NSNP_HAP_N="4096", NSNP_CAT_N="2048", NSNP_DEEP_WINDOWS="40960",
                   NSNP_HAPE2E_SITES="6000", NSNP_E2E_COLS="400000", NSNP_E2E_CHUNK_MB="4", NSNP_PDE2E_SITES="70000")


def test_bench_line_schema():
    d = _run("--gpus", "1", "--steps", "4", "--warmup", "1", "--windows", "131072", "--cpu-seconds", "1.5", "--hap-batch", "2048", env=dict(os.environ, **SMALL_POOLS))
    # the other BASELINE configurations ride in the same line (tools/workloads.py): every one with a value, a parity sample that holds, a CPU
    # baseline, and - where a device kernel dominates - a roofline fraction
    w = d["workloads"]
    assert set(w) == {"haplotype", "two_stage", "deep60", "hap_e2e", "e2e", "pd_e2e"}
    lw = d["_line"]["workloads"]                                  # the driver's line: numbers only per sub-workload
    assert set(lw) == set(w)
    for name, c in lw.items():
        assert c["parity_ok"] is True and c["value"] > 0 and c["ms_per_step"] > 0 and c["cpu"] > 0 and "error" not in c, (name, c)
        assert abs(c["value"] - w[name]["value"]) <= 1e-5 * c["value"] and "metric" not in c
        assert (0 < c["frac"] <= 1) if name in ("haplotype", "two_stage", "deep60") else c["frac"] is None
    assert d["_line"]["dtype"] == "f32" and d["_line"]["bf16x3"]["parity_ok"] is True and d["_line"]["roofline_encode"]["bound"] == "hbm"
    for name, line in w.items():
        assert "error" not in line, (name, line.get("error"))
        sm = line["summary"]
        assert sm["value"] > 0 and sm["ms_per_step"] > 0 and sm["parity_ok"] is True and sm["cpu_baseline_value"] > 0, (name, sm)
        assert line["parity_sample"]["ok"] and line["n_gpus"] == 1
        if name in ("haplotype", "two_stage", "deep60"):
            assert 0 < sm["dominant_kernel_frac"] <= 1 and _fractions_are_physical(line) >= 2, name
    ar = w["haplotype"]["roofline_arrange"]
    assert ar["bound"] == "hbm" and 0 < ar["frac"] <= 1 and ar["parity"]["ok"] and ar["sites_per_launch"] == 2048
    he = w["hap_e2e"]
    assert he["parity_sample"]["timed_run_equals_the_one_pass_run"] and he["parity_sample"]["two_stage_fixture"]["ok"] and 0 < he["fraction_of_hbm_resident_rate"] <= 1.2
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["unit"] == "sites/s"
    # one step = one sweep of the resident pool
    assert d["dtype"] == "f32" and d["config"]["windows_resident_per_gpu"] == 131072 and d["config"]["batches_per_step"] == 32
    # three timed passes, the median one reported; the clock the chip held; the run's own outputs against the oracle
    assert len(d["repeats"]["values"]) == 3 and sorted(d["repeats"]["values"])[1] == round(d["value"]) and d["repeats"]["reported"] == "median"
    assert 1500 < d["shader_clock_mhz"]["value"] < 2600
    p = d["parity_sample"]
    assert p["ok"] and p["encode_bit_exact"] and p["calls_equal_own_argmax"] and p["max_abs_dp"] < p["tolerance"] == 1e-4
    assert p["sites"] == 32 * 2048 and p["ring_slots_checked"] == 3 and p["ring_columns_checked"] > 0 and p["batches_sampled"] == 32
    assert all(r is None or r["traffic"] is None or "not from this run" in r["traffic_source"] for k, r in d.items() if k.startswith("roofline"))
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert _fractions_are_physical(d) == 3                      # roofline (MFMA, dominant forward kernel), the other recurrence layer, roofline_encode (HBM)
    r = d["roofline"]
    assert {r["kernel"]} | {d[k]["kernel"] for k in d if k.startswith("roofline_pileup_l")} == {"pileup_l0", "pileup_l1f"}
    assert r["bound"] == "mfma" and r["kernel"] in d["kernel_exclusive_ms"] and "chip" in r and "achieved_algorithmic" in r
    assert r["launches_timed"] >= 16
    assert d["roofline_encode"]["bound"] == "hbm" and d["roofline_encode"]["batches_per_launch"] == d["config"]["encode_batches_per_launch"] == 32
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0
    # value = sites of the timed steps / wall time; and far above the north-star target of 50k sites/s/GPU
    assert abs(d["value"] - d["config"]["sites_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["value"] > 1e6
    # the chip-level figure is the driver-clock cross-check: executed forward flops of the timed sites / wall time
    exec_flop_site = 2 * 33 * 256 * 84 * 2 + 2 * 17 * 256 * 192 * 2 + (128 * 128 + 256 * 128 + 32 * 256) * 2
    assert abs(r["chip"]["achieved"] - exec_flop_site * d["value"] / 1e12) / r["chip"]["achieved"] < 1e-6
    # the opt-in arithmetic is a second, labelled value measured on the same pool, within the port's tolerance of fp32
    assert d["f16x3"]["value"] > 1e6 and d["f16x3"]["max_abs_dp_vs_fp32_on_the_pool"] < 1e-4
    # bf16x3: full fp32 operand width on the bf16 pipe - its error against a float64 evaluation of the model is the fp32 path's own (the
    # two differ from each other by fp32 summation-order noise, a few 1e-6 on the pool), its own roofline against the bf16 peak with the
    # six MFMAs of a product priced as executed, its own outputs against the oracle
    b = d["bf16x3"]
    assert b["value"] > 1e6 and b["max_abs_dp_vs_fp32_on_the_pool"] < 5e-6 and b["parity_sample"]["ok"] and b["parity_sample"]["max_abs_dp"] < 1e-4
    e = d["error_vs_float64"]["modes"]
    assert e["fp32"]["sites"] == e["bf16x3"]["sites"] == e["f16x3"]["sites"] == 2048
    assert e["bf16x3"]["max_abs_error"] <= max(1.5 * e["fp32"]["max_abs_error"], 1e-6) and e["bf16x3"]["max_abs_error"] < 3e-6
    assert b["roofline"]["peak"] == 2500.0 and 0 < b["roofline"]["frac"] <= 1 and 0 < b["roofline"]["chip"]["frac"] <= 1
    nl = b["sites_per_forward_launch"]                            # bf16x3: 16,384 sites per forward launch (four pool batches)
    assert nl == 4 * 4096 and b["roofline"]["executed_flop_per_launch"] in (nl * 2 * 33 * 256 * (32 * 3 + 64 * 6) * 2, nl * 2 * 17 * 256 * 192 * 6 * 2)


def test_two_stage_workload_line():
    """bench.py --workload two-stage (BASELINE configs[3]) on a reduced candidate set: both stages run, calls are gathered"""
    env = dict(os.environ, NSNP_TWO_STAGE_N2="40000", NSNP_TWO_STAGE_N5="5000")
    d = _run("--workload", "two-stage", "--steps", "1", "--warmup", "1", "--cpu-seconds", "1.5", env=env)
    assert d["config"]["stage2_sites"] == 36864 and d["config"]["stage5_sites"] == 5000 and d["scaling"] == "strong"      # whole batches of 4096
    assert d["value"] > 1e5 and d["stage5"]["sites_per_s"] > 1e4 and d["dtype"] == "f32"
    assert _fractions_are_physical(d) == 4                      # hap LSTM chain, hap features, stage-2 forward kernel, encode
    assert d["roofline"]["bound"] == "mfma" and d["roofline_features"]["bound"] == "hbm"
    assert 0 < d["stage5"]["frac_of_fp32_mfma_peak"] <= 1 and 0 < d["stage2"]["frac_of_fp32_mfma_peak"] <= 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    p = d["parity_sample"]
    assert p["ok"] and p["stage2"]["ok"] and p["stage2"]["encode_bit_exact"] and p["stage5"]["ok"] and p["stage5"]["max_abs_dp"] < 1e-4


def test_haplotype_workload_line():
    """bench.py --workload haplotype (BASELINE configs[2]) on a reduced pool: features + HaplotypeModel forward, the labelled second values
    (int8 planes, f16x3, legacy CatModel), rooflines of the fused step launch (MFMA) and of the feature reduction (HBM), CPU baseline"""
    env = dict(os.environ, NSNP_HAP_N="6000", NSNP_CAT_N="4096")
    d = _run("--workload", "haplotype", "--steps", "3", "--warmup", "1", "--hap-batch", "2048", "--cpu-seconds", "1.5", env=env)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["unit"] == "sites/s"
    assert d["config"]["hap_sites_resident_per_gpu"] == 6000 and d["config"]["D"] == 90 and "workload" in d["config"]
    assert d["sites_timed"]["haplotype_sites"] == 2048 + 1904 + 2048          # batches 1, 2 (the ragged last one), 0 of the pool
    assert abs(d["value"] - d["sites_timed"]["haplotype_sites"] / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert _fractions_are_physical(d) == 3                      # fused LSTM step (MFMA), feature reduction (HBM), read arrangement (HBM)
    ar = d["roofline_arrange"]
    assert ar["bound"] == "hbm" and ar["parity"]["ok"] and ar["sites_per_launch"] == 2048 and ar["kernel"].startswith("k_hap_arrange")
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["launches_per_pass"] == 83
    assert d["roofline_features"]["bound"] == "hbm"
    s = d["second_values"]
    assert s["forward_only_f16x3"]["max_abs_dp_vs_fp32"] < 1e-4 and s["features_int8_planes"]["sites_per_s"] > 1e5
    assert 0 < s["legacy_CatModel_forward_fp32"]["roofline"]["frac"] <= 1 and s["legacy_CatModel_forward_fp32"]["sites_per_s"] > 1e4
    assert 0 < s["forward_only_fp32"]["frac_of_fp32_mfma_peak"] <= 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["parity_sample"]["ok"] and d["parity_sample"]["haplotype"]["sites"] >= 192 and d["parity_sample"]["haplotype"]["max_abs_dp"] < 1e-4


def test_deep60_workload_line():
    """bench.py --workload deep60 (BASELINE configs[4]) on reduced pools: 60x windows, D = 180 read planes, fp16-split CatModel weights"""
    env = dict(os.environ, NSNP_HAP_N="4096", NSNP_CAT_N="4096", NSNP_DEEP_WINDOWS="32768")
    d = _run("--workload", "deep60", "--steps", "2", "--warmup", "1", "--hap-batch", "2048", "--cpu-seconds", "1.5", env=env)
    assert d["config"]["D"] == 180 and d["config"]["coverage"] == 60.0 and d["config"]["window_batches_per_step"] == 8
    assert d["sites_timed"] == {"windows_60x": 2 * 8 * 4096, "haplotype_sites": 4096, "cat_sites": 4096}
    assert _fractions_are_physical(d) == 5                      # hap LSTM chain, features, 60x forward kernel, 60x encode, f16x3 conv chain
    assert d["roofline_cat_conv_f16x3"]["peak"] == 2500.0 and d["roofline_features"]["D"] == 180
    assert d["value"] > 1e4 and d["cpu_baseline"]["value"] > 0
    p = d["parity_sample"]
    assert p["ok"] and p["haplotype"]["ok"] and p["pileup_60x"]["ok"] and p["pileup_60x"]["encode_bit_exact"] and p["cat_f16x3"]["ok"]


def test_e2e_text_to_vcf_line():
    """bench.py --workload e2e (a labelled measurement, never the headline): mpileup text on the page cache -> chunked, double-buffered
    parse / H2D / encode / forward -> VCF; per-stage busy times, the bounding stage, parity against the one-chunk run"""
    env = dict(os.environ, NSNP_E2E_COLS="600000", NSNP_E2E_CHUNK_MB="8")
    d = _run("--workload", "e2e", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1.5", env=env)
    assert d["scaling"] == "strong" and d["unit"] == "sites/s" and d["config"]["columns"] == 600000 and "NOT the headline" in d["config"]["workload"]
    assert d["parity_sample"]["ok"] and d["parity_sample"]["sites"] == d["config"]["candidate_sites"] > 5000
    # the text is cut into columns on the device (nsnp_mpileup_tokenise); the same run with the host tokeniser rides along as a second value
    # and its VCF is part of the parity sample; the tokeniser's launches are priced against HBM
    assert d["tokenise"] == "device" and d["parity_sample"]["vcf_equals_the_host_parsed_run"] is True and d["host_parsed"]["value"] > 1e4
    assert d["roofline_tokenise"]["bound"] == "hbm" and 0 < d["roofline_tokenise"]["frac"] <= 1 and 80 < d["bytes_over_pcie_per_column"] < 100
    assert d["value"] > 1e4 and d["columns_per_s"] > 1e6 and abs(d["value"] - d["config"]["candidate_sites"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert len(d["stage_busy_s_per_step"]) == 4 and d["bound_by"] in d["stage_busy_s_per_step"] and d["cpu_baseline"]["value"] > 0
    assert d["config"]["text_bytes"] // d["config"]["chunk_bytes"] >= 5          # several chunks in flight
    m = d["main_thread_s_per_step"]
    # (the rows are formatted and written on a writer thread while the next contig of the run streams: pipeline.call_contigs)
    assert set(m) >= {"wait_parse_s", "issue_s", "wait_counts_s", "drain_s", "wait_rows_s"} and set(d["writer_thread_s_per_step"]) == {"vcf_s", "write_s"}
    assert sum(m[k] for k in ("wait_parse_s", "issue_s", "drain_s", "wait_rows_s")) <= d["ms_per_step"] * 1e-3 * 1.05


def test_pd_e2e_site_files_to_vcf_line():
    """bench.py --workload pd-e2e (a labelled measurement): .pd.bin window files on the page cache -> pinned staging / H2D / forward with the
    calls written into pinned memory -> VCF; both on-disk layouts, the bounding station, the host-CPU account, parity against the one-pass run
    and the oracle"""
    env = dict(os.environ, NSNP_PDE2E_SITES="150000")
    d = _run("--workload", "pd-e2e", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1.5", env=env)
    assert d["scaling"] == "strong" and d["unit"] == "sites/s" and d["config"]["sites"] == 150000 and "NOT the headline" in d["config"]["workload"]
    par = d["parity_sample"]
    assert par["ok"] and par["timed_run_equals_the_one_pass_run"] and par["the_K_files_gave_equal_rows"] and par["file_windows_equal_the_oracle_encode"]
    assert par["max_abs_dp_vs_oracle"] <= par["tolerance"] == 1e-4
    assert abs(d["value"] - 150000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6 and 0.0 < d["fraction_of_hbm_resident_rate"] <= 1.05
    assert len(d["stage_busy_s_per_step"]) == 4 and d["bound_by"] in d["stage_busy_s_per_step"] and d["cpu_baseline"]["value"] > 0
    assert d["bytes_over_pcie_per_site"] == 1188
    sv = d["second_values"]
    assert set(sv) == {"int32_counts_on_disk_narrowed_while_staged", "int32_counts_on_disk_sent_as_int32"}
    assert all(v["vcf_equals_the_int16_run"] and v["value"] > 0 for v in sv.values())
    assert sv["int32_counts_on_disk_sent_as_int32"]["bytes_over_pcie_per_site"] == 2376


@pytest.mark.parametrize("workload", ["pileup", "two-stage", "haplotype"])
def test_two_ranks_through_the_launcher_on_one_gpu(workload):
    """`bench.py --gpus 2` end to end on the one-GPU box: the parent starts the ranks, both run the real kernels on GPU 0, the
    barrier / max-over-ranks timing / rooted gather run over gloo (TEST configuration: everything of the N > 1 path except
    the RCCL transport itself)"""
    env = dict(os.environ, NSNP_TWO_STAGE_N2="30000", NSNP_TWO_STAGE_N5="3001", NSNP_HAP_N="3000", NSNP_CAT_N="2048")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "65536",
                          "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--no-second-precision", "--hap-batch", "2048", "--workload", workload,
                          "--workloads", "none"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _parse(out.stdout)
    assert d["n_gpus"] == 2 and d["config"]["world_size_observed"] == 2 and d["value"] > 1e5
    if workload == "haplotype":
        assert d["scaling"] == "weak" and d["config"]["hap_sites_resident_per_gpu"] == 3000 and "TEST_CONFIGURATION" in d["config"]
        assert abs(d["value"] - 2 * d["sites_timed"]["haplotype_sites"] / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6          # whole-job aggregate
    elif workload == "pileup":
        assert d["scaling"] == "weak" and d["config"]["windows_resident_per_gpu"] == 65536 and "TEST_CONFIGURATION" in d["config"]
        assert abs(d["value"] - 2 * d["config"]["sites_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6      # whole-job aggregate
    else:
        assert d["scaling"] == "strong" and d["config"]["stage2_sites"] == 2 * 12288 and d["config"]["stage5_sites"] == 3001     # whole batches per rank


def test_two_ranks_pd_e2e_on_one_gpu():
    """--workload pd-e2e under two ranks sharing the GPU (gloo): every rank streams its shard of every file, rank 0 formats and writes;
    the line's parity sample compares the sharded, streamed VCF with the one-pass run and the oracle"""
    env = dict(os.environ, NSNP_PDE2E_SITES="100000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo", "--share-gpu",
                          "--no-cpu-baseline", "--workload", "pd-e2e"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _parse(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["sites"] == 100000 and d["parity_sample"]["ok"]
    assert all(v["vcf_equals_the_int16_run"] for v in d["second_values"].values())


def test_bench_with_the_library_gather_entry():
    """--gather rccl-abi: the final merge through nsnp_comm_init + nsnp_gather_results (one rank here)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--windows", "65536", "--gather", "rccl-abi",
                          "--no-cpu-baseline", "--no-second-precision", "--workloads", "none"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _parse(out.stdout)
    assert d["config"]["gather"] == "rccl-abi" and d["value"] > 1e6


def test_two_ranks_with_the_other_configurations_in_the_same_line():
    """the default line's "workloads" under two ranks (sharing GPU 0, collectives over gloo): the sub-runs reuse the process group of the
    headline, every rank takes part in their gathers, rank 0 reports them; hap_e2e and pd_e2e shard the sites of every FILE over the ranks"""
    env = dict(os.environ, **SMALL_POOLS)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "65536",
                          "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--no-second-precision", "--hap-batch", "2048",
                          "--workloads", "two_stage,hap_e2e,haplotype,pd_e2e"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _parse(out.stdout)
    assert d["n_gpus"] == 2 and set(d["workloads"]) == {"two_stage", "hap_e2e", "haplotype", "pd_e2e"}
    for name, line in d["workloads"].items():
        assert "error" not in line and line["n_gpus"] == 2 and line["config"]["world_size_observed"] == 2 and line["summary"]["parity_ok"] is True, name
    assert d["workloads"]["hap_e2e"]["parity_sample"]["rows"] == 6000


def test_direct_torchrun_keeps_the_line_last_on_the_jobs_stdout():
    """the driver's N > 1 command: `python -m torch.distributed.run ... bench.py --gpus N` - every rank inherits the job's stdout, so ranks
    other than 0 send theirs to stderr and rank 0 flushes native buffers before it prints: the LAST line of the merged stdout is the line
    (here: two ranks sharing GPU 0 over gloo, the TEST configuration)"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "65536",
                          "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--no-second-precision", "--workloads", "none"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _parse(out.stdout)
    assert d["n_gpus"] == 2 and d["config"]["world_size_observed"] == 2 and d["_line"]["parity_ok"] is True
