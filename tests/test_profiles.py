"""The bench lines committed under profiles/ (the evidence the round is judged by) keep the contract: required keys, every roofline
fraction physical (executed work / time / peak, 0 < frac <= 1), value consistent with ms_per_step, the run's own outputs checked
against the oracle (parity_sample), HBM traffic of the HBM-bound kernels not below the bytes they provably move."""
import glob
import json
import os

import pytest

from tests.helpers import ROOT

ROUND = "r06"
LINES = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", f"{ROUND}_*_line.json")) if "e2e" not in p)


def _load(path):
    """a committed *_line.json is a run's FULL result object (bench_details*.json); the driver's compact line is *_stdout.txt beside it"""
    return json.load(open(path))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_line(path):
    d = _load(path)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity_sample", "shader_clock_mhz"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    n = 0
    for k, r in d.items():
        if k.startswith("roofline") and r:
            assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1, k
            if "second_bound" in r:                              # the encode's truthful bound beside its HBM fraction
                sb = r["second_bound"]
                assert sb["bound"] == "valu-issue" and r["frac"] < sb["frac"] <= 1 and "not this run" in sb["counters_from"]
            if "chip" in r:
                assert 0 < r["chip"]["frac"] <= 1
            # traffic is labelled as coming from the committed PMC passes, and an HBM-bound kernel cannot have moved fewer bytes than
            # its algorithmic ones (the feature reduction's 712 MB against 1,005 MB provable was an average over two
            # launch shapes under a correction calibrated for another access width)
            if r.get("traffic") is not None:
                assert "not from this run" in r["traffic_source"], k
                if r["bound"] == "hbm":
                    assert r["traffic"] >= 0.95 * r["algorithmic_bytes_per_launch"], (k, r["traffic"], r["algorithmic_bytes_per_launch"])
            n += 1
    assert n >= 2
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["value"] > 50 * c["value"]
    # the line's own outputs were compared with the oracle, and a bf16x3 second value rides along with its own parity
    assert d["parity_sample"]["ok"] is True and d["parity_sample"].get("tolerance", 1e-4) == 1e-4
    assert 1500 < d["shader_clock_mhz"]["value"] < 2600
    b3 = d.get("bf16x3") or (d.get("second_values") or {}).get("forward_only_bf16x3")
    assert b3, "every line carries a bf16x3 second value"
    ps = b3.get("parity_sample")
    assert ps and ps["ok"] is True


def test_pileup_line_is_the_headline_configuration():
    d = _load(os.path.join(ROOT, "profiles", f"{ROUND}_default_line.json"))
    # the default command carries every other configuration as a sub-line (tools/workloads.py)
    w = d["workloads"]
    assert set(w) == {"haplotype", "two_stage", "deep60", "hap_e2e", "e2e", "pd_e2e"}
    for name, line in w.items():
        sm = line["summary"]
        assert "error" not in line and sm["value"] > 0 and sm["parity_ok"] is True and sm["cpu_baseline_value"] > 0, name
    assert 0 < w["haplotype"]["roofline_arrange"]["frac"] <= 1 and w["haplotype"]["roofline_arrange"]["parity"]["ok"]
    assert "second_bound" in d["roofline_encode"]
    assert "configs[1]" in d["config"]["workload"] and d["config"]["windows_resident_per_gpu"] == 1 << 20 and d["config"]["batch"] == 4096
    assert d["config"]["batches_per_step"] == 256 and d["timed_region_s"] >= 1.0              # one step = one sweep of the pool, >= 1 s timed
    assert len(d["repeats"]["values"]) == 3 and sorted(d["repeats"]["values"])[1] == round(d["value"])
    p = d["parity_sample"]
    assert p["sites"] >= 65536 and p["encode_bit_exact"] and p["calls_equal_own_argmax"] and p["max_abs_dp"] < 1e-4
    e = d["error_vs_float64"]["modes"]
    assert e["bf16x3"]["max_abs_error"] <= max(1.5 * e["fp32"]["max_abs_error"], 1e-6)      # full fp32 operand width: the fp32 path's own error
    assert d["bf16x3"]["roofline"]["peak"] == 2500.0 and d["bf16x3"]["value"] > d["value"]


def test_all_workloads_have_a_line():
    names = {os.path.basename(p) for p in LINES}
    assert {f"{ROUND}_default_line.json", f"{ROUND}_haplotype_line.json", f"{ROUND}_two_stage_line.json", f"{ROUND}_deep60_line.json"} <= names
    e2e = _load(os.path.join(ROOT, "profiles", f"{ROUND}_e2e_line.json"))
    assert e2e["parity_sample"]["ok"] and "NOT the headline" in e2e["config"]["workload"] and e2e["bound_by"] in e2e["stage_busy_s_per_step"]
    assert e2e["bf16x3"]["parity_sample"]["ok"]
    # stage 5 from host memory: >= 0.8 x the HBM-resident rate, csv identical across pass sizes, dtypes and to the reference's rows
    h = _load(os.path.join(ROOT, "profiles", f"{ROUND}_hap_e2e_line.json"))
    assert h["parity_sample"]["ok"] and h["parity_sample"]["timed_run_equals_the_one_pass_run"] and h["parity_sample"]["two_stage_fixture"]["ok"]
    assert h["fraction_of_hbm_resident_rate"] >= 0.8 and h["bound_by"].startswith("device")
    # the text path with the tokeniser on the device (round 6): device-bound, the host-parsed VCF is part of its parity sample, >= 12 M sites/s
    assert e2e["tokenise"] == "device" and e2e["bound_by"].startswith("device") and e2e["parity_sample"]["vcf_equals_the_host_parsed_run"] is True
    assert e2e["value"] >= 12e6 and e2e["host_parsed"]["value"] < e2e["value"] and 0 < e2e["roofline_tokenise"]["frac"] <= 1
    for k in ("int32_file_narrowed_while_staged", "int32_file_sent_as_int32"):
        assert h["second_values"][k]["fraction_of_hbm_resident_rate"] >= 0.8, k
    assert h["second_values"]["int32_file_sent_as_int32"]["bytes_over_pcie_per_site"] > 63360


def test_host_scaling_table_is_committed():
    t = json.load(open(os.path.join(ROOT, "profiles", f"{ROUND}_host_scaling.json")))
    rows = {(r["workload"], r["ranks"]) for r in t["rows"]}
    assert rows == {(w, n) for w in ("e2e", "hap-e2e", "pd-e2e") for n in (1, 2, 4, 8)} and "not a scaling number" in t["what"]


def test_fetch_calibration_is_committed():
    c = json.load(open(os.path.join(ROOT, "profiles", "r04_fetch_calibration.json")))["shapes"]
    assert abs(c["k_calib_b128"]["factor_bytes_per_counted_byte"] - 2.0) < 0.02                # the documented wide-read case reproduces
    assert 1.3 < c["k_calib_rows33<unsigned int>"]["factor_bytes_per_counted_byte"] < 2.0      # k_hap_features' shape does not follow it


def test_pd_e2e_line_is_committed():
    """the streamed window-file path's line: parity against the one-pass run and the oracle, both on-disk layouts, >= 0.75 of the
    HBM-resident rate with int16 counts on disk, the compute stream idle < 2 ms per file between passes"""
    h = json.load(open(os.path.join(ROOT, "profiles", f"{ROUND}_pd_e2e_line.json")))
    assert h["parity_sample"]["ok"] and h["parity_sample"]["timed_run_equals_the_one_pass_run"] and h["parity_sample"]["file_windows_equal_the_oracle_encode"]
    assert h["fraction_of_hbm_resident_rate"] >= 0.75 and h["compute_stream_idle_between_passes_s_per_step"] < 0.002
    assert set(h["second_values"]) == {"int32_counts_on_disk_narrowed_while_staged", "int32_counts_on_disk_sent_as_int32"}
    assert all(v["vcf_equals_the_int16_run"] for v in h["second_values"].values())


def test_the_drivers_lines_are_committed_beside_the_full_objects():
    """profiles/<round>_*_stdout.txt: what `bench.py` printed - the last line is the flat object the driver parses (a 58 KB
    line could not be read back): under 4 KB, one "metric" key, its numbers the full object's"""
    n = 0
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{ROUND}_*_stdout.txt"))):
        last = open(p).read().strip().splitlines()[-1]
        line = json.loads(last)
        assert len(last) < 4096 and last.count('"metric"') == 1, p
        full = json.load(open(p.replace("_stdout.txt", "_line.json")))
        assert abs(line["value"] - full["value"]) <= 1e-6 * full["value"] and line["metric"] == full["metric"], p
        for k in ("roofline", "cpu_baseline", "parity_ok", "config", "timed_region_s"):
            assert k in line, (p, k)
        n += 1
    assert n >= 7
    d = json.loads(open(os.path.join(ROOT, "profiles", f"{ROUND}_default_stdout.txt")).read().strip().splitlines()[-1])
    assert set(d["workloads"]) == {"haplotype", "two_stage", "deep60", "hap_e2e", "e2e", "pd_e2e"} and d["parity_ok"] is True
    assert all(w["parity_ok"] is True and w["value"] > 0 for w in d["workloads"].values())
    assert d["workloads"]["haplotype"]["sites_per_step"] == 15000 and d["workloads"]["two_stage"]["sites_per_step"] > 1_400_000      # BASELINE sizes


def test_pipeline_traces_are_committed():
    """rocprofv3 kernel (+ memory-copy) traces of the three host-fed bench runs, cut to their timed regions: the line's device-busy figure
    is within 5 % of the trace's union of dispatch intervals"""
    for name in ("e2e", "hap_e2e", "pd_e2e"):
        o = json.load(open(os.path.join(ROOT, "profiles", f"{ROUND}_{name}_overlap.json")))
        assert o["workload"] == name and o["per_step_ms"]["kernel_busy"] > 0 and o["dispatches_in_region"] > 50
        assert abs(o["line_against_trace"]["device_busy_ms_per_step"]["relative_difference"]) <= 0.05, name
        assert os.path.exists(os.path.join(ROOT, "profiles", f"{ROUND}_{name}_kernel_stats.csv"))
    t = json.load(open(os.path.join(ROOT, "profiles", f"{ROUND}_tokenise.json")))
    assert t["roofline"]["bound"] == "hbm" and 0 < t["roofline"]["frac"] <= 1 and t["roofline"]["traffic"] >= t["algorithmic_bytes_per_call"]
