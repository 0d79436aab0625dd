"""Host-side arithmetic of the bench lines (tools/bench_common.py): work-per-unit tables, roofline objects, traffic lookup."""
import json
import os

import pytest

from tools import bench_common as bc


def test_pileup_work_tables_add_up_to_the_survey_figures():
    # 6,274,560 MAC = 12.55 MFLOP/site as the reference computes it
    assert bc.PILEUP_ALG_FLOP_FORWARD == 2 * 6_274_560
    assert bc.PILEUP_ALG_FLOP["pileup_l1f"] == bc.PILEUP_ALG_FLOP["pileup_proj1"] + bc.PILEUP_ALG_FLOP["pileup_l1"]
    # executed: 33 steps of K = 20 + 64 in layer 0, 17 steps of K = 128 + 64 in layer 1, heads at one position
    assert bc.PILEUP_EXEC_FLOP["pileup_l0"] == 2 * 33 * 256 * 84 * 2
    assert bc.PILEUP_EXEC_FLOP_FORWARD == 2 * 33 * 256 * 84 * 2 + 2 * 17 * 256 * 192 * 2 + (128 * 128 + 256 * 128 + 32 * 256) * 2
    assert bc.PILEUP_EXEC_FLOP_FORWARD < bc.PILEUP_ALG_FLOP_FORWARD


def test_haplotype_and_catmodel_work():
    # model_dev.py:59-84: per step and encoder 2 directions x 1024 gate rows x (K_in + 256); K_in = 112 (105 padded) / 512 / 512,
    # the last layer only up to the centre step (17 of 33, 6 of 11)
    per = lambda kin: 2 * 1024 * (kin + 256) * 2
    want = sum(L * per(112) + L * per(512) + (L // 2 + 1) * per(512) for L in (33, 11))
    assert bc.hap_exec_flop() == want and 276e6 < want < 278e6 < bc.HAP_ALG_FLOP
    assert bc.hap_lstm_launches() == 83
    # crnn.py:92-190: 297.8 MFLOP/site of convolutions as the reference computes them; the tile GEMM pads K and rows a little
    assert 297e6 < bc.cat_conv_alg_flop() < 299e6 and bc.cat_conv_alg_flop() <= bc.cat_conv_exec_flop() < 1.02 * bc.cat_conv_alg_flop()


def test_roofline_objects_are_physical_fractions():
    r = bc.roofline_mfma("k", 13.69e9, 0.1088, 32, alg_flop_per_launch=26.58e9)
    assert r["bound"] == "mfma" and abs(r["achieved"] - 13.69e9 / 0.1088e-3 / 1e12) < 1e-9
    assert abs(r["frac"] - r["achieved"] / bc.PEAK_F32_MFMA_TFLOPS) < 1e-12 and 0 < r["frac"] <= 1
    assert r["achieved_algorithmic"]["tflops"] > bc.PEAK_F32_MFMA_TFLOPS        # the reference schedule's figure is NOT a fraction
    for k in ("peak", "unit", "traffic", "avg_launch_ms", "launches_timed"):
        assert k in r
    h = bc.roofline_hbm("enc", 116.6e6, 0.0717, 4)
    assert h["bound"] == "hbm" and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-12 and 0 < h["frac"] <= 1


def test_committed_traffic_lookup(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    json.dump({"workloads": {"pileup": {"batch": 4096, "precision": 0, "kernels": {
        "pileup_l0": {"hbm_bytes_per_launch": 86.7e6},
        "encode_columns": {"hbm_bytes_per_launch": 140e6, "hbm_bytes_per_column": 129.7}}}}}, open(prof / "roofline_traffic.json", "w"))
    monkeypatch.setattr(bc, "ROOT", str(tmp_path))
    assert bc.committed_traffic("pileup", "pileup_l0", batch=4096, precision=0) == 86.7e6
    assert bc.committed_traffic("pileup", "pileup_l0", batch=8192, precision=0) is None            # another configuration: no figure
    assert bc.committed_traffic("haplotype", "pileup_l0") is None
    # the column encode is launched with groups of batches of any size: priced per column
    assert abs(bc.committed_traffic("pileup", "encode_columns", batch=4096, columns=1000) - 129.7e3) < 1e-6


def test_usable_cores_is_positive_and_bounded():
    n = bc.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_the_drivers_line_is_flat_and_small():
    """the 58 KB default line (six nested full lines, each with its own "metric") could not be read back by the driver.
    The same result object through compact_line: one flat object under 4 KB, one "metric" key, no prose, the contract's keys all there."""
    full = json.load(open(os.path.join(bc.ROOT, "profiles", "r05_default_line.json")))
    assert len(json.dumps(full)) > 50_000
    line = bc.compact_line(full, "/somewhere/bench_details.json")
    s = bc.dump_compact(line)
    assert len(s) < bc.COMPACT_LINE_MAX_BYTES and s.count('"metric"') == 1 and "\n" not in s
    back = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "timed_region_s", "parity_ok", "workloads"):
        assert k in back, k
    assert back["dtype"] == "f32" and back["config"]["workload"].startswith("BASELINE configs[1]") and back["details"] == "bench_details.json"
    assert abs(back["value"] - full["value"]) < 1 and abs(back["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert set(back["roofline"]) == {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed", "chip_frac"}
    assert set(back["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample", "host_cpu"} and len(back["cpu_baseline"]["sample"]) <= 100
    assert set(back["workloads"]) == set(full["workloads"])
    for name, w in back["workloads"].items():
        assert w["parity_ok"] is True and w["cpu"] > 0 and all(not isinstance(v, (dict, list)) for v in w.values()), name
    # a sub-workload that raised is a short error entry with parity_ok false, not a traceback
    full["workloads"]["e2e"] = {"error": "RuntimeError: " + "x" * 500}
    e = bc.compact_line(full)["workloads"]["e2e"]
    assert e["parity_ok"] is False and len(e["error"]) <= 80
    # a line that would not fit is refused rather than printed
    full["config"]["workload"] = "w" * 100
    big = bc.compact_line(full)
    big["workloads"] = {f"w{i}": big["workloads"]["haplotype"] for i in range(40)}
    with pytest.raises(RuntimeError):
        bc.dump_compact(big)


def test_write_details_round_trip(tmp_path, monkeypatch):
    monkeypatch.setattr(bc, "ROOT", str(tmp_path))
    p = bc.write_details({"metric": "m", "value": 1.5, "nested": {"metric": "inner"}})
    assert p == str(tmp_path / "bench_details.json") and json.load(open(p))["nested"]["metric"] == "inner"
