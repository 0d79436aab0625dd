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
