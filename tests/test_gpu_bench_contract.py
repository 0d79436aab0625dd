"""bench.py prints one JSON line with the contract's keys (short run on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "1",
                          "--windows", "131072", "--cpu-seconds", "1.5"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 1 and d["unit"] == "sites/s"
    assert d["dtype"] == "f32" and d["config"]["windows_resident_per_gpu"] == 131072 and d["config"]["batches_per_step"] == 4
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0
    # value = sites of the timed steps / wall time; and far above the north-star target of 50k sites/s/GPU
    assert abs(d["value"] - d["config"]["sites_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["value"] > 1e6
    # the dominant kernel is chosen by total time over >= 32 timed launches, the executed-flop view rides along
    assert r["launches_timed"] >= 16 and r["kernel"] in d["kernel_avg_ms"] and "chip" in r and "executed" in r
    assert d["roofline_encode"]["bound"] == "hbm"
    # the opt-in arithmetic is a second, labelled value measured on the same pool, within the port's tolerance of fp32
    assert d["f16x3"]["value"] > 1e6 and d["f16x3"]["max_abs_dp_vs_fp32_on_the_pool"] < 1e-4


def test_two_stage_workload_line():
    """bench.py --workload two-stage (BASELINE configs[3]) on a reduced candidate set: both stages run, calls are gathered"""
    env = dict(os.environ, NSNP_TWO_STAGE_N2="40000", NSNP_TWO_STAGE_N5="5000")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "two-stage", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["stage2_sites"] == 40000 and d["config"]["stage5_sites"] == 5000 and d["scaling"] == "strong"
    assert d["value"] > 1e5 and d["stage5"]["sites_per_s"] > 1e4 and d["dtype"] == "f32"


@pytest.mark.parametrize("workload", ["pileup", "two-stage"])
def test_two_ranks_through_the_launcher_on_one_gpu(workload):
    """`bench.py --gpus 2` end to end on the one-GPU box: the parent starts the ranks, both run the real kernels on GPU 0, the
    barrier / max-over-ranks timing / rooted gather run over gloo (TEST configuration: everything of the N > 1 path except
    the RCCL transport itself)"""
    env = dict(os.environ, NSNP_TWO_STAGE_N2="30000", NSNP_TWO_STAGE_N5="3001")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "65536",
                          "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--no-second-precision", "--workload", workload],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size_observed"] == 2 and d["value"] > 1e5
    if workload == "pileup":
        assert d["scaling"] == "weak" and d["config"]["windows_resident_per_gpu"] == 65536 and "TEST_CONFIGURATION" in d["config"]
        assert abs(d["value"] - 2 * d["config"]["sites_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6      # whole-job aggregate
    else:
        assert d["scaling"] == "strong" and d["config"]["stage2_sites"] == 30000 and d["config"]["stage5_sites"] == 3001


def test_bench_with_the_library_gather_entry():
    """--gather rccl-abi: the final merge through nsnp_comm_init + nsnp_gather_results (one rank here)"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--windows", "65536", "--gather", "rccl-abi",
                          "--no-cpu-baseline", "--no-second-precision"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["gather"] == "rccl-abi" and d["value"] > 1e6
