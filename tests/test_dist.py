"""Site sharding + result gather (nanosnp_amd/dist.py) with world_size 2 on CPU (gloo)."""
import os
import socket

import numpy as np
import pytest

from nanosnp_amd.dist import shard_columns, shard_range


def test_shard_range_is_a_contiguous_order_preserving_partition():
    for n in (0, 1, 7, 8, 9, 4096, 1_500_001):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_shard_columns_halo_keeps_every_owned_window_complete():
    m, world = 1000, 4
    for r in range(world):
        lo_h, hi_h, lo, hi = shard_columns(m, r, world)
        assert lo_h == max(0, lo - 16) and hi_h == min(m, hi + 16)
        for c in (lo, hi - 1):                 # a window centred on an owned column
            if c - 16 >= 0 and c + 16 < m:
                assert lo_h <= c - 16 and c + 16 < hi_h


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, q):
    import torch
    import torch.distributed as dist
    from nanosnp_amd.dist import gather_results, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank "computes" its own site range: payload = f(site index), so order is checkable
        lo, hi = shard_range(n_total, rank, world)
        idx = torch.arange(lo, hi, dtype=torch.float32)
        local = torch.stack([idx, idx * 2 + 1, torch.full_like(idx, rank)], dim=1)
        merged = gather_results(local, n_total, root=0)
        if rank == 0:
            q.put(merged.numpy())
        else:
            assert merged is None
            q.put(None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [0, 1, 5, 4097])
def test_gather_results_world2_gloo(n_total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    merged = next(o for o in outs if o is not None)
    assert merged.shape == (n_total, 3)
    want = np.arange(n_total, dtype=np.float32)
    assert np.array_equal(merged[:, 0], want) and np.array_equal(merged[:, 1], want * 2 + 1)
    lo1, _ = shard_range(n_total, 1, 2)
    assert np.array_equal(merged[:, 2], (np.arange(n_total) >= lo1).astype(np.float32))


def _pipeline_worker(rank, world, port, q):
    """Sharded oracle pipeline == single-process pipeline (the N>1 path, with the CPU checker
    standing in for the device kernels)."""
    import torch
    import torch.distributed as dist
    from nanosnp_amd import host
    from nanosnp_amd.dist import gather_results, shard_range
    from oracle import oracle
    from tests.helpers import load_pileup_weights
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 37
        cols = host.synth_columns(5, n * 33, window=33)
        counts, _, _ = oracle.encode_columns(cols.bases, cols.col_off, cols.ref)
        x = counts.reshape(n, 33, 18)
        lo, hi = shard_range(n, rank, world)
        gt, zy = oracle.pileup_forward(load_pileup_weights(), x[lo:hi])
        merged = gather_results(torch.from_numpy(np.concatenate([gt, zy], 1)), n)
        if rank == 0:
            full_gt, full_zy = oracle.pileup_forward(load_pileup_weights(), x)
            q.put(bool(np.array_equal(merged.numpy(), np.concatenate([full_gt, full_zy], 1))))
        else:
            q.put(None)
    finally:
        dist.destroy_process_group()


def test_sharded_pipeline_equals_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert any(o is True for o in outs)


# ---- bench.py's own launcher: `--gpus N` without a torchrun environment starts the N ranks itself -----------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    import subprocess
    import sys
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)


def test_bench_gpus2_spawns_two_ranks_and_relays_one_line():
    """the driver's `python bench.py --gpus N`: N ranks are started (CPU / gloo dry run of the same skeleton: barrier,
    max-over-ranks timing, rooted gather), rank 0's single JSON line comes back through the parent"""
    import json
    out = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-launcher")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_observed"] == 2 and d["steps"] == 5 and d["selftest"] is True
    assert d["gather_ok"] is True and d["slowest_rank_bound_ok"] is True      # the time is the slowest rank's
    # the ranks share the box's cores explicitly (torch.distributed.run would force OMP_NUM_THREADS=1 for nproc > 1)
    assert d["omp_num_threads"] == str(max(1, d["usable_cores"] // 2))


def test_a_rank_that_dies_before_the_gather_fails_the_whole_job_closed():
    """one rank exits before the rooted gather - the launcher's parent must come back NON-ZERO, without a
    result line and without hanging on the surviving rank's receive (torch.distributed.run tears the survivors down)"""
    import time
    t0 = time.time()
    out = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-launcher", env={"NSNP_SELFTEST_DIE_RANK": "1"})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "failed" in out.stderr and time.time() - t0 < 240
    # the root dying is no different
    out = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-launcher", env={"NSNP_SELFTEST_DIE_RANK": "0"})
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _empty_shard_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from nanosnp_amd.dist import gather_results
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # n_total < world: rank 2 owns nothing, and gather_results is the FIRST operation on the group
        from nanosnp_amd.dist import shard_range
        lo, hi = shard_range(2, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float32)[:, None]
        out = gather_results(local, 2)
        q.put(out.numpy() if rank == 0 else None)
    finally:
        dist.destroy_process_group()


def test_gather_results_with_an_empty_shard_as_the_first_call_on_the_group():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_shard_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    merged = next(o for o in outs if o is not None)
    assert merged.shape == (2, 1) and merged[:, 0].tolist() == [0.0, 1.0]


def test_bench_never_reports_fewer_ranks_than_asked_for():
    # a torchrun environment that disagrees with --gpus is an error, not a silent n_gpus: 1
    out = _bench("--gpus", "4", "--selftest-launcher", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
    # N > 1 on a machine without N GPUs: the ranks start, find no device, the parent exits non-zero without a result line
    import torch
    if torch.cuda.device_count() >= 2:
        return
    out = _bench("--gpus", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline")
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _varlen_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from nanosnp_amd.dist import gather_varlen
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = [5, 0, 3][rank]                                   # different lengths, one of them empty
        local = torch.full((n, 2), float(rank), dtype=torch.float64)
        local[:, 1] = torch.arange(n, dtype=torch.float64)
        out = gather_varlen(local)
        q.put(out.numpy() if rank == 0 else None)
    finally:
        dist.destroy_process_group()


def test_gather_varlen_world3_gloo():
    """the rooted gather of the column-sharded pipeline: per-rank site lists of different lengths, rank order kept"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_varlen_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    merged = next(o for o in outs if o is not None)
    assert merged.shape == (8, 2)
    assert merged[:, 0].tolist() == [0.0] * 5 + [2.0] * 3 and merged[:, 1].tolist() == [0, 1, 2, 3, 4, 0, 1, 2]


def _text_worker(rank, world, port, sizes, bs, q):
    import torch.distributed as dist
    from nanosnp_amd import host
    from nanosnp_amd.dist import batch_heads, gather_text, site_offsets
    from tests.test_vcf import _random_calls
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table, a = _random_calls(77, sum(sizes))
        lo = sum(sizes[:rank])
        mine = [x[lo:lo + sizes[rank]] for x in a]
        first, n_total = site_offsets(sizes[rank])
        assert (first, n_total) == (lo, sum(sizes))
        heads = batch_heads(mine[3], first, n_total, bs)
        text, rows = host.vcf_format_batches(table, *mine, batch_size=bs, first=first, n_total=n_total, heads=heads, as_view=True)
        out = gather_text(text)
        q.put(bytes(out) if rank == 0 else None)
        assert (out is None) == (rank != 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes,bs", [((1204, 0, 803), 1000), ((7, 2, 1300), 64), ((995, 8, 4), 1000), ((40, 3, 25), 7), ((0, 0, 0), 1000)])
def test_rows_formatted_per_rank_equal_the_single_process_text(sizes, bs):
    """every rank formats ITS rows (site_offsets + batch_heads: the ten argmax values a batch's rows read from one another), the text
    is gathered: the bytes of one process formatting the whole list - with an empty rank, a rank inside another's first ten rows
    and a last batch shorter than ten rows"""
    import torch.multiprocessing as mp
    from nanosnp_amd import host
    from tests.test_vcf import _random_calls
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_text_worker, args=(r, 3, port, sizes, bs, q)) for r in range(3)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table, a = _random_calls(77, sum(sizes))
    assert next(o for o in outs if o is not None) == host.vcf_format_batches(table, *a, batch_size=bs)[0]
