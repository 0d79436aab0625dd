#!/usr/bin/env python3
"""bench.py -- candidate SNP sites/sec (pileup encode + PileupModel forward) on synthetic 30x windows.

Workload = BASELINE.json configs[1]: a pool of 1M stand-alone 33-column windows (generator G2,
SURVEY.md 8(d)) resident in HBM, processed in batches of 4096 windows.  One *step* = one batch
through the hot path: column encode (mpileup bytes -> int32 [M,18] counts) + PileupModel forward
reading the windows in place (-> softmax probabilities) + argmax/max/depth post-processing.
Batches are independent, so steps are issued round-robin over `--streams` HIP streams (one
nsnp_ctx each) to keep all 256 CUs busy at this batch size.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 4096] [--streams 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: one process per GPU, every rank owns its own pool (weak scaling, no data-path
collective); the only exchange is the rooted gather of the compact per-site results at the end
(inside the timed region).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

METRIC = "candidate SNP sites/sec (pileup encode + model fwd), 30x windows"

# algorithmic work per site of the reference schedule
ALG_FLOP_PER_SITE = {
    "pileup_l0": 2 * 1_385_472,      # layer-0 BiLSTM, 33 steps x 2 directions (model.py:34-35)
    "pileup_proj1": 2 * 2_162_688,   # layer-1 input GEMMs, 33 steps x 2 directions
    "pileup_l1": 2 * 1_081_344,      # layer-1 recurrent GEMMs
    "pileup_head": 2 * 1_645_056,    # output_proj + dense on 33 positions + 4 heads (model.py:37,67-72)
}
ALG_FLOP_PER_SITE["pileup_l1f"] = ALG_FLOP_PER_SITE["pileup_proj1"] + ALG_FLOP_PER_SITE["pileup_l1"]   # fused kernel
assert sum(v for k, v in ALG_FLOP_PER_SITE.items() if k != "pileup_l1f") == 2 * 6_274_560          # 12.55 MFLOP/site
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak (the f16x3 path issues 3 fp16 MFMAs per fp32 product)
PEAK_HBM_GBS = 8000.0


# f16x3 recurrence kernels: SIMD cycles per site = MFMAs x 16.3 / 16 sites + LSTM cells x 82 / 64 lanes
SERIAL_SIMD_CYCLES_PER_SITE = {"pileup_l0": 66 * 128 * 16.3 / 16 + 66 * 64 * 82 / 64,
                               "pileup_l1f": 34 * 288 * 16.3 / 16 + 34 * 64 * 82 / 64}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--windows", type=int, default=1 << 20, help="windows resident per GPU")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (0 = runtime default of 4). 16 lets more "
                    "of the 32 streams run kernels side by side: +8 %% sites/s (34 M vs 31.5 M), but every launch is then stretched by the "
                    "launches it shares the chip with and the per-launch durations stop being comparable with a rocprofv3 trace of "
                    "the same command (tracing itself lowers the concurrency); 24 and more collapse")
    ap.add_argument("--precision", type=int, default=1, help="PileupModel forward: 0 exact fp32 MFMA, 1 f16x3 split")
    ap.add_argument("--fused-l1", type=int, default=1, help="f16x3: fused projection + layer-1 recurrence kernel")
    ap.add_argument("--fused-waves", type=int, default=8, help="waves per workgroup of the fused kernel (0 = auto)")
    ap.add_argument("--l0-rs", type=int, default=1, help="f16x3: register-stationary layer-0 kernel (0 = LDS-image kernel)")
    ap.add_argument("--l1-rs", type=int, default=1, help="f16x3 fused layer 1: register-stationary kernel (0 = LDS-image / ring kernel)")
    ap.add_argument("--l1-groups", type=int, default=0, help="16-site groups per layer-1 workgroup (0 = picked from the batch size)")
    ap.add_argument("--l0-groups", type=int, default=0, help="16-site groups per layer-0 workgroup (0 = picked from the batch size, 1 at 4096 sites)")
    ap.add_argument("--proj1-tiles", type=int, default=0, help="tiles per wave of the projection kernel (0 = library default)")
    ap.add_argument("--rec-waves", type=int, default=0, help="force waves per recurrence workgroup (0 = auto)")
    ap.add_argument("--repeat", type=int, default=1, help="repeat the timed region (extra values are informational)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--timing-streams", type=int, default=8, help="record per-kernel HIP events on this many of the streams "
                    "(every kernel launch of those streams inside the timed region; event records cost ~6 %% when on all 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    return ap.parse_args()


def cpu_baseline(cols, batch, weights, target_s):
    """The oracle (plain-C port of the reference algorithm, full reference schedule) on this box's
    host cores, on a bounded sample of the same windows: encode + forward, all cores."""
    from oracle import oracle
    cores = os.cpu_count() or 1

    def run(n):
        m = n * 33
        b1 = int(cols.col_off[m])
        t0 = time.perf_counter()
        counts, depth, flags = oracle.encode_columns(cols.bases[:b1], cols.col_off[:m + 1], cols.ref[:m])
        gt, zy = oracle.pileup_forward(weights, counts.reshape(n, 33, 18), nthreads=cores)
        return time.perf_counter() - t0

    n0 = min(1024, batch)
    t = run(n0)
    n = int(min(max(n0, n0 * target_s / max(t, 1e-6)), 262144, cols.n_cols // 33))
    n = max(n0, (n // 64) * 64)
    t = run(n)
    return {"value": n / t, "unit": "sites/s", "cores": cores, "kind": "port",
            "sample": f"{n} of the same synthetic windows (encode + full-schedule fp32 forward), "
                      f"oracle/liboracle.so with OpenMP over {cores} threads, {t:.1f} s"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)      # must be set before HIP initialises
    # the host-side generator / CPU baseline use OpenMP: share the cores between the ranks of a node
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, world))))
    import torch
    import torch.distributed as dist
    from nanosnp_amd import _lib, host
    from nanosnp_amd.dist import gather_results
    from tests.helpers import load_pileup_weights

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    batch, S = args.batch, max(1, args.streams)
    n_windows = max(batch, (min(args.windows, max(args.steps, 1) * batch) // batch) * batch)
    n_batches = n_windows // batch
    weights = load_pileup_weights()                      # the shipped ont_pileup weights (fixture)

    # ---- synthetic pool, resident in HBM before the clock starts ---------------------------------
    cols = host.synth_columns(20260000 + rank, n_windows * 33, coverage=args.coverage, window=33)
    d_bases = torch.from_numpy(cols.bases).to(dev)
    d_off = torch.from_numpy(cols.col_off).to(dev)
    d_ref = torch.from_numpy(cols.ref).to(dev)
    centers = (torch.arange(batch, dtype=torch.int64, device=dev) * 33 + 16).contiguous()
    mcols = batch * 33

    lib = _lib.load()
    ctxs, streams, bufs = [], [], []
    for s in range(S):
        ctx = _lib.Context(local_rank, chunk_sites=batch)
        ctx.pileup_load_weights(weights)
        ctx.enable_timing(not args.no_kernel_timing and s < max(1, args.timing_streams))
        ctx.set_option("pileup_precision", args.precision)
        if args.rec_waves:
            ctx.set_option("recurrence_waves", args.rec_waves)
        if args.proj1_tiles:
            ctx.set_option("proj1_tiles", args.proj1_tiles)
        ctx.set_option("fused_l1", args.fused_l1)
        ctx.set_option("l0_register_stationary", args.l0_rs)
        ctx.set_option("l1_register_stationary", args.l1_rs)
        ctx.set_option("l1_site_groups", args.l1_groups)
        ctx.set_option("l0_site_groups", args.l0_groups)
        if args.fused_waves:
            ctx.set_option("fused_waves", args.fused_waves)
        ctxs.append(ctx)
        streams.append(torch.cuda.Stream(device=dev))
        bufs.append(dict(
            counts=torch.empty((mcols, 18), dtype=torch.int32, device=dev),
            depth=torch.empty(mcols, dtype=torch.int32, device=dev),
            flags=torch.empty(mcols, dtype=torch.uint8, device=dev)))
    # results of every batch of the pool stay resident (24 fp32 + compact calls per site)
    gt_all = torch.empty((n_windows, 21), dtype=torch.float32, device=dev)
    zy_all = torch.empty((n_windows, 3), dtype=torch.float32, device=dev)
    res = dict(ga=torch.empty(n_windows, dtype=torch.uint8, device=dev), za=torch.empty(n_windows, dtype=torch.uint8, device=dev),
               gm=torch.empty(n_windows, dtype=torch.float32, device=dev), zm=torch.empty(n_windows, dtype=torch.float32, device=dev))

    P = C.c_void_p

    def make_step(i, s=None):
        """pre-built argument lists: the timed loop is three C-ABI calls per step"""
        b = i % n_batches
        s = i % S if s is None else s
        c0 = b * mcols
        st = P(streams[s].cuda_stream)
        h = ctxs[s].handle
        bf = bufs[s]
        enc = (h, P(d_bases.data_ptr()), P(d_off.data_ptr() + 8 * c0), P(d_ref.data_ptr() + c0), mcols,
               C.c_double(0.12), 6, P(bf["counts"].data_ptr()), P(bf["depth"].data_ptr()), P(bf["flags"].data_ptr()), st)
        n0 = b * batch
        gt_p, zy_p = P(gt_all.data_ptr() + 4 * 21 * n0), P(zy_all.data_ptr() + 4 * 3 * n0)
        fwd = (h, P(bf["counts"].data_ptr()), P(centers.data_ptr()), batch, gt_p, zy_p, st)
        post = (h, gt_p, zy_p, None, batch, P(res["ga"].data_ptr() + n0), P(res["za"].data_ptr() + n0),
                P(res["gm"].data_ptr() + 4 * n0), P(res["zm"].data_ptr() + 4 * n0), None, st)
        return enc, fwd, post

    def run_steps(first, count, table=None):
        for i in range(first, first + count):
            enc, fwd, post = (table or steps)[i]
            rc = lib.nsnp_pileup_encode_columns(*enc)
            rc = rc or lib.nsnp_pileup_forward_windows(*fwd)
            rc = rc or lib.nsnp_pileup_postprocess(*post)
            if rc:
                _lib.check(rc, None, "bench step")

    def sync_all():
        for st in streams:
            st.synchronize()
        torch.cuda.synchronize(dev)

    W, K = args.warmup, args.steps
    steps = [make_step(i) for i in range(W + K)]

    def barrier():
        if world > 1:
            dist.barrier()

    def merge_results(n_done):
        """final merge: compact per-site calls of this rank -> rank 0 (RCCL gather over xGMI)"""
        compact = torch.stack([res["ga"][:n_done].float(), res["za"][:n_done].float(),
                               res["gm"][:n_done], res["zm"][:n_done]], dim=1)
        return gather_results(compact, n_done * world) if world > 1 else compact

    n_done = min(K, n_batches) * batch
    run_steps(0, W)
    sync_all()
    merge_results(n_done)               # warm the merge path (first-use module loads) outside the clock
    sync_all()
    extra = []
    for rep in range(max(0, args.repeat - 1)):      # informational repeats BEFORE the reported region
        sync_all(); t0 = time.perf_counter(); run_steps(W, K); sync_all()
        extra.append(world * K * batch / (time.perf_counter() - t0))
    for ctx in ctxs:
        ctx.read_timing()               # drop warm-up launches
    barrier(); sync_all()
    t0 = time.perf_counter()
    run_steps(W, K)
    t_issue = time.perf_counter() - t0          # host time to enqueue the K steps (informational)
    sync_all()
    merged = merge_results(n_done)
    sync_all(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- per-kernel durations from HIP events recorded on the launch streams ---------------------
    tot = {}
    for ctx in ctxs:
        for k, (ms, n) in ctx.read_timing().items():
            a = tot.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += n
    # the same kernels with the chip to themselves (one stream, after the timed region): in the timed region up to
    # --hw-queues launches share the chip, which stretches every launch without saying anything about the kernel
    excl = {}
    if not args.no_kernel_timing:
        solo = [make_step(W + i, 0) for i in range(32)]
        run_steps(0, 4, solo); sync_all(); ctxs[0].read_timing()
        run_steps(4, 28, solo); sync_all()
        excl = {k: ms / n for k, (ms, n) in ctxs[0].read_timing().items() if n}
    if rank == 0:
        if args.precision == 1 and args.fused_l1 and "pileup_l1" in tot:      # the fused kernel is timed in the l1 slot
            tot["pileup_l1f"] = tot.pop("pileup_l1")
        avg_ms = {k: (v[0] / v[1]) for k, v in tot.items() if v[1]}
        if not avg_ms:
            print(json.dumps({"metric": METRIC, "value": world * K * batch / dt, "unit": "sites/s", "n_gpus": world,
                              "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "host_issue_ms_per_step": t_issue / K * 1e3, "note": "kernel timing disabled",
                              "repeats_before": [round(v) for v in extra]}))
            return
        dom = max(avg_ms, key=lambda k: tot[k][0])
        if dom in ALG_FLOP_PER_SITE:
            achieved = ALG_FLOP_PER_SITE[dom] * batch / (avg_ms[dom] * 1e-3) / 1e12
            peak = PEAK_F16_MFMA_TFLOPS if args.precision == 1 else PEAK_F32_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak}
            if args.precision == 1:
                roof["note"] = ("f16x3: 3 fp16 MFMAs per fp32 product; achieved = algorithmic fp32-equivalent flops, priced against "
                                "the dense fp16 peak.  A SIMD of this chip runs the MFMAs (16.3 cycles per 16x16x32) and the LSTM "
                                "cell's exp2 / rcp work (82 cycles per 64-lane cell) one after the other, not side by side "
                                "(tools/probes/cell_rate.hip): serial_bound is the launch time that sum allows (DESIGN.md section 4)")
        else:
            nbytes = (int(cols.col_off[mcols]) + mcols * (1 + 72))        # bytes in + ref + 18 int32 out
            achieved = nbytes / (avg_ms[dom] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS}
        if args.precision == 1 and dom in SERIAL_SIMD_CYCLES_PER_SITE:
            # measured serial bound of a recurrence kernel on this chip: a SIMD spends 16.3 cycles per
            # 16x16x32 fp16 MFMA and 82 cycles per 64-lane LSTM cell, and the two do not overlap
            cyc = SERIAL_SIMD_CYCLES_PER_SITE[dom]
            bound_ms = cyc * batch / (1024 * 2.0e9) * 1e3
            roof["serial_bound"] = {"simd_cycles_per_site": cyc, "clock_ghz": 2.0, "simds": 1024, "bound_ms_per_launch": bound_ms,
                                    "frac": bound_ms / avg_ms[dom],
                                    "note": "whole-chip bound; launches of 4096 sites share the chip with the other streams' kernels"}
        xk = "pileup_l1" if dom == "pileup_l1f" else dom
        if xk in excl:
            unit_work = ALG_FLOP_PER_SITE[dom] * batch / 1e12 if dom in ALG_FLOP_PER_SITE else nbytes / 1e9
            roof["exclusive"] = {"avg_launch_ms": excl[xk], "achieved": unit_work / (excl[xk] * 1e-3), "frac": unit_work / (excl[xk] * 1e-3) / roof["peak"],
                                 "note": "same kernel, same batch, one stream: no other launch shares the chip (28 launches after the timed region)"}
        roof["concurrency"] = "up to %d launches of %d streams share the chip in the timed region; avg_launch_ms is per launch, not exclusive" % (args.hw_queues or 4, S)
        roof["avg_launch_ms"] = avg_ms[dom]
        roof["traffic"] = None
        tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tp):                       # HBM bytes per launch from the committed PMC passes
            try:
                tj = json.load(open(tp))
                if tj.get("batch") == batch and dom in tj.get("kernels", {}):
                    roof["traffic"] = tj["kernels"][dom]["hbm_bytes_per_launch"]
            except Exception:
                pass
        # whole-forward view (all four forward kernels, same events)
        fwd_ms = sum(avg_ms.get(k, 0.0) for k in ALG_FLOP_PER_SITE)      # (either l1f or proj1 + l1 is present)
        out = {
            "metric": METRIC, "value": world * K * batch / dt, "unit": "sites/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "host_issue_ms_per_step": t_issue / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3 (fp32 split into two fp16, 3 MFMAs per product, fp32 accumulate)" if args.precision == 1 else "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: pileup encode + PileupModel fwd, synthetic 30x "
                                   "windows (G2) resident in HBM, batch=4096",
                       "batch": batch, "windows_resident_per_gpu": n_windows, "streams": S,
                       "coverage": args.coverage, "precision": "f16x3" if args.precision == 1 else "fp32",
                       "weights": "ont_pileup.chkpt values (tests/golden fixture)",
                       "parallelism": f"site-sharded x{world}, rooted gather of calls"},
            "roofline": roof,
            "kernel_avg_ms": {k: round(v, 5) for k, v in sorted(avg_ms.items())},
            "kernel_exclusive_ms": {("pileup_l1f" if (k == "pileup_l1" and args.precision == 1 and args.fused_l1) else k): round(v, 5)
                                    for k, v in sorted(excl.items())},
            "kernel_timing": {"streams_with_events": min(S, max(1, args.timing_streams)),
                              "launches_timed": {k: v[1] for k, v in sorted(tot.items()) if v[1]}},
            "forward_alg_tflops": (2 * 6_274_560 * batch / (fwd_ms * 1e-3) / 1e12) if fwd_ms else None,
        }
        if extra:
            out["repeats_before"] = [round(v) for v in extra]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cols, batch, weights, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        assert merged is not None and merged.shape[0] == n_done * world
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
