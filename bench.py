#!/usr/bin/env python3
"""bench.py -- candidate SNP sites/sec (pileup encode + PileupModel forward) on synthetic 30x windows.

Workload = BASELINE.json configs[1]: a pool of 1,048,576 stand-alone 33-column windows (generator G2,
SURVEY.md 8(d)) resident in HBM on every GPU, processed in batches of 4096 windows.  One batch through
the hot path = column encode (mpileup bytes -> int32 [M,18] counts) + PileupModel forward reading the
windows in place (-> softmax probabilities) + argmax/max post-processing.  One *step* = one sweep of the
whole pool (256 batches = 1,048,576 sites), issued round-robin over `--streams` HIP streams (one nsnp_ctx
each); the timed region (W warm-up steps, then exactly K steps between barrier + synchronize) is run
`--repeat` times (default 3) and the line reports the MEDIAN pass, all values under "repeats".  After the
timed passes and outside the clock, the outputs the last fp32 pass left behind are compared with the oracle
("parity_sample"; the process exits non-zero when that fails).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no torchrun environment: this process starts the N ranks itself (a child
`python -m torch.distributed.run`, before anything touches the GPU), relays rank 0's JSON line and exits
with the children's status.  One process per GPU, every rank owns its own pool (weak scaling, no
data-path collective); the only exchange is the rooted gather of the compact per-site calls at the end,
inside the timed region (RCCL over xGMI).  Rank 0 prints ONE JSON line, the LAST line of stdout: a flat object under 4 KB
(tools/bench_common.py::compact_line - the contract's keys, `roofline`, `cpu_baseline`, numbers per sub-workload; exactly one
"metric" key, no prose).  Everything else the run knows - per-kernel tables, notes, the full result object of every sub-workload -
is written to bench_details.json beside this file (path on stderr).

The headline `value` is the exact-fp32 path (the library default, the reference's arithmetic); the bf16x3
mode (three bf16 terms per operand: the full fp32 significand on the bf16 matrix pipe) and the opt-in f16x3
mode are timed afterwards on the same pool and reported under "bf16x3" / "f16x3" in the same line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "candidate SNP sites/sec (pileup encode + model fwd), 30x windows"
N_POOL = 1 << 20


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--windows", type=int, default=N_POOL, help="windows resident per GPU (BASELINE configs[1]: 1M)")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--workload", default="pileup", choices=["pileup", "haplotype", "two-stage", "deep60", "e2e", "hap-e2e", "pd-e2e"],
                    help="pileup = BASELINE configs[1] (the metric's configuration); haplotype = configs[2]: haplotype features + "
                         "HaplotypeModel fwd (+ the legacy crnn.py CatModel fwd) on 150 k G3 sites; two-stage = configs[3]: stage 2 + stage 5 "
                         "on a chr20-sized candidate set, sites sharded over the ranks, gathered calls merged on rank 0; deep60 = configs[4]: "
                         "60x columns + D = 180 read planes + fp16-split conv weights; e2e = a labelled text-to-VCF measurement: samtools-mpileup text of a synthetic contig "
                         "on the page cache -> pinned staging beside H2D beside tokenise + encode + forward on the device -> pileup.vcf (tools/e2e_bench.py); hap-e2e = the same for stage 5: a haplotype "
                         "site file on the page cache -> pinned staging beside H2D beside features + HaplotypeModel fwd -> haplotype.csv (tools/hap_e2e_bench.py); pd-e2e = stage 2 from "
                         ".pd.bin window files: pread + int16 narrowing beside H2D beside PileupModel fwd -> pileup.vcf (tools/pd_e2e_bench.py; one rank)")
    ap.add_argument("--encode-group", type=int, default=32, help="batches encoded per column-encode launch (on the encode stream, into a "
                    "ring of count buffers; the kernel is twice as efficient per byte at >= 1 M columns)")
    ap.add_argument("--hap-sites", type=int, default=0, help="haplotype / deep60 workloads: sites resident per job (0 = the workload's default)")
    ap.add_argument("--hap-batch", type=int, default=16384, help="haplotype / deep60 workloads: sites per step")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (0 = runtime default of 4)")
    ap.add_argument("--precision", type=int, default=0, help="headline arithmetic of the PileupModel forward: 0 exact fp32 MFMA "
                    "(library default), 1 f16x3 split, 2 bf16x3")
    ap.add_argument("--no-second-precision", action="store_true", help="skip the labelled f16x3 pass after the fp32 headline")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="nsnp_ctx_set_option on every context (tuning)")
    ap.add_argument("--repeat", type=int, default=3, help="timed passes of the headline arithmetic; the line reports the median pass")
    ap.add_argument("--no-parity-sample", action="store_true", help="skip the comparison of the run's own outputs with the oracle")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--timing-streams", type=int, default=4, help="record per-kernel HIP events on this many of the streams "
                    "(every kernel launch of those streams inside the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU-baseline sample time")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (the measured configuration). "
                    "gloo + --share-gpu is a TEST configuration: the ranks share GPU 0 and the gathered calls cross host memory, so that the "
                    "whole multi-rank path runs on a one-GPU box (tests/test_gpu_bench_contract.py); its value is not a scaling number")
    ap.add_argument("--gather", default="torch", choices=["torch", "rccl-abi"], help="final merge through one torch.distributed collective "
                    "(default) or through the library's own RCCL entry nsnp_gather_results (validated at world size 1 only)")
    ap.add_argument("--share-gpu", action="store_true", help="test configuration: every rank uses GPU 0 (needs --dist-backend gloo)")
    ap.add_argument("--workloads", default="all", help="pileup workload only: after the headline's timed region, short runs of the other BASELINE "
                    "configurations in the same process, reported under \"workloads\" in the same line (tools/workloads.py): 'all' (default), 'none', or a "
                    "comma-separated subset of haplotype,two_stage,deep60,hap_e2e,e2e,pd_e2e")
    ap.add_argument("--selftest-launcher", action="store_true", help="CPU/gloo dry run of the multi-rank plumbing (spawn, barrier, "
                    "max-over-ranks timing, rooted gather, one JSON line); no kernels, value is null -- tests/test_dist.py")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# launcher: `bench.py --gpus N` without a torchrun environment starts the N ranks itself
# ---------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks(args):
    """Parent of a multi-rank run.  Nothing here imports torch or touches the GPU: the ranks are children of a child
    `python -m torch.distributed.run`, rank 0's JSON line is relayed, the exit status is the children's."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # torch.distributed.run force-sets OMP_NUM_THREADS=1 for nproc > 1 unless the variable is already in its environment; the
    # ranks generate their synthetic pools with OpenMP, so the cores of the box are shared between them explicitly
    env["OMP_NUM_THREADS"] = str(max(1, usable_cores() // max(1, args.gpus)))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in p.stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if p.returncode != 0 or len(lines) != 1:
        print(f"bench.py: {args.gpus}-rank run failed (exit {p.returncode}, {len(lines)} result lines)", file=sys.stderr)
        return p.returncode or 1
    print(lines[0])
    return 0


from tools.bench_common import emit_line, flush_native_stdout, usable_cores  # noqa: E402


def selftest_launcher(args, rank, world):
    """The distributed skeleton of main() on CPU tensors over gloo: same barrier / max-over-ranks timing / rooted gather /
    single JSON line, the timed region sleeps instead of launching kernels."""
    import torch
    import torch.distributed as dist
    from nanosnp_amd.dist import gather_results, shard_range
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    n_total = 1000 * world + 3
    lo, hi = shard_range(n_total, rank, world)
    if world > 1: dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.002 * args.steps * (1 + rank))                     # rank r is r+1 times slower: MAX over ranks must pick the last
    if os.environ.get("NSNP_SELFTEST_DIE_RANK") == str(rank):       # tests/test_dist.py: a rank that dies before the gather must take
        os._exit(7)                                                 # the whole job down with a non-zero status, not leave it hanging
    idx = torch.arange(lo, hi, dtype=torch.float32)
    merged = gather_results(torch.stack([idx, idx * 3], 1), n_total)
    if world > 1: dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        ok = merged.shape[0] == n_total and bool((merged[:, 0] == torch.arange(n_total)).all())
        print(json.dumps({"metric": METRIC, "value": None, "unit": "sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / max(args.steps, 1) * 1e3, "selftest": True, "gather_ok": ok,
                          "world_size_observed": dist.get_world_size() if world > 1 else 1,
                          "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "usable_cores": usable_cores(),
                          "slowest_rank_bound_ok": dt >= 0.002 * args.steps * world}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def cpu_baseline(cols, batch, weights, target_s):
    """The oracle's cache-blocked arrangement of the reference algorithm (full reference schedule, AVX2) on this box's
    host cores, on a bounded sample of the same windows: column encode + forward."""
    from oracle import oracle
    from tools.bench_common import host_cpu_name, reference_cpu
    cores = usable_cores()

    def run(n):
        m = n * 33
        b1 = int(cols.col_off[m])
        t0 = time.perf_counter()
        counts, depth, flags = oracle.encode_columns(cols.bases[:b1], cols.col_off[:m + 1], cols.ref[:m])
        t1 = time.perf_counter()
        oracle.pileup_forward(weights, counts.reshape(n, 33, 18), nthreads=cores, blocked=True)
        t2 = time.perf_counter()
        return t2 - t0, t1 - t0

    n0 = min(2048, batch)
    t, _ = run(n0)
    n = int(min(max(n0, n0 * target_s / max(t, 1e-6)), 1 << 19, cols.n_cols // 33))
    n = max(n0, (n // 64) * 64)
    t, t_enc = run(n)
    out = {"value": n / t, "unit": "sites/s", "cores": cores, "kind": "port",
           "sample": f"{n} of the same synthetic windows: column encode (single thread, {t_enc:.1f} s) + full-schedule fp32 forward "
                     f"blocked for L1 with AVX2 FMA, OpenMP over {cores} threads ({t - t_enc:.1f} s); oracle/liboracle.so",
           "host_cpu": host_cpu_name(), "logical_cpus": os.cpu_count()}
    rj, fwd = reference_cpu("forward")               # the reference itself, timed in the development container (cannot travel)
    if rj and fwd:
        try:
            f64 = next(r for r in fwd if r["batch"] == 64 and r["threads"] == rj["host"]["logical_cpus"])
            out["reference_in_dev_container"] = {
                "value": f64["sites_per_s"], "unit": "sites/s", "cores": f64["threads"], "cpu": rj["host"]["cpu"],
                "what": "the reference's own LSTMNetwork.predict on CPU torch, batch 64, 1,000 windows (BASELINE configs[0]); forward only; "
                        "tests/manual/time_reference_cpu.py"}
        except Exception:
            pass
    return out


def main():
    args = parse_args()
    # an application's choice, made before any OpenMP runtime loads (nanosnp_amd.host.recommend_omp_env): idle OpenMP workers spin briefly,
    # then sleep - measured on the e2e workload under a 16-core quota: always spinning 50 ms per contig, never spinning 30-33, 5k-100k spins 22;
    # the site-file pipelines are throttled by the quota at 100k (idle teams spin 1.6 ms per region) and not at 5k
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("GOMP_SPINCOUNT", "5000")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if rank != 0:
        # torchrun hands every rank the job's stdout: whatever a rank other than 0 (or a native library inside it: RCCL's banner) writes there
        # would land around - or behind - rank 0's line.  Their stdout IS their stderr from here on.
        sys.stdout.flush()
        os.dup2(2, 1)
    if args.selftest_launcher:
        sys.exit(selftest_launcher(args, rank, world))
    if args.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)      # must be set before HIP initialises
    # the host-side generator / CPU baseline use OpenMP: share the cores between the ranks of a node (launch_ranks passes the
    # same value to its children explicitly; an external torchrun that force-set 1 for nproc > 1 is overridden here)
    if world > 1 and os.environ.get("OMP_NUM_THREADS", "1") == "1":
        os.environ["OMP_NUM_THREADS"] = str(max(1, usable_cores() // world))
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // max(1, world))))
    if args.workload == "two-stage":
        from tools.two_stage_bench import run as run_two_stage
        sys.exit(run_two_stage(args, rank, world, local_rank))
    if args.workload == "e2e":
        from tools.e2e_bench import run as run_e2e
        sys.exit(run_e2e(args, rank, world, local_rank))
    if args.workload == "hap-e2e":
        from tools.hap_e2e_bench import run as run_hap_e2e
        sys.exit(run_hap_e2e(args, rank, world, local_rank))
    if args.workload == "pd-e2e":
        from tools.pd_e2e_bench import run as run_pd_e2e
        sys.exit(run_pd_e2e(args, rank, world, local_rank))
    if args.workload in ("haplotype", "deep60"):
        from tools.hap_bench import run as run_hap
        sys.exit(run_hap(args, rank, world, local_rank, deep60=args.workload == "deep60"))

    import torch
    import torch.distributed as dist
    if args.share_gpu:
        if args.dist_backend != "gloo":
            print("bench.py: --share-gpu needs --dist-backend gloo (RCCL refuses two ranks on one device)", file=sys.stderr)
            sys.exit(2)
        local_rank = 0
    elif torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        print(f"bench.py: {world} ranks asked for, {torch.cuda.device_count()} GPUs visible", file=sys.stderr)
        sys.exit(3)
    from nanosnp_amd.dist import gather_results, gather_results_abi
    from tools.pileup_stage import PileupStage, pileup_rooflines

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")       # where collective payloads live

    stage = PileupStage(local_rank, args.windows, batch=args.batch, streams=args.streams, coverage=args.coverage, seed=20260000 + rank,
                        precision=args.precision, opts=args.opt, timing_streams=0 if args.no_kernel_timing else args.timing_streams,
                        enc_group=args.encode_group)
    batch, n_windows, n_batches = stage.batch, stage.n_windows, stage.n_batches
    W, K = max(0, args.warmup), max(1, args.steps)
    bps = n_batches                                     # batches per step: one step sweeps the pool once

    def barrier():
        if world > 1:
            dist.barrier()

    n_done = min(K * bps, n_batches) * batch                # distinct sites whose calls exist after the timed region

    def merge_results():
        """final merge: compact per-site calls of this rank -> rank 0 (RCCL gather over xGMI)"""
        compact = stage.compact_calls(n_done)
        if args.gather == "rccl-abi":
            return gather_results_abi(stage.ctxs[0], compact, n_done * world)
        return gather_results(compact.to(cdev), n_done * world) if world > 1 else compact

    def timed_pass():
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks"""
        stage.run(0, W * bps)
        stage.sync()
        merge_results()                      # warm the merge path (first-use module loads) outside the clock
        stage.sync()
        stage.read_timing()                  # drop warm-up launches
        barrier(); stage.sync()
        t0 = time.perf_counter()
        stage.run(W * bps, K * bps)
        t_issue = time.perf_counter() - t0          # host time to enqueue the K steps (informational)
        stage.sync()
        merged = merge_results()
        stage.sync(); barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, t_issue, merged, stage.read_timing()

    passes = [timed_pass() for _ in range(max(1, args.repeat))]
    order = sorted(range(len(passes)), key=lambda i: passes[i][0])
    dt, t_issue, merged, tot = passes[order[len(order) // 2]]            # the median pass is the one reported
    sites_timed = world * K * bps * batch
    repeats = [sites_timed / p[0] for p in passes]
    # what the last fp32 pass left behind, for the parity sample (compared with the oracle after all timing, outside every clock)
    snap = None
    if rank == 0 and not args.no_parity_sample and args.precision == 0:
        snap = stage.snapshot(stage.parity_ranges(n_done))
    clock_mhz = stage.ctxs[0].shader_clock_mhz(stage.streams[0]) if rank == 0 else None

    # the same kernels with the chip to themselves (one stream, after the timed region)
    excl, excl_n = stage.exclusive_pass()

    # ---- labelled second values on the same pool: bf16x3 (full fp32 operand width on the bf16 pipe) and the opt-in f16x3 ----------
    seconds = {}
    f64_err = {"fp32": stage.float64_error(snap)} if snap is not None else {}
    if not args.no_second_precision and args.precision == 0:
        ref_gt = stage.gt_all[:n_done].clone(); ref_zy = stage.zy_all[:n_done].clone()
        torch.cuda.synchronize(dev)                     # (the copies run on torch's current stream, the forwards on the stage's)
        for prec, label, text in ((2, "bf16x3", "bf16x3 (every fp32 operand as three bf16 terms = its full 24-bit significand and exponent range, "
                                                "six bf16 MFMAs per product, fp32 accumulate)"),
                                  (1, "f16x3", "f16x3 (every fp32 operand split into two fp16, 3 fp16 MFMAs per product, fp32 accumulate; opt-in)")):
            stage.set_precision(prec, fwd_group=int(os.environ.get("NSNP_B3_FWD_GROUP", "4")) if prec == 2 else 1)     # bf16x3: 16,384 sites per forward launch (its kernels fill the chip there)
            dt2, _, _, tot2 = timed_pass()
            d = max((stage.gt_all[:n_done] - ref_gt).abs().max().item(), (stage.zy_all[:n_done] - ref_zy).abs().max().item())
            sv = {"value": sites_timed / dt2, "unit": "sites/s", "ms_per_step": dt2 / K * 1e3, "dtype": text, "sites_per_forward_launch": stage.batch * stage.F,
                  "max_abs_dp_vs_fp32_on_the_pool": d, "tolerance": 1e-4,
                  "kernel_avg_ms_in_region": {k: round(v[0] / v[1], 5) for k, v in sorted(tot2.items())}}
            if rank == 0 and not args.no_parity_sample:
                f64_err[label] = stage.float64_error(stage.snapshot(stage.parity_ranges(n_done, per_batch=64), ring_batches=0))
            if prec == 2:
                # its own rooflines against the dense bf16 MFMA peak, the six MFMAs of a product priced as executed, and its own
                # outputs against the oracle
                snap2 = stage.snapshot(stage.parity_ranges(n_done, per_batch=512)) if (rank == 0 and not args.no_parity_sample) else None
                excl2, excl2_n = stage.exclusive_pass()
                if rank == 0:
                    pr = pileup_rooflines(stage, tot2, excl2, excl2_n, sites_timed / world, dt2, 2, stage.G) if tot2 else {}
                    sv["roofline"] = pr.get("roofline"); sv["roofline_other_layer"] = next((v for k, v in pr.items() if k.startswith("roofline_pileup_l")), None)
                    sv["kernel_exclusive_ms"] = {k: round(v, 5) for k, v in sorted(excl2.items())}
                    sv["parity_sample"] = stage.parity_check(snap2) if snap2 is not None else None
            seconds[label] = sv
        stage.set_precision(0, fwd_group=1)
        del ref_gt, ref_zy

    # ---- the other BASELINE configurations, short runs in this process (every rank takes part: their merges are collectives) ----
    sub, sub_ok = {}, True
    if args.workloads != "none":
        from tools.workloads import PLAN, run_all
        only = None if args.workloads == "all" else set(args.workloads.split(","))
        if only is not None and not only <= {p[0] for p in PLAN}:
            print(f"bench.py: --workloads: unknown name in {sorted(only)}", file=sys.stderr)
            sys.exit(2)
        sub, sub_ok = run_all(args, rank, world, local_rank, only)

    # every rank empties its native stdout buffers (RCCL's banner) BEFORE rank 0 prints: the line is the last thing on the job's stdout
    flush_native_stdout()
    if world > 1:
        dist.barrier()
    exit_code = 0
    if rank == 0:
        out = {
            "metric": METRIC, "value": sites_timed / dt, "unit": "sites/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {0: "f32", 1: "f16x3 (fp32 split into two fp16, 3 MFMAs per product, fp32 accumulate)",
                      2: "bf16x3 (fp32 split into three bf16 = 24 significand bits, 6 MFMAs per product, fp32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: pileup encode + PileupModel fwd, 1M synthetic 30x windows (G2) resident in HBM, batch=4096",
                       "batch": batch, "windows_resident_per_gpu": n_windows, "batches_per_step": bps, "sites_per_step": bps * batch,
                       "streams": stage.S, "encode_batches_per_launch": stage.G, "coverage": args.coverage,
                       "precision": {0: "fp32", 1: "f16x3", 2: "bf16x3"}[args.precision],
                       "weights": "ont_pileup.chkpt values (nanosnp_amd/data/ont_pileup_weights.npz)",
                       "parallelism": f"site-sharded x{world}, rooted gather of calls",
                       "world_size_observed": dist.get_world_size() if world > 1 else 1, "gather": args.gather,
                       **({"TEST_CONFIGURATION": "ranks share GPU 0, gather over gloo: not a scaling number"} if args.share_gpu else {})},
            "host_issue_ms_per_step": t_issue / K * 1e3,
        }
        if tot:
            out.update(pileup_rooflines(stage, tot, excl, excl_n, sites_timed / world, dt, args.precision, stage.G))
            out.setdefault("roofline", None)
            out["kernel_avg_ms_in_region"] = {k: round(v[0] / v[1], 5) for k, v in sorted(tot.items())}
            out["kernel_exclusive_ms"] = {k: round(v, 5) for k, v in sorted(excl.items())}
            out["kernel_exclusive_note"] = ("HIP-event time per launch with the chip to the kernel (one stream, after the timed region); ~3 % above the "
                                            "rocprofv3 kernel-trace durations of the same launches (event overhead), and NOT additive to ms_per_step: in "
                                            "the timed region the launches of 32 streams overlap each other's tails")
            out["kernel_timing"] = {"streams_with_events": stage.timed_streams, "launches_timed_in_region": {k: v[1] for k, v in sorted(tot.items())}}
        else:
            out["roofline"] = None
        out.update(seconds)
        if f64_err:
            out["error_vs_float64"] = {"modes": f64_err, "what": "max |p - p64| on the same windows, p64 = LSTMNetwork.predict evaluated in float64 (numpy, "
                                       "oracle.pileup_forward_f64): the fp32 MFMA path's own distance from exact arithmetic is the yardstick for the split modes"}
        out["repeats"] = {"values": [round(v) for v in repeats], "reported": "median", "spread": (max(repeats) - min(repeats)) / out["value"]}
        out["timed_region_s"] = dt
        out["shader_clock_mhz"] = {"value": clock_mhz, "how": "s_memtime / s_memrealtime in every workgroup of a ~2 ms full-chip fp32 MFMA "
                                   "probe right after the timed passes (nsnp_ctx_shader_clock); the MFMA peaks are priced at 2400"}
        if snap is not None:
            out["parity_sample"] = stage.parity_check(snap)
        else:
            out["parity_sample"] = None
        out["cpu_baseline"] = cpu_baseline(stage.cols, batch, stage.weights, args.cpu_seconds) if (not args.no_cpu_baseline and world == 1) else None
        assert merged is not None and merged.shape[0] == n_done * world
        if sub:
            out["workloads"] = sub
            out["workloads_note"] = ("short runs of the other BASELINE configurations in this process after the headline's timed region "
                                     "(tools/workloads.py); each is the full result object of `bench.py --workload NAME`; `value` above is configs[1] alone")
        emit_line(out)                                # the driver's line (flat, < 4 KB) last on stdout; the whole object -> bench_details.json
        for who in (out, out.get("bf16x3") or {}):
            if who.get("parity_sample") is not None and not who["parity_sample"]["ok"]:
                print("bench.py: parity_sample FAILED: " + json.dumps(who["parity_sample"]), file=sys.stderr)
                exit_code = 1
    if not sub_ok:
        exit_code = 1                                # (a sub-workload's parity sample failed or it raised: its own message is on stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
