#!/usr/bin/env python3
"""bench.py --workload two-stage: BASELINE configs[3], "full two-stage (s2 + s5) on a synthetic chr20-sized candidate set,
sites sharded across the GPUs, RCCL gather" (run_caller.sh:109-136).

    stage 2  1,500,000 candidate windows (chr20: 64.4 Mbp x 2.36 % candidates): column encode + PileupModel
             forward (exact fp32) + argmax/max, batches of 4096 windows (tools/pileup_stage.py: encode of 8 batches per launch on its
             own stream, forwards on 32 streams)
    stage 5    150,000 low-confidence sites (10 % of stage 2; the reference gives no ratio): the reference's int32 read planes
             (generator G3) resident in HBM -> haplotype features -> HaplotypeModel forward (exact fp32) -> argmax/max, passes of
             16384 sites (tools/hap_bench.py HapStage)
    merge    compact per-site calls of both stages gathered to rank 0 in site order (one rooted transfer batch each)

The TOTAL work is fixed and statically sharded by nanosnp_amd.dist.shard_range (strong scaling; a rank's stage-2 shard is cut to
whole batches of 4096 windows, the line reports the windows actually processed); a *step* is one sweep of a rank's shard of
both stages.  Text output (pileup.vcf / haplotype.csv / merged VCF) is host work outside the metric:
tests/test_gpu_two_stage.py covers it against the reference, tools/probes/two_stage_probe.py times it.
HaplotypeModel weights are seeded (the trained ones are absent upstream)."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_STAGE2 = 1_500_000
N_STAGE5 = 150_000


def run(args, rank, world, local_rank, emit=None):
    import torch
    import torch.distributed as dist
    if args.share_gpu:
        if args.dist_backend != "gloo":
            print("bench.py: --share-gpu needs --dist-backend gloo", file=sys.stderr)
            return 2
        local_rank = 0
    elif torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        print(f"bench.py: {world} ranks asked for, {torch.cuda.device_count()} GPUs visible", file=sys.stderr)
        return 3
    from nanosnp_amd.dist import gather_results, gather_varlen, shard_range
    from tools import bench_common as bc
    from tools.hap_bench import HapStage, cpu_baseline_hap, hap_rooflines
    from tools.pileup_stage import PileupStage, pileup_rooflines
    if world > 1 and emit is None:                      # (embedded in the default bench line: the process group exists already)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    batch = args.batch
    n2_tot = int(os.environ.get("NSNP_TWO_STAGE_N2", getattr(args, "two_stage_n2", 0) or N_STAGE2))
    n5_tot = int(os.environ.get("NSNP_TWO_STAGE_N5", getattr(args, "two_stage_n5", 0) or N_STAGE5))
    lo2, hi2 = shard_range(n2_tot, rank, world)
    lo5, hi5 = shard_range(n5_tot, rank, world); n5 = hi5 - lo5

    # ---- inputs of this rank's shard, resident in HBM before the clock starts ----
    ps = PileupStage(local_rank, max(hi2 - lo2, batch), batch=batch, streams=args.streams, coverage=args.coverage, seed=20260300 + rank,
                     enc_group=args.encode_group)
    n2 = ps.n_windows                                       # whole batches
    hs = HapStage(local_rank, max(n5, 1), min(args.hap_batch, max(n5, 1)), 30.0, 90, 20260400 + 1000 * rank)

    def stage2():
        ps.run(0, ps.n_batches)

    def stage5():
        for i in range(hs.n_batches):
            hs.run_batch(i)

    def sync_all():
        ps.sync(); hs.sync()
        torch.cuda.synchronize(dev)

    def merge():
        a = ps.compact_calls(n2)
        if world == 1:
            return a, hs.res
        return gather_varlen(a.to(cdev)), gather_results(hs.res[:n5].to(cdev), n5_tot)

    def barrier():
        if world > 1:
            dist.barrier()

    W, K = max(1, args.warmup), max(1, args.steps)
    for _ in range(W):
        stage2(); sync_all(); stage5(); sync_all(); merge(); sync_all()
    ps.read_timing(); hs.ctx.read_timing()
    barrier(); sync_all()
    t0 = time.perf_counter()
    t2 = t5 = 0.0
    hs.sites_in_chain = 0
    for _ in range(K):
        ta = time.perf_counter(); stage2(); sync_all()
        tb = time.perf_counter(); stage5(); sync_all()
        tc = time.perf_counter(); merged = merge(); sync_all()
        t2 += tb - ta; t5 += tc - tb
        hs.sites_in_chain += n5
    barrier()
    dt = time.perf_counter() - t0
    n2_all = n2
    if world > 1:
        tm = torch.tensor([dt, t2, t5], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt, t2, t5 = (float(v) for v in tm.tolist())
        cnt = torch.tensor([n2], dtype=torch.int64, device=cdev)
        dist.all_reduce(cnt)
        n2_all = int(cnt.item())
    ptot = ps.read_timing()
    tim = hs.ctx.read_timing()
    exit_code = 0
    parity = None
    if rank == 0 and not args.no_parity_sample:          # what the last timed step left behind (compared with the oracle below, outside every clock)
        parity = (ps.snapshot(ps.parity_ranges(n2, per_batch=1024, n_ranges=32)),
                  hs.snapshot(range(hs.n_batches), per_batch=max(64, 1024 // hs.n_batches)))
    # ---- labelled second value: both stages in the bf16x3 arithmetic (full fp32 operand width on the bf16 matrix pipe) ----
    second = None
    if not args.no_second_precision:
        ref2 = ps.gt_all[:n2].clone(); ref5 = hs.gt[:max(n5, 1)].clone()
        torch.cuda.synchronize(dev)
        ps.set_precision(2); hs.ctx.set_option("hap_precision", 2)
        stage2(); sync_all(); stage5(); sync_all(); merge(); sync_all()
        barrier(); sync_all()
        tb0 = time.perf_counter(); tb2 = tb5 = 0.0
        for _ in range(K):
            ta = time.perf_counter(); stage2(); sync_all()
            tb = time.perf_counter(); stage5(); sync_all()
            tc = time.perf_counter(); merge(); sync_all()
            tb2 += tb - ta; tb5 += tc - tb
        barrier()
        dtb = time.perf_counter() - tb0
        if world > 1:
            tm = torch.tensor([dtb, tb2, tb5], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dtb, tb2, tb5 = (float(v) for v in tm.tolist())
        second = {"value": n2_all * K / dtb, "unit": "sites/s", "ms_per_step": dtb / K * 1e3,
                  "dtype": "bf16x3 (every fp32 operand as three bf16 terms = 24 significand bits, six bf16 MFMAs per product, fp32 accumulate)",
                  "stage2_ms_per_step": tb2 / K * 1e3, "stage5_ms_per_step": tb5 / K * 1e3,
                  "max_abs_dp_vs_fp32": {"stage2": float((ps.gt_all[:n2] - ref2).abs().max().item()),
                                         "stage5": float((hs.gt[:max(n5, 1)] - ref5).abs().max().item())}, "tolerance": 1e-4}
        if rank == 0 and not args.no_parity_sample:
            second["parity_sample"] = {"stage2": ps.parity_check(ps.snapshot(ps.parity_ranges(n2, per_batch=256, n_ranges=16))),
                                       "stage5": hs.parity_check(hs.snapshot(range(hs.n_batches), per_batch=max(32, 256 // hs.n_batches)))}
            second["parity_sample"]["ok"] = all(v["ok"] for v in second["parity_sample"].values())
        ps.set_precision(0); hs.ctx.set_option("hap_precision", 0)
        del ref2, ref5
    if rank == 0:
        assert merged[0].shape[0] == n2_all and merged[1].shape[0] == (n5_tot if world > 1 else hs.n)
        # rooflines: the dominant kernel of the whole job is the HaplotypeModel's fused step launch (80 % of the time)
        chain_ms, chain_n = tim["hap_lstm_chain"]
        b0, b1 = hs.batch_range(0)
        feat_ms, feat_n = hs.features_alone(b0, b1)      # (cleared first: the region's launches are L = 33 AND L = 11)
        roofs = hap_rooflines(hs, chain_ms, chain_n, feat_ms, feat_n, b1 - b0, "two-stage")
        excl, excl_n = ps.exclusive_pass(groups=2)
        pr = pileup_rooflines(ps, ptot, excl, excl_n, n2 * K, t2, 0, ps.G, "two-stage")
        roofs["roofline_stage2"] = pr.get("roofline")
        roofs["roofline_encode"] = pr.get("roofline_encode")
        out = {
            "metric": "candidate SNP sites/sec, two-stage (s2 pileup + s5 haplotype) on a chr20-sized synthetic candidate set",
            "value": n2_all * K / dt, "unit": "sites/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: %s stage-2 windows (encode + PileupModel fwd) + %s stage-5 sites (haplotype "
                                   "features on int32 read planes + HaplotypeModel fwd, seeded weights), sites sharded over the ranks, calls gathered to rank 0"
                                   % (("1.5 M", "150 k") if (n2_tot, n5_tot) == (N_STAGE2, N_STAGE5) else (n2_tot, n5_tot)),
                       **({"REDUCED_POOL": "a short run inside the default bench line: the configuration is 1.5 M + 150 k sites"} if n2_tot < N_STAGE2 else {}),
                       "stage2_sites": n2_all, "stage2_sites_nominal": n2_tot, "stage5_sites": n5_tot, "batch": batch, "streams": ps.S,
                       "encode_batches_per_launch": ps.G, "hap_sites_per_pass": hs.batch,
                       "parallelism": f"site-sharded x{world}, rooted gathers of calls", "world_size_observed": dist.get_world_size() if world > 1 else 1,
                       **({"TEST_CONFIGURATION": "ranks share GPU 0, gather over gloo: not a scaling number"} if args.share_gpu else {})},
            "stage2": {"ms_per_step": t2 / K * 1e3, "sites_per_s": n2_all * K / t2,
                       "executed_tflops_per_gpu": bc.PILEUP_EXEC_FLOP_FORWARD * n2 * K / t2 / 1e12,
                       "frac_of_fp32_mfma_peak": bc.PILEUP_EXEC_FLOP_FORWARD * n2 * K / t2 / 1e12 / bc.PEAK_F32_MFMA_TFLOPS},
            "stage5": {"ms_per_step": t5 / K * 1e3, "sites_per_s": n5_tot * K / t5,
                       "executed_tflops_per_gpu": bc.hap_exec_flop() * n5 * K / t5 / 1e12,
                       "frac_of_fp32_mfma_peak": bc.hap_exec_flop() * n5 * K / t5 / 1e12 / bc.PEAK_F32_MFMA_TFLOPS,
                       "algorithmic_tflops_per_gpu": bc.HAP_ALG_FLOP * n5 * K / t5 / 1e12,
                       "note": "per GPU; the time includes the feature reduction (HBM-bound) and the argmax; the algorithmic figure prices the "
                               "reference's 353.7 MFLOP/site and is not a fraction of the peak"},
        }
        out.update(roofs)
        out.setdefault("roofline", None)
        if second:
            out["bf16x3"] = second
        out["timed_region_s"] = dt
        out["shader_clock_mhz"] = {"value": hs.ctx.shader_clock_mhz(hs.stream), "how": "s_memtime / s_memrealtime in every workgroup of a ~2 ms full-chip "
                                   "fp32 MFMA probe after the timed region (nsnp_ctx_shader_clock); the MFMA peaks are priced at 2400"}
        out["parity_sample"] = None
        if parity is not None:
            par = {"stage2": ps.parity_check(parity[0]), "stage5": hs.parity_check(parity[1])}
            par["ok"] = par["stage2"]["ok"] and par["stage5"]["ok"]
            par["tolerance"] = 1e-4
            out["parity_sample"] = par
        out["cpu_baseline"] = None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_two_stage(ps, hs, args.cpu_seconds)
        if emit is not None:
            emit(out)
        else:
            bc.emit_line(out, "two_stage")
        if (out["parity_sample"] is not None and not out["parity_sample"]["ok"]) or (second and second.get("parity_sample") and not second["parity_sample"]["ok"]):
            print("bench.py: parity_sample FAILED: " + json.dumps(out["parity_sample"]), file=sys.stderr)
            exit_code = 1
    if world > 1:
        dist.barrier()
        if emit is None:
            dist.destroy_process_group()
    return exit_code


def cpu_baseline_two_stage(ps, hs, target_s):
    """both stages through the oracle on a bounded sample in the workload's 10 : 1 proportion"""
    from oracle import oracle
    from tools import bench_common as bc
    cores = bc.usable_cores()

    def run(nb):
        na = 10 * nb
        m = na * 33; b1 = int(ps.cols.col_off[m])
        t0 = time.perf_counter()
        counts, _, _ = oracle.encode_columns(ps.cols.bases[:b1], ps.cols.col_off[:m + 1], ps.cols.ref[:m])
        oracle.pileup_forward(ps.weights, counts.reshape(na, 33, 18), nthreads=cores, blocked=True)
        t1 = time.perf_counter()
        pp = [p[:nb].cpu().numpy() for p in hs.planes[0]]; ph = [p[:nb].cpu().numpy() for p in hs.planes[1]]
        xp = oracle.hap_features_batch(*pp, nthreads=cores); xh = oracle.hap_features_batch(*ph, nthreads=cores)
        oracle.hap_forward(hs.weights, xp, xh, nthreads=cores)
        return na, (t1 - t0, time.perf_counter() - t1)

    nb0 = max(cores, 32)
    if hs.n < nb0 or ps.n_windows < 10 * nb0:
        return None
    _, ts = run(nb0)
    nb = int(min(max(nb0, nb0 * target_s / max(sum(ts), 1e-6)), 4096, hs.n, ps.n_windows // 10))
    na, ts = run(nb)
    out = {"value": na / sum(ts), "unit": "sites/s", "cores": cores, "kind": "port",
           "sample": f"{na} stage-2 windows (encode + full-schedule forward, {ts[0]:.1f} s) + {nb} stage-5 sites (features + HaplotypeModel forward, "
                     f"{ts[1]:.1f} s), the workload's 10 : 1 proportion, OpenMP over {cores} threads; oracle/liboracle.so",
           "host_cpu": bc.host_cpu_name(), "logical_cpus": os.cpu_count()}
    return out
