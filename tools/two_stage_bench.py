#!/usr/bin/env python3
"""bench.py --workload two-stage: BASELINE configs[3], "full two-stage (s2 + s5) on a synthetic chr20-sized candidate set,
sites sharded across the GPUs, RCCL gather" (run_caller.sh:109-136).

    stage 2  1,500,000 candidate windows (chr20: 64.4 Mbp x 2.36 % candidates): column encode + PileupModel
             forward (exact fp32) + argmax/max, batches of 4096 windows
    stage 5    150,000 low-confidence sites (10 % of stage 2; the reference gives no ratio): read planes (generator G3, int8,
             resident in HBM) -> haplotype features -> HaplotypeModel forward (exact fp32) -> argmax/max, batches of 4096 sites
    merge    compact per-site calls of both stages gathered to rank 0 in site order (one rooted collective each)

The TOTAL work is fixed and statically sharded by nanosnp_amd.dist.shard_range (strong scaling); a *step* is one sweep of a
rank's shard of both stages.  Text output (pileup.vcf / haplotype.csv / merged VCF) is host work outside the metric
(SURVEY.md 8(d)): tests/test_gpu_two_stage.py covers it against the reference, tools/two_stage_probe.py times it.
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


def run(args, rank, world, local_rank):
    import numpy as np
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
    from nanosnp_amd import _lib, host
    from nanosnp_amd.dist import gather_results, shard_range
    from nanosnp_amd.fixtures import load_pileup_weights, seeded_hap_weights
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    batch = args.batch
    n2_tot = int(os.environ.get("NSNP_TWO_STAGE_N2", N_STAGE2)); n5_tot = int(os.environ.get("NSNP_TWO_STAGE_N5", N_STAGE5))
    lo2, hi2 = shard_range(n2_tot, rank, world); n2 = hi2 - lo2
    lo5, hi5 = shard_range(n5_tot, rank, world); n5 = hi5 - lo5

    # ---- inputs of this rank's shard, resident in HBM before the clock starts ----
    cols = host.synth_columns(20260300 + rank, max(n2, 1) * 33, coverage=args.coverage, window=33)
    d_bases = torch.from_numpy(cols.bases).to(dev); d_off = torch.from_numpy(cols.col_off).to(dev); d_ref = torch.from_numpy(cols.ref).to(dev)
    planes = []
    for L, seed in ((33, 20260400), (11, 20260500)):
        ps = [[], [], [], [], []]
        for c0 in range(0, max(n5, 1), 16384):                      # generated in chunks: the int32 planes of 150k sites would be 2 x 9.5 GB of host memory
            pl = host.synth_hap_planes(seed + 97 * rank + c0, min(16384, max(n5, 1) - c0), 30, 90, L)
            for k in range(4):
                ps[k].append(torch.from_numpy(pl[k].astype(np.int8)).to(dev))
            ps[4].append(torch.from_numpy(pl[4]).to(dev))
        planes.append([torch.cat(p) for p in ps])
    S = max(1, min(args.streams, 8))
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    w_pile = load_pileup_weights(); w_hap = seeded_hap_weights(12, H=256)
    ctxs = []
    for s in range(S):
        c = _lib.Context(local_rank, chunk_sites=batch)
        c.pileup_load_weights(w_pile); c.hap_load_weights(w_hap)
        ctxs.append(c)
    centers = (torch.arange(batch, dtype=torch.int64, device=dev) * 33 + 16).contiguous()
    res2 = torch.empty((max(n2, 1), 4), dtype=torch.float32, device=dev)
    res5 = torch.empty((max(n5, 1), 2), dtype=torch.float32, device=dev)

    def stage2():
        for i, b0 in enumerate(range(0, n2, batch)):
            b1 = min(n2, b0 + batch); s = i % S
            with torch.cuda.stream(streams[s]):
                c0, c1 = b0 * 33, b1 * 33
                base0 = int(cols.col_off[c0])
                counts, depth, flags = ctxs[s].pileup_encode_columns(d_bases[base0:], d_off[c0:c1 + 1] - base0, d_ref[c0:c1], stream=streams[s])
                gt, zy = ctxs[s].pileup_forward_windows(counts, centers[:b1 - b0], stream=streams[s])
                ga, za, gm, zm, _ = ctxs[s].pileup_postprocess(gt, zy, stream=streams[s])
                res2[b0:b1] = torch.stack([ga.float(), za.float(), gm, zm], 1)

    def stage5():
        for i, b0 in enumerate(range(0, n5, batch)):
            b1 = min(n5, b0 + batch); s = i % S
            with torch.cuda.stream(streams[s]):
                xp = ctxs[s].hap_features(*[p[b0:b1] for p in planes[0]], stream=streams[s])
                xh = ctxs[s].hap_features(*[p[b0:b1] for p in planes[1]], stream=streams[s])
                gt, _ = ctxs[s].hap_forward(xp, xh, stream=streams[s])
                gm, ga = gt.max(dim=1)
                res5[b0:b1] = torch.stack([ga.float(), gm], 1)

    def sync_all():
        for st in streams:
            st.synchronize()
        torch.cuda.synchronize(dev)

    def merge():
        a = gather_results(res2[:n2].to(cdev), n2_tot) if world > 1 else res2[:n2]
        b = gather_results(res5[:n5].to(cdev), n5_tot) if world > 1 else res5[:n5]
        return a, b

    def barrier():
        if world > 1:
            dist.barrier()

    W, K = max(1, args.warmup), max(1, args.steps)
    for _ in range(W):
        stage2(); stage5(); sync_all(); merge(); sync_all()
    barrier(); sync_all()
    t0 = time.perf_counter()
    t2 = t5 = 0.0
    for _ in range(K):
        ta = time.perf_counter(); stage2(); sync_all()
        tb = time.perf_counter(); stage5(); sync_all()
        tc = time.perf_counter(); merged = merge(); sync_all()
        t2 += tb - ta; t5 += tc - tb
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([dt, t2, t5], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt, t2, t5 = (float(v) for v in tm.tolist())
    if rank == 0:
        assert merged[0].shape[0] == n2_tot and merged[1].shape[0] == n5_tot
        print(json.dumps({
            "metric": "candidate SNP sites/sec, two-stage (s2 pileup + s5 haplotype) on a chr20-sized synthetic candidate set",
            "value": n2_tot * K / dt, "unit": "sites/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: 1.5 M stage-2 windows (encode + PileupModel fwd) + 150 k stage-5 sites (haplotype "
                                   "features + HaplotypeModel fwd, seeded weights), sites sharded over the ranks, calls gathered to rank 0",
                       "stage2_sites": n2_tot, "stage5_sites": n5_tot, "batch": batch, "streams": S,
                       "parallelism": f"site-sharded x{world}, rooted gathers of calls", "world_size_observed": dist.get_world_size() if world > 1 else 1},
            "stage2": {"ms_per_step": t2 / K * 1e3, "sites_per_s": n2_tot * K / t2},
            "stage5": {"ms_per_step": t5 / K * 1e3, "sites_per_s": n5_tot * K / t5,
                       "algorithmic_tflops": 353.7e6 * n5_tot * K / t5 / 1e12 / world, "peak_tflops_f32_mfma": 157.3,
                       "note": "per GPU; includes the feature reduction and the int8 -> fp32 feature write (HBM-bound part)"},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0
