#!/usr/bin/env python3
"""bench.py --workload pd-e2e: stage 2 FROM SITE FILES - `.pd.bin` window matrices to pileup.vcf, the path of
PileupModel/predict.py:37-195 over PredictDataset files (dataset.py:118-149: the whole HDF5 array into numpy, a Python loop over the
position strings, a DataLoader with four workers, a blocking .to(device) per batch of 1,000, a Python loop per site).  A labelled
measurement, never the headline `value` (BASELINE's metric is quoted with inputs in HBM; the text path `--workload e2e` is the
designed replacement of stages s1 + s2 - this one serves callers that keep the reference's intermediate files).

    NSNP_PDE2E_SITES windows (default 524,288; G2, 30x: encoded and gathered on the device once, written as a site file on the page
    cache with int16 counts, sitefile.write_pileup_bin's default) -> nanosnp_amd.pipeline.predict_pileup_bins: passes of 65,536
    windows, pread into pinned buffers on all host cores beside H2D on a copy stream beside PileupModel forward (fp32) whose heads
    kernel writes argmax / max into pinned host memory (the coverage slice is taken from the staged pass) ->
    nsnp_vcf_format_batches -> pileup.vcf written.  One *step* = the whole file; the K timed steps are K files of one run.

Beside `value`: the same windows as a reference-layout file (int32 counts: twice the page-cache bytes, narrowed to int16 while
staged), that file sent as int32 (PCIe: 2,376 B per site), and the HBM-RESIDENT rate of the same forward + calls.
parity_sample = the VCF of the timed run byte-identical to the one-pass run and to the runs from the int32 file."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_SITES = 524_288


def run(args, rank, world, local_rank, emit=None):
    created = []
    try:
        return _run(args, rank, world, local_rank, emit, created)
    finally:
        if world > 1 and emit is None:
            import torch.distributed as dist
            if dist.is_initialized():
                try:
                    dist.barrier()                       # nobody removes a file another rank still reads
                    dist.destroy_process_group()
                except Exception:
                    pass
        for pth in created:
            try:
                if os.path.exists(pth):
                    os.remove(pth)
            except OSError:
                pass


def _run(args, rank, world, local_rank, emit, created):
    import numpy as np
    import torch
    import torch.distributed as dist
    if args.share_gpu:
        local_rank = 0                               # TEST configuration: every rank on GPU 0 (one-GPU boxes), collectives over gloo
    if world > 1 and emit is None and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    cdev = torch.device("cuda", local_rank) if (world > 1 and args.dist_backend == "nccl") else torch.device("cpu")
    from nanosnp_amd import host, sitefile
    from nanosnp_amd.fixtures import load_pileup_weights
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import predict_pileup_bins
    from tools import bench_common as bc
    from tools.hap_e2e_bench import scratch_dir
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n = int(os.environ.get("NSNP_PDE2E_SITES", getattr(args, "pd_sites", 0) or N_SITES))
    P = 65536
    weights = load_pileup_weights()
    model = LSTMNetwork(device=local_rank).load_weight_list(weights)
    ctx = model.ctx
    tmp = scratch_dir(n * (2376 + 1188) * 2)
    path = os.path.join(tmp, f"nsnp_pde2e_{n}.pd.bin")
    path32 = os.path.join(tmp, f"nsnp_pde2e_{n}_int32.pd.bin")
    if rank == 0:
        created += [path, path32]
    # ---- the site file: G2 windows encoded + gathered on the device, chunk by chunk (rank 0 writes, every rank reads the page cache) ----
    t_gen = time.perf_counter()
    cols_keep = None
    if rank == 0:
        cols_keep = _make_files(args, ctx, dev, n, path, path32)
    if world > 1:
        dist.barrier()
    t_gen = time.perf_counter() - t_gen
    fai = "chrP\t%d\t6\t60\t61\n" % (n * 40 + 100)
    out_path = os.path.join(tmp, f"nsnp_pde2e_{rank}.vcf")
    created.append(out_path)
    W, K = max(1, args.warmup), max(1, args.steps)
    return _measure(args, rank, world, local_rank, emit, created, model, ctx, dev, cdev, n, P, path, path32, out_path, fai, weights, cols_keep, t_gen, tmp, W, K)


def _make_files(args, ctx, dev, n, path, path32):
    import numpy as np
    import torch
    from nanosnp_amd import host, sitefile
    chunk = 65536
    position = np.char.add(np.char.add("chrP:", (np.arange(n) * 40 + 17).astype(str)), ":" + "N" * 16 + "A" + "N" * 16).astype("S83")
    pf = np.frombuffer(position.tobytes(), np.uint8).reshape(n, 83)
    maps = sitefile.create_arrays(path, {"position_matrix": (np.int16, (n, 33, 18)), "position": (np.uint8, (n, 83))})
    maps32 = sitefile.create_arrays(path32, {"position_matrix": (np.int32, (n, 33, 18)), "position": (np.uint8, (n, 83))})
    cols_keep = None
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        cols = host.synth_columns(20260800 + c0, m * 33, coverage=args.coverage, window=33)
        counts, _, _ = ctx.pileup_encode_columns(torch.from_numpy(cols.bases).to(dev), torch.from_numpy(cols.col_off).to(dev), torch.from_numpy(cols.ref).to(dev))
        x = ctx.pileup_gather_windows(counts, torch.arange(m, dtype=torch.int64, device=dev) * 33 + 16)
        xh = x.cpu().numpy()
        if host.stage_values(maps["position_matrix"][c0:c0 + m].reshape(-1), xh.size, src=xh.reshape(-1)):
            raise RuntimeError("a synthetic count beyond int16")
        maps32["position_matrix"][c0:c0 + m] = xh
        if c0 == 0:
            cols_keep = cols
    maps["position"][:] = pf; maps32["position"][:] = pf
    for a in list(maps.values()) + list(maps32.values()):
        a.flush()
    del maps, maps32
    return cols_keep


def _measure(args, rank, world, local_rank, emit, created, model, ctx, dev, cdev, n, P, path, path32, out_path, fai, weights, cols_keep, t_gen, tmp, W, K):
    import numpy as np
    import torch
    import torch.distributed as dist
    from nanosnp_amd import host, sitefile
    from nanosnp_amd.pipeline import predict_pileup_bins
    from tools import bench_common as bc

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(steps, narrow=True, pass_sites=P, path=path):
        for _ in range(W):                           # a warm-up run is a few files long: the HIP runtime opens its SDMA copy engines one by one on
            # first use (6-8 ms each inside hipMemcpyAsync: docs/rounds/r05.md), a once-per-process cost that a one-file run does not reach
            predict_pileup_bins(model, [path] * min(4, steps), fai, out_path, pass_sites=pass_sites, narrow=narrow)
        torch.cuda.synchronize(dev)
        if os.path.exists(out_path):
            os.remove(out_path)                      # (truncating the previous run's half gigabyte of tmpfs pages is not part of a run)
        bc.settle_collector()
        barrier()
        st = {"trace": []} if os.environ.get("NSNP_PD_TRACE") == "1" else {}
        c0 = bc.cgroup_cpu_stat()
        prof = None
        if os.environ.get("NSNP_PD_CPROFILE") == "1":      # development aid: where the main thread spends the run (stderr)
            import cProfile
            prof = cProfile.Profile()
        clk0 = bc.clocks_ns()
        t0 = time.perf_counter()
        if prof:
            prof.enable()
        predict_pileup_bins(model, [path] * steps, fai, out_path, pass_sites=pass_sites, narrow=narrow, stats=st)
        t_ret = time.perf_counter()
        torch.cuda.synchronize(dev); barrier()
        dt = time.perf_counter() - t0
        bc.mark_region(clk0, bc.clocks_ns(), steps, {"workload": "pd_e2e", "stats": {k: v for k, v in st.items() if isinstance(v, (int, float))}})   # (the first timed run only)
        if "trace" in st:                                    # development aid: where the issuing thread spends a pass (stderr)
            tr = st.pop("trace")
            sys.stderr.write("pd trace, ms per pass: wait for staging %.2f, H2D issue %.2f, wait for the set two ahead + submit %.2f, compute issue %.2f (%d passes)\n" % (
                sum(t[3] - t[2] for t in tr) / len(tr) * 1e3, sum(t[4] - t[3] for t in tr) / len(tr) * 1e3, sum(t[5] - t[4] for t in tr) / len(tr) * 1e3,
                sum(t[6] - t[5] for t in tr) / len(tr) * 1e3, len(tr)))
            sys.stderr.write("  per pass compute issue ms: " + " ".join("%.1f" % ((t[6] - t[5]) * 1e3) for t in tr[:40]) + "\n")
            sys.stderr.write("  per pass H2D issue ms: " + " ".join("%.1f" % ((t[4] - t[3]) * 1e3) for t in tr[:40]) + "\n")
        if prof:
            import pstats
            prof.disable()
            sys.stderr.write("predict_pileup_bins returned after %.1f ms, timed region %.1f ms\n" % ((t_ret - t0) * 1e3, dt * 1e3))
            pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(18)
        if world > 1:
            tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt = float(tm.item())
        if world > 1:
            mine = {k: round(st.get(k, 0.0) / steps, 4) for k in ("stage_s", "wait_stage_s", "h2d_s", "gpu_s", "issue_s", "drain_s", "vcf_s", "gather_s")}
            mine["rank"] = rank
            per = [None] * world
            dist.all_gather_object(per, mine)
            st["per_rank"] = per
        c1 = bc.cgroup_cpu_stat()
        if c0 and c1:
            st["host_cpu"] = {"core_seconds_used": round((c1[2] - c0[2]) * 1e-6, 3), "average_cores_busy": round((c1[2] - c0[2]) * 1e-6 / dt, 1),
                              "quota_throttled_periods": c1[0] - c0[0], "quota_throttled_thread_ms": round((c1[1] - c0[1]) * 1e-3, 1)}
        return dt, st

    def describe(dt, st, steps, bps):
        per = {k: st.get(k, 0.0) / steps for k in ("stage_s", "h2d_s", "gpu_s", "vcf_s")}
        names = {"stage_s": "host staging (pread into pinned buffers, coverage slice, position fields)", "h2d_s": "H2D copies",
                 "gpu_s": "device: int16 -> int32 + PileupModel forward + argmax / max into pinned memory", "vcf_s": "VCF rows + file write (writer thread)"}
        return {"value": n * steps / dt, "unit": "sites/s", "ms_per_step": dt / steps * 1e3,
                "stage_busy_s_per_step": {names[k]: round(v, 4) for k, v in per.items()}, "bound_by": names[max(per, key=per.get)],
                "h2d_GB_per_s": st.get("bytes_h2d", 0.0) / max(st.get("h2d_s", 0.0), 1e-9) / 1e9, "bytes_over_pcie_per_site": bps,
                "pcie_bound_sites_per_s_at_the_measured_h2d_rate": st.get("bytes_h2d", 0.0) / max(st.get("h2d_s", 0.0), 1e-9) / bps,
                "main_thread_s_per_step": {k: round(st.get(k, 0.0) / steps, 4) for k in ("setup_s", "wait_stage_s", "issue_s", "drain_s", "account_s")},
                "staging_thread_s_per_step": {k: round(st.get(k, 0.0) / steps, 4) for k in ("stage_values_s", "stage_coverage_s", "stage_fields_s")},
                "compute_stream_idle_between_passes_s_per_step": round(st.get("gpu_idle_s", 0.0) / steps, 4), "host_cpu_over_the_timed_region": st.get("host_cpu"),
                **({"per_rank_s_per_step": st["per_rank"]} if "per_rank" in st else {})}

    dt, st = timed(K)
    header = host.vcf_header(fai).encode()
    vcf_one_file, files_equal = b"", True
    if rank == 0:
        body = open(out_path, "rb").read()[len(header):]
        vcf_one_file = body[:len(body) // K]
        files_equal = body == vcf_one_file * K
    head = describe(dt, st, K, 594 * 2)
    K2 = max(1, min(K, 4))
    second = {}
    if not args.no_second_precision:
        for key, nrw, bps in (("int32_counts_on_disk_narrowed_while_staged", True, 594 * 2), ("int32_counts_on_disk_sent_as_int32", False, 594 * 4)):
            d2, s2 = timed(K2, narrow=nrw, path=path32)
            second[key] = describe(d2, s2, K2, bps)
            if rank == 0:
                b2 = open(out_path, "rb").read()[len(header):]
                second[key]["vcf_equals_the_int16_run"] = bool(b2[:len(b2) // K2] == vcf_one_file)
    one = os.path.join(tmp, f"nsnp_pde2e_one_{rank}.vcf")
    if not args.no_parity_sample:
        created.append(one)
        predict_pileup_bins(model, [path], fai, one, pass_sites=n)          # (every rank: its shard of the file as ONE pass)
    if rank != 0:
        barrier()
        return 0
    # ---- HBM-resident rate of the same forward + calls (rank 0) ----
    m = min(P, n)
    xr = torch.from_numpy(np.asarray(sitefile.read_arrays(path)["position_matrix"][:m], np.int32)).to(dev)
    centers = (torch.arange(m, dtype=torch.int64, device=dev) * 33 + 16).contiguous()
    for _ in range(2):
        ctx.pileup_forward_windows_calls(xr.view(m * 33, 18), centers)
    torch.cuda.synchronize(dev)
    reps = max(4, n // m)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.pileup_forward_windows_calls(xr.view(m * 33, 18), centers)
    torch.cuda.synchronize(dev)
    resident = m * reps / (time.perf_counter() - t0)

    parity = None
    if not args.no_parity_sample:
        v1 = open(one, "rb").read()[len(header):]
        # and the first windows against the oracle: encode -> forward on the same columns
        from oracle import oracle
        k = 2048
        b1 = int(cols_keep.col_off[k * 33])
        oc, _, _ = oracle.encode_columns(cols_keep.bases[:b1], cols_keep.col_off[:k * 33 + 1], cols_keep.ref[:k * 33])
        x0 = np.asarray(sitefile.read_arrays(path)["position_matrix"][:k], np.int32)
        enc_ok = bool(np.array_equal(np.asarray(x0), oc.reshape(k, 33, 18)))
        ogt, ozy = oracle.pileup_forward(weights, oc.reshape(k, 33, 18), nthreads=bc.usable_cores())
        gt, zy = ctx.pileup_forward(torch.from_numpy(np.ascontiguousarray(x0)).to(dev))
        dp = float(max(np.abs(gt.cpu().numpy() - ogt).max(), np.abs(zy.cpu().numpy() - ozy).max()))
        parity = {"vcf_bytes_per_file": len(vcf_one_file), "the_K_files_gave_equal_rows": bool(files_equal), "timed_run_equals_the_one_pass_run": bool(v1 == vcf_one_file),
                  "file_windows_equal_the_oracle_encode": enc_ok, "max_abs_dp_vs_oracle": dp, "tolerance": 1e-4, "sites_vs_oracle": k,
                  "what": "pileup.vcf of the timed, streamed run byte-identical to the run that takes the whole file as one pass (and to the runs from the int32 file); the "
                          "file's first windows and their probabilities against oracle/liboracle.so (the row formatter against the reference: tests/test_vcf.py)"}
        parity["ok"] = bool(files_equal and parity["timed_run_equals_the_one_pass_run"] and enc_ok and dp <= 1e-4 and
                            all(v.get("vcf_equals_the_int16_run", True) for v in second.values()))
    out = {
        "metric": "candidate SNP sites/sec, .pd.bin site file to pileup.vcf (windows on the page cache: staging + H2D + PileupModel fwd + VCF)",
        "value": head["value"], "unit": "sites/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "stage 2 from site files: %d G2 windows (30x) as a .pd.bin site file (%.2f GB, int16 counts, page cache) -> passes of %d: "
                               "pread into pinned buffers beside H2D beside PileupModel fwd (fp32) + calls written to pinned memory -> pileup.vcf; NOT the headline "
                               "configuration (BASELINE configs[1] has its inputs in HBM)" % (n, os.path.getsize(path) / 1e9, P),
                   "sites": n, "sites_per_pass": P, "file_bytes": os.path.getsize(path), "world_size_observed": world,
                   **({"TEST_CONFIGURATION": "ranks share GPU 0, gather over gloo: device time is serialised, not a scaling number"} if args.share_gpu else {})},
        **{k: head[k] for k in ("stage_busy_s_per_step", "bound_by", "h2d_GB_per_s", "bytes_over_pcie_per_site", "pcie_bound_sites_per_s_at_the_measured_h2d_rate",
                                "main_thread_s_per_step", "staging_thread_s_per_step", "compute_stream_idle_between_passes_s_per_step", "host_cpu_over_the_timed_region")},
        **({"per_rank_s_per_step": head["per_rank_s_per_step"]} if "per_rank_s_per_step" in head else {}),
        "hbm_resident_sites_per_s": resident, "fraction_of_hbm_resident_rate": head["value"] / resident,
        "second_values": second, "usable_cores": bc.usable_cores(), "roofline": None, "parity_sample": parity, "timed_region_s": dt,
        "file_generation_s": round(t_gen, 1), "cpu_baseline": None,
    }
    for v in second.values():
        v["fraction_of_hbm_resident_rate"] = v["value"] / resident
    if not args.no_cpu_baseline:
        from oracle import oracle
        k = min(n, 65536)
        x0 = np.asarray(sitefile.read_arrays(path)["position_matrix"][:k], np.int32)
        cores = bc.usable_cores()
        t0 = time.perf_counter()
        oracle.pileup_forward(weights, x0[:4096], nthreads=cores, blocked=True)
        t1 = time.perf_counter() - t0
        kk = int(min(k, max(4096, 4096 * min(args.cpu_seconds, 12.0) / max(t1, 1e-6)))) // 64 * 64
        t0 = time.perf_counter()
        oracle.pileup_forward(weights, x0[:kk], nthreads=cores, blocked=True)
        t2 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": kk / t2, "unit": "sites/s", "cores": cores, "kind": "port",
                               "sample": f"{kk} of the file's windows through the oracle's blocked full-schedule fp32 forward, OpenMP over {cores} threads ({t2:.1f} s); "
                                         "no file I/O, no row formatting; oracle/liboracle.so", "host_cpu": bc.host_cpu_name(), "logical_cpus": os.cpu_count()}
    barrier()                                        # (the other ranks wait here while rank 0 measured the resident rate and checked)
    if emit is not None:
        emit(out)
    else:
        bc.emit_line(out, "pd_e2e")
    if parity is not None and not parity["ok"]:
        print("bench.py: parity_sample FAILED: " + json.dumps(parity), file=sys.stderr)
        return 1
    return 0
