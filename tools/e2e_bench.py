#!/usr/bin/env python3
"""bench.py --workload e2e: mpileup TEXT to VCF, the path that replaces make_predict_data.sh:184-234 + PileupModel/predict.py:37-195
(DNA_CreateCanSnpTensor -> DNA_CreatePredictData -> make_bin_predict_data.py -> predict.py and the four text / HDF5 files between
them) - a labelled measurement, never the headline `value` of the repository (BASELINE's metric is quoted with inputs in HBM).

    a synthetic contig of NSNP_E2E_COLS columns (default 6,000,000: generator G1, 30x) written as samtools-mpileup text to a file that
    sits in the page cache -> memory-mapped -> nanosnp_amd.pipeline.call_contigs: chunks of whole lines copied into pinned buffers
    (libnanosnp_host.so, OpenMP) beside their H2D copies beside nsnp_mpileup_tokenise + column encode + site selection + PileupModel
    forward + argmax of earlier chunks -> call rows -> nsnp_vcf_format_batches -> pileup.vcf written.  One *step* = the whole contig.
    (NSNP_TOKENISE=host: the text is cut into columns by the host tokeniser instead, the path of rounds 1-5; it rides along as the
    `host_parsed` second value and its VCF must equal the device-tokenised one.)

Under N ranks the TEXT is sharded by byte range (every rank parses only its lines) and the calls are gathered to rank 0 (strong
scaling).  The line carries the per-stage busy times and names the stage that bounds the pipeline; parity = the VCF of the chunked
run is byte-identical to the one-chunk run of the same text."""
from __future__ import annotations

import json
import mmap
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_COLS = 6_000_000


def run(args, rank, world, local_rank, emit=None):
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
    from nanosnp_amd import host
    from nanosnp_amd.fixtures import load_pileup_weights
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_contig, call_contigs, tokenise_mode
    from tools import bench_common as bc
    if world > 1 and emit is None:                      # (embedded in the default bench line: the process group exists already)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    n_cols = int(os.environ.get("NSNP_E2E_COLS", getattr(args, "e2e_cols", 0) or N_COLS))
    chunk = int(os.environ.get("NSNP_E2E_CHUNK_MB", 64)) << 20
    weights = load_pileup_weights()
    model = LSTMNetwork(device=local_rank).load_weight_list(weights)
    # ---- the contig: G1 columns -> mpileup text on a file (page cache), memory-mapped by every rank ----
    cols = host.synth_columns(20260900, n_cols, coverage=args.coverage, het_rate=0.03)
    seq = cols.ref
    path = os.path.join(tempfile.gettempdir(), f"nsnp_e2e_{n_cols}_{int(args.coverage)}.mpileup")
    if rank == 0:
        text_np = cols.mpileup_text_native("chr20s")
        with open(path + ".tmp", "wb") as f:
            f.write(memoryview(text_np))
        os.replace(path + ".tmp", path)
        del text_np
    if world > 1:
        dist.barrier()
    f = open(path, "rb")
    text = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    text_bytes = len(text)
    W, K = max(1, args.warmup), max(1, args.steps)
    out_path = os.path.join(tempfile.gettempdir(), f"nsnp_e2e_{rank}.vcf")

    header = host.vcf_header("chr20s\t%d\t8\t60\t61\n" % n_cols).encode()

    def run(k, stats=None):
        """k contigs = ONE run (pipeline.call_contigs, what call_variants does with a genome's contigs: the rows of contig c are formatted
        and written on a writer thread while contig c + 1 streams) -> pileup.vcf = header + the k contigs' rows.  (sites, rows) per contig."""
        if rank == 0 and os.path.exists(out_path):
            os.remove(out_path)
        g = open(out_path, "wb") if rank == 0 else None
        try:
            if g:
                g.write(header)
            ns, nr = call_contigs(model, [("chr20s", text, seq)] * k, g, chunk_bytes=chunk, stats=stats)
        finally:
            if g:
                g.close()
        return ns // k, nr // k

    def rows_of_first_contig(k):
        body = open(out_path, "rb").read()[len(header):]
        assert len(body) % k == 0 and body[:len(body) // k] * k == body, "the contigs of one run gave different rows"
        return body[:len(body) // k]

    run(max(W, min(3, K)))
    torch.cuda.synchronize(dev)
    bc.settle_collector()
    if world > 1:
        dist.barrier()
    stats = {}
    clk0 = bc.clocks_ns()
    t0 = time.perf_counter()
    ms0 = torch.cuda.memory_stats(dev)
    cg0 = bc.cgroup_cpu_stat()
    n_sites, n_rows = run(K, stats)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    bc.mark_region(clk0, bc.clocks_ns(), K, {"workload": "e2e", "stats": {k: v for k, v in stats.items() if isinstance(v, (int, float))}})
    rows_text = rows_of_first_contig(K) if rank == 0 else b""
    cg1 = bc.cgroup_cpu_stat()
    ms1 = torch.cuda.memory_stats(dev)
    allocator = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")}
    host_cpu = None
    if cg0 and cg1:
        host_cpu = {"core_seconds_used": round((cg1[2] - cg0[2]) * 1e-6, 3), "average_cores_busy": round((cg1[2] - cg0[2]) * 1e-6 / dt, 1),
                    "quota_throttled_periods": cg1[0] - cg0[0], "quota_throttled_thread_ms": round((cg1[1] - cg0[1]) * 1e-3, 1),
                    "note": "a throttled period freezes every thread of the cgroup until the next 100 ms period starts: the steps far above the median"}
    if world > 1:
        tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
    per_rank = None
    if world > 1:
        mine = {k: round(stats.get(k, 0.0) / K, 4) for k in ("parse_s", "wait_parse_s", "h2d_s", "gpu_s", "issue_s", "drain_s", "vcf_s", "gather_s")}
        mine["rank"] = rank
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    # ---- labelled second value: the same text with the PileupModel forward in the bf16x3 arithmetic ----
    second = None
    if not args.no_second_precision:
        model.ctx.set_option("pileup_precision", 2)
        run(min(2, K))
        torch.cuda.synchronize(dev)
        bc.settle_collector()
        if world > 1:
            dist.barrier()
        st2 = {}
        t0 = time.perf_counter()
        n_sites2, n_rows2 = run(K, st2)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        dt2 = time.perf_counter() - t0
        rows2 = rows_of_first_contig(K) if rank == 0 else b""
        if world > 1:
            tm = torch.tensor([dt2], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt2 = float(tm.item())
        model.ctx.set_option("pileup_precision", 0)
        if rank == 0:
            per2 = {k: st2.get(k, 0.0) / K for k in ("parse_s", "h2d_s", "gpu_s", "vcf_s")}
            # the same sites; a row may differ from the fp32 run's in QUAL / GQ only, by a unit of the last decimal (probabilities differ by ~1e-6)
            a_rows, b_rows = bytes(rows_text).split(b"\n"), bytes(rows2).split(b"\n")
            same_shape = len(a_rows) == len(b_rows) and n_sites2 == n_sites
            differing = changed = 0
            worst = 0.0
            ok2 = same_shape
            if same_shape:
                for ra, rb in zip(a_rows, b_rows):
                    if ra == rb:
                        continue
                    differing += 1
                    fa, fb = ra.split(b"\t"), rb.split(b"\t")
                    if len(fa) != len(fb) or fa[:5] != fb[:5] or fa[6:9] != fb[6:9]:
                        changed += 1                    # another call: the two largest probabilities of the site are within ~1e-6 of each other
                    else:
                        worst = max(worst, abs(float(fa[5]) - float(fb[5])))
                ok2 = worst <= 0.0101 and changed <= max(1, len(a_rows) // 20000)
            second = {"value": n_sites2 * K / dt2, "unit": "sites/s", "ms_per_step": dt2 / K * 1e3,
                      "dtype": "bf16x3 (every fp32 operand as three bf16 terms = 24 significand bits, six bf16 MFMAs per product, fp32 accumulate)",
                      "stage_busy_s_per_step": {k: round(v, 4) for k, v in per2.items()}, "bound_by": max(per2, key=per2.get),
                      "vcf_rows_differing_from_the_fp32_run": differing, "of_them_with_another_call": changed, "max_QUAL_difference_of_the_others": worst,
                      "parity_sample": {"ok": bool(ok2), "what": "same sites and rows as the fp32 run; a differing row differs in QUAL / GQ only, by at most one unit of QUAL's second "
                                                                 "decimal - except sites whose two largest probabilities tie within the arithmetic's ~1e-6 (counted; at most 1 in "
                                                                 "20,000 rows); the probabilities themselves: the bf16x3 parity samples of the pileup line"}}

    # ---- labelled second value: the same text cut into columns on the HOST cores (nsnp_mpileup_parse_into), the path of rounds 1-5 ----
    mode = tokenise_mode()
    host_parsed = None
    if mode == "device" and not getattr(args, "no_host_parsed", False):
        os.environ["NSNP_TOKENISE"] = "host"
        try:
            run(min(2, K))
            torch.cuda.synchronize(dev)
            bc.settle_collector()
            if world > 1:
                dist.barrier()
            st3 = {}
            t0 = time.perf_counter()
            n_sites3, n_rows3 = run(K, st3)
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            dt3 = time.perf_counter() - t0
            rows3 = rows_of_first_contig(K) if rank == 0 else b""
        finally:
            os.environ["NSNP_TOKENISE"] = "device"
        if world > 1:
            tm = torch.tensor([dt3], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt3 = float(tm.item())
        if rank == 0:
            per3 = {k: st3.get(k, 0.0) / K for k in ("parse_s", "h2d_s", "gpu_s", "vcf_s")}
            host_parsed = {"value": n_sites3 * K / dt3, "unit": "sites/s", "ms_per_step": dt3 / K * 1e3,
                           "stage_busy_s_per_step": {k: round(v, 4) for k, v in per3.items()}, "bound_by": max(per3, key=per3.get),
                           "vcf_equals_the_device_tokenised_run": bool(bytes(rows3) == bytes(rows_text) and n_sites3 == n_sites),
                           "what": "NSNP_TOKENISE=host: nsnp_mpileup_parse_into on the host cores writes pos / col_off / bases into pinned memory, those cross PCIe"}

    exit_code = 0
    if rank == 0:
        # parity: the chunked run against the one-chunk run of the same text, byte for byte (outside the clock)
        parity = None
        if not args.no_parity_sample and world == 1:
            whole, ns1, nr1 = call_contig(model, text, "chr20s", seq, chunk_bytes=1 << 40)
            parity = {"ok": bool(bytes(whole) == rows_text and ns1 == n_sites), "vcf_bytes": len(rows_text), "sites": n_sites,
                      "what": "pileup.vcf rows of the chunked, double-buffered run byte-identical to the one-chunk run of the same text "
                              "(nanosnp_amd.pipeline.call_contig; against the reference's own rows: tests/test_gpu_predict.py)"}
        if host_parsed and parity is not None:             # the device-tokenised VCF equals the host-parsed one, byte for byte: part of the sample
            parity["vcf_equals_the_host_parsed_run"] = host_parsed["vcf_equals_the_device_tokenised_run"]
            parity["ok"] = bool(parity["ok"] and host_parsed["vcf_equals_the_device_tokenised_run"])
        per = {k: stats.get(k, 0.0) / K for k in ("parse_s", "h2d_s", "gpu_s", "vcf_s")}
        bound = max(per, key=per.get)
        dev_tok = stats.get("tokenise") == "device"
        names = {"parse_s": "host: text into pinned memory (nsnp_stage_values, OpenMP)" if dev_tok else "host parse (nsnp_mpileup_parse_into, OpenMP)",
                 "h2d_s": "H2D copies", "gpu_s": ("device: tokenise + " if dev_tok else "device: ") + "encode + select + forward + calls",
                 "vcf_s": "D2H + VCF formatting (nsnp_vcf_format_batches)"}
        h2d_bytes = text_bytes if dev_tok else int(cols.col_off[-1]) + 16 * n_cols
        # the tokeniser against HBM: algorithmic bytes = the text read once + what it writes (column-5 bytes, position 8 + offset 8 + reference
        # byte 1 per line); its launches read the text twice, so HBM traffic is ~1.7x this figure
        tok = None
        if dev_tok and stats.get("tok_s"):
            chunks_per_step = stats.get("chunks", 0) / K
            alg = stats.get("text_bytes", 0) / K * (1 + int(cols.col_off[-1]) / text_bytes) + 17 * (stats.get("columns", 0) / K)
            tok = bc.roofline_hbm("mpileup_tokenise (3 launches)", alg / max(chunks_per_step, 1), stats["tok_s"] / K / max(chunks_per_step, 1) * 1e3,
                                  int(stats.get("chunks", 0)), how="HIP events around the three launches of every chunk on the compute stream, inside the timed run "
                                  "(other streams' copies run beside them)", chunk_bytes=chunk, text_bytes_per_step=stats.get("text_bytes", 0) / K)
        cols_per_pass = stats.get("columns", 0) / K * (world if world > 1 else 1)
        out = {
            "metric": "candidate SNP sites/sec, mpileup text to VCF (text on the page cache: staging + H2D + tokenise + encode + forward + VCF)",
            "value": n_sites * K / dt, "unit": "sites/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "text to VCF: a synthetic G1 contig of %d columns at %gx as samtools-mpileup text (%.0f MB, page cache) -> "
                                   "chunks of %d MB: staging into pinned memory beside H2D beside tokenise + column encode + site selection + PileupModel fwd (fp32) -> pileup.vcf; "
                                   "NOT the headline configuration (BASELINE configs[1] has its inputs in HBM)" % (n_cols, args.coverage, text_bytes / 1e6, chunk >> 20),
                       "columns": n_cols, "text_bytes": text_bytes, "chunk_bytes": chunk, "candidate_sites": n_sites, "vcf_rows": n_rows,
                       "parallelism": f"text sharded by byte range x{world}, calls gathered to rank 0",
                       "world_size_observed": dist.get_world_size() if world > 1 else 1,
                       **({"TEST_CONFIGURATION": "ranks share GPU 0, gather over gloo: not a scaling number"} if args.share_gpu else {})},
            "columns_per_s": n_cols * K / dt, "text_MB_per_s": text_bytes * K / dt / 1e6,
            "stage_busy_s_per_step": {names[k]: round(v, 4) for k, v in per.items()},
            "stage_rates": {"parse_MB_per_s": text_bytes / world / max(per["parse_s"], 1e-9) / 1e6,
                            "h2d_GB_per_s": h2d_bytes / world / max(per["h2d_s"], 1e-9) / 1e9,
                            "device_columns_per_s": n_cols / world / max(per["gpu_s"], 1e-9),
                            "vcf_rows_per_s": n_rows / max(per["vcf_s"], 1e-9)},
            "bound_by": names[bound],
            "overlap": {"sum_of_stage_busy_s": round(sum(per.values()), 4), "wall_s_per_step": round(dt / K, 4),
                        "note": "several things at a time: the worker thread stages (or parses) chunk k + 1 / k + 2, the copy stream sends chunk k, the "
                                "compute stream tokenises chunk k, encodes + selects chunk k - 1 and runs the forward of chunk k - 2 (line and site counts are "
                                "read one chunk late: no host round trip in the loop); wall < sum when they overlap.  main_thread_s_per_step: where the "
                                "issuing thread spends the step (issue_s includes wait_counts_s)"},
            "main_thread_s_per_step": {k: round(stats.get(k, 0.0) / K, 4) for k in ("setup_s", "wait_parse_s", "issue_s", "wait_counts_s", "drain_s", "wait_rows_s")},
            "writer_thread_s_per_step": {k: round(stats.get(k, 0.0) / K, 4) for k in ("vcf_s", "write_s")},
            "host_cpu_over_the_timed_region": host_cpu, "torch_allocator_over_the_timed_region": allocator,
            "step": "one contig of a run of `steps` contigs (pipeline.call_contigs: the rows of contig c are formatted and written on a writer thread "
                    "while contig c + 1 streams); ms_per_step = the run / steps",
            "tokenise": stats.get("tokenise", "host"), "tokenise_s_per_step": round(stats.get("tok_s", 0.0) / K, 5),
            "bytes_over_pcie_per_column": h2d_bytes / n_cols, "roofline_tokenise": tok,
            **({"bf16x3": second} if second else {}),
            **({"host_parsed": host_parsed} if host_parsed else {}),
            **({"per_rank_s_per_step": per_rank} if per_rank else {}),
            "usable_cores": bc.usable_cores(), "roofline": None, "parity_sample": parity, "timed_region_s": dt,
            "cpu_baseline": None,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cols, weights, args.cpu_seconds)
        if emit is not None:
            emit(out)
        else:
            bc.emit_line(out, "e2e")
        if (parity is not None and not parity["ok"]) or (second and not second["parity_sample"]["ok"]):
            print("bench.py: parity_sample FAILED: " + json.dumps([parity, second and second["parity_sample"]]), file=sys.stderr)
            exit_code = 1
    text.close(); f.close()
    if world > 1:
        dist.barrier()
        if emit is None:
            dist.destroy_process_group()
    if rank == 0:
        for pth in (path, out_path):
            try:
                os.remove(pth)
            except OSError:
                pass
    return exit_code


def cpu_baseline(cols, weights, target_s):
    """the oracle on a bounded prefix of the same contig: text parse is not part of it (the oracle starts from the column arrays): column
    encode (one thread) + site selection + full-schedule forward of the selected windows (blocked, OpenMP)"""
    import numpy as np
    from oracle import oracle
    from tools import bench_common as bc
    cores = bc.usable_cores()

    def run(m):
        b1 = int(cols.col_off[m])
        t0 = time.perf_counter()
        counts, depth, flags = oracle.encode_columns(cols.bases[:b1], cols.col_off[:m + 1], cols.ref[:m])
        centers = oracle.select_sites(cols.pos[:m], flags)
        x = np.stack([counts[c - 16:c + 17] for c in centers]) if len(centers) else np.zeros((0, 33, 18), np.int32)
        oracle.pileup_forward(weights, x, nthreads=cores, blocked=True)
        return time.perf_counter() - t0, len(centers)

    m0 = min(200_000, cols.n_cols)
    t, n = run(m0)
    m = int(min(cols.n_cols, max(m0, m0 * target_s / max(t, 1e-6))))
    t, n = run(m)
    return {"value": n / t, "unit": "sites/s", "cores": cores, "kind": "port",
            "sample": f"the first {m} columns of the same contig as arrays (no text parse): column encode + site selection + full-schedule fp32 forward of "
                      f"the {n} selected windows, OpenMP over {cores} threads ({t:.1f} s); oracle/liboracle.so",
            "host_cpu": bc.host_cpu_name(), "logical_cpus": os.cpu_count()}
