#!/usr/bin/env python3
"""bench.py --workload hap-e2e: stage 5 FROM HOST MEMORY - haplotype site file to haplotype.csv, the path that replaces
HaplotypeModel/predict_dev.py:27-48 (TestDataset reading a whole HDF5 bin, a DataLoader with four workers calling
get_frequency_feature per site, a blocking .to(device) per batch, a Python loop per site: dataset_dev.py:92-172,337-349,
write_to_bins.py:44-63).  A labelled measurement, never the headline `value` of the repository (BASELINE's metric is quoted with
inputs in HBM).

    NSNP_HAPE2E_SITES G3 sites (default 65,536; D = 90, 30x) in a haplotype site file on the page cache - int8 read planes, the
    writer's default, and the same sites as int32 planes (the reference's dtype) - and a synthetic reference contig resident in HBM
    -> nanosnp_amd.hap_pipeline.predict_haplotype_bins: passes of 16,384 sites, pread into pinned buffers on all host cores beside H2D
    on a copy stream beside reference rows + haplotype features x 2 + HaplotypeModel forward (fp32) + argmax / max on the compute
    stream -> calls D2H -> nsnp_hap_csv_format -> haplotype.csv written.  One *step* = the whole file; the K timed steps are K files
    of ONE run (a directory of K bins, as predict_dev.py loops over os.listdir): one pipeline across them.

value = the int8 file.  Beside it: the int32 file narrowed to int8 while it is staged (what a file of reference dtype costs), the
int32 file sent as int32 (PCIe-bound: 63,360 B per site), and the HBM-RESIDENT rate of the same passes (planes already on the device,
the loop of `--workload haplotype`) that the streamed rates are measured against.  parity_sample = the csv of the timed run
byte-identical to the one-pass run of the same file, to the int32 runs, and (a second context with the fixture's seeded weights) the
19 stage-5 sites of tests/golden/two_stage.npz through the same file path against the rows the reference's predict_dev.py wrote."""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_SITES = 65_536
REF_LEN = 4_000_000


def scratch_dir(need_bytes):
    """where the site files go: /dev/shm (tmpfs = page-cache pages, written at memory speed) when it has room for three times the
    files, else the temporary directory (a file just written sits in the page cache there as well)"""
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 3 * need_bytes and os.access("/dev/shm", os.W_OK):
            return "/dev/shm"
    except OSError:
        pass
    return tempfile.gettempdir()


def make_files(tmp, n, D, seed, want_int32=True, chunk=16384):
    """the same G3 sites as an int8 and as an int32 site file, written piece by piece (sitefile.create_arrays); positions ascending on
    one synthetic contig -> (path8, path32, {contig: sequence})"""
    import numpy as np
    from nanosnp_amd import host, sitefile
    rng = np.random.default_rng(seed)
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), REF_LEN)
    seq[rng.random(REF_LEN) < 0.01] = ord("N")
    pos = np.sort(rng.choice(np.arange(100, REF_LEN - 100), n, replace=False)).astype(np.int64)
    cand = np.char.add("chrH:", pos.astype(str)).astype("S")
    hp = pos[:, None] + (np.arange(11)[None, :] - 5) * 29
    hpos = np.char.add("chrH:", hp.astype(str)).astype("S")

    def fields(a):
        w = a.dtype.itemsize
        return np.frombuffer(a.tobytes(), np.uint8).reshape(a.shape + (w,))
    cf, hf = fields(cand), fields(hpos)
    paths = []
    for dt in ([np.int8, np.int32] if want_int32 else [np.int8]):
        p = os.path.join(tmp, f"nsnp_hape2e_{n}_{D}_{np.dtype(dt).name}.bin")
        specs = {k: (dt, (n, D, 11 if k.startswith("haplotype") else 33)) for k in sitefile.HAP_PLANES}
        specs["candidate_positions"] = (np.uint8, cf.shape); specs["haplotype_positions"] = (np.uint8, hf.shape)
        paths.append((p, sitefile.create_arrays(p, specs)))
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        pp = host.synth_hap_planes(seed + c0, m, 30.0, D, 33); ph = host.synth_hap_planes(seed + 100 + c0, m, 30.0, D, 11)
        src = dict(zip(sitefile.HAP_PLANES, (ph[0], ph[3], ph[1], ph[2], pp[0], pp[3], pp[1], pp[2])))
        for _, maps in paths:
            for k in sitefile.HAP_PLANES:                       # (all host threads: numpy's astype + assignment takes 1 s per plane and chunk)
                dst = maps[k][c0:c0 + m].reshape(-1)
                assert host.stage_values(dst, dst.size, src=src[k].reshape(-1)) == 0
    for _, maps in paths:
        maps["candidate_positions"][:] = cf; maps["haplotype_positions"][:] = hf
        for a in maps.values():
            if hasattr(a, "flush"):
                a.flush()
    out = [p for p, _ in paths]
    return out[0], (out[1] if want_int32 else None), {"chrH": seq}


def two_stage_fixture_check(local_rank, tmp):
    """the reference's own stage-5 rows (tests/golden/two_stage.npz, written by HaplotypeModel/predict_dev.py with seeded weights) through
    the file path: site file from the fixture's planes + group positions, reference rows from the fixture's FASTA on the device"""
    import gzip
    import numpy as np
    from nanosnp_amd import _lib, host, sitefile
    from nanosnp_amd.fixtures import TWO_STAGE_HAP_WEIGHTS, seeded_hap_weights
    from nanosnp_amd.hap_pipeline import predict_haplotype_bins
    gold = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(gold, "two_stage.npz"))
    fa = os.path.join(tmp, "nsnp_hape2e_ref.fa")
    with open(fa, "wb") as f:
        f.write(gzip.open(os.path.join(gold, "encode_g1.fa.gz")).read())
    seq = host.fasta_load_contig(fa, "chrS")
    gpos = z["group_pos"]
    cands = [f"chrS:{p}" for p in gpos[:, 5]]
    planes = {f"{a}_{b}": z[f"{a[0]}_{c}"] for a in ("pileup", "haplotype") for b, c in (("sequences", "seq"), ("baseq", "bq"), ("mapq", "mq"), ("hap", "hap"))}
    p = os.path.join(tmp, "nsnp_hape2e_fixture.bin")
    sitefile.write_haplotype_bin(p, cands, [[f"chrS:{q}" for q in row] for row in gpos], planes)
    ctx = _lib.Context(local_rank)
    ctx.hap_load_weights(seeded_hap_weights(**TWO_STAGE_HAP_WEIGHTS))
    out = os.path.join(tmp, "nsnp_hape2e_fixture.csv")
    n = predict_haplotype_bins(ctx, [p], {"chrS": seq}, out, distributed=False)
    ctx.close()
    got = open(out).read().splitlines(); want = bytes(z["csv"]).decode().splitlines()
    same = n == len(want) == len(got)
    moved = 0
    for g, w in zip(got, want):
        gf, wf = g.split("\t"), w.split("\t")
        if gf[:3] != wf[:3] or abs(float(gf[3]) - float(wf[3])) > 0.0101:
            same = False
        moved += g != w
    for q in (fa, fa + ".fai", p, out):
        try:
            os.remove(q)
        except OSError:
            pass
    return {"ok": bool(same), "rows": len(want), "rows_whose_QUAL_differs_in_its_second_decimal": moved,
            "what": "the 19 stage-5 sites of tests/golden/two_stage.npz (planes + group positions + FASTA) through site file -> stream -> csv against "
                    "the rows the reference's predict_dev.py wrote: contig, position and genotype equal, QUAL within one unit of its second decimal "
                    "(probabilities differ from CPU torch by ~1e-7; tests/test_gpu_two_stage.py explains every such row)"}


def run(args, rank, world, local_rank, emit=None):
    """(the site files are 1-5 GB of memory-backed pages: removed whatever happens)"""
    created = []
    try:
        return _run(args, rank, world, local_rank, emit, created)
    finally:
        if rank == 0:
            for pth in created:
                try:
                    if pth and os.path.exists(pth):
                        os.remove(pth)
                except OSError:
                    pass


def _run(args, rank, world, local_rank, emit, created):
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
    from nanosnp_amd.fixtures import seeded_hap_weights
    from nanosnp_amd.hap_pipeline import DeviceReference, HapBinSource, predict_haplotype_bins, stream_haplotype
    from tools import bench_common as bc
    embedded = emit is not None
    if world > 1 and not embedded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    n = int(os.environ.get("NSNP_HAPE2E_SITES", args.hap_sites or N_SITES))
    P = int(args.hap_batch)
    D = 90
    want32 = not args.no_second_precision
    tmp = scratch_dir(n * D * 44 * (5 if want32 else 1))
    # ---- the site files (rank 0 writes, every rank reads through the page cache) ----
    t_gen = time.perf_counter()
    created += [os.path.join(tmp, f"nsnp_hape2e_{n}_{D}_int8.bin"), os.path.join(tmp, f"nsnp_hape2e_{n}_{D}_int32.bin"),
                os.path.join(tmp, f"nsnp_hape2e_{rank}.csv"), os.path.join(tmp, "nsnp_hape2e_onepass.csv")]
    if rank == 0:
        p8, p32, refs = make_files(tmp, n, D, 20260500, want_int32=want32)
    if world > 1:
        dist.barrier()
    if rank != 0:
        rng = np.random.default_rng(20260500)
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), REF_LEN); seq[rng.random(REF_LEN) < 0.01] = ord("N")
        refs = {"chrH": seq}
        p8 = os.path.join(tmp, f"nsnp_hape2e_{n}_{D}_int8.bin"); p32 = os.path.join(tmp, f"nsnp_hape2e_{n}_{D}_int32.bin") if want32 else None
    t_gen = time.perf_counter() - t_gen
    weights = seeded_hap_weights(12, H=256)
    ctx = _lib.Context(local_rank)
    ctx.set_option("hap_pass_sites", min(max(128, -(-P // 128) * 128), 131072))
    ctx.hap_load_weights(weights)
    ref = DeviceReference(refs, local_rank)
    out_path = os.path.join(tmp, f"nsnp_hape2e_{rank}.csv")
    W, K = max(1, args.warmup), max(1, args.steps)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(path, steps, narrow=True, pass_sites=P):
        """W warm-up files, then exactly `steps` files between barrier + synchronize on both sides; max over ranks -> (seconds, stats)"""
        for _ in range(W):
            predict_haplotype_bins(ctx, [path], ref, out_path, pass_sites=pass_sites, narrow=narrow)
        torch.cuda.synchronize(dev); bc.settle_collector(); barrier()
        st = {}
        clk0 = bc.clocks_ns()
        t0 = time.perf_counter()
        # the K steps = K files of one run (a directory of bins): ONE pipeline over all of them, as predict_dev.py's loop over
        # os.listdir is one run; the csv rows of file k are formatted and written while file k + 1 computes
        predict_haplotype_bins(ctx, [path] * steps, ref, out_path, pass_sites=pass_sites, narrow=narrow, stats=st)
        torch.cuda.synchronize(dev); barrier()
        dt = time.perf_counter() - t0
        bc.mark_region(clk0, bc.clocks_ns(), steps, {"workload": "hap_e2e", "stats": {k: v for k, v in st.items() if isinstance(v, (int, float))}})   # (the first timed run only)
        if world > 1:
            tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt = float(tm.item())
        return dt, st

    def describe(dt, st, steps, bytes_per_site):
        per = {k: st.get(k, 0.0) / steps for k in ("stage_s", "h2d_s", "gpu_s")}
        per["csv_s"] = st.get("csv_s", 0.0) / steps
        names = {"stage_s": "host staging (pread from the page cache into pinned buffers, OpenMP)", "h2d_s": "H2D copies",
                 "gpu_s": "device: reference rows + haplotype features x 2 + HaplotypeModel forward + argmax", "csv_s": "csv formatting + file write (writer thread)"}
        bound = max(per, key=per.get)
        return {"value": n * steps / dt, "unit": "sites/s", "ms_per_step": dt / steps * 1e3,
                "stage_busy_s_per_step": {names[k]: round(v, 4) for k, v in per.items()}, "bound_by": names[bound],
                "h2d_GB_per_s": st.get("bytes_h2d", 0.0) / max(st.get("h2d_s", 0.0), 1e-9) / 1e9,
                "staging_GB_per_s_of_bytes_written": st.get("bytes_staged", 0.0) / max(st.get("stage_s", 0.0), 1e-9) / 1e9,
                "bytes_over_pcie_per_site": bytes_per_site,
                "pcie_bound_sites_per_s_at_the_measured_h2d_rate": st.get("bytes_h2d", 0.0) / max(st.get("h2d_s", 0.0), 1e-9) / bytes_per_site,
                "main_thread_s_per_step": {k: round(st.get(k, 0.0) / steps, 4) for k in ("setup_s", "wait_stage_s", "issue_s", "drain_s")},
                "passes_per_step": st.get("passes", 0) / steps, "int8_passes_per_step": st.get("passes_int8", 0) / steps}

    dt, st = timed(p8, K)
    per_rank = None
    if world > 1:
        mine = {k: round(st.get(k, 0.0) / K, 4) for k in ("stage_s", "wait_stage_s", "h2d_s", "gpu_s", "issue_s", "drain_s", "csv_s", "gather_s")}
        mine["rank"] = rank
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    csv_all = open(out_path, "rb").read() if rank == 0 else b""
    csv_timed = csv_all[:len(csv_all) // K]                     # the rows of the first of the K files (all K must be equal)
    files_equal = csv_all == csv_timed * K
    head = describe(dt, st, K, 15_840 + 12 * 12)
    K2 = max(1, min(K, 4))
    seconds = {}
    if want32:
        d2, s2 = timed(p32, K2, narrow=True)
        csv_narrow = open(out_path, "rb").read()[:len(csv_timed)] if rank == 0 else b""
        seconds["int32_file_narrowed_while_staged"] = describe(d2, s2, K2, 15_840 + 12 * 12)
        d3, s3 = timed(p32, K2, narrow=False, pass_sites=min(P, 8192))
        csv_i32 = open(out_path, "rb").read()[:len(csv_timed)] if rank == 0 else b""
        seconds["int32_file_sent_as_int32"] = describe(d3, s3, K2, 63_360 + 12 * 12)
        seconds["int32_file_sent_as_int32"]["sites_per_pass"] = min(P, 8192)
    # ---- the HBM-resident rate of the same passes: planes of one pass on the device, features + forward + argmax in a loop ----
    src = HapBinSource(p8)
    m = min(P, n)
    pl = {}
    for name in ("pileup_sequences", "pileup_baseq", "pileup_mapq", "pileup_hap", "haplotype_sequences", "haplotype_baseq", "haplotype_mapq", "haplotype_hap"):
        L = 33 if name.startswith("pileup") else 11
        a = np.empty(m * D * L, np.int8); src.stage_plane(name, 0, m, a)
        pl[name] = torch.from_numpy(a.reshape(m, D, L)).to(dev)
    cf, hf = src.position_fields(0, m)
    cp, cc = host.parse_ctg_pos(cf.reshape(m, -1), ref.table); hp, hc = host.parse_ctg_pos(hf, ref.table)
    off33 = torch.arange(-16, 17, dtype=torch.int64, device=dev)[None, :]
    rp = ref.rows(torch.from_numpy(cc).to(dev)[:, None].expand(m, 33), torch.from_numpy(cp).to(dev)[:, None] - 1 + off33)
    rh = ref.rows(torch.from_numpy(hc).to(dev), torch.from_numpy(hp).to(dev) - 1)
    src.close()

    def resident_pass():
        xp = ctx.hap_features(pl["pileup_sequences"], pl["pileup_baseq"], pl["pileup_mapq"], pl["pileup_hap"], rp)
        xh = ctx.hap_features(pl["haplotype_sequences"], pl["haplotype_baseq"], pl["haplotype_mapq"], pl["haplotype_hap"], rh)
        gt, _ = ctx.hap_forward(xp, xh)
        return gt.max(dim=1)
    for _ in range(2):
        resident_pass()
    torch.cuda.synchronize(dev)
    reps = max(4, n // m)
    t0 = time.perf_counter()
    for _ in range(reps):
        resident_pass()
    torch.cuda.synchronize(dev)
    resident = m * reps / (time.perf_counter() - t0)
    del pl

    exit_code = 0
    if rank == 0:
        parity = None
        if not args.no_parity_sample:
            one = os.path.join(tmp, "nsnp_hape2e_onepass.csv")
            c1 = _lib.Context(local_rank)
            c1.set_option("hap_pass_sites", 16384); c1.hap_load_weights(weights)
            if world == 1:
                predict_haplotype_bins(c1, [p8], ref, one, pass_sites=n, distributed=False)                       # ONE pass over the whole file (2 GB of int8 planes)
                csv_one = open(one, "rb").read(); os.remove(one)
            else:
                csv_one = None
            c1.close()
            fx = two_stage_fixture_check(local_rank, tmp)
            parity = {"csv_bytes": len(csv_timed), "rows": csv_timed.count(b"\n"), "the_K_files_of_the_timed_run_gave_equal_rows": bool(files_equal),
                      "timed_run_equals_the_one_pass_run": (csv_one == csv_timed) if csv_one is not None else None,
                      "int32_file_narrowed_equals_int8_file": (csv_narrow == csv_timed) if want32 else None,
                      "int32_file_as_int32_equals_int8_file": (csv_i32 == csv_timed) if want32 else None,
                      "two_stage_fixture": fx,
                      "what": "haplotype.csv of the timed, three-stations-in-flight run byte-identical to the run that takes the whole file as ONE pass, to "
                              "the runs of the int32 file (narrowed while staged / sent as int32); and the reference-written rows of the two-stage fixture "
                              "through the same file path"}
            parity["ok"] = bool(parity["rows"] == n and files_equal and fx["ok"] and all(v is not False for v in (
                parity["timed_run_equals_the_one_pass_run"], parity["int32_file_narrowed_equals_int8_file"], parity["int32_file_as_int32_equals_int8_file"])))
        out = {
            "metric": "haplotype sites/sec, site file to haplotype.csv (read planes on the page cache: staging + H2D + features + HaplotypeModel fwd + csv)",
            **{k: head[k] for k in ("value", "unit")}, "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "stage 5 from host memory: %d G3 sites (D = %d, 30x) in a haplotype site file with int8 read planes (%.2f GB, page cache) + a "
                                   "%d-base reference contig in HBM -> passes of %d sites: pread into pinned buffers beside H2D beside reference rows + haplotype "
                                   "features x 2 + HaplotypeModel fwd (fp32) + argmax -> haplotype.csv; NOT the headline configuration (BASELINE configs[2] has its "
                                   "inputs in HBM: --workload haplotype)" % (n, D, os.path.getsize(p8) / 1e9, REF_LEN, P),
                       "sites": n, "sites_per_pass": P, "D": D, "file_bytes_int8": os.path.getsize(p8), "file_bytes_int32": os.path.getsize(p32) if want32 else None,
                       "weights": "seeded (trained HaplotypeModel checkpoints are absent upstream)",
                       "parallelism": f"sites sharded x{world} by shard_range, calls gathered to rank 0",
                       "world_size_observed": dist.get_world_size() if world > 1 else 1,
                       **({"TEST_CONFIGURATION": "ranks share GPU 0, gather over gloo: device time is serialised, not a scaling number"} if args.share_gpu else {})},
            **{k: head[k] for k in ("stage_busy_s_per_step", "bound_by", "h2d_GB_per_s", "staging_GB_per_s_of_bytes_written", "bytes_over_pcie_per_site",
                                    "pcie_bound_sites_per_s_at_the_measured_h2d_rate", "main_thread_s_per_step", "passes_per_step", "int8_passes_per_step")},
            **({"per_rank_s_per_step": per_rank} if per_rank else {}),
            "hbm_resident_sites_per_s": resident,
            "fraction_of_hbm_resident_rate": head["value"] / world / resident,
            "second_values": seconds,
            "usable_cores": bc.usable_cores(), "roofline": None, "parity_sample": parity, "timed_region_s": dt, "file_generation_s": round(t_gen, 1),
            "cpu_baseline": None,
        }
        for v in seconds.values():
            v["fraction_of_hbm_resident_rate"] = v["value"] / world / resident
        if not args.no_cpu_baseline and world == 1:
            from tools.hap_bench import cpu_baseline_hap

            class _S:                                       # the sample the oracle is timed on: the first sites of the same file
                pass
            s_ = _S()
            a = HapBinSource(p8)
            k = min(n, 8192)
            s_.planes = []
            for names_, L in ((("pileup_sequences", "pileup_baseq", "pileup_mapq", "pileup_hap"), 33), (("haplotype_sequences", "haplotype_baseq", "haplotype_mapq", "haplotype_hap"), 11)):
                arrs = []
                for nm in names_:
                    buf = np.empty(k * D * L, np.int32 if a.elem == 4 else np.int8); a.stage_plane(nm, 0, k, buf)
                    arrs.append(torch.from_numpy(buf.astype(np.int32).reshape(k, D, L)))
                cf, hf = a.position_fields(0, k)
                cands = [bytes(r).rstrip(b"\0").decode() for r in cf.reshape(k, -1)]
                rows = host.haplotype_ref_rows(refs, cands, 33) if L == 33 else \
                    host.haplotype_ref_rows(refs, cands, 11, position_lists=[[bytes(q).rstrip(b"\0").decode() for q in r] for r in hf])
                arrs.append(torch.from_numpy(rows))
                s_.planes.append(arrs)
            a.close()
            s_.weights, s_.D, s_.n = weights, D, k
            out["cpu_baseline"] = cpu_baseline_hap(s_, args.cpu_seconds)
        if emit is not None:
            emit(out)
        else:
            bc.emit_line(out, "hap_e2e")
        if parity is not None and not parity["ok"]:
            print("bench.py: parity_sample FAILED: " + json.dumps(parity), file=sys.stderr)
            exit_code = 1
    ctx.close()
    if world > 1:
        dist.barrier()
        if not embedded:
            dist.destroy_process_group()
    return exit_code
