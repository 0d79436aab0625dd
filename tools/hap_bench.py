#!/usr/bin/env python3
"""bench.py --workload haplotype | deep60

haplotype = BASELINE configs[2], "HaplotypeModel (crnn.py) fwd on 1 MI355X, paired-haplotype windows":
    150,000 low-confidence sites (generator G3): the reference's int32 read planes [N,90,33] + [N,90,11] resident in
    HBM -> haplotype features (dataset_dev.get_frequency_feature, :55-87) -> model_dev.LSTMNetwork.predict (:133-143, exact fp32)
    -> argmax / max; one *step* = one batch of --hap-batch sites (16384).  After the timed region, labelled second values on the same
    pool: int8 read planes, the opt-in f16x3 arithmetic, and the LEGACY network config 2 names literally (model.CatModel = crnn.ResCRNN
    + percentage RNN, HaplotypeModel/model.py:332-358) on synthetic group tensors.

deep60 = BASELINE configs[4], "60x deep-coverage pileups (depth-bucket LDS spill path) + fp16 conv weights":
    one step = 40 batches of 4096 60x windows (column encode + PileupModel fwd) + 16,384 sites with D = 180 read planes (features +
    HaplotypeModel fwd, fp32) + the same number of sites through the legacy CatModel with its conv / LSTM weights split into fp16
    pairs (cat_precision 1: the reference has no fp16 path, the port's f16x3 split keeps the 1e-4 contract).  value = 60x windows / s
    over the whole step (the stage-5 : stage-2 ratio of 10 % is the two-stage workload's).

Every rank owns its own pools (weak scaling); the only exchange is the rooted gather of the compact calls inside the timed region.
HaplotypeModel / CatModel weights are seeded (the trained ones are absent upstream: .MISSING_LARGE_BLOBS)."""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_HAP = 150_000
N_HAP_DEEP = 65_536
N_WIN_DEEP = 655_360
WIN_BATCHES_PER_STEP = 40


class HapStage:
    """read planes of n sites resident in HBM; run_batch(i) = features (L = 33, L = 11) + HaplotypeModel forward + argmax / max"""

    def __init__(self, local_rank, n_sites, batch, coverage, D, seed, timing=True, int8_copy=False):
        import numpy as np
        import torch
        from nanosnp_amd import _lib, host
        from nanosnp_amd.fixtures import seeded_hap_weights
        self.torch, self._lib, self.lib = torch, _lib, _lib.load()
        self.dev = dev = torch.device("cuda", local_rank)
        self.n, self.batch, self.D = int(n_sites), int(batch), int(D)
        self.n_batches = -(-self.n // self.batch)
        self.planes, self.planes8 = [], []
        for L, sd in ((33, seed), (11, seed + 100)):
            ps = [[], [], [], [], []]
            for c0 in range(0, self.n, 16384):          # generated in chunks: the int32 planes of 150 k sites are ~10 GB
                pl = host.synth_hap_planes(sd + c0, min(16384, self.n - c0), coverage, D, L)
                for k in range(5):
                    ps[k].append(torch.from_numpy(pl[k]).to(dev))
            self.planes.append([torch.cat(p) for p in ps])
            if int8_copy:
                self.planes8.append([p.to(torch.int8) for p in self.planes[-1][:4]] + [self.planes[-1][4]])
        self.weights = seeded_hap_weights(12, H=256)
        self.ctx = _lib.Context(local_rank)
        self.ctx_pass_sites = min(max(128, -(-self.batch // 128) * 128), 131072)
        self.ctx.set_option("hap_pass_sites", self.ctx_pass_sites)
        self.ctx.hap_load_weights(self.weights)
        self.ctx.enable_timing(timing)
        self.stream = torch.cuda.Stream(device=dev)
        b = self.batch
        self.xp = torch.empty((b, 105, 33), dtype=torch.float32, device=dev)
        self.xh = torch.empty((b, 105, 11), dtype=torch.float32, device=dev)
        self.gt = torch.empty((self.n, 10), dtype=torch.float32, device=dev)
        self.zy = torch.empty((self.n, 3), dtype=torch.float32, device=dev)
        self.res = torch.empty((self.n, 2), dtype=torch.float32, device=dev)

    def batch_range(self, i):
        b0 = (i % self.n_batches) * self.batch
        return b0, min(self.n, b0 + self.batch)

    def features(self, b0, b1, int8=False, which=(0, 1)):
        P, lib = C.c_void_p, self.lib
        sp = P(self.stream.cuda_stream)
        for e in which:
            pl = (self.planes8 if int8 else self.planes)[e]
            L = 33 if e == 0 else 11
            out = self.xp if e == 0 else self.xh
            esz = 1 if int8 else 4
            args = [P(p.data_ptr() + esz * b0 * self.D * L) for p in pl[:4]] + [P(pl[4].data_ptr() + 4 * b0 * L)]
            fn = lib.nsnp_hap_features_i8 if int8 else lib.nsnp_hap_features
            rc = fn(self.ctx.handle, *args, b1 - b0, self.D, L, P(out.data_ptr()), sp)
            if rc:
                self._lib.check(rc, self.ctx.handle, "nsnp_hap_features")

    def features_alone(self, b0, b1, int8=False, warm=32, timed=16):
        """the L = 33 feature launch alone on the chip -> (total ms, launches) of `timed` launches behind `warm` untimed ones of the
        same shape: the memory side of the chip takes tens of milliseconds to reach its clocks after an idle period (the first eight
        launches after a synchronisation run 0.34 ms, the 130th 0.245: tools/probes/feat_warm_probe.py), and inside a job the kernel never
        meets a cold chip.  The timing accumulator is cleared first: launches of other shapes (L = 11) must not enter the average"""
        for _ in range(warm):
            self.features(b0, b1, int8=int8, which=(0,))
        self.sync(); self.ctx.read_timing()
        for _ in range(timed):
            self.features(b0, b1, int8=int8, which=(0,))
        self.sync()
        return self.ctx.read_timing()["hap_features"]

    def forward(self, b0, b1):
        P = C.c_void_p
        rc = self.lib.nsnp_hap_forward(self.ctx.handle, P(self.xp.data_ptr()), P(self.xh.data_ptr()), b1 - b0,
                                       P(self.gt.data_ptr() + 40 * b0), P(self.zy.data_ptr() + 12 * b0), P(self.stream.cuda_stream))
        if rc:
            self._lib.check(rc, self.ctx.handle, "nsnp_hap_forward")

    def run_batch(self, i, int8=False):
        b0, b1 = self.batch_range(i)
        self.features(b0, b1, int8)
        self.forward(b0, b1)
        with self.torch.cuda.stream(self.stream):
            gm, ga = self.gt[b0:b1].max(dim=1)                  # predict_dev.py:40-43
            self.res[b0:b1, 0] = ga.float(); self.res[b0:b1, 1] = gm
        return b1 - b0

    def sync(self):
        self.stream.synchronize()

    # ---- parity of what a run wrote (the lines' "parity_sample") ---------------------------------------------------------
    def snapshot(self, batch_ids, per_batch=256):
        """host copies of the probabilities / calls the last run_batch() of the given pool batches left behind (first per_batch sites each)"""
        t = self.torch
        self.sync(); t.cuda.synchronize(self.dev)
        ranges = []
        for i in sorted({b % self.n_batches for b in batch_ids}):
            b0, b1 = self.batch_range(i)
            ranges.append((b0, min(per_batch, b1 - b0)))
        idx = t.cat([t.arange(a, a + c, device=self.dev) for a, c in ranges])
        return {"ranges": ranges, "gt": self.gt[idx].cpu().numpy(), "zy": self.zy[idx].cpu().numpy(), "res": self.res[idx].cpu().numpy()}

    def parity_check(self, snap, tolerance=1e-4, nthreads=None):
        """the snapshot against the oracle's chain on the same sites: haplotype features (float64 sums, fp32 cast) + HaplotypeModel
        forward (oracle/hap_features_oracle.c, hap_forward_oracle.c); calls = argmax / max of the run's own probabilities"""
        import numpy as np
        from oracle import oracle
        from tools.bench_common import usable_cores
        nt = nthreads or usable_cores()
        ogt, ozy = [], []
        for a, c in snap["ranges"]:
            pp = [p[a:a + c].cpu().numpy() for p in self.planes[0]]; ph = [p[a:a + c].cpu().numpy() for p in self.planes[1]]
            xp = oracle.hap_features_batch(*pp, nthreads=nt); xh = oracle.hap_features_batch(*ph, nthreads=nt)
            g, z = oracle.hap_forward(self.weights, xp, xh, nthreads=nt)
            ogt.append(g); ozy.append(z)
        ogt, ozy = np.concatenate(ogt), np.concatenate(ozy)
        gt, zy, res = snap["gt"], snap["zy"], snap["res"]
        dp = float(max(np.abs(gt - ogt).max(), np.abs(zy - ozy).max()))
        calls_self = res is None or bool(np.array_equal(res[:, 0], gt.argmax(1).astype(np.float32)) and np.array_equal(res[:, 1], gt.max(1)))
        ok = bool(np.isfinite(gt).all() and np.isfinite(zy).all() and dp <= tolerance and calls_self)
        return {"ok": ok, "sites": int(gt.shape[0]), "max_abs_dp": dp, "tolerance": tolerance, "calls_equal_own_argmax": calls_self,
                "batches_sampled": len(snap["ranges"]),
                "what": "probabilities and calls the timed region's own run left behind for `sites` sites (read planes -> haplotype features -> "
                        "HaplotypeModel forward) against oracle/liboracle.so on the same read planes"}

    def arrange_alone(self, b0, b1, R=64, warm=8, timed=16):
        """k_hap_arrange (H1 + H2: centre filter, stable sort by centre HP, pad with -2, depth cut) alone on the chip, on read matrices
        made from the sites [b0, b1) of the pool: the site's reads in a random order in the first rows of an [R, 33] matrix, n_reads of
        them valid (the rest zeros), the way the stage-4 builder hands them over.  -> roofline dict with its own parity: the arranged
        planes give the features of the pool's planes bit for bit (the reduction is invariant under the order of equal-HP reads)."""
        import numpy as np
        t, dev, D, L = self.torch, self.dev, self.D, 33
        n = b1 - b0
        g = t.Generator(device=dev); g.manual_seed(7)
        seq, bq, mq, hap, ref = [p[b0:b1] for p in self.planes[0]]
        valid = seq[:, :, L // 2] > -2                                         # padding rows of the pool's planes
        depth = valid.sum(1).to(t.int32)
        R = int(max(R, int(depth.max().item())))
        key = t.rand((n, D), generator=g, device=dev) + (~valid).float() * 2    # a random order of the real reads, padding behind them
        perm = key.argsort(1)
        def mat(p):
            q = p.gather(1, perm[:, :, None].expand(-1, -1, L))
            q = t.where((q == -2), t.zeros_like(q), q)                          # (rows behind n_reads; never looked at)
            out = t.zeros((n, R, L), dtype=t.int32, device=dev)
            out[:, :min(R, D)] = q[:, :R]
            return out
        ms_, mb, mm, mh = mat(seq), mat(bq), mat(mq), mat(hap)
        outs = [t.empty((n, D, L), dtype=t.int32, device=dev) for _ in range(4)]
        dout = t.empty(n, dtype=t.int32, device=dev)
        P = C.c_void_p
        def launch():
            rc = self.lib.nsnp_hap_arrange_reads(self.ctx.handle, P(ms_.data_ptr()), P(mb.data_ptr()), P(mm.data_ptr()), P(mh.data_ptr()), P(depth.data_ptr()),
                                                 n, R, L, D, *[P(o.data_ptr()) for o in outs], P(dout.data_ptr()), P(self.stream.cuda_stream))
            if rc:
                self._lib.check(rc, self.ctx.handle, "nsnp_hap_arrange_reads")
        t.cuda.synchronize(dev)
        for _ in range(warm):
            launch()
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        e0.record(self.stream)
        for _ in range(timed):
            launch()
        e1.record(self.stream)
        self.sync()
        avg_ms = e0.elapsed_time(e1) / timed
        # parity: depth, and the features of the arranged planes against the features of the pool's own planes
        with t.cuda.stream(self.stream):
            fa = self.ctx.hap_features(outs[0], outs[1], outs[2], outs[3], ref.contiguous(), stream=self.stream)
            fb = self.ctx.hap_features(seq.contiguous(), bq.contiguous(), mq.contiguous(), hap.contiguous(), ref.contiguous(), stream=self.stream)
        self.sync()
        # centre-HP order of the arranged rows: non-decreasing over the kept rows, padding behind
        hp_c = outs[3][:, :, L // 2]
        kept = t.arange(D, device=dev)[None, :] < dout[:, None]
        sorted_ok = bool((((hp_c[:, 1:] >= hp_c[:, :-1]) | ~kept[:, 1:]).all()).item()) and bool(((hp_c == -2) == ~kept).all().item())
        ok = bool(t.equal(fa, fb)) and bool(t.equal(dout, t.minimum(depth, t.full_like(depth, D)))) and sorted_ok
        nbytes = int(depth.sum().item()) * 4 * L * 4 + n * (4 * D * L * 4 + 8)
        from tools import bench_common as bc
        r = bc.roofline_hbm("k_hap_arrange (L = 33: centre filter + stable HP sort + pad / cut to D)", nbytes, avg_ms, timed,
                            how="one HIP event pair around %d back-to-back launches on one stream, nothing else running, behind %d untimed ones" % (timed, warm),
                            sites_per_launch=n, D_out=D, rows_of_the_read_matrices=R, mean_reads_per_site=float(depth.float().mean().item()))
        r["algorithmic_bytes"] = "the valid rows of the four int32 read matrices in (n_reads x 33 x 4 B x 4) + the four padded [D, 33] int32 planes out + n_reads and depth"
        r["parity"] = {"ok": ok, "what": "depth = min(n_reads, D); kept rows in non-decreasing centre-HP order with the -2 padding behind them; haplotype features of the "
                                         "arranged planes bit-identical to the features of the pool's own planes (against the reference's "
                                         "single_group_pileup_haplotype_feature: tests/test_gpu_hap.py, tests/golden/hap_arrange.npz)"}
        r["sites_per_s"] = n / (avg_ms * 1e-3)
        return r

    def feature_bytes(self, n, L, int8=False):
        """algorithmic bytes of one feature launch: four read planes + the reference row in, [105, L] fp32 out"""
        return n * (4 * (1 if int8 else 4) * self.D * L + 4 * L + 105 * L * 4)


class CatStage:
    """legacy CatModel forward on synthetic group tensors (8192 distinct sites tiled to n)"""

    def __init__(self, local_rank, n_sites, batch, precision, timing=True):
        import torch
        from nanosnp_amd import _lib
        from nanosnp_amd.fixtures import seeded_cat_weights, synth_cat_groups
        self.torch, self._lib, self.lib = torch, _lib, _lib.load()
        self.dev = dev = torch.device("cuda", local_rank)
        self.n, self.batch, self.precision = int(n_sites), int(batch), int(precision)
        self.n_batches = -(-self.n // self.batch)
        distinct = min(self.n, 8192)
        g0, g1 = synth_cat_groups(20260600, distinct)
        rep = -(-self.n // distinct)
        self.g0 = torch.from_numpy(g0).to(dev).repeat(rep, 1, 1, 1)[:self.n].contiguous()
        self.g1 = torch.from_numpy(g1).to(dev).repeat(rep, 1, 1, 1)[:self.n].contiguous()
        self.weights = seeded_cat_weights(21)
        self.ctx = _lib.Context(local_rank)
        self.ctx.cat_load_weights(self.weights)
        self.ctx.set_option("cat_precision", precision)
        self.ctx.enable_timing(timing)
        self.stream = torch.cuda.Stream(device=dev)
        self.gt = torch.empty((self.n, 10), dtype=torch.float32, device=dev)

    def run_batch(self, i):
        P = C.c_void_p
        b0 = (i % self.n_batches) * self.batch; b1 = min(self.n, b0 + self.batch)
        row = 40 * 11 * 5 * 4
        rc = self.lib.nsnp_cat_forward(self.ctx.handle, P(self.g0.data_ptr() + row * b0), P(self.g1.data_ptr() + row * b0), b1 - b0,
                                       P(self.gt.data_ptr() + 40 * b0), P(self.stream.cuda_stream))
        if rc:
            self._lib.check(rc, self.ctx.handle, "nsnp_cat_forward")
        return b1 - b0

    def sync(self):
        self.stream.synchronize()

    def parity_check(self, b0, n, tolerance=1e-4, nthreads=None):
        """gt[b0 : b0 + n] as the last run_batch() left it against oracle/cat_forward_oracle.c on the same group tensors"""
        import numpy as np
        from oracle import oracle
        from tools.bench_common import usable_cores
        self.sync(); self.torch.cuda.synchronize(self.dev)
        gt = self.gt[b0:b0 + n].cpu().numpy()
        ogt = oracle.cat_forward(self.weights, self.g0[b0:b0 + n].cpu().numpy(), self.g1[b0:b0 + n].cpu().numpy(), nthreads=nthreads or usable_cores())
        dp = float(np.abs(gt - ogt).max())
        return {"ok": bool(np.isfinite(gt).all() and dp <= tolerance), "sites": int(n), "max_abs_dp": dp, "tolerance": tolerance,
                "precision": "f16x3" if self.precision == 1 else ("bf16x3" if self.precision == 2 else "fp32")}


def _timed(fn, sync, reps):
    fn(); sync()
    t0 = time.perf_counter()
    n = 0
    for _ in range(reps):
        n += fn()
    sync()
    return n, time.perf_counter() - t0


def hap_rooflines(hs, chain_ms, chain_n, feat_ms, feat_n, feat_sites, workload="haplotype"):
    """roofline of the fused LSTM step launches (MFMA) and of the feature reduction (HBM)"""
    from tools import bench_common as bc
    out = {}
    n_launch = bc.hap_lstm_launches()
    if chain_n:
        sites_per_pass = hs.sites_in_chain / chain_n
        avg = chain_ms / chain_n / n_launch
        out["roofline"] = bc.roofline_mfma(
            "k_hap_gemm<LSTM> (fused step: gates GEMM + cell)", bc.hap_exec_flop() * sites_per_pass / n_launch, avg, chain_n * n_launch,
            alg_flop_per_launch=bc.HAP_ALG_FLOP * sites_per_pass / n_launch,
            traffic=bc.committed_traffic(workload, "hap_gemm_lstm", D=hs.D),
            how="one HIP event pair around the %d dependent step launches of a pass (one stream, nothing else running), divided by %d; "
                "launches differ in size (K = 368 / 768, 2 or 4 direction slices), so flops and time are both per AVERAGE launch"
                % (n_launch, n_launch), launches_per_pass=n_launch, sites_per_pass=sites_per_pass)
    if feat_n:
        nbytes = hs.feature_bytes(feat_sites, 33)
        out["roofline_features"] = bc.roofline_hbm("k_hap_features (L = 33, int32 planes)", nbytes, feat_ms / feat_n, feat_n,
                                                   traffic=bc.committed_traffic(workload, "hap_features", D=hs.D),
                                                   how="HIP events around every launch, one stream, nothing else running, behind 32 untimed launches of the same shape (warm memory clocks)", sites_per_launch=feat_sites,
                                                   D=hs.D)
    return out


def cpu_baseline_hap(hs, target_s, deep=None):
    """the oracle's restatement (oracle/hap_features_oracle.c, hap_forward_oracle.c; plain loops, OpenMP over sites) on this box's
    usable cores, on a bounded sample of the same sites; beside it the reference ITSELF as timed in the development container"""
    import numpy as np
    from oracle import oracle
    from tools import bench_common as bc
    cores = bc.usable_cores()

    def run(n):
        pp = [p[:n].cpu().numpy() for p in hs.planes[0]]; ph = [p[:n].cpu().numpy() for p in hs.planes[1]]
        t0 = time.perf_counter()
        xp = oracle.hap_features_batch(*pp, nthreads=cores); xh = oracle.hap_features_batch(*ph, nthreads=cores)
        t1 = time.perf_counter()
        oracle.hap_forward(hs.weights, xp, xh, nthreads=cores)
        return time.perf_counter() - t0, t1 - t0

    n0 = max(cores, 32)
    t, _ = run(n0)
    n = int(min(max(n0, n0 * target_s / max(t, 1e-6)), 8192, hs.n))
    t, tf = run(n)
    out = {"value": n / t, "unit": "sites/s", "cores": cores, "kind": "port",
           "sample": f"{n} of the same synthetic sites (D = {hs.D}): haplotype features ({tf:.2f} s) + HaplotypeModel forward, full reference "
                     f"schedule, plain fp32 loops, OpenMP over {cores} threads ({t - tf:.1f} s); oracle/liboracle.so",
           "host_cpu": bc.host_cpu_name(), "logical_cpus": os.cpu_count()}
    rj, hp = bc.reference_cpu("haplotype")
    if rj and hp:
        try:
            f = max((r for r in hp["forward"] if r["threads"] == rj["host"]["logical_cpus"]), key=lambda r: r["sites_per_s"])
            out["reference_in_dev_container"] = {
                "value": f["sites_per_s"], "unit": "sites/s", "cores": f["threads"], "cpu": rj["host"]["cpu"],
                "what": "the reference's own model_dev.LSTMNetwork.predict on CPU torch (MKL), batch %d, seeded weights, forward only; its "
                        "get_frequency_feature takes %.2f ms per call (two calls per site, one DataLoader worker); tests/manual/time_reference_cpu.py"
                        % (f["batch"], hp["features"][0]["ms_per_call"]),
                "legacy_CatModel_predict_sites_per_s": max(r["sites_per_s"] for r in hp["cat"])}
        except Exception:
            pass
    return out


def run(args, rank, world, local_rank, deep60=False, emit=None):
    import torch
    import torch.distributed as dist
    from tools import bench_common as bc
    if args.share_gpu:
        if args.dist_backend != "gloo":
            print("bench.py: --share-gpu needs --dist-backend gloo", file=sys.stderr)
            return 2
        local_rank = 0
    elif torch.cuda.device_count() < world or local_rank >= torch.cuda.device_count():
        print(f"bench.py: {world} ranks asked for, {torch.cuda.device_count()} GPUs visible", file=sys.stderr)
        return 3
    from nanosnp_amd.dist import gather_results
    if world > 1 and emit is None:                      # (embedded in the default bench line: the process group exists already)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    cov, D = (60.0, 180) if deep60 else (30.0, 90)
    n_hap = int(os.environ.get("NSNP_HAP_N", args.hap_sites or (N_HAP_DEEP if deep60 else N_HAP)))
    reduced = n_hap < (N_HAP_DEEP if deep60 else N_HAP)
    hb = min(args.hap_batch, n_hap)
    hs = HapStage(local_rank, n_hap, hb, cov, D, 20260400 + 1000 * rank, int8_copy=not deep60)
    n_cat = int(os.environ.get("NSNP_CAT_N", getattr(args, "cat_sites", 0) or min(n_hap, 65_536)))
    cs = CatStage(local_rank, n_cat, min(hb, n_cat), 1 if deep60 else 0)
    ps = None
    if deep60:
        from tools.pileup_stage import PileupStage
        n_win = int(os.environ.get("NSNP_DEEP_WINDOWS", getattr(args, "deep_windows", 0) or N_WIN_DEEP))
        ps = PileupStage(local_rank, n_win, batch=args.batch, streams=args.streams, coverage=cov, seed=20260700 + rank,
                         enc_group=args.encode_group)
    wb = min(WIN_BATCHES_PER_STEP, ps.n_batches) if ps else 0
    W, K = max(1, args.warmup), max(1, args.steps)

    def barrier():
        if world > 1:
            dist.barrier()

    def sync_all():
        hs.sync(); cs.sync()
        if ps:
            ps.sync()
        torch.cuda.synchronize(dev)

    def step(i):
        n = [0, 0, 0]
        if ps:
            ps.run(i * wb, wb); n[0] = wb * ps.batch
            ps.sync()                                   # the stages of a step follow each other (stage-2 calls select the stage-5 sites)
        n[1] = hs.run_batch(i)
        if deep60:
            hs.sync()
            n[2] = cs.run_batch(i); cs.sync()
        return n

    def merge():
        outs = [hs.res]
        if ps:
            outs.append(ps.compact_calls(min(K * wb, ps.n_batches) * ps.batch))
        if deep60:
            outs.append(cs.gt.max(dim=1)[0][:, None])
        if world == 1:
            return outs
        return [gather_results(o.to(cdev), o.shape[0] * world) for o in outs]

    exit_code = 0
    for i in range(W):
        step(i)
    sync_all(); merge(); sync_all()
    hs.ctx.read_timing(); cs.ctx.read_timing()
    if ps:
        ps.read_timing()
    barrier(); sync_all()
    t0 = time.perf_counter()
    done = [0, 0, 0]
    hs.sites_in_chain = 0
    for i in range(W, W + K):
        n = step(i)
        done = [a + b for a, b in zip(done, n)]
        hs.sites_in_chain += n[1]
    sync_all()
    merged = merge()
    sync_all(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
    tim = hs.ctx.read_timing()
    chain_ms, chain_n = tim["hap_lstm_chain"]
    feat_in_region = tim["hap_features"]
    ctim = cs.ctx.read_timing()
    ptot = ps.read_timing() if ps else {}
    # what the timed steps left behind, for the parity sample (compared with the oracle after everything else, outside every clock)
    parity = None
    if rank == 0 and not args.no_parity_sample:
        parity = {"hap_snap": hs.snapshot(range(W, W + K), per_batch=max(64, 1024 // min(K, hs.n_batches)))}
        if ps:
            parity["ps_snap"] = ps.snapshot(ps.parity_ranges(min(K * wb, ps.n_batches) * ps.batch, per_batch=512, n_ranges=16))
        if deep60:
            cb0 = ((W + K - 1) % cs.n_batches) * cs.batch
            parity["cat"] = cs.parity_check(cb0, min(256, cs.n - cb0))

    # ---- after the timed region, alone on the chip: feature launches by window length, second / labelled values ----
    b0, b1 = hs.batch_range(0)
    nfe = b1 - b0
    feat_ms, feat_n = hs.features_alone(b0, b1)
    second = {}
    if rank == 0 or world > 1:
        # int8 read planes: same features bit for bit from a quarter of the input bytes
        if hs.planes8:
            ms8, n8 = hs.features_alone(b0, b1, int8=True)
            second["features_int8_planes"] = {"avg_launch_ms": ms8 / n8, "sites_per_s": nfe / (ms8 / n8 * 1e-3),
                                              "GB_per_s_of_its_own_bytes": hs.feature_bytes(nfe, 33, True) / (ms8 / n8 * 1e-3) / 1e9,
                                              "note": "nsnp_hap_features_i8, L = 33: own packed format, a quarter of the input bytes; the roofline above prices the int32 layout"}
        # the forward alone, fp32 and the opt-in f16x3 arithmetic, on the features of batch 0
        hs.features(b0, b1); hs.sync()
        nf, tf = _timed(lambda: (hs.forward(b0, b1), nfe)[1], hs.sync, 3)
        second["forward_only_fp32"] = {"sites_per_s": nf / tf, "executed_tflops": bc.hap_exec_flop() * nf / tf / 1e12,
                                       "frac_of_fp32_mfma_peak": bc.hap_exec_flop() * nf / tf / 1e12 / bc.PEAK_F32_MFMA_TFLOPS}
        if not args.no_second_precision:
            ref = hs.gt[b0:b1].clone()
            torch.cuda.synchronize(dev)                 # (the copy runs on torch's current stream, the forwards on the stage's)
            for prec, label, text in ((2, "bf16x3", "bf16x3 (weights as three bf16 planes, fp32 activations split on their way into LDS: 24 significand bits "
                                                    "per operand, six bf16 MFMAs per product, fp32 accumulate)"),
                                      (1, "f16x3", "f16x3 (every fp32 operand split into two fp16, 3 fp16 MFMAs per product, fp32 accumulate; opt-in)")):
                hs.ctx.set_option("hap_precision", prec)
                hs.sync(); hs.ctx.read_timing()
                nf, tf = _timed(lambda: (hs.forward(b0, b1), nfe)[1], hs.sync, 3)
                ch_ms, ch_n = hs.ctx.read_timing()["hap_lstm_chain"]
                d = (hs.gt[b0:b1] - ref).abs().max().item()
                sv = {"sites_per_s": nf / tf, "max_abs_dp_vs_fp32": d, "tolerance": 1e-4, "dtype": text}
                if prec == 2 and ch_n:
                    # its own roofline against the dense bf16 MFMA peak, the six MFMAs of a product priced as executed
                    n_launch = bc.hap_lstm_launches()
                    spp = min(nfe, int(hs.ctx_pass_sites))
                    sv["roofline"] = bc.roofline_mfma("k_hap_lstm_b3x (bf16x3 LSTM step, 256 x 256 tiles)", bc.hap_exec_flop() * 6 * spp / n_launch, ch_ms / ch_n / n_launch,
                                                      ch_n * n_launch, peak=bc.PEAK_F16_MFMA_TFLOPS, launches_per_pass=n_launch, sites_per_pass=spp,
                                                      how="one HIP event pair around the %d step launches of a pass, divided by %d" % (n_launch, n_launch))
                    if rank == 0 and not args.no_parity_sample:
                        t = torch
                        m = min(256, nfe)
                        sv["parity_sample"] = hs.parity_check({"ranges": [(b0, m)], "gt": hs.gt[b0:b0 + m].cpu().numpy(),
                                                               "zy": hs.zy[b0:b0 + m].cpu().numpy(), "res": None})
                second["forward_only_" + label] = sv
            hs.ctx.set_option("hap_precision", 0)
        if not deep60:
            # config 2 names crnn.py: the legacy CatModel forward, exact fp32, on its own pool
            cs.ctx.read_timing()
            nc, tc = _timed(lambda: cs.run_batch(0), cs.sync, 3)
            ct = cs.ctx.read_timing()
            second["legacy_CatModel_forward_fp32"] = cat_report(cs, nc, tc, ct, bc)
            if not args.no_second_precision:
                cs.ctx.set_option("cat_precision", 2); cs.precision = 2
                cs.ctx.read_timing()
                nc, tc = _timed(lambda: cs.run_batch(0), cs.sync, 3)
                ct = cs.ctx.read_timing()
                second["legacy_CatModel_forward_bf16x3"] = cat_report(cs, nc, tc, ct, bc)
                if rank == 0 and not args.no_parity_sample:
                    second["legacy_CatModel_forward_bf16x3"]["parity_sample"] = cs.parity_check(0, min(256, cs.n))
                cs.ctx.set_option("cat_precision", 0); cs.precision = 0

    if rank == 0:
        roofs = hap_rooflines(hs, chain_ms, chain_n, feat_ms, feat_n, nfe, "deep60" if deep60 else "haplotype")
        if not deep60:
            roofs["roofline_arrange"] = hs.arrange_alone(b0, b1)
        if deep60:
            from tools.pileup_stage import pileup_rooflines
            excl, excl_n = ps.exclusive_pass()
            pr = pileup_rooflines(ps, ptot, excl, excl_n, done[0], dt, 0, ps.G, "deep60")
            roofs["roofline_pileup_60x"] = pr.get("roofline")
            if roofs["roofline_pileup_60x"]:
                roofs["roofline_pileup_60x"].pop("chip", None)      # the step also holds the other two stages
            roofs["roofline_encode_60x"] = pr.get("roofline_encode")
            roofs["roofline_cat_conv_f16x3"] = cat_report(cs, done[2], None, ctim, bc).get("roofline")
        value_sites = done[0] if deep60 else done[1]
        out = {
            "metric": ("candidate SNP sites/sec, 60x deep coverage: 60x windows (encode + PileupModel fwd) + 10 % of them as D=180 sites through the "
                       "HaplotypeModel and the legacy CatModel with fp16-split weights" if deep60 else
                       "haplotype sites/sec (haplotype features + HaplotypeModel fwd), 30x paired-haplotype windows"),
            "value": world * value_sites / dt, "unit": "sites/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4]: 60x deep coverage - per step %d x %d 60x windows (column encode + PileupModel fwd, fp32) + %d sites "
                                    "with D = 180 read planes (haplotype features + HaplotypeModel fwd, fp32) + %d sites through the legacy CatModel with "
                                    "fp16-split conv / LSTM weights (cat_precision 1)" % (wb, ps.batch, hb, cs.batch)) if deep60 else
                                   ("BASELINE configs[2]: HaplotypeModel fwd on paired-haplotype windows - %d G3 sites (int32 read planes [N,90,33] + [N,90,11]) "
                                    "resident in HBM, haplotype features + model_dev.LSTMNetwork.predict (fp32), %d sites per step; legacy crnn.py CatModel "
                                    "forward reported beside it" % (n_hap, hb)),
                       **({"REDUCED_POOL": "a short run inside the default bench line: the configuration's pool is %d sites" % (N_HAP_DEEP if deep60 else N_HAP)} if reduced else {}),
                       "hap_sites_resident_per_gpu": n_hap, "hap_sites_per_step": hb, "D": D, "coverage": cov,
                       "weights": "seeded (trained HaplotypeModel / CatModel checkpoints are absent upstream)",
                       "parallelism": f"site-sharded x{world} (every rank its own pool), rooted gather of calls",
                       "world_size_observed": dist.get_world_size() if world > 1 else 1,
                       **({"windows_resident_per_gpu": ps.n_windows, "window_batches_per_step": wb, "batch": ps.batch, "cat_sites_per_step": cs.batch} if deep60 else {}),
                       **({"TEST_CONFIGURATION": "ranks share GPU 0, gather over gloo: not a scaling number"} if args.share_gpu else {})},
            "sites_timed": {"windows_60x": done[0], "haplotype_sites": done[1], "cat_sites": done[2]},
            "stage_ms_in_region": {"hap_lstm_chain_ms_per_pass": chain_ms / max(chain_n, 1), "hap_features_ms_total": feat_in_region[0],
                                   "cat_forward_ms_per_pass": (ctim["cat_forward_pass"][0] / max(ctim["cat_forward_pass"][1], 1))},
        }
        out.update(roofs)
        out.setdefault("roofline", None)
        out["second_values"] = second
        out["timed_region_s"] = dt
        out["shader_clock_mhz"] = {"value": hs.ctx.shader_clock_mhz(hs.stream), "how": "s_memtime / s_memrealtime in every workgroup of a ~2 ms full-chip "
                                   "fp32 MFMA probe after the timed region (nsnp_ctx_shader_clock); the MFMA peaks are priced at 2400"}
        out["parity_sample"] = None
        if parity is not None:
            par = {"haplotype": hs.parity_check(parity["hap_snap"])}
            if ps:
                par["pileup_60x"] = ps.parity_check(parity["ps_snap"])
            if "cat" in parity:
                par["cat_f16x3"] = parity["cat"]
            par["ok"] = all(v["ok"] for v in par.values())
            par["tolerance"] = 1e-4
            out["parity_sample"] = par
        out["cpu_baseline"] = None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_deep(ps, hs, cs, args.cpu_seconds) if deep60 else cpu_baseline_hap(hs, args.cpu_seconds)
        assert merged[0].shape[0] == n_hap * world
        if emit is not None:
            emit(out)
        else:
            bc.emit_line(out, "deep60" if deep60 else "haplotype")
        if out["parity_sample"] is not None and not out["parity_sample"]["ok"]:
            print("bench.py: parity_sample FAILED: " + json.dumps(out["parity_sample"]), file=sys.stderr)
            exit_code = 1
    if world > 1:
        dist.barrier()
        if emit is None:
            dist.destroy_process_group()
    return exit_code


def cat_report(cs, n_sites, seconds, ctim, bc):
    """rates and the conv-chain roofline of CatModel passes recorded in ctim (passes of <= 4096 sites each)"""
    conv_ms, conv_n = ctim["cat_conv_chain"]
    pass_ms, pass_n = ctim["cat_forward_pass"]
    out = {}
    if seconds:
        out["sites_per_s"] = n_sites / seconds
    if conv_n:
        sites_per_pass = min(4096, cs.batch)
        mult = {0: 1, 1: 3, 2: 6}[cs.precision]         # fp16 / bf16 MFMAs executed per fp32 product
        peak = bc.PEAK_F16_MFMA_TFLOPS if mult > 1 else bc.PEAK_F32_MFMA_TFLOPS
        out["roofline"] = bc.roofline_mfma(
            "k_cat_conv (3x3 convolution + 1x1 shortcut as a GEMM, pixel block + halo staged in LDS once per channel chunk)", bc.cat_conv_exec_flop() * mult * sites_per_pass / 12, conv_ms / conv_n / 12,
            conv_n * 12, alg_flop_per_launch=bc.cat_conv_alg_flop() * sites_per_pass / 12, peak=peak,
            how="one HIP event pair around the 12 convolution launches (+ 4 pooling launches) of a pass of %d sites, divided by 12" % sites_per_pass,
            launches_per_pass=12, sites_per_pass=sites_per_pass)
        out["conv_share_of_pass"] = conv_ms / max(pass_ms, 1e-9)
    return out


def cpu_baseline_deep(ps, hs, cs, target_s):
    """all three stages of a deep60 step through the oracle on a bounded sample in the step's proportions (10 : 1 : 1)"""
    import numpy as np
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
        t2 = time.perf_counter()
        oracle.cat_forward(cs.weights, cs.g0[:nb].cpu().numpy(), cs.g1[:nb].cpu().numpy(), nthreads=cores)
        t3 = time.perf_counter()
        return na, (t1 - t0, t2 - t1, t3 - t2)

    nb0 = max(cores, 32)
    _, ts = run(nb0)
    nb = int(min(max(nb0, nb0 * target_s / max(sum(ts), 1e-6)), 4096, hs.n, cs.n, ps.n_windows // 10))
    na, ts = run(nb)
    return {"value": na / sum(ts), "unit": "sites/s", "cores": cores, "kind": "port",
            "sample": f"{na} 60x windows (encode + full-schedule forward, {ts[0]:.1f} s) + {nb} D = 180 sites (features + HaplotypeModel forward, {ts[1]:.1f} s) + "
                      f"{nb} CatModel sites (fp32 loops, {ts[2]:.1f} s), the step's 10 : 1 : 1 proportions, OpenMP over {cores} threads; oracle/liboracle.so",
            "host_cpu": bc.host_cpu_name(), "logical_cpus": os.cpu_count()}
