"""The pileup stage of the bench workloads: a pool of stand-alone 33-column windows (generator G2) resident in HBM,
processed in batches through the hot path of PileupModel/predict.py:44-65 --

    column encode (mpileup bytes -> int32 [M,18] counts)  ->  PileupModel forward reading the windows in place  ->  argmax / max

Used by bench.py (BASELINE configs[1]), tools/two_stage_bench.py (configs[3], stage 2) and tools/hap_bench.py (configs[4], the 60x
stage).  Scheduling: the column encode is twice as efficient per byte at >= 1 M columns than at the 135 k columns of one 4096-window
batch (launch ramp + tail), and the forward only needs `counts`, so the encode runs on ITS OWN stream, `enc_group` consecutive
batches per launch (default 32 = 4.3 M columns: 18.13 / 18.37 / 18.50 / 18.53 M sites/s at 8 / 16 / 32 / 64), into a ring of count buffers; the forward + post-processing of each batch follow on one of `streams` streams
(one nsnp_ctx each) behind the group's encode event.  Every batch issued by run() is encoded by a launch issued by the same run()
call (no encode work is carried into or out of a timed region).
"""
from __future__ import annotations

import ctypes as C


class PileupStage:
    def __init__(self, local_rank, n_windows, batch=4096, streams=32, coverage=30.0, seed=20260000, precision=0, opts=(),
                 timing_streams=4, enc_group=32, ring=3, weights=None):
        import torch
        from nanosnp_amd import _lib, host
        from nanosnp_amd.fixtures import load_pileup_weights
        self.torch, self._lib = torch, _lib
        self.lib = _lib.load()
        self.dev = torch.device("cuda", local_rank)
        self.batch, self.S = int(batch), max(1, int(streams))
        self.n_windows = max(self.batch, (int(n_windows) // self.batch) * self.batch)
        self.n_batches = self.n_windows // self.batch
        self.G = max(1, min(int(enc_group), self.n_batches))                  # batches per encode launch (cut at the pool's wrap-around)
        self.R = max(2, int(ring))
        self.coverage = coverage
        self.weights = weights if weights is not None else load_pileup_weights()   # the shipped ont_pileup weights (fixture)
        self.mcols = self.batch * 33
        dev = self.dev
        # ---- synthetic pool, resident in HBM ----
        self.cols = host.synth_columns(seed, self.n_windows * 33, coverage=coverage, window=33)
        self.d_bases = torch.from_numpy(self.cols.bases).to(dev)
        self.d_off = torch.from_numpy(self.cols.col_off).to(dev)
        self.d_ref = torch.from_numpy(self.cols.ref).to(dev)
        self.centers = (torch.arange(self.batch, dtype=torch.int64, device=dev) * 33 + 16).contiguous()
        self.F = 1                                       # pool batches per forward launch (set_precision: the bf16x3 second value uses 4)
        # ---- contexts: one per forward stream + one for the encode stream ----
        self.timed_streams = min(self.S, max(0, int(timing_streams)))
        # the timed streams are spread over the stream indices (batch i runs on stream i mod S: a short region still meets one), stream 0
        # among them (the exclusive pass runs there); every recorded event costs the stream an extra packet, hence few of them (-0.5 % at 16)
        self.timed_idx = sorted({(k * self.S) // self.timed_streams for k in range(self.timed_streams)}) if self.timed_streams else []
        self.ctxs, self.streams = [], []
        for s in range(self.S):
            ctx = _lib.Context(local_rank, chunk_sites=self.batch)
            ctx.pileup_load_weights(self.weights)
            ctx.enable_timing(s in self.timed_idx)
            ctx.set_option("pileup_precision", precision)
            for o in opts:
                name, val = o.split("=")
                ctx.set_option(name, int(val))
            self.ctxs.append(ctx)
            self.streams.append(torch.cuda.Stream(device=dev))
        self.enc_ctx = _lib.Context(local_rank)
        self.enc_ctx.enable_timing(self.timed_streams > 0)
        self.enc_stream = torch.cuda.Stream(device=dev)
        gm = self.G * self.mcols
        self.ring = [dict(counts=torch.empty((gm, 18), dtype=torch.int32, device=dev),
                          depth=torch.empty(gm, dtype=torch.int32, device=dev),
                          flags=torch.empty(gm, dtype=torch.uint8, device=dev),
                          enc_done=torch.cuda.Event(), users=[]) for _ in range(self.R)]
        self.gseq = 0
        # results of every batch of the pool stay resident (24 fp32 + compact calls per site)
        n = self.n_windows
        self.gt_all = torch.empty((n, 21), dtype=torch.float32, device=dev)
        self.zy_all = torch.empty((n, 3), dtype=torch.float32, device=dev)
        self.res = dict(ga=torch.empty(n, dtype=torch.uint8, device=dev), za=torch.empty(n, dtype=torch.uint8, device=dev),
                        gm=torch.empty(n, dtype=torch.float32, device=dev), zm=torch.empty(n, dtype=torch.float32, device=dev))
        self._ev_pool = []

    # ---- scheduling -----------------------------------------------------------------------------------------------
    def set_precision(self, precision, fwd_group=1):
        """arithmetic of the forward on every stream; fwd_group = consecutive pool batches handed to ONE forward call (the bf16x3 kernels
        fill the chip at 16 k sites per launch, not at 4 k: VERDICT r4 #3; the fp32 headline keeps BASELINE's batch per launch)"""
        F = max(1, min(int(fwd_group), self.G))
        while self.G % F:
            F -= 1
        if F != self.F:
            self.sync()
            self.F = F
            self.centers = (self.torch.arange(self.batch * F, dtype=self.torch.int64, device=self.dev) * 33 + 16).contiguous()
            for ctx in self.ctxs:
                ctx.reserve(self.batch * F)
        for ctx in self.ctxs:
            ctx.set_option("pileup_precision", precision)

    def _event(self):
        return self._ev_pool.pop() if self._ev_pool else self.torch.cuda.Event()

    def run(self, first, count, single_stream=False):
        """issues batches first .. first + count - 1 of the endless batch sequence (batch i = pool batch i mod n_batches)"""
        P, lib, check = C.c_void_p, self.lib, self._lib.check
        i, end = first, first + count
        while i < end:
            b = i % self.n_batches
            g = min(self.G - (b % self.G), end - i, self.n_batches - b)      # up to G consecutive pool batches, cut at a group boundary / the pool's end
            slot = self.ring[self.gseq % self.R]; self.gseq += 1
            es = self.enc_stream
            for ev in slot["users"]:                                         # the forwards that last read this ring slot
                es.wait_event(ev)
                self._ev_pool.append(ev)
            slot["users"] = []
            c0 = b * self.mcols
            slot["c0"], slot["ncols"] = c0, g * self.mcols          # what the slot holds (parity_check compares it with the oracle)
            rc = lib.nsnp_pileup_encode_columns(self.enc_ctx.handle, P(self.d_bases.data_ptr()), P(self.d_off.data_ptr() + 8 * c0),
                                                P(self.d_ref.data_ptr() + c0), g * self.mcols, C.c_double(0.12), 6,
                                                P(slot["counts"].data_ptr()), P(slot["depth"].data_ptr()), P(slot["flags"].data_ptr()),
                                                P(es.cuda_stream))
            if rc:
                check(rc, self.enc_ctx.handle, "encode")
            slot["enc_done"].record(es)
            F = self.F
            for j in range(0, g, F):
                fcount = min(F, g - j) * self.batch                                # (a group cut at the pool's end may hold fewer batches)
                s = 0 if single_stream else ((i + j) // F) % self.S
                st = self.streams[s]
                st.wait_event(slot["enc_done"])
                h, sp = self.ctxs[s].handle, P(st.cuda_stream)
                n0 = (b + j) * self.batch
                gt_p, zy_p = P(self.gt_all.data_ptr() + 84 * n0), P(self.zy_all.data_ptr() + 12 * n0)
                # forward + argmax / max (predict.py:51-57) in one call: the fp32 heads kernel writes both
                rc = lib.nsnp_pileup_forward_windows_calls(h, P(slot["counts"].data_ptr() + 72 * j * self.mcols), P(self.centers.data_ptr()),
                                                           fcount, gt_p, zy_p, P(self.res["ga"].data_ptr() + n0),
                                                           P(self.res["za"].data_ptr() + n0), P(self.res["gm"].data_ptr() + 4 * n0),
                                                           P(self.res["zm"].data_ptr() + 4 * n0), sp)
                if rc:
                    check(rc, h, "forward / postprocess")
                ev = self._event(); ev.record(st); slot["users"].append(ev)
            i += g

    def sync(self):
        self.enc_stream.synchronize()
        for st in self.streams:
            st.synchronize()
        self.torch.cuda.synchronize(self.dev)

    # ---- timing ---------------------------------------------------------------------------------------------------
    def read_timing(self):
        """{kernel: [total_ms, launches]} of the launches recorded since the last read (forward kernels: the timed streams; encode:
        every launch).  The fused layer-1 kernel is reported as pileup_l1f."""
        tot = {}
        for ctx in [self.ctxs[s] for s in self.timed_idx] + [self.enc_ctx]:
            for k, (ms, n) in ctx.read_timing().items():
                if n:
                    a = tot.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += n
        if "pileup_proj1" not in tot and "pileup_l1" in tot:
            tot["pileup_l1f"] = tot.pop("pileup_l1")
        return tot

    def exclusive_pass(self, groups=4):
        """the same launches with the chip to themselves: everything on one forward stream behind the encode stream, `groups`
        encode launches of enc_group batches each after one untimed group -> {kernel: avg ms per launch}, launches"""
        if not self.timed_streams:
            return {}, {}
        self.sync(); self.read_timing()
        self.run(0, self.G, single_stream=True); self.sync(); self.read_timing()
        for k in range(groups):                      # one group at a time: no encode runs beside a forward
            self.run(k * self.G, self.G, single_stream=True); self.sync()
        tot = self.read_timing()
        return {k: v[0] / v[1] for k, v in tot.items()}, {k: v[1] for k, v in tot.items()}

    def encode_bytes(self, n_batches_in_launch):
        """algorithmic bytes of one encode launch over the first batches of the pool: column bytes + ref + 18 int32 out"""
        m = n_batches_in_launch * self.mcols
        return int(self.cols.col_off[m]) + m * (1 + 72)

    # ---- parity of what a run wrote (bench.py "parity_sample"; tests/test_gpu_stage_parity.py) --------------------------
    def parity_ranges(self, n_done, per_batch=2048, n_ranges=32):
        """site ranges a parity sample looks at: the first `per_batch` windows of n_ranges batches spread over the pool, so that
        every forward stream (batch i runs on stream i mod S) and every encode group of a sweep is represented"""
        nb = max(1, min(int(n_done) // self.batch, self.n_batches))
        step = max(1, nb // n_ranges) | 1                                 # odd stride: walks through the residues mod 32
        picks = sorted({(k * step) % nb for k in range(min(n_ranges, nb))})
        return [(b * self.batch, min(per_batch, self.batch)) for b in picks]

    def snapshot(self, ranges, ring_batches=4):
        """host copies of what the last run() calls left behind: probabilities and calls of the windows in `ranges`
        [(first site, count)], and the first `ring_batches` batches of every ring slot's count buffer together with the column
        range the slot was encoded from"""
        t = self.torch
        self.sync()
        idx = t.cat([t.arange(a, a + c, device=self.dev) for a, c in ranges])
        snap = {"ranges": list(ranges), "gt": self.gt_all[idx].cpu().numpy(), "zy": self.zy_all[idx].cpu().numpy(),
                "res": {k: v[idx].cpu().numpy() for k, v in self.res.items()}, "ring": []}
        for slot in self.ring:
            if "c0" not in slot or ring_batches <= 0:
                continue
            m = min(slot["ncols"], ring_batches * self.mcols)
            snap["ring"].append({"c0": slot["c0"], "m": m, "counts": slot["counts"][:m].cpu().numpy(),
                                 "depth": slot["depth"][:m].cpu().numpy(), "flags": slot["flags"][:m].cpu().numpy()})
        return snap

    def _oracle_encode(self, c0, m):
        from oracle import oracle
        off = self.cols.col_off[c0:c0 + m + 1]
        return oracle.encode_columns(self.cols.bases[int(off[0]):int(off[-1])], off - off[0], self.cols.ref[c0:c0 + m])

    def float64_error(self, snap, per_range=64):
        """max |p - p64| of a snapshot()'s probabilities on the first `per_range` windows of each of its ranges, p64 = the model evaluated
        in float64 (oracle.pileup_forward_f64): separates the arithmetic error of a mode from fp32 summation-order noise"""
        import numpy as np
        from oracle import oracle
        key = tuple((a, min(c, per_range)) for a, c in snap["ranges"])
        if getattr(self, "_f64_key", None) != key:
            xs = [self._oracle_encode(a * 33, c * 33)[0].reshape(c, 33, 18) for a, c in key]
            self._f64 = oracle.pileup_forward_f64(self.weights, np.concatenate(xs))
            self._f64_key = key
        sel, off = [], 0
        for (a, c), (_, ck) in zip(snap["ranges"], key):
            sel.append(np.arange(off, off + ck)); off += c
        sel = np.concatenate(sel)
        return {"sites": int(sel.size), "max_abs_error": float(max(np.abs(snap["gt"][sel] - self._f64[0]).max(), np.abs(snap["zy"][sel] - self._f64[1]).max()))}

    def parity_check(self, snap, tolerance=1e-4, nthreads=None):
        """compares a snapshot() with the oracle (the plain restatement - the pinned checker, not the blocked arrangement bench.py
        times) on the same windows: probabilities within `tolerance` (BASELINE north_star: 1e-4 abs), calls = np.argmax / np.max
        of the run's own probabilities (predict.py:54-57), calls vs the oracle's except where its two best classes are closer than
        2 x tolerance, every ring slot's counts / depth / flags bit for bit -> the "parity_sample" object of a bench line"""
        import numpy as np
        from oracle import oracle
        from tools.bench_common import usable_cores
        xs = []
        for a, c in snap["ranges"]:
            counts, _, _ = self._oracle_encode(a * 33, c * 33)
            xs.append(counts.reshape(c, 33, 18))
        x = np.concatenate(xs)
        n = x.shape[0]
        ogt, ozy = oracle.pileup_forward(self.weights, x, nthreads=nthreads or usable_cores())
        gt, zy = snap["gt"], snap["zy"]
        dp = float(max(np.abs(gt - ogt).max(), np.abs(zy - ozy).max()))
        finite = bool(np.isfinite(gt).all() and np.isfinite(zy).all())
        r = snap["res"]
        calls_self = bool(np.array_equal(r["ga"], gt.argmax(1).astype(np.uint8)) and np.array_equal(r["za"], zy.argmax(1).astype(np.uint8))
                          and np.array_equal(r["gm"], gt.max(1)) and np.array_equal(r["zm"], zy.max(1)))
        flips = 0
        for mine, ref in ((r["ga"], ogt), (r["za"], ozy)):
            diff = np.nonzero(mine != ref.argmax(1))[0]
            if diff.size:
                top2 = np.sort(ref[diff], axis=1)[:, -2:]
                if bool(((top2[:, 1] - top2[:, 0]) > 2 * tolerance).any()):
                    flips = -1                                          # a call differs where the oracle has a clear winner
                    break
                flips += int(diff.size)
        enc_ok, ring_cols = True, 0
        for s in snap["ring"]:
            oc, od, of = self._oracle_encode(s["c0"], s["m"])
            enc_ok &= bool(np.array_equal(s["counts"], oc) and np.array_equal(s["depth"], od) and np.array_equal(s["flags"], of))
            ring_cols += s["m"]
        ok = finite and dp <= tolerance and calls_self and flips >= 0 and enc_ok and n > 0
        return {"ok": bool(ok), "sites": int(n), "max_abs_dp": dp, "tolerance": tolerance, "encode_bit_exact": enc_ok,
                "ring_slots_checked": len(snap["ring"]), "ring_columns_checked": int(ring_cols), "calls_equal_own_argmax": calls_self,
                "calls_differing_from_oracle_at_near_ties": int(max(flips, 0)), "call_differs_at_a_clear_winner": flips < 0,
                "batches_sampled": len(snap["ranges"]),
                "what": "what the timed region's own run left behind (all streams, ring of count buffers) against oracle/liboracle.so "
                        "(plain restatement): softmax probabilities of `sites` pool windows spread over the batches, argmax / max calls, "
                        "and the count buffers still in the ring"}

    def compact_calls(self, n_done):
        t = self.torch
        return t.stack([self.res["ga"][:n_done].float(), self.res["za"][:n_done].float(), self.res["gm"][:n_done], self.res["zm"][:n_done]], dim=1)


def pileup_rooflines(stage, tot, excl, excl_n, sites_per_gpu, dt, precision, enc_group, workload="pileup"):
    """roofline objects of the pileup stage.  Headline = the dominant forward kernel - the one whose launches take longest with
    the chip to themselves (in-region durations depend on which launches happened to share the chip and let the choice flip between
    two kernels of nearly equal weight; ties go to the one with more executed flops) - priced
    on its EXCLUSIVE launches (one stream, nothing else on the chip; the duration a rocprofv3 kernel trace shows for the same
    launches): executed flops per launch / average launch duration / peak.  `chip` = executed forward flops of all timed sites
    over the wall time of the timed region (encode, post-processing and the gather included in the time)."""
    from tools import bench_common as bc
    batch = stage.batch * stage.F                       # sites per forward launch
    peak = bc.PEAK_F32_MFMA_TFLOPS if precision == 0 else bc.PEAK_F16_MFMA_TFLOPS       # fp16 and bf16 MFMAs share one dense peak
    fwd_keys = [k for k in tot if k in bc.PILEUP_EXEC_FLOP and (precision != 2 or k in bc.PILEUP_EXEC_FLOP_BF16X3)]
    xf = lambda k: bc.pileup_exec_flop(k, precision)   # executed MFMA flops per site (f16x3: 3, bf16x3: 6 MFMAs per fp32 product)
    out = {}
    if fwd_keys:
        dom = max(fwd_keys, key=lambda k: (round(excl.get(k, 0.0), 5), bc.PILEUP_EXEC_FLOP[k]))
        if dom in excl:
            how = ("HIP events around every launch of the kernel, one stream, nothing else running (after the timed region; %d launches)"
                   % excl_n[dom])
            roof = bc.roofline_mfma(dom, xf(dom) * batch, excl[dom], excl_n[dom],
                                    alg_flop_per_launch=bc.PILEUP_ALG_FLOP[dom] * batch, peak=peak, how=how,
                                    traffic=bc.committed_traffic(workload, dom, batch=batch, precision=precision))
            fwd_exec = sum(xf(k) for k in fwd_keys)
            chip = fwd_exec * sites_per_gpu / dt / 1e12
            roof["chip"] = {"achieved": chip, "frac": chip / peak, "unit": "TFLOP/s",
                            "achieved_algorithmic_tflops": bc.PILEUP_ALG_FLOP_FORWARD * sites_per_gpu / dt / 1e12,
                            "note": "per GPU: executed forward flops of all timed sites / wall time of the timed region (encode, post-processing "
                                    "and the gather included in the time); the algorithmic figure prices the reference's 12.55 MFLOP/site and is not a fraction"}
            roof["in_region_avg_launch_ms"] = tot[dom][0] / tot[dom][1]
            roof["in_region_note"] = ("in the timed region launches of several streams share the chip, so their durations overlap and are not "
                                      "additive; no fraction is derived from them")
            out["roofline"] = roof
            # the other recurrence layer beside it, the same way (layers 0 and 1 are within 5 % of each other in time)
            for k in fwd_keys:
                if k != dom and k in excl and k in ("pileup_l0", "pileup_l1f", "pileup_l1"):
                    out["roofline_" + k] = bc.roofline_mfma(k, xf(k) * batch, excl[k], excl_n[k],
                                                            alg_flop_per_launch=bc.PILEUP_ALG_FLOP[k] * batch, peak=peak, how=how,
                                                            traffic=bc.committed_traffic(workload, k, batch=batch, precision=precision))
    if "encode_columns" in excl:
        nbytes = stage.encode_bytes(enc_group)
        e = bc.roofline_hbm("encode_columns", nbytes, excl["encode_columns"], excl_n["encode_columns"],
                            traffic=bc.committed_traffic(workload, "encode_columns", batch=batch, columns=enc_group * stage.mcols),
                            how="HIP events around every launch, nothing else running; one launch encodes %d batches = %d columns"
                                % (enc_group, enc_group * stage.mcols), batches_per_launch=enc_group, columns_per_launch=enc_group * stage.mcols)
        vb = bc.valu_issue_bound("k_encode_columns", stage.coverage, enc_group * stage.mcols, excl["encode_columns"])
        if vb:
            e["second_bound"] = vb
        out["roofline_encode"] = e
    return out
