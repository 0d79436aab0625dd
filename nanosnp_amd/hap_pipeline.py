"""Stage s5 of the reference pipeline, streamed from host memory:

    haplotype bins (read planes)  ->  haplotype features x 2 -> HaplotypeModel forward -> argmax / max  ->  haplotype.csv

replacing the loop of ``HaplotypeModel/predict_dev.py:27-48``: a ``TestDataset`` (``dataset_dev.py:337-349``) that reads a whole
HDF5 bin into numpy (``:92-172``), a ``DataLoader`` whose four worker processes reduce one site per ``__getitem__`` call
(``get_frequency_feature``, 1.6 ms each), a blocking ``.to(device)`` per batch and a Python loop over the sites of every batch.

Here the read planes of a pass of sites (default 16,384) move through three stations that work at the same time:

    worker thread   reads passes k + 1 and k + 2 from the site file (pread from the page cache, all host cores, straight into one
                    of three pinned buffer sets; the reference's int32 planes are narrowed to int8 on the way when every value
                    fits - a quarter of the bytes over PCIe - and the fixed-width "ctg:pos" fields become integer arrays)
    copy stream     sends pass k to one of three device buffer sets
    compute stream  reference rows (a gather from the reference sequence resident in HBM) -> nsnp_hap_features(_i8) x 2 ->
                    nsnp_hap_forward -> argmax / max of pass k - 1, the calls (5 bytes per site) on their way back to a pinned array

and the csv rows of all sites are written by one native call (``nsnp_hap_csv_format``) at the end.  Under torch.distributed every
rank takes the contiguous ``shard_range`` of the sites formats its own rows, and the text is gathered to rank 0 (rank order = position order).
There is no CPU path: a missing HIP library or GPU raises.
"""
from __future__ import annotations

import os
from collections import namedtuple

import numpy as np

from . import host, sitefile

PILEUP_PLANES = ("pileup_sequences", "pileup_baseq", "pileup_mapq", "pileup_hap")            # argument order of nsnp_hap_features
HAPLOTYPE_PLANES = ("haplotype_sequences", "haplotype_baseq", "haplotype_mapq", "haplotype_hap")


HapCalls = namedtuple("HapCalls", "table contig_id pos gt_arg gt_max probabilities")
HapCalls.__doc__ = """calls of stream_haplotype in site order: table (host.ContigTable) names the contigs contig_id (int32) indexes; pos int64;
gt_arg uint8 (index into options.gt_decoded_labels[0:10]); gt_max float32; probabilities float32 [n, 10] or None"""


class _LocalNames:
    """contig names met in candidate fields when no DeviceReference is in play (the source carries its own reference rows)"""
    def __init__(self):
        self.names, self.table = [], host.ContigTable([])

    def add_names(self, names):
        new = [n for n in dict.fromkeys(names) if n not in self.table.index]
        if new:
            self.names += new
            self.table = host.ContigTable(self.names)


# ---- the reference sequence in HBM ------------------------------------------------------------------------------------------
class DeviceReference:
    """Reference contigs as BASE2INT codes (dataset_dev.py:9: A C G T -> 1..4, everything else - N, lower case, IUPAC - 0, which is
    what the reference's bare ``except`` makes of a KeyError) in one uint8 device tensor, with the per-contig offsets and lengths the
    row gather needs.  Build it once per run and hand it to every stream_haplotype call."""

    def __init__(self, references: dict, device=0):
        import torch
        self.dev = device if isinstance(device, torch.device) else (torch.device(device) if isinstance(device, str) else torch.device("cuda", int(device)))
        self.names = list(references)
        lut = np.zeros(256, np.uint8)
        for k, v in host._BASE2INT.items():
            lut[k] = v
        seqs = [np.frombuffer(v, np.uint8) if isinstance(v, (bytes, bytearray)) else
                (np.frombuffer(v.encode(), np.uint8) if isinstance(v, str) else np.ascontiguousarray(v, np.uint8)) for v in references.values()]
        lens = np.array([s.size for s in seqs], np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        codes = np.concatenate([lut[s] for s in seqs]) if seqs and offs[-1] else np.zeros(1, np.uint8)
        self.codes = torch.from_numpy(codes).to(self.dev)
        self._lens, self._offs = lens, offs[:-1].copy()
        self._upload_tables()

    def _upload_tables(self):
        import torch
        self.table = host.ContigTable(self.names)
        n = max(1, len(self.names))
        lens = np.zeros(n, np.int64); offs = np.zeros(n, np.int64)
        lens[:len(self._lens)] = self._lens; offs[:len(self._offs)] = self._offs
        self.length = torch.from_numpy(lens).to(self.dev)
        self.offset = torch.from_numpy(offs).to(self.dev)

    def add_names(self, names):
        """contigs that appear in position fields but not in the reference: known by name (the csv rows carry it), length 0 (every
        lookup fails -> 0, as references[ctg] raising KeyError does)"""
        new = [n for n in dict.fromkeys(names) if n not in self.table.index]
        if new:
            self.names += new
            self._lens = np.concatenate([self._lens, np.zeros(len(new), np.int64)])
            self._offs = np.concatenate([self._offs, np.zeros(len(new), np.int64)])
            self._upload_tables()

    def rows(self, ctg, rp):
        """ctg int32/int64 [n, L] contig ids (-1 = unknown), rp int64 [n, L] 0-based positions (device tensors) -> int32 [n, L]:
        references[ctg][rp] through BASE2INT, 0 where the reference's lookup raises - unknown contig, rp >= len, rp < -len; a negative
        rp inside [-len, -1] wraps around as Python indexing does (dataset_dev.py:111-118,155-160)."""
        import torch
        c = ctg.to(torch.int64)
        known = c >= 0
        cc = torch.where(known, c, torch.zeros_like(c))
        L = self.length[cc]
        valid = known & (rp < L) & (rp >= -L)
        idx = torch.where(rp < 0, rp + L, rp)
        g = torch.where(valid, self.offset[cc] + idx, torch.zeros_like(idx))
        return (self.codes[g].to(torch.int32) * valid.to(torch.int32)).contiguous()


# ---- sources of read planes --------------------------------------------------------------------------------------------------
class HapBinSource:
    """A haplotype site file (nanosnp_amd.sitefile, the flat stand-in of haplotype_bins/<ctg>_<s>_<e>.bin): planes are pread into
    the staging buffers, position fields are parsed natively per pass."""

    def __init__(self, path):
        self.path = str(path)
        idx = sitefile.array_index(self.path)
        missing = [k for k in sitefile.HAP_PLANES + ("candidate_positions", "haplotype_positions") if k not in idx]
        if missing:
            raise sitefile.SiteFileError(f"{path}: not a haplotype bin (missing {missing})")
        self.idx = idx
        dts = {idx[k][0] for k in sitefile.HAP_PLANES}
        if len(dts) != 1 or dts.pop() not in (np.dtype(np.int8), np.dtype(np.int32)):
            raise sitefile.SiteFileError(f"{path}: the eight read planes must share one dtype, int8 or int32")
        self.elem = idx["pileup_sequences"][0].itemsize
        self.n, self.Dp, Lp = idx["pileup_sequences"][1]
        nh, self.Dh, Lh = idx["haplotype_sequences"][1]
        if (Lp, Lh) != (33, 11) or nh != self.n:
            raise sitefile.SiteFileError(f"{path}: expected [N,D,33] pileup and [N,D,11] haplotype planes")
        for k in sitefile.HAP_PLANES:
            want = (self.n, self.Dp, 33) if k.startswith("pileup") else (self.n, self.Dh, 11)
            if idx[k][1] != want:
                raise sitefile.SiteFileError(f"{path}: {k} has shape {idx[k][1]}, expected {want}")
        self.fd = os.open(self.path, os.O_RDONLY)
        arrs = sitefile.read_arrays(self.path, mmap=True)
        self._cand, self._hpos = arrs["candidate_positions"], arrs["haplotype_positions"]
        self.ref_rows = None

    def close(self):
        if self.fd >= 0:
            os.close(self.fd); self.fd = -1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stage_plane(self, name, lo, hi, dst):
        """sites [lo, hi) of plane `name` -> dst (numpy view of a pinned buffer, int8 or int32); returns values that did not fit int8"""
        _, (_, D, L), off, _ = self.idx[name]
        return host.stage_values(dst, (hi - lo) * D * L, fd=self.fd, src_off=off + lo * D * L * self.elem,
                                 src_dtype=np.int8 if self.elem == 1 else np.int32)

    def position_fields(self, lo, hi):
        return np.asarray(self._cand[lo:hi]), np.asarray(self._hpos[lo:hi])


class HapArraySource:
    """Read planes already in host memory (numpy arrays or memmaps): planes_pileup / planes_haplotype = (seq, baseq, mapq, hap
    [, ref_row]) as predict_haplotype takes them.  With ref rows given they are uploaded as they are; without, candidate_positions /
    haplotype_positions ("ctg:pos" strings or byte fields) are needed and the rows come from a DeviceReference."""

    def __init__(self, planes_pileup, planes_haplotype, candidate_positions, haplotype_positions=None):
        pp, ph = list(planes_pileup), list(planes_haplotype)
        dts = {np.asarray(a).dtype for a in pp[:4] + ph[:4]}
        if len(dts) != 1 or next(iter(dts)) not in (np.dtype(np.int8), np.dtype(np.int32)):
            raise ValueError("the eight read planes must share one dtype, int8 or int32")
        self.elem = next(iter(dts)).itemsize
        self.planes = {k: np.ascontiguousarray(a) for k, a in zip(PILEUP_PLANES + HAPLOTYPE_PLANES, pp[:4] + ph[:4])}
        self.n, self.Dp, Lp = self.planes["pileup_sequences"].shape
        nh, self.Dh, Lh = self.planes["haplotype_sequences"].shape
        if (Lp, Lh) != (33, 11) or nh != self.n or len(candidate_positions) != self.n:
            raise ValueError("expected [N,D,33] pileup planes, [N,D,11] haplotype planes and N candidate positions")
        self.ref_rows = None
        if len(pp) > 4 and len(ph) > 4:
            self.ref_rows = (np.ascontiguousarray(pp[4], np.int32), np.ascontiguousarray(ph[4], np.int32))
        self._cand = _as_fields(candidate_positions)
        self._hpos = _as_fields(haplotype_positions) if haplotype_positions is not None else None
        if self.ref_rows is None and self._hpos is None:
            raise ValueError("either reference rows or haplotype_positions are needed")

    def close(self):
        pass

    def stage_plane(self, name, lo, hi, dst):
        a = self.planes[name]
        per = a.shape[1] * a.shape[2]
        return host.stage_values(dst, (hi - lo) * per, src=a.reshape(-1), src_off=lo * per * self.elem, src_dtype=a.dtype)

    def position_fields(self, lo, hi):
        return self._cand[lo:hi], (self._hpos[lo:hi] if self._hpos is not None else None)


def _as_fields(strings):
    """list of str / bytes (possibly nested one level) or a uint8 array -> zero-padded uint8 fields [..., width]"""
    if isinstance(strings, np.ndarray) and strings.dtype == np.uint8:
        return strings
    a = np.array(strings, dtype=np.bytes_)                   # 'S<width>', zero-padded like the HDF5 atoms
    w = max(a.dtype.itemsize, 1)
    return np.frombuffer(a.tobytes(), np.uint8).reshape(a.shape + (w,)) if a.size else np.zeros(a.shape + (1,), np.uint8)


class _NarrowOverflow(Exception):
    pass


# ---- buffer sets ---------------------------------------------------------------------------------------------------------------
class _Set:
    """buffers of one pass in flight: the eight read planes (as bytes: int8 or int32 views are taken per pass), the position arrays
    and / or the reference rows; pinned on the host side, plain on the device side"""

    def __init__(self, P, Dp, Dh, elem, dev=None):
        import torch
        kw = dict(pin_memory=True) if dev is None else dict(device=dev)
        self.P, self.Dp, self.Dh, self.elem = P, Dp, Dh, elem
        self.planes = {k: torch.empty(P * (Dp * 33 if k.startswith("pileup") else Dh * 11) * elem, dtype=torch.uint8, **kw)
                       for k in PILEUP_PLANES + HAPLOTYPE_PLANES}
        self.cand_pos = torch.empty(P, dtype=torch.int64, **kw); self.cand_ctg = torch.empty(P, dtype=torch.int32, **kw)
        self.hap_pos = torch.empty((P, 11), dtype=torch.int64, **kw); self.hap_ctg = torch.empty((P, 11), dtype=torch.int32, **kw)
        self.ref_p = torch.empty((P, 33), dtype=torch.int32, **kw); self.ref_h = torch.empty((P, 11), dtype=torch.int32, **kw)
        self.h2d_done = None            # host sets: the copy engine has read this set
        self.free = None                # device sets: the last kernel reading this set is done

    def fits(self, P, Dp, Dh, elem):
        return self.P >= P and self.P * self.Dp * self.elem >= P * Dp * elem and self.P * self.Dh * self.elem >= P * Dh * elem

    def nbytes(self):
        return sum(t.numel() for t in self.planes.values())


def stream_haplotype(ctx, source, reference=None, lo=0, hi=None, pass_sites=16384, narrow=True, stats=None, keep_probabilities=False):
    """Sites [lo, hi) of `source` (HapBinSource / HapArraySource) through haplotype features + HaplotypeModel forward + argmax / max,
    pass by pass, staging / H2D / compute overlapped (module docstring).  ctx: a Context with hap weights loaded (its hap_precision
    option selects the arithmetic).  reference: a DeviceReference (needed unless the source carries its own reference rows).
    narrow: int32 planes are narrowed to int8 while they are staged when every value fits (bit-identical features from a quarter of
    the bytes); False sends them as int32.

    Returns HapCalls (numpy arrays in site order; with keep_probabilities also the [n, 10] probabilities).  stats (dict) receives
    per-station busy times and byte counts.  The buffer sets are kept on ctx between calls: one call at a time per context."""
    return stream_segments(ctx, [(source, lo, hi)], reference, pass_sites, narrow, stats, keep_probabilities)[0]


def stream_segments(ctx, segments, reference=None, pass_sites=16384, narrow=True, stats=None, keep_probabilities=False, on_segment=None):
    with host.gc_paused():
        return _stream_segments(ctx, segments, reference, pass_sites, narrow, stats, keep_probabilities, on_segment)


def _stream_segments(ctx, segments, reference, pass_sites, narrow, stats, keep_probabilities, on_segment):
    """stream_haplotype over several (source, lo, hi) segments - the bins of a directory - as ONE pipeline: the first pass of segment
    f + 1 is staged and copied while the last passes of segment f compute, so a directory of bins pays the pipeline's fill and drain
    once, not once per file.  on_segment(index, HapCalls), when given, is called on a writer thread as soon as the calls of a segment
    are back on the host, in segment order, while later segments are still computing (predict_haplotype_bins formats and writes the
    csv rows there).  Returns the list of HapCalls."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    import torch
    dev = torch.device("cuda", ctx.device)
    st = stats if stats is not None else {}
    for k in ("stage_s", "h2d_s", "gpu_s", "bytes_staged", "bytes_h2d", "sites", "passes", "passes_int8", "wait_stage_s", "issue_s", "drain_s", "setup_s"):
        st.setdefault(k, 0.0)
    t_enter = time.perf_counter()
    segs = []
    for source, lo, hi in segments:
        hi = source.n if hi is None else min(int(hi), source.n)
        lo = max(0, int(lo or 0))
        if source.ref_rows is None and reference is None:
            raise ValueError("stream_haplotype: the source carries no reference rows: a DeviceReference is needed")
        segs.append((source, lo, max(lo, hi)))
    names = reference if reference is not None else _LocalNames()
    seg_off = np.concatenate([[0], np.cumsum([hi - lo for _, lo, hi in segs])]).astype(np.int64)
    n = int(seg_off[-1])
    P = int(max(1, pass_sites))
    # the passes of all segments in order; the very first one is a quarter pass (the pipeline's fill: nothing computes while it is
    # staged and copied)
    passes = []
    for f, (source, lo, hi) in enumerate(segs):
        a = lo
        while a < hi:
            b = min(hi, a + (max(1, P // 4) if not passes and hi - lo > P else P))
            passes.append((f, a, b, int(seg_off[f]) + a - lo))
            a = b
    last_pass_of = {f: k for k, (f, _, _, _) in enumerate(passes)}
    empty = lambda: HapCalls(names.table, np.empty(0, np.int32), np.empty(0, np.int64), np.empty(0, np.uint8), np.empty(0, np.float32),
                             np.empty((0, 10), np.float32) if keep_probabilities else None)
    if not passes:
        out = [empty() for _ in segs]
        if on_segment is not None:
            for f, c in enumerate(out):
                on_segment(f, c)
        return out
    Pmax = max(b - a for _, a, b, _ in passes)
    Dp, Dh = max(s_.Dp for s_, _, _ in segs), max(s_.Dh for s_, _, _ in segs)
    e_of = lambda src: 1 if (narrow or src.elem == 1) else src.elem              # bytes per value in the staging sets and over PCIe
    elem = max(e_of(s_) for s_, lo, hi in segs if hi > lo)
    n_sets = min(3, len(passes))
    # pinned buffers are expensive to create (page-locking): kept on the context between calls, with their device twins
    hsets = getattr(ctx, "_hap_host_sets", None)
    if not hsets or len(hsets) < n_sets or not all(s_.fits(Pmax, Dp, Dh, elem) for s_ in hsets):
        hsets = [_Set(Pmax, Dp, Dh, elem) for _ in range(n_sets)]
        ctx._hap_host_sets = hsets
    dsets = getattr(ctx, "_hap_dev_sets", None)
    if not dsets or len(dsets) < n_sets or not all(s_.fits(Pmax, Dp, Dh, elem) for s_ in dsets):
        dsets = [_Set(Pmax, Dp, Dh, elem, dev) for _ in range(n_sets)]
        ctx._hap_dev_sets = dsets
    if getattr(ctx, "_hap_copy_stream", None) is None:
        ctx._hap_copy_stream = host.copy_stream(dev)
    copy_stream = ctx._hap_copy_stream
    main = torch.cuda.current_stream(dev)
    for s_ in hsets:
        s_.h2d_done = None
    for s_ in dsets:
        s_.free = None
    copy_stream.wait_stream(main)
    # the calls of all passes: device arrays, copied back once per pass into pinned arrays (5 bytes per site)
    res = getattr(ctx, "_hap_results", None)
    if res is None or res[0].numel() < n:
        res = (torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
               torch.empty(n, dtype=torch.uint8, pin_memory=True), torch.empty(n, dtype=torch.float32, pin_memory=True))
        ctx._hap_results = res
    d_ga, d_gm, h_ga, h_gm = res
    probs = torch.empty((n, 10), dtype=torch.float32, device=dev) if keep_probabilities else None
    cand_pos_all = np.empty(n, np.int64); cand_ctg_all = np.empty(n, np.int32)
    off33 = torch.arange(-16, 17, dtype=torch.int64, device=dev)[None, :]

    def stage(k):
        """pass k -> host set k % n_sets (worker thread)"""
        t0 = time.perf_counter()
        f, a, b, o = passes[k]
        source = segs[f][0]
        m = b - a
        hs = hsets[k % n_sets]
        e_out = e_of(source)
        nbytes = 0
        for name in PILEUP_PLANES + HAPLOTYPE_PLANES:
            cnt = m * (source.Dp * 33 if name.startswith("pileup") else source.Dh * 11)
            v = hs.planes[name].numpy()[:cnt * e_out].view(np.int8 if e_out == 1 else np.int32)
            if source.stage_plane(name, a, b, v):
                raise _NarrowOverflow(name)              # a value outside int8 (e.g. a mapping quality of 255)
            nbytes += v.nbytes
        cand_f, hap_f = source.position_fields(a, b)
        cand_f = cand_f.reshape(m, -1)
        cp, cc = host.parse_ctg_pos(cand_f, names.table)
        if (cc < 0).any():
            # a candidate on a contig the table does not hold yet: known by name from here on (ids already handed out keep their meaning)
            names.add_names([bytes(r).rstrip(b"\0").split(b":")[0].decode() for r in cand_f[cc < 0]])
            cp, cc = host.parse_ctg_pos(cand_f, names.table)
        hs.cand_pos.numpy()[:m] = cp; hs.cand_ctg.numpy()[:m] = cc
        cand_pos_all[o:o + m] = cp; cand_ctg_all[o:o + m] = cc
        if source.ref_rows is not None:
            hs.ref_p.numpy()[:m] = source.ref_rows[0][a:b]; hs.ref_h.numpy()[:m] = source.ref_rows[1][a:b]
        else:
            hp, hc = host.parse_ctg_pos(hap_f.reshape(m, 11, -1), names.table)
            hs.hap_pos.numpy()[:m] = hp; hs.hap_ctg.numpy()[:m] = hc
        return e_out, nbytes, time.perf_counter() - t0

    def calls_of(f, done=None):
        """the calls of segment f as host arrays (writer thread: waits for the segment's last copies first)"""
        if done is not None:
            done.synchronize()
        o0, o1 = int(seg_off[f]), int(seg_off[f + 1])
        c = HapCalls(names.table, cand_ctg_all[o0:o1].copy(), cand_pos_all[o0:o1].copy(), h_ga[o0:o1].numpy().copy(), h_gm[o0:o1].numpy().copy(),
                     probs[o0:o1].cpu().numpy() if probs is not None else None)
        if on_segment is not None:
            on_segment(f, c)
        return c

    tev = lambda: torch.cuda.Event(enable_timing=True)
    ev = [dict(h0=tev(), h1=tev(), c0=tev(), c1=tev()) for _ in passes]
    st["setup_s"] += time.perf_counter() - t_enter
    seg_futs = [None] * len(segs)
    next_seg = 0                                         # segments up to here have had their writer job submitted (in order)
    with ThreadPoolExecutor(max_workers=1) as pool, ThreadPoolExecutor(max_workers=1) as writer:
        futs = [pool.submit(stage, j) for j in range(min(2, len(passes)))]
        for k, (f, a, b, o) in enumerate(passes):
            source = segs[f][0]
            m = b - a
            t_w = time.perf_counter()
            try:
                e_out, nbytes, t_stage = futs[k].result()
            except _NarrowOverflow:
                # int32 planes that do not fit int8 after all: the whole call is done again with the planes as they are (the writer
                # of nanosnp_amd.sitefile stores int8 whenever it can, so this is a file of reference dtype holding e.g. mapq 255).
                # Segments whose rows have already gone out through on_segment are not repeated.
                for f_ in futs[k + 1:]:
                    f_.cancel()
                pool.shutdown(wait=True)
                writer.shutdown(wait=True)
                torch.cuda.synchronize(dev)
                st["narrow_restarts"] = st.get("narrow_restarts", 0) + 1
                head = [sf.result() for sf in seg_futs[:next_seg]]
                cb = (lambda i, c: on_segment(i + next_seg, c)) if on_segment is not None else None
                return head + stream_segments(ctx, segs[next_seg:], reference, pass_sites, False, stats, keep_probabilities, cb)
            t_i = time.perf_counter()
            st["wait_stage_s"] += t_i - t_w; st["stage_s"] += t_stage; st["bytes_staged"] += nbytes
            st["passes"] += 1; st["passes_int8"] += int(e_out == 1); st["sites"] += m
            hs, ds = hsets[k % n_sets], dsets[k % n_sets]
            Dps, Dhs = source.Dp, source.Dh
            # ---- H2D on the copy stream, behind the last readers of this device set (pass k - 3) ----
            if ds.free is not None:
                copy_stream.wait_event(ds.free)
            with torch.cuda.stream(copy_stream):
                ev[k]["h0"].record(copy_stream)
                for name in PILEUP_PLANES + HAPLOTYPE_PLANES:
                    cnt = m * (Dps * 33 if name.startswith("pileup") else Dhs * 11) * e_out
                    ds.planes[name][:cnt].copy_(hs.planes[name][:cnt], non_blocking=True)
                ds.cand_pos[:m].copy_(hs.cand_pos[:m], non_blocking=True); ds.cand_ctg[:m].copy_(hs.cand_ctg[:m], non_blocking=True)
                if source.ref_rows is not None:
                    ds.ref_p[:m].copy_(hs.ref_p[:m], non_blocking=True); ds.ref_h[:m].copy_(hs.ref_h[:m], non_blocking=True)
                else:
                    ds.hap_pos[:m].copy_(hs.hap_pos[:m], non_blocking=True); ds.hap_ctg[:m].copy_(hs.hap_ctg[:m], non_blocking=True)
                ev[k]["h1"].record(copy_stream)
            hs.h2d_done = ev[k]["h1"]
            st["bytes_h2d"] += nbytes + m * (44 * 4 if source.ref_rows is not None else 12 * 12)
            if k + 2 < len(passes):
                nxt = hsets[(k + 2) % n_sets]
                if nxt.h2d_done is not None:
                    nxt.h2d_done.synchronize()          # the copy engine is done with the set the worker is about to overwrite
                futs.append(pool.submit(stage, k + 2))
            # ---- compute stream ----
            main.wait_event(ev[k]["h1"])
            ev[k]["c0"].record(main)
            if source.ref_rows is not None:
                ref_p, ref_h = ds.ref_p[:m], ds.ref_h[:m]
            else:
                ref_p = reference.rows(ds.cand_ctg[:m, None].expand(m, 33), ds.cand_pos[:m, None] - 1 + off33)
                ref_h = reference.rows(ds.hap_ctg[:m], ds.hap_pos[:m] - 1)
            tdt = torch.int8 if e_out == 1 else torch.int32
            pl = lambda name, D, L: ds.planes[name][:m * D * L * e_out].view(tdt).view(m, D, L)
            xp = ctx.hap_features(*[pl(nm, Dps, 33) for nm in PILEUP_PLANES], ref_p)
            xh = ctx.hap_features(*[pl(nm, Dhs, 11) for nm in HAPLOTYPE_PLANES], ref_h)
            gt, _ = ctx.hap_forward(xp, xh)
            gm, ga = gt.max(dim=1)                                                # predict_dev.py:40-43
            d_ga[o:o + m] = ga.to(torch.uint8); d_gm[o:o + m] = gm
            if probs is not None:
                probs[o:o + m] = gt
            h_ga[o:o + m].copy_(d_ga[o:o + m], non_blocking=True); h_gm[o:o + m].copy_(d_gm[o:o + m], non_blocking=True)
            ev[k]["c1"].record(main)
            ds.free = torch.cuda.Event(); ds.free.record(main)
            # ---- segments that are complete with this pass (and empty ones in front of the next): their calls go to the writer thread ----
            if last_pass_of[f] == k:
                while next_seg <= f:
                    seg_futs[next_seg] = writer.submit(calls_of, next_seg, ds.free if next_seg == f else None)
                    next_seg += 1
            st["issue_s"] += time.perf_counter() - t_i
        t_d = time.perf_counter()
        torch.cuda.synchronize(dev)
        while next_seg < len(segs):                      # (empty segments behind the last pass)
            seg_futs[next_seg] = writer.submit(calls_of, next_seg, None)
            next_seg += 1
        out = [sf.result() for sf in seg_futs]
        st["drain_s"] += time.perf_counter() - t_d
    for e in ev:
        st["h2d_s"] += e["h0"].elapsed_time(e["h1"]) * 1e-3
        st["gpu_s"] += e["c0"].elapsed_time(e["c1"]) * 1e-3
    return out


stream_segments.__doc__ = _stream_segments.__doc__


def predict_haplotype_bins(ctx, bin_paths, reference, output_file, pass_sites=16384, narrow=True, score_mode=host.SCORE_FLOAT64, stats=None,
                           distributed=True):
    """The reference's ``predict(model, test_data, reference_path, ...)`` (predict_dev.py:27-48) over haplotype site files: the files of
    bin_paths (a directory - os.listdir order, as the reference iterates it - or a list of paths) go through ONE pipeline
    (stream_segments: the first pass of the next file is staged while the last passes of this one compute) and the rows
    ``ctg \t pos \t GT \t qual`` of every file are formatted and appended to output_file on a writer thread as soon as the file's calls
    are back, while later files compute.  reference: a DeviceReference, a dict {contig: sequence} (uploaded once) or the path of a FASTA
    file (read as get_truth.load_reference_file reads it: host.load_reference_file).
    Under torch.distributed every rank works on its shard_range of every file and formats its own rows, the text travels to rank 0 in
    one rooted gather and rank 0 writes (distributed=False: this process alone does the whole job even inside a process group).  Returns the number of rows
    written (on rank 0; 0 elsewhere)."""
    import time
    import torch
    import torch.distributed as tdist
    from .dist import gather_text, shard_range
    if isinstance(bin_paths, (str, os.PathLike)):
        d = str(bin_paths)
        paths = [os.path.join(d, f) for f in os.listdir(d)] if os.path.isdir(d) else [d]
    else:
        paths = [str(p) for p in bin_paths]
    if isinstance(reference, (str, os.PathLike)):      # predict_dev.py:28: references = load_reference_file(reference_path)
        reference = host.load_reference_file(reference)
    ref = reference if isinstance(reference, DeviceReference) else DeviceReference(reference, ctx.device)
    sharded = bool(distributed) and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    rank, world = (tdist.get_rank(), tdist.get_world_size()) if sharded else (0, 1)
    st = stats if stats is not None else {}
    st.setdefault("csv_s", 0.0)
    sources = []
    total = 0
    f = open(output_file, "wb") if rank == 0 else None
    try:
        for p in paths:
            sources.append(HapBinSource(p))
        segments = [(s_,) + shard_range(s_.n, rank, world) for s_ in sources]

        texts = {}                                      # sharded: this rank's rows of every file (final text) until the gather

        def write_rows(i, tbl, ctg, pos, ga, gm):
            nonlocal total
            t0 = time.perf_counter()
            if pos.size:
                text = host.hap_csv_format(tbl, ctg, pos, ga, gm, score_mode)
                if sharded:
                    texts[i] = text
                else:
                    f.write(text)
                    total += int(pos.size)
            st["csv_s"] += time.perf_counter() - t0

        if not sharded:
            stream_segments(ctx, segments, ref, pass_sites, narrow, st, on_segment=lambda i, c: write_rows(i, *c[:5]))
        else:
            # a csv row depends on its own site alone (predict_dev.py:40-47): every rank formats its rows beside its compute, as the
            # single process does, and the TEXT of all files travels to rank 0 in one rooted gather (the sizes first, as one small
            # object; a rank whose rows cannot be written - calculate_score on p == 1 under score_mode 0 - fails every rank)
            err = None
            try:
                stream_segments(ctx, segments, ref, pass_sites, narrow, st, on_segment=lambda i, c: write_rows(i, *c[:5]))
            except (host.HostError, ValueError, ArithmeticError) as e_:
                err = f"rank {rank}: {e_}"
            t_g = time.perf_counter()
            mine = [len(texts.get(i, b"")) for i in range(len(sources))]
            sizes = [None] * world
            tdist.all_gather_object(sizes, (mine, err))
            errs = [e_ for _, e_ in sizes if e_]
            if errs:
                raise host.HostError("; ".join(errs))
            backend_dev = torch.device("cuda", ctx.device) if tdist.get_backend() == "nccl" else "cpu"
            allt = gather_text(b"".join(texts[i] for i in sorted(texts)), backend_dev)
            if rank == 0:
                start = np.concatenate([[0], np.cumsum([sum(sz) for sz, _ in sizes])])
                within = [np.concatenate([[0], np.cumsum(sz)]) for sz, _ in sizes]
                for i in range(len(sources)):
                    for r in range(world):
                        f.write(allt[int(start[r] + within[r][i]):int(start[r] + within[r][i + 1])])
                        lo_, hi_ = shard_range(sources[i].n, r, world)
                        total += hi_ - lo_
            st["gather_s"] = st.get("gather_s", 0.0) + time.perf_counter() - t_g
    finally:
        for s_ in sources:
            s_.close()
        if f:
            f.close()
    return total
