"""Stage s1 + s2 of the reference pipeline in one pass on the device:

    <chr>.mpileup text + FASTA  ->  column encode -> candidate windows -> PileupModel -> pileup.vcf

replacing DNA_CreateCanSnpTensor -> DNA_CreatePredictData -> make_bin_predict_data.py ->
PileupModel/predict.py (dna_sv_tensor/src/scripts/make_predict_data.sh:184-234,
scripts/s2_pileup_model_predict.sh:11-16) and the four text/HDF5 files between them.
File reading and VCF writing are host work (native readers / writer in libnanosnp_host.so);
everything between lives in HBM.
"""
from __future__ import annotations

import mmap
import os

import numpy as np

from . import host
from .predict import COV_CHANNELS


# ---- text ranges -------------------------------------------------------------------------------------------------------------
def _as_bytes_like(text):
    """bytes / bytearray / mmap / numpy uint8 -> (object with find / rfind over the whole text, numpy uint8 view of it)"""
    if isinstance(text, np.ndarray):
        text = memoryview(np.ascontiguousarray(text, np.uint8)).cast("B")
        return bytes(text) if len(text) < (1 << 20) else _MvFind(text), np.frombuffer(text, np.uint8)
    return text, np.frombuffer(text, np.uint8)


class _MvFind:
    """find / rfind of a single byte over a memoryview, through numpy (large numpy inputs only)"""
    def __init__(self, mv):
        self.a = np.frombuffer(mv, np.uint8)

    def __len__(self):
        return int(self.a.size)

    def find(self, ch, lo, hi=None):
        hi = self.a.size if hi is None else hi
        step = 1 << 16
        for s0 in range(lo, hi, step):
            w = np.flatnonzero(self.a[s0:min(hi, s0 + step)] == ch[0])
            if w.size:
                return s0 + int(w[0])
        return -1

    def rfind(self, ch, lo, hi):
        step = 1 << 16
        e = hi
        while e > lo:
            s0 = max(lo, e - step)
            w = np.flatnonzero(self.a[s0:e] == ch[0])
            if w.size:
                return s0 + int(w[-1])
            e = s0
        return -1


def line_cuts(text, n_parts, lo=0, hi=None):
    """n_parts + 1 offsets cutting text[lo:hi] into parts of whole lines of about equal bytes (lo and hi themselves must be line
    boundaries: 0, len(text) or an offset just behind a newline)."""
    hi = len(text) if hi is None else hi
    cuts = [lo]
    for k in range(1, n_parts):
        g = max(cuts[-1], lo + (hi - lo) * k // n_parts)
        nl = text.find(b"\n", g, hi)
        cuts.append(hi if nl < 0 else nl + 1)
    cuts.append(hi)
    return cuts


def halo_range(text, lo, hi, halo=16):
    """[lo, hi) grown by up to `halo` whole lines on either side -> (lo_ext, hi_ext, lines added in front, lines added behind)"""
    n_txt = len(text)
    a, n_lo = lo, 0
    while n_lo < halo and a > 0:
        nl = text.rfind(b"\n", 0, a - 1)
        a = nl + 1                                   # (-1 + 1 = 0 when the first line is reached)
        n_lo += 1
    b, n_hi = hi, 0
    while n_hi < halo and b < n_txt:
        nl = text.find(b"\n", b, n_txt)
        b = n_txt if nl < 0 else nl + 1
        n_hi += 1
    return a, b, n_lo, n_hi


class _HostSet:
    """pinned host buffers of one text chunk in flight (the parser writes straight into them, the copy engine reads them)"""
    def __init__(self, cap_bytes):
        import torch
        cap_cols = cap_bytes // 8 + 2
        self.pos = torch.empty(cap_cols, dtype=torch.int64, pin_memory=True)
        self.off = torch.empty(cap_cols + 1, dtype=torch.int64, pin_memory=True)
        self.bases = torch.empty(cap_bytes, dtype=torch.uint8, pin_memory=True)
        self.np = (self.pos.numpy(), self.off.numpy(), self.bases.numpy())
        self.h2d_done = None


def stream_contig(model, text, contig, chr_seq, lo=0, hi=None, chunk_bytes=64 << 20, min_af=0.12, min_coverage=6, stats=None, on_rows=None):
    """The device part of stages s1 + s2 over the lines of text[lo:hi], chunk by chunk: a worker thread parses chunk k + 1
    (libnanosnp_host.so, OpenMP, straight into pinned buffers) while this thread sends chunk k to the device and runs column encode
    -> site selection -> PileupModel forward + argmax / max on it.  Every chunk is parsed with 16 lines of halo on either side
    (re-parsed, not exchanged) and calls the sites centred in its own lines, so the result does not depend on where the cuts fall.
    Returns the call rows [n, 13] float64 (position, argmax / max of both heads, the eight coverage channels: all exact in float64)
    as a device tensor in position order - or, with on_rows, hands every chunk's rows to that callback as soon as they are issued
    (call_contig formats the VCF rows of complete batches meanwhile) and returns None.  stats (a dict) receives per-stage busy times."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    import torch
    ctx = model.ctx
    dev = torch.device("cuda", ctx.device)
    finder, arr = _as_bytes_like(text)
    hi = arr.size if hi is None else hi
    st = stats if stats is not None else {}
    for k in ("parse_s", "h2d_s", "gpu_s", "text_bytes", "columns", "chunks"):
        st.setdefault(k, 0.0)
    if hi <= lo:
        return torch.zeros((0, 13), dtype=torch.float64, device=dev)
    n_chunks = max(1, -(-(hi - lo) // int(chunk_bytes)))
    cuts = line_cuts(finder, n_chunks, lo, hi)
    ranges = [halo_range(finder, cuts[k], cuts[k + 1]) for k in range(n_chunks) if cuts[k + 1] > cuts[k]]
    cap = max(b - a for a, b, _, _ in ranges) + 64
    # pinned buffers are expensive to create (page-locking): kept on the model between calls
    sets = getattr(model, "_host_sets", None)
    if not sets or sets[0].bases.numel() < cap or (len(sets) < 2 and len(ranges) > 1):
        sets = [_HostSet(cap), _HostSet(cap)] if len(ranges) > 1 else [_HostSet(cap)]
        model._host_sets = sets
    for hs_ in sets:
        hs_.h2d_done = None
    d_seq = torch.from_numpy(np.ascontiguousarray(chr_seq)).to(dev)
    stream = torch.cuda.current_stream(dev)

    def parse(k):
        t0 = time.perf_counter()
        a, b, _, _ = ranges[k]
        out = host.mpileup_parse_range(arr, a, b, out=sets[k % len(sets)].np)
        return out, time.perf_counter() - t0

    rows_all = []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ranges]
    with ThreadPoolExecutor(max_workers=1) as pool:
        fut = pool.submit(parse, 0)
        for k, (a, b, n_lo, n_hi) in enumerate(ranges):
            (pos, col_off, bases), t_parse = fut.result()
            hs = sets[k % len(sets)]
            st["parse_s"] += t_parse; st["text_bytes"] += b - a; st["chunks"] += 1
            M = int(pos.size)
            if M and (int(pos.max()) > chr_seq.size or int(pos.min()) < 1):
                raise ValueError(f"{contig}: position outside the reference sequence")
            ev[k][0].record(stream)
            d_pos = hs.pos[:M].to(dev, non_blocking=True)
            d_off = hs.off[:M + 1].to(dev, non_blocking=True)
            d_bases = hs.bases[:max(int(bases.size), 1)].to(dev, non_blocking=True)
            ev[k][1].record(stream)
            hs.h2d_done = ev[k][1]
            if k + 1 < len(ranges):
                nxt = sets[(k + 1) % len(sets)]
                if nxt.h2d_done is not None:
                    nxt.h2d_done.synchronize()          # the copy engine is done with the set the parser is about to overwrite
                fut = pool.submit(parse, k + 1)
            own = M - n_lo - n_hi
            st["columns"] += own
            if own > 0:
                d_ref = d_seq[d_pos - 1]
                counts, depth, flags = ctx.pileup_encode_columns(d_bases, d_off, d_ref, min_af, min_coverage)
                centers, n_loc = ctx.pileup_select_sites(d_pos, flags)
                if n_loc:
                    centers = centers[(centers >= n_lo) & (centers < M - n_hi)].contiguous()     # halo columns belong to the neighbours
                    n_loc = int(centers.shape[0])
                if n_loc:
                    gt, zy, ga, za, gm, zm = ctx.pileup_forward_windows_calls(counts, centers)
                    cov = counts[centers][:, COV_CHANNELS].to(torch.float64)              # predict.py:63
                    f64 = lambda t: t.to(torch.float64)[:, None]
                    rows_k = torch.cat([f64(d_pos[centers]), f64(ga), f64(za), f64(gm), f64(zm), cov], dim=1)
                    ev[k][2].record(stream)
                    if on_rows is not None:
                        on_rows(rows_k)
                    else:
                        rows_all.append(rows_k)
            if own <= 0 or not n_loc:
                ev[k][2].record(stream)
    torch.cuda.synchronize(dev)
    for e0, e1, e2 in ev:
        st["h2d_s"] += e0.elapsed_time(e1) * 1e-3
        st["gpu_s"] += e1.elapsed_time(e2) * 1e-3
    if on_rows is not None:
        return None
    return torch.cat(rows_all) if rows_all else torch.zeros((0, 13), dtype=torch.float64, device=dev)


def _format_rows(r, contig, chr_seq, batch_size, score_mode):
    """call rows [n, 13] (host float64) -> (VCF text, rows written) of the reference's predict loop over consecutive batches"""
    n = r.shape[0]
    site_pos = r[:, 0].astype(np.int64)
    site_ref = chr_seq[site_pos - 1] & 0xDF                                  # make_predict_data/main.cpp:91 upper-cases
    # the VCF rows depend on the batch boundary: one native call formats every batch (OpenMP over the batches)
    return host.vcf_format_batches(host.ContigTable([contig]), np.zeros(n, np.int32), site_pos, site_ref, r[:, 1].astype(np.uint8),
                                   r[:, 2].astype(np.uint8), r[:, 3].astype(np.float32), r[:, 4].astype(np.float32),
                                   r[:, 5:13].astype(np.float32), batch_size=batch_size, score_mode=score_mode)


def call_contig(model, mpileup_text, contig: str, chr_seq: np.ndarray, min_af=0.12, min_coverage=6,
                batch_size=1000, score_mode=host.SCORE_FLOAT64, chunk_bytes=64 << 20, stats=None):
    """One contig: returns (vcf_rows: bytes, n_sites, n_rows).  model: pileup_model.LSTMNetwork; mpileup_text: bytes, mmap or a
    numpy uint8 array holding the contig's samtools-mpileup text.

    The text is worked off in chunks of whole lines (stream_contig: parse of chunk k + 1 on the host beside the device work of
    chunk k).  Under an initialised torch.distributed process group (one process per GPU, torchrun) the TEXT is statically sharded:
    rank r parses and calls only the lines of its byte range (cut at line boundaries; 16 lines of halo re-parsed, not exchanged), and
    the per-site calls are gathered to rank 0 in rank = position order, where the rows are formatted exactly as a single process
    would format them (the reference's batches of `batch_size` sites run over the whole site list).  Ranks other than 0 return
    (b"", n_sites_total, 0)."""
    import time
    import torch
    import torch.distributed as tdist
    from .dist import gather_varlen
    ctx = model.ctx
    finder, arr = _as_bytes_like(mpileup_text)
    sharded = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    rank, world = (tdist.get_rank(), tdist.get_world_size()) if sharded else (0, 1)
    cuts = line_cuts(finder, world, 0, arr.size)
    rows = stream_contig(model, mpileup_text, contig, chr_seq, cuts[rank], cuts[rank + 1], chunk_bytes, min_af, min_coverage, stats)
    if sharded:
        backend_dev = torch.device("cuda", ctx.device) if tdist.get_backend() == "nccl" else "cpu"
        rows = gather_varlen(rows.to(backend_dev))
        if rank != 0:
            n_tot = torch.zeros(1, dtype=torch.int64, device=backend_dev)
            tdist.broadcast(n_tot, src=0)
            return b"", int(n_tot.item()), 0
        tdist.broadcast(torch.tensor([rows.shape[0]], dtype=torch.int64, device=backend_dev), src=0)
    # (formatting the rows of finished chunks on a worker thread while later chunks are parsed was measured: parse and formatter are
    # both OpenMP-parallel host work on the same cores - 24 + 34 ms per 6 M-column contig on 16 cores - and run slower side by side
    # than one after the other: 162 against 75 ms per contig)
    n_sites = int(rows.shape[0])
    if n_sites == 0:
        return b"", 0, 0
    t0 = time.perf_counter()
    text, n_rows = _format_rows(rows.cpu().numpy(), contig, chr_seq, batch_size, score_mode)
    if stats is not None:
        stats["vcf_s"] = stats.get("vcf_s", 0.0) + time.perf_counter() - t0
        stats["sites"] = stats.get("sites", 0) + n_sites
        stats["vcf_rows"] = stats.get("vcf_rows", 0) + n_rows
    return text, n_sites, n_rows


def call_variants(model, contigs, fasta_path, fai_text, output_file, **kw):
    """contigs: iterable of (name, path to <name>.mpileup).  Writes pileup.vcf (rank 0 only under torch.distributed: see
    call_contig); returns total rows."""
    import torch.distributed as tdist
    root = not (tdist.is_available() and tdist.is_initialized()) or tdist.get_rank() == 0
    total = 0
    f = open(output_file, "wb") if root else None
    try:
        if root:
            f.write(host.vcf_header(fai_text).encode())
        for name, path in contigs:
            seq = host.fasta_load_contig(fasta_path, name)
            with open(path, "rb") as g:
                size = os.fstat(g.fileno()).st_size
                text = mmap.mmap(g.fileno(), 0, access=mmap.ACCESS_READ) if size else b""    # parsed in place, never copied
                try:
                    rows_text, _, rows = call_contig(model, text, name, seq, **kw)
                finally:
                    if size:
                        try:
                            text.close()
                        except BufferError:          # (an exception on its way up still holds views of the mapping)
                            pass
            if root:
                f.write(rows_text)
            total += rows
    finally:
        if f:
            f.close()
    return total
