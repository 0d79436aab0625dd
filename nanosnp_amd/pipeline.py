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


def ramp_cuts(text, lo, hi, chunk_bytes, first=None, growth=1.5):
    """offsets cutting text[lo:hi] into chunks of whole lines whose sizes grow from `first` bytes (default chunk_bytes / 4, at least 1 MB) by
    `growth` per chunk up to chunk_bytes: the pipeline's fill - nothing computes before the first chunk has been staged, copied, tokenised
    and encoded, and the forward of chunk 0 is issued behind the tokeniser of chunk 2 - shrinks with the first chunks (with equal 64 MB
    chunks the first forward of a 6 M-column contig started 4.0 ms into an 18.4 ms pass); growth 1.5 keeps the copy of the next, larger
    chunk shorter than the compute of the current one (H2D 0.018 ms / MB against 0.029 ms / MB of device work).  Where to start is a
    trade against the per-chunk issue cost (~0.3 ms of host time, ~30 launches): measured per contig of a run of contigs (the previous
    contig's last forwards cover most of the fill there) 18.9 / 17.6 / 17.3 / 17.6 / 18.0 ms starting at 4 / 8 / 16 / 32 / 64 MB."""
    chunk_bytes = max(1, int(chunk_bytes))
    if not first and os.environ.get("NSNP_RAMP_FIRST_MB"):           # (A/B measurements)
        first = int(float(os.environ["NSNP_RAMP_FIRST_MB"]) * (1 << 20))
    first = int(first) if first else max(min(chunk_bytes, 1 << 20), chunk_bytes // 4)
    cuts, size, target = [lo], float(min(first, chunk_bytes)), float(lo)
    while cuts[-1] < hi:
        target += size                               # (targets accumulate: the chunks average `size` bytes however long the lines are)
        g = max(int(target), cuts[-1] + 1)
        if g >= hi or hi - g < size / 2:             # (what is left is smaller than half a chunk: it joins this one)
            cuts.append(hi)
            break
        nl = text.find(b"\n", g - 1, hi)
        cuts.append(hi if nl < 0 else nl + 1)
        size = min(size * growth, float(chunk_bytes))
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


def _cols_for(cap_bytes):
    """columns budgeted for a chunk of cap_bytes of text: one per 24 bytes (a samtools line at 30x is ~90 bytes; a valid line cannot
    be shorter than 10).  A chunk with more lines than that grows its buffer sets once, when the parser reports it (NSNP_HOST_ERANGE)."""
    return cap_bytes // 24 + 1024


class _HostSet:
    """pinned host buffers of one text chunk in flight (the parser writes straight into them, the copy engine reads them)"""
    def __init__(self, cap_bytes, cap_cols=None):
        import torch
        cap_cols = int(cap_cols or _cols_for(cap_bytes))
        self.pos = torch.empty(cap_cols, dtype=torch.int64, pin_memory=True)
        self.off = torch.empty(cap_cols + 1, dtype=torch.int64, pin_memory=True)
        self.bases = torch.empty(cap_bytes, dtype=torch.uint8, pin_memory=True)
        self.np = (self.pos.numpy(), self.off.numpy(), self.bases.numpy())
        self.h2d_done = None


class _DevSet:
    """device buffers of one text chunk in flight (filled by the copy stream, read by the chunk's encode / select / call rows)"""
    def __init__(self, cap_bytes, dev, cap_cols=None):
        import torch
        cap_cols = int(cap_cols or _cols_for(cap_bytes))
        self.pos = torch.empty(cap_cols, dtype=torch.int64, device=dev)
        self.off = torch.empty(cap_cols + 1, dtype=torch.int64, device=dev)
        self.bases = torch.empty(cap_bytes, dtype=torch.uint8, device=dev)
        self.free = None                               # event on the compute stream: the last kernels reading this set are done


def tokenise_mode(tokenise=None):
    """where the mpileup text is cut into columns: "device" (default: the raw text crosses PCIe and nsnp_mpileup_tokenise cuts it in HBM; the
    host only copies the text into pinned memory) or "host" (nsnp_mpileup_parse_into on the host cores; NSNP_TOKENISE=host)"""
    t = tokenise or os.environ.get("NSNP_TOKENISE", "device")
    if t not in ("device", "host"):
        raise ValueError(f"tokenise: 'device' or 'host', not {t!r}")
    return t


def stream_contig(model, text, contig, chr_seq, lo=0, hi=None, chunk_bytes=64 << 20, min_af=0.12, min_coverage=6, stats=None, on_rows=None,
                  tokenise=None):
    with host.gc_paused():
        if tokenise_mode(tokenise) == "device":
            return _stream_contig_dev(model, text, contig, chr_seq, lo, hi, chunk_bytes, min_af, min_coverage, stats, on_rows)
        return _stream_contig(model, text, contig, chr_seq, lo, hi, chunk_bytes, min_af, min_coverage, stats, on_rows)


class _TextSet:
    """one chunk of raw mpileup text in flight: pinned on the host (filled by the staging thread, read by the copy engine) or on the device
    (filled by the copy stream, read by the tokeniser)"""
    def __init__(self, cap_bytes, dev=None):
        import torch
        self.buf = torch.empty(cap_bytes, dtype=torch.uint8, **(dict(pin_memory=True) if dev is None else dict(device=dev)))
        self.np = self.buf.numpy() if dev is None else None
        self.h2d_done = None                           # host set: the copy engine has read it
        self.free = None                               # device set: the tokeniser has read it


class _ColSet:
    """the columns of one chunk on the device, as nsnp_mpileup_tokenise writes them and the encode reads them.  Sized for any text of
    cap_bytes (a line is at least 10 bytes, its column 5 shorter than the line): no growth path, nothing to re-run."""
    def __init__(self, cap_bytes, dev):
        import torch
        cc = cap_bytes // 10 + 2
        self.pos = torch.empty(cc, dtype=torch.int64, device=dev)
        self.off = torch.empty(cc + 1, dtype=torch.int64, device=dev)
        self.ref = torch.empty(cc, dtype=torch.uint8, device=dev)
        self.bases = torch.empty(cap_bytes, dtype=torch.uint8, device=dev)


def _stream_contig_dev(model, text, contig, chr_seq, lo, hi, chunk_bytes, min_af, min_coverage, stats, on_rows, defer=False):
    """stream_contig with the text cut into columns ON THE DEVICE (nsnp_mpileup_tokenise).  The host touches every byte of the text once - a
    multi-threaded copy of the chunk (whole lines, 16 lines of halo either side, found by a few find / rfind calls) from the page cache
    into pinned memory - and the chunks are worked off four things at a time:

        worker thread   copies chunks k + 2 and k + 3 into two of four pinned text buffers (libnanosnp_host.so: nsnp_stage_values)
        copy stream     sends the text of chunk k + 1 (three device text buffers: a copy waits for the tokeniser three chunks back)
        compute stream  tokenise chunk k -> positions, reference bytes, column-5 strings (three column sets on the device)
                        encode + select chunk k - 1 (its line count came back through pinned memory while chunk k was being issued)
                        PileupModel forward + argmax / max of chunk k - 2 (its site count likewise)

    so this thread never waits for work it has just issued, and the device never waits for this thread.  Same rows as the host-parsed
    path (tests/test_gpu_predict.py); text the reference's reader aborts on is refused with the same errors.
    defer=True (call_contigs): returns (rows, done, finalize) as soon as the last chunk is ISSUED - `done` is an event behind the last kernel,
    finalize() waits for it and adds the per-stage times to stats - so that the next contig's text is staged, copied and tokenised while
    this one's last forward (1.4 ms of a 6 M-column contig's 15) still runs; the buffer sets carry their events from call to call."""
    import time
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    import torch
    ctx = model.ctx
    dev = torch.device("cuda", ctx.device)
    finder, arr = _as_bytes_like(text)
    hi = arr.size if hi is None else hi
    st = stats if stats is not None else {}
    for k in ("parse_s", "h2d_s", "gpu_s", "tok_s", "text_bytes", "columns", "chunks", "setup_s", "wait_parse_s", "issue_s", "wait_counts_s", "drain_s"):
        st.setdefault(k, 0.0)
    st["tokenise"] = "device"
    t_enter = time.perf_counter()
    if hi <= lo:
        empty = torch.zeros((0, 13), dtype=torch.float64, device=dev)
        return (empty, None, lambda: None) if defer else empty
    cuts = ramp_cuts(finder, lo, hi, int(chunk_bytes))
    ranges = [halo_range(finder, cuts[k], cuts[k + 1]) for k in range(len(cuts) - 1) if cuts[k + 1] > cuts[k]]
    cap = max(b - a for a, b, _, _ in ranges) + 64
    # FOUR pinned text buffers (the staging thread works up to three chunks ahead of the tokeniser), THREE device text buffers (the copy of
    # chunk k + 1 is issued before the tokeniser of chunk k: the compute stream never waits for a copy that was only just issued - with
    # two buffers and the copy issued in the tokeniser's own iteration a 24 M-column contig ran at 65.7 ms against 48.7 ms of device time)
    hsets = getattr(model, "_text_host_sets", None)
    n_hsets, n_tsets, n_sets = min(4, len(ranges)), min(3, len(ranges)), min(3, len(ranges))
    if not hsets or min(s_.buf.numel() for s_ in hsets) < cap or len(hsets) < n_hsets:
        hsets = model._text_host_sets = [_TextSet(cap) for _ in range(n_hsets)]
        model._text_dev_sets = None
    tsets = getattr(model, "_text_dev_sets", None)
    if not tsets or tsets[0].buf.device != dev or min(t_.buf.numel() for t_ in tsets) < cap or len(tsets) < n_tsets:
        tsets = model._text_dev_sets = [_TextSet(cap, dev) for _ in range(n_tsets)]
        model._col_dev_sets = [_ColSet(cap, dev) for _ in range(n_sets)]
        model._copy_stream = getattr(model, "_copy_stream", None) or host.copy_stream(dev)
    csets, copy_stream = model._col_dev_sets, model._copy_stream
    if len(getattr(model, "_meta_pin", ())) < len(ranges):
        model._meta_pin = torch.zeros((len(ranges), 4), dtype=torch.int64, pin_memory=True)
    if len(getattr(model, "_tok_meta_pin", ())) < len(ranges):
        model._tok_meta_pin = torch.zeros((len(ranges), 4), dtype=torch.int64, pin_memory=True)
    meta_pin, tok_pin = model._meta_pin, model._tok_meta_pin
    main = torch.cuda.current_stream(dev)
    # (the sets keep their events from the previous call: a pinned buffer is rewritten only behind the copy that read it, a device text buffer
    # behind the tokeniser that read it - whichever call issued those; the column sets are written and read on the compute stream alone)
    # the reference sequence: through a pinned buffer on the copy stream (a copy from pageable memory on the compute stream would wait for
    # everything queued there - the previous contig's last forward - and stall this thread for as long)
    n_seq = int(chr_seq.size)
    sp = getattr(model, "_seq_pin", None)
    if sp is None or sp.numel() < n_seq:
        sp = model._seq_pin = torch.empty(max(n_seq + n_seq // 8, 1 << 20), dtype=torch.uint8, pin_memory=True)
        model._seq_pin_free = None
    if getattr(model, "_seq_pin_free", None) is not None:
        model._seq_pin_free.synchronize()              # (the previous contig's sequence has left the pinned buffer)
    sp.numpy()[:n_seq] = np.ascontiguousarray(chr_seq)
    with torch.cuda.stream(copy_stream):               # (allocated as the copy stream's memory: a block the compute stream has just freed may
        d_seq = torch.empty(max(n_seq, 1), dtype=torch.uint8, device=dev)[:n_seq]      # still be read by work queued there)
        d_seq.copy_(sp[:n_seq], non_blocking=True)
        model._seq_pin_free = torch.cuda.Event(); model._seq_pin_free.record(copy_stream)
    d_seq.record_stream(main)
    main.wait_event(model._seq_pin_free)
    cov_idx = getattr(model, "_cov_idx", None)
    if cov_idx is None or cov_idx.device != dev:
        cov_idx = model._cov_idx = torch.tensor(list(COV_CHANNELS), dtype=torch.int64, device=dev)
    if getattr(model, "_stream_main_id", None) != (main.device, main.stream_id):
        copy_stream.wait_stream(main)                  # (another compute stream than last time: its queued work may still read the device sets)
        model._stream_main_id = (main.device, main.stream_id)

    trace = st.get("trace")

    def stage(k):
        t0 = time.perf_counter()
        a, b, _, _ = ranges[k]
        host.stage_values(hsets[k % len(hsets)].np, b - a, src=arr, src_off=a, src_dtype=np.uint8)
        t1 = time.perf_counter()
        if trace is not None:
            trace.append(("stage", k, t0, t1))
        return t1 - t0

    rows_all = []
    ev0 = torch.cuda.Event(enable_timing=True)
    if trace is not None:
        torch.cuda.synchronize(dev); ev0.record(main); torch.cuda.synchronize(dev)
    t_ev0 = time.perf_counter()
    st["setup_s"] += time.perf_counter() - t_enter
    tev = lambda: torch.cuda.Event(enable_timing=True)
    ev = [dict(h0=tev(), h1=tev(), t0=tev(), t1=tev(), a0=tev(), a1=tev(), b0=tev(), b1=tev()) for _ in ranges]

    def encode_of(tk):
        """the second third of a chunk: its line count is on the host by now"""
        k, n_lo, n_hi, cs, tok_done = tk
        t_w = time.perf_counter()
        tok_done.synchronize()
        st["wait_counts_s"] += time.perf_counter() - t_w
        M, nb, status, _ = tok_pin[k].tolist()
        if status & ctx.TOK_EFORMAT:
            raise host.HostError(f"{contig}: malformed input (a line with fewer than five tab-separated fields)")
        if status & ctx.TOK_BLANK:
            raise host.HostError(f"{contig}: mpileup text holds empty line(s): malformed input (every line must be one pileup column)")
        if status & ctx.TOK_EPOS:
            raise ValueError(f"{contig}: position outside the reference sequence")
        if status:
            raise host.HostError(f"{contig}: tokeniser status {status}")
        own = M - n_lo - n_hi
        st["columns"] += own
        ev[k]["a0"].record(main)
        job = None
        if own > 0:
            d_pos = cs.pos[:M]
            counts, depth, flags = ctx.pileup_encode_columns(cs.bases[:max(nb, 1)], cs.off[:M + 1], cs.ref[:M], min_af, min_coverage)
            # selection + the run of the chunk's own sites in the list, written into pinned memory by the last of its four launches
            center = ctx.pileup_select_sites_range(d_pos, flags, n_lo, M - n_hi, meta_pin[k], stream=main)
            sel_done = torch.cuda.Event(); sel_done.record(main)
            job = (k, M, cs, counts, center, sel_done)
        ev[k]["a1"].record(main)
        return job

    def calls_of(job):
        """the last third of a chunk: its site count is on the host by now"""
        k, M, cs, counts, center, sel_done = job
        t_w = time.perf_counter()
        sel_done.synchronize()
        st["wait_counts_s"] += time.perf_counter() - t_w
        _, c_lo, c_hi, _ = meta_pin[k].tolist()
        ev[k]["b0"].record(main)
        if c_hi > c_lo:
            centers = center[c_lo:c_hi]
            gt, zy, ga, za, gm, zm = ctx.pileup_forward_windows_calls(counts, centers)
            rows_k = ctx.pileup_call_rows(counts, centers, cs.pos, ga, za, gm, zm)                  # predict.py:52-65, one launch
            if on_rows is not None:
                on_rows(rows_k)
            else:
                rows_all.append(rows_k)
        ev[k]["b1"].record(main)

    pending, job = deque(), None
    with ThreadPoolExecutor(max_workers=1) as pool:
        futs = [pool.submit(stage, j) for j in range(min(3, len(ranges)))]

        def send(j):
            """the text of chunk j on its way: copy stream, behind the tokeniser of chunk j - 3 (the last reader of its device text buffer)"""
            t_w = time.perf_counter()
            t_stage = futs[j].result()
            t_s = time.perf_counter()
            st["wait_parse_s"] += t_s - t_w
            if trace is not None:
                trace.append(("main: wait stage", j, t_w, t_s))
            a_, b_, _, _ = ranges[j]
            n_ = b_ - a_
            hs, ts = hsets[j % len(hsets)], tsets[j % len(tsets)]
            st["parse_s"] += t_stage; st["text_bytes"] += n_; st["chunks"] += 1
            if ts.free is not None:
                copy_stream.wait_event(ts.free)
            with torch.cuda.stream(copy_stream):
                ev[j]["h0"].record(copy_stream)
                ts.buf[:n_].copy_(hs.buf[:n_], non_blocking=True)
                ev[j]["h1"].record(copy_stream)
            hs.h2d_done = ev[j]["h1"]
            if j + 3 < len(ranges):
                nxt = hsets[(j + 3) % len(hsets)]
                if nxt.h2d_done is not None:
                    nxt.h2d_done.synchronize()          # the copy engine is done with the buffer the staging thread is about to overwrite (chunk j - 1)
                futs.append(pool.submit(stage, j + 3))

        send(0)
        for k, (a, b, n_lo, n_hi) in enumerate(ranges):
            t_i = time.perf_counter()
            if k + 1 < len(ranges):
                send(k + 1)                             # one chunk ahead of the tokeniser
            ts, cs = tsets[k % len(tsets)], csets[k % len(csets)]
            n = b - a
            # ---- first third of chunk k on the compute stream: the tokeniser; lines / bytes / status land in pinned memory ----
            main.wait_event(ev[k]["h1"])
            ev[k]["t0"].record(main)
            ctx.mpileup_tokenise_into(ts.buf[:n], d_seq, cs.pos, cs.off, cs.bases, cs.ref, tok_pin[k], stream=main)
            ev[k]["t1"].record(main)
            ts.free = ev[k]["t1"]
            tok_done = torch.cuda.Event(); tok_done.record(main)
            pending.append((k, n_lo, n_hi, cs, tok_done))
            # ---- second third of chunk k - 1, last third of chunk k - 2 ----
            nxt_job = encode_of(pending.popleft()) if len(pending) > 1 else None
            if job is not None:
                calls_of(job)
            job = nxt_job
            st["issue_s"] += time.perf_counter() - t_i
            if trace is not None:
                trace.append(("main: issue", k, t_i, time.perf_counter()))
        t_i = time.perf_counter()
        while pending or job is not None:
            nxt_job = encode_of(pending.popleft()) if pending else None
            if job is not None:
                calls_of(job)
            job = nxt_job
        st["issue_s"] += time.perf_counter() - t_i
    rows = None if on_rows is not None else (torch.cat(rows_all) if rows_all else torch.zeros((0, 13), dtype=torch.float64, device=dev))
    done = torch.cuda.Event(); done.record(main)

    def finalize():
        t_d = time.perf_counter()
        done.synchronize()
        copy_stream.synchronize()
        st["drain_s"] += time.perf_counter() - t_d
        if trace is not None:
            for k, e in enumerate(ev):
                for what, x0, x1 in (("h2d", "h0", "h1"), ("tokenise", "t0", "t1"), ("encode+select", "a0", "a1"), ("forward+rows", "b0", "b1")):
                    try:
                        trace.append((what, k, t_ev0 + ev0.elapsed_time(e[x0]) * 1e-3, t_ev0 + ev0.elapsed_time(e[x1]) * 1e-3))
                    except (RuntimeError, ValueError):
                        pass
        for e in ev:
            st["h2d_s"] += e["h0"].elapsed_time(e["h1"]) * 1e-3
            tk = e["t0"].elapsed_time(e["t1"]) * 1e-3
            st["tok_s"] += tk; st["gpu_s"] += tk
            for x0, x1 in (("a0", "a1"), ("b0", "b1")):
                try:
                    st["gpu_s"] += e[x0].elapsed_time(e[x1]) * 1e-3
                except (RuntimeError, ValueError):
                    pass                               # (a chunk without columns of its own never recorded its last third)

    if defer:
        return rows, done, finalize
    finalize()
    return rows


def _stream_contig(model, text, contig, chr_seq, lo, hi, chunk_bytes, min_af, min_coverage, stats, on_rows):
    """The device part of stages s1 + s2 over the lines of text[lo:hi], chunk by chunk, three things at a time:

        worker thread   parses chunks k + 1 and k + 2 (libnanosnp_host.so, OpenMP, straight into one of three pinned buffer sets)
        copy stream     sends chunk k to the device (three device buffer sets: chunk k + 3 waits for the last readers of chunk k)
        compute stream  column encode -> site selection of chunk k, then PileupModel forward + argmax / max of chunk k - 1

    The number of selected sites is data: it comes back through a pinned buffer and is read ONE CHUNK LATER (the forward of chunk
    k - 1 is issued behind the encode of chunk k), so this thread never waits for work it has just issued and the device never
    waits for this thread.  Every chunk is parsed with 16 lines of halo on either side (re-parsed, not exchanged) and calls the
    sites centred in its own lines, so the result does not depend on where the cuts fall.
    Returns the call rows [n, 13] float64 (position, argmax / max of both heads, the eight coverage channels: all exact in float64)
    as a device tensor in position order - or, with on_rows, hands every chunk's rows to that callback as soon as they are issued
    and returns None.  stats (a dict) receives per-stage busy times (and, with a list under stats["trace"], the spans of every chunk's
    parse / copy / encode / forward on one clock: tools/probes/e2e_timeline.py).  The pinned and device buffer sets live on `model` between
    calls: one call at a time per model (use one model per thread)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    import torch
    ctx = model.ctx
    dev = torch.device("cuda", ctx.device)
    finder, arr = _as_bytes_like(text)
    hi = arr.size if hi is None else hi
    st = stats if stats is not None else {}
    for k in ("parse_s", "h2d_s", "gpu_s", "text_bytes", "columns", "chunks", "setup_s", "wait_parse_s", "issue_s", "wait_counts_s", "drain_s"):
        st.setdefault(k, 0.0)
    t_enter = time.perf_counter()
    if hi <= lo:
        return torch.zeros((0, 13), dtype=torch.float64, device=dev)
    n_chunks = max(1, -(-(hi - lo) // int(chunk_bytes)))
    cuts = line_cuts(finder, n_chunks, lo, hi)
    if n_chunks >= 3:
        # the pipeline's fill: nothing computes while the first chunk is parsed and copied - it is half a chunk (the other half goes
        # to a chunk of its own behind it)
        half = finder.find(b"\n", lo + (cuts[1] - lo) // 2, cuts[1])
        if half >= 0 and lo < half + 1 < cuts[1]:
            cuts = [lo, half + 1] + cuts[1:]
    ranges = [halo_range(finder, cuts[k], cuts[k + 1]) for k in range(len(cuts) - 1) if cuts[k + 1] > cuts[k]]
    cap = max(b - a for a, b, _, _ in ranges) + 64
    # pinned buffers are expensive to create (page-locking): kept on the model between calls, with their device twins
    sets = getattr(model, "_host_sets", None)
    # THREE host sets: the parser works two chunks ahead of the copy engine (it never waits for this thread between two chunks)
    n_sets = min(3, len(ranges))
    if not sets or min(s_.bases.numel() for s_ in sets) < cap or len(sets) < n_sets:
        sets = [_HostSet(cap) for _ in range(n_sets)]
        model._host_sets = sets
        model._dev_sets = None
    dsets = getattr(model, "_dev_sets", None)
    # THREE device sets as well: with two, the copy of chunk k + 2 waits for the forward of chunk k (the last reader of its set) and the
    # encode of chunk k + 2 - in front of the forward of chunk k + 1 in stream order - waits for that copy: copies and forwards
    # alternated (tools/probes/e2e_timeline.py: 2.4 ms per chunk = copy 0.8 + encode 0.15 + forward 1.45); with three the copy runs beside
    # the forward of chunk k + 1
    if not dsets or len(dsets) < min(3, len(ranges)) or dsets[0].bases.device != dev or min(d_.bases.numel() for d_ in dsets) < cap:
        dsets = [_DevSet(cap, dev) for _ in range(min(3, len(ranges)))]
        model._dev_sets = dsets
        model._copy_stream = host.copy_stream(dev)
    if len(getattr(model, "_meta_pin", ())) < len(ranges):
        model._meta_pin = torch.zeros((len(ranges), 4), dtype=torch.int64, pin_memory=True)
    meta_pin = model._meta_pin
    copy_stream = model._copy_stream
    main = torch.cuda.current_stream(dev)
    for hs_ in sets:
        hs_.h2d_done = None
    for ds_ in dsets:
        ds_.free = None
    d_seq = torch.from_numpy(np.ascontiguousarray(chr_seq)).to(dev)
    seq_len = int(chr_seq.size)
    cov_idx = torch.tensor(list(COV_CHANNELS), dtype=torch.int64, device=dev)
    copy_stream.wait_stream(main)                      # (whatever the caller queued before us may still read the device sets)

    def parse(k):
        t0 = time.perf_counter()
        a, b, _, _ = ranges[k]
        try:
            out = host.mpileup_parse_range(arr, a, b, out=sets[k % len(sets)].np, strict_lines=True)
        except host.HostRangeError as e:
            # more (shorter) lines than budgeted: this set grows - the parser owns it right now (the copy engine was waited for before
            # this parse was submitted) - and the chunk is parsed again; the device twin grows on the main thread before the copy
            sets[k % len(sets)] = _HostSet(max(cap, e.n_bytes), e.n_cols + e.n_cols // 4 + 1024)
            out = host.mpileup_parse_range(arr, a, b, out=sets[k % len(sets)].np, strict_lines=True)
        pos = out[0]
        bad = bool(pos.size) and (int(pos.max()) > seq_len or int(pos.min()) < 1)
        t1 = time.perf_counter()
        if trace is not None:
            trace.append(("parse", k, t0, t1))
        return out, bad, t1 - t0

    rows_all = []
    trace = st.get("trace")                            # a list: receives (what, chunk, start, end) in host seconds; device spans are mapped onto the host clock
    ev0 = torch.cuda.Event(enable_timing=True)
    if trace is not None:
        torch.cuda.synchronize(dev); ev0.record(main); torch.cuda.synchronize(dev)
    t_ev0 = time.perf_counter()
    st["setup_s"] += time.perf_counter() - t_enter
    tev = lambda: torch.cuda.Event(enable_timing=True)
    ev = [dict(h0=tev(), h1=tev(), a0=tev(), a1=tev(), b0=tev(), b1=tev()) for _ in ranges]

    def calls_of(job):
        """the second half of a chunk: its site count is on the host by now"""
        k, M, n_lo, n_hi, ds, counts, center, sel_done = job
        t_w = time.perf_counter()
        sel_done.synchronize()
        st["wait_counts_s"] += time.perf_counter() - t_w
        _, c_lo, c_hi, _ = meta_pin[k].tolist()
        ev[k]["b0"].record(main)
        if c_hi > c_lo:
            centers = center[c_lo:c_hi]                                                   # ascending: the chunk's own sites are one run
            gt, zy, ga, za, gm, zm = ctx.pileup_forward_windows_calls(counts, centers)
            # (index tensors made once, on the device: indexing with a Python list copies it from pageable memory every time, and
            # that copy waits for everything queued on the device - the forward just issued included)
            cov = counts.index_select(0, centers).index_select(1, cov_idx).to(torch.float64)      # predict.py:63
            f64 = lambda t: t.to(torch.float64)[:, None]
            rows_k = torch.cat([f64(ds.pos[:M].index_select(0, centers)), f64(ga), f64(za), f64(gm), f64(zm), cov], dim=1)
            if on_rows is not None:
                on_rows(rows_k)
            else:
                rows_all.append(rows_k)
        ev[k]["b1"].record(main)
        ds.free = torch.cuda.Event(); ds.free.record(main)

    job = None
    with ThreadPoolExecutor(max_workers=1) as pool:
        futs = [pool.submit(parse, j) for j in range(min(2, len(ranges)))]       # (one worker: the parses run one after the other)
        for k, (a, b, n_lo, n_hi) in enumerate(ranges):
            if job is not None and not futs[k].done() and os.environ.get("NSNP_PIPE_NO_EARLY_CALLS") != "1":      # (the variable: A/B measurements)
                # the parser is still busy with chunk k: the calls of chunk k - 1 go out now instead of behind the encode of chunk k
                # (at the front of the text this starts the first forward a parse earlier)
                t_i0 = time.perf_counter()
                calls_of(job); job = None
                st["issue_s"] += time.perf_counter() - t_i0
            t_w = time.perf_counter()
            (pos, col_off, bases), bad, t_parse = futs[k].result()
            t_i = time.perf_counter()
            st["wait_parse_s"] += t_i - t_w
            if trace is not None:
                trace.append(("main: wait parse", k, t_w, t_i))
            if bad:
                raise ValueError(f"{contig}: position outside the reference sequence")
            hs, ds = sets[k % len(sets)], dsets[k % len(dsets)]
            st["parse_s"] += t_parse; st["text_bytes"] += b - a; st["chunks"] += 1
            M, nb = int(pos.size), int(bases.size)
            if M + 1 > ds.off.numel() or nb > ds.bases.numel():
                torch.cuda.synchronize(dev)                       # (rare: every reader of the old set is done before it is dropped)
                ds = dsets[k % len(dsets)] = _DevSet(max(nb, ds.bases.numel()), dev, max(M + M // 4 + 1024, ds.pos.numel()))
            # ---- H2D on the copy stream, behind the last readers of this device set (chunk k - 2) ----
            if ds.free is not None:
                copy_stream.wait_event(ds.free)
            with torch.cuda.stream(copy_stream):
                ev[k]["h0"].record(copy_stream)
                ds.pos[:M].copy_(hs.pos[:M], non_blocking=True)
                ds.off[:M + 1].copy_(hs.off[:M + 1], non_blocking=True)
                ds.bases[:max(nb, 1)].copy_(hs.bases[:max(nb, 1)], non_blocking=True)
                ev[k]["h1"].record(copy_stream)
            hs.h2d_done = ev[k]["h1"]
            if k + 2 < len(ranges):
                nxt = sets[(k + 2) % len(sets)]
                if nxt.h2d_done is not None:
                    nxt.h2d_done.synchronize()          # the copy engine is done with the set the parser is about to overwrite (chunk k - 1)
                futs.append(pool.submit(parse, k + 2))
            # ---- first half of chunk k on the compute stream: encode + select, the counts on their way to the host ----
            main.wait_event(ev[k]["h1"])
            ev[k]["a0"].record(main)
            own = M - n_lo - n_hi
            st["columns"] += own
            nxt_job = None
            if own > 0:
                d_pos = ds.pos[:M]
                d_ref = d_seq[d_pos - 1]
                counts, depth, flags = ctx.pileup_encode_columns(ds.bases[:max(nb, 1)], ds.off[:M + 1], d_ref, min_af, min_coverage)
                center, n_sel = ctx.pileup_select_sites_async(d_pos, flags)
                # halo columns belong to the neighbours: the chunk's own centres are [c_lo, c_hi) of the ascending list
                meta = torch.stack([n_sel[0], (center < n_lo).sum(), (center < M - n_hi).sum(), n_sel[0]])
                meta_pin[k].copy_(meta, non_blocking=True)
                sel_done = torch.cuda.Event(); sel_done.record(main)
                nxt_job = (k, M, n_lo, n_hi, ds, counts, center, sel_done)
            else:
                ds.free = torch.cuda.Event(); ds.free.record(main)
            ev[k]["a1"].record(main)
            # ---- second half of chunk k - 1 ----
            if job is not None:
                calls_of(job)
            job = nxt_job
            st["issue_s"] += time.perf_counter() - t_i
            if trace is not None:
                trace.append(("main: issue", k, t_i, time.perf_counter()))
        if job is not None:
            t_i = time.perf_counter()
            calls_of(job)
            st["issue_s"] += time.perf_counter() - t_i
    t_d = time.perf_counter()
    torch.cuda.synchronize(dev)
    st["drain_s"] += time.perf_counter() - t_d
    if trace is not None:
        for k, e in enumerate(ev):
            for what, x0, x1 in (("h2d", "h0", "h1"), ("encode+select", "a0", "a1"), ("forward+rows", "b0", "b1")):
                try:
                    trace.append((what, k, t_ev0 + ev0.elapsed_time(e[x0]) * 1e-3, t_ev0 + ev0.elapsed_time(e[x1]) * 1e-3))
                except RuntimeError:
                    pass
    for e in ev:
        st["h2d_s"] += e["h0"].elapsed_time(e["h1"]) * 1e-3
        st["gpu_s"] += e["a0"].elapsed_time(e["a1"]) * 1e-3
        if e["b0"].query() and e["b1"].query():
            try:
                st["gpu_s"] += e["b0"].elapsed_time(e["b1"]) * 1e-3
            except RuntimeError:
                pass                                   # (a chunk without columns of its own never recorded its second half)
    if on_rows is not None:
        return None
    return torch.cat(rows_all) if rows_all else torch.zeros((0, 13), dtype=torch.float64, device=dev)


stream_contig.__doc__ = _stream_contig.__doc__


def _format_rows(r, contig, chr_seq, batch_size, score_mode, as_view=False, shard_dev=None, nthreads=None, ctx=None):
    """call rows [n, 13] float64 (a device tensor, a host tensor or a numpy array) -> (VCF text, rows written) of the reference's
    predict loop over consecutive batches.  shard_dev (the device the process group's collectives take their tensors on): the rows are
    one rank's share of the contig -> (this rank's text, its rows, sites of all ranks); every rank must call.  Device rows are cut into their typed columns ON the device (six small kernels, 41 B per
    site over the bus instead of 104 B and nine numpy passes)."""
    import torch
    if isinstance(r, torch.Tensor) and r.is_cuda and ctx is not None:
        # one kernel cuts the rows into their typed columns and writes them (41 B per site) straight into pinned host memory: no
        # device-to-host copy, so a text run keeps the copy engines to the text's way in (a D2H copy beside the H2D stream makes the HIP
        # runtime open further SDMA engines - 6-8 ms each - and was seen to leave later H2D traffic of the process on a slower one)
        n = int(r.shape[0])
        hb = getattr(ctx, "_rows_host", None)
        if hb is None or hb[0].numel() < n:
            cap = max(n + n // 4, 65536)
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
            hb = ctx._rows_host = (mk(cap, torch.int64), mk(cap, torch.uint8), mk(cap, torch.uint8), mk(cap, torch.float32), mk(cap, torch.float32),
                                   mk((cap, 8), torch.float32))
        s_ = torch.cuda.current_stream(r.device)
        ctx.pileup_rows_unpack(r.contiguous(), hb, stream=s_)
        s_.synchronize()
        site_pos, ga, za, gm, zm = (t[:n].numpy() for t in hb[:5])
        cov = hb[5][:n].numpy()
    elif isinstance(r, torch.Tensor):
        site_pos = r[:, 0].to(torch.int64).cpu().numpy()
        ga, za = r[:, 1].to(torch.uint8).cpu().numpy(), r[:, 2].to(torch.uint8).cpu().numpy()
        gm, zm = r[:, 3].to(torch.float32).cpu().numpy(), r[:, 4].to(torch.float32).cpu().numpy()
        cov = r[:, 5:13].to(torch.float32).contiguous().cpu().numpy()
    else:
        site_pos = r[:, 0].astype(np.int64)
        ga, za, gm, zm = r[:, 1].astype(np.uint8), r[:, 2].astype(np.uint8), r[:, 3].astype(np.float32), r[:, 4].astype(np.float32)
        cov = r[:, 5:13].astype(np.float32)
    n = site_pos.shape[0]
    site_ref = chr_seq[site_pos - 1] & 0xDF                                  # make_predict_data/main.cpp:91 upper-cases
    first, n_total, heads = 0, n, None
    if shard_dev is not None:
        # one rank's rows of a sharded contig: the batches run over the site list of ALL ranks - where this rank's rows start in it
        # and the ten argmax values its rows may read from a batch that starts on another rank (two small collectives)
        from .dist import batch_heads, site_offsets
        first, n_total = site_offsets(n, shard_dev)
        heads = batch_heads(ga, first, n_total, batch_size, shard_dev)
    # the VCF rows depend on the batch boundary: one native call formats every batch (OpenMP over the batches)
    text, n_rows = host.vcf_format_batches(host.ContigTable([contig]), np.zeros(n, np.int32), site_pos, site_ref, ga, za, gm, zm, cov,
                                           batch_size=batch_size, score_mode=score_mode, as_view=as_view, first=first, n_total=n_total, heads=heads,
                                           nthreads=nthreads)
    return (text, n_rows) if shard_dev is None else (text, n_rows, n_total)


def _call_contig_rows_beside(model, mpileup_text, contig, chr_seq, min_af, min_coverage, batch_size, score_mode, chunk_bytes, stats, tokenise=None):
    """call_contig for one process with the rows of finished chunks formatted on a writer thread while later chunks compute: every
    chunk's call rows travel to a pinned buffer of their own behind the chunk's forward; the writer formats the COMPLETE batches of
    `batch_size` sites that have arrived (the reference's rows depend on the batch a site falls into: predict.py:102-125) with a
    quarter of the host threads and carries the rest over; what is left when the last chunk is back is formatted on all threads.
    Byte-identical to formatting all rows at the end (tests/test_gpu_predict.py)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    import torch
    pins = getattr(model, "_rows_pins", None)
    if pins is None:
        pins = model._rows_pins = []
    events, sizes, pieces = [], [], []
    state = dict(carry=np.empty((0, 13), np.float64), rows=0, busy=0.0)
    few = max(1, host.lib().nsnp_host_threads() // 4)

    def fmt(r, nthreads):
        t0 = time.perf_counter()
        site_pos = r[:, 0].astype(np.int64)
        text, n_rows = host.vcf_format_batches(host.ContigTable([contig]), np.zeros(len(r), np.int32), site_pos, chr_seq[site_pos - 1] & 0xDF,
                                               r[:, 1].astype(np.uint8), r[:, 2].astype(np.uint8), r[:, 3].astype(np.float32), r[:, 4].astype(np.float32),
                                               r[:, 5:13].astype(np.float32), batch_size=batch_size, score_mode=score_mode, nthreads=nthreads)
        pieces.append(text); state["rows"] += n_rows; state["busy"] += time.perf_counter() - t0

    def work(k):
        events[k].synchronize()
        got = pins[k][:sizes[k]].numpy()
        c = np.concatenate([state["carry"], got]) if len(state["carry"]) else got
        full = len(c) // batch_size * batch_size
        if full:
            fmt(c[:full], few)
        state["carry"] = c[full:].copy()

    with ThreadPoolExecutor(max_workers=1) as writer:
        futs = []

        def on_rows(rows_k):
            k, n = len(events), int(rows_k.shape[0])
            if k >= len(pins):
                pins.append(torch.empty((max(n + n // 4, 4096), 13), dtype=torch.float64, pin_memory=True))
            elif pins[k].shape[0] < n:
                pins[k] = torch.empty((n + n // 4, 13), dtype=torch.float64, pin_memory=True)
            pins[k][:n].copy_(rows_k, non_blocking=True)
            ev = torch.cuda.Event(blocking=True); ev.record()
            events.append(ev); sizes.append(n)
            futs.append(writer.submit(work, k))

        stream_contig(model, mpileup_text, contig, chr_seq, 0, None, chunk_bytes, min_af, min_coverage, stats, on_rows=on_rows, tokenise=tokenise)
        t0 = time.perf_counter()
        for f in futs:
            f.result()
        if len(state["carry"]):
            fmt(state["carry"], 0)
        n_sites = int(sum(sizes))
        text = b"".join(pieces)
        if stats is not None:
            stats["vcf_s"] = stats.get("vcf_s", 0.0) + time.perf_counter() - t0          # what the caller still waits for behind the last chunk
            stats["vcf_beside_s"] = stats.get("vcf_beside_s", 0.0) + state["busy"]
            stats["sites"] = stats.get("sites", 0) + n_sites
            stats["vcf_rows"] = stats.get("vcf_rows", 0) + state["rows"]
    return text, n_sites, state["rows"]


def call_contig(model, mpileup_text, contig: str, chr_seq: np.ndarray, min_af=0.12, min_coverage=6,
                batch_size=1000, score_mode=host.SCORE_FLOAT64, chunk_bytes=64 << 20, stats=None, rows_beside=None, tokenise=None):
    """One contig: returns (vcf_rows: bytes-like - a memoryview of the formatter's buffer, no copy; bytes(...) it to keep it -, n_sites, n_rows).  model: pileup_model.LSTMNetwork; mpileup_text: bytes, mmap or a
    numpy uint8 array holding the contig's samtools-mpileup text.  tokenise: "device" (default) / "host" (tokenise_mode).

    The text is worked off in chunks of whole lines (stream_contig: parse of chunk k + 1 on the host beside the device work of
    chunk k).  Under an initialised torch.distributed process group (one process per GPU, torchrun) the TEXT is statically sharded:
    rank r parses and calls only the lines of its byte range (cut at line boundaries; 16 lines of halo re-parsed, not exchanged) and
    formats the rows of ITS sites exactly as a single process would format them (the reference's batches of `batch_size` sites run
    over the whole site list: dist.site_offsets + dist.batch_heads hand a rank the little it needs of the others); the text is
    gathered to rank 0 in rank = position order.  Ranks other than 0 return (b"", n_sites_total, 0)."""
    import time
    import torch
    import torch.distributed as tdist
    from .dist import gather_text
    ctx = model.ctx
    finder, arr = _as_bytes_like(mpileup_text)
    sharded = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    rank, world = (tdist.get_rank(), tdist.get_world_size()) if sharded else (0, 1)
    if rows_beside is None:
        rows_beside = os.environ.get("NSNP_ROWS_BESIDE", "0") == "1"
    if rows_beside and not sharded:
        return _call_contig_rows_beside(model, mpileup_text, contig, chr_seq, min_af, min_coverage, batch_size, score_mode, chunk_bytes, stats, tokenise)
    cuts = line_cuts(finder, world, 0, arr.size)
    rows = stream_contig(model, mpileup_text, contig, chr_seq, cuts[rank], cuts[rank + 1], chunk_bytes, min_af, min_coverage, stats, tokenise=tokenise)
    if sharded:
        # every rank formats ITS rows (on its own host cores, exactly as the single process would format them: _format_rows), the text
        # - about 60 B per row, half of what the calls take - travels to rank 0 in one rooted gather
        backend_dev = torch.device("cuda", ctx.device) if tdist.get_backend() == "nccl" else "cpu"
        t0 = time.perf_counter()
        text, n_rows, n_sites = _format_rows(rows, contig, chr_seq, batch_size, score_mode, as_view=True, shard_dev=backend_dev, ctx=ctx)
        t1 = time.perf_counter()
        cnt = torch.tensor([n_rows], dtype=torch.int64, device=backend_dev)
        tdist.all_reduce(cnt)
        text = gather_text(text, backend_dev)
        if stats is not None:
            stats["vcf_s"] = stats.get("vcf_s", 0.0) + t1 - t0
            stats["gather_s"] = stats.get("gather_s", 0.0) + time.perf_counter() - t1
            stats["sites"] = stats.get("sites", 0) + int(rows.shape[0])
            stats["vcf_rows"] = stats.get("vcf_rows", 0) + n_rows
        return (text, n_sites, int(cnt.item())) if rank == 0 else (b"", n_sites, 0)
    # (formatting the rows of finished chunks on a worker thread while later chunks compute - rows_beside=True,
    # _call_contig_rows_beside - is built, byte-identical and SLOWER: round 3 with the 34 ms formatter 162 against 75 ms per 6 M-column
    # contig; round 5 with the 2.4 ms formatter on a quarter of the threads 27.6-28.4 against 22.5 ms (median of 30 steps, A/B/A/B on
    # one box): the per-chunk D2H copies and the writer's numpy passes cost the issuing thread 5 ms (GIL, copy-engine calls) to save
    # 2.4 ms behind the last chunk.  Off by default.)
    n_sites = int(rows.shape[0])
    if n_sites == 0:
        return b"", 0, 0
    t0 = time.perf_counter()
    text, n_rows = _format_rows(rows, contig, chr_seq, batch_size, score_mode, as_view=True, ctx=ctx)
    if stats is not None:
        stats["vcf_s"] = stats.get("vcf_s", 0.0) + time.perf_counter() - t0
        stats["sites"] = stats.get("sites", 0) + n_sites
        stats["vcf_rows"] = stats.get("vcf_rows", 0) + n_rows
    return text, n_sites, n_rows


def call_contigs(model, items, out, min_af=0.12, min_coverage=6, batch_size=1000, score_mode=host.SCORE_FLOAT64, chunk_bytes=64 << 20, stats=None):
    """A run over several contigs.  items: iterable of (name, mpileup text - bytes / mmap / uint8 array -, reference sequence uint8); out:
    a binary file object the rows are appended to (None on ranks other than 0).  Returns (sites, rows).
    One process: the rows of contig c are cut, brought to the host (on a stream of their own), formatted and written on a WRITER thread
    while contig c + 1 streams - formatting + writing is 4-5 ms behind every 6 M-column contig otherwise, a fifth of its time - on a
    quarter of the host threads (the parser keeps the rest busy; the last contig's rows get them all).  Under a process group the
    contigs run one after the other through call_contig (its collectives stay on the issuing thread)."""
    import time
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    import torch
    import torch.distributed as tdist
    sharded = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    st = stats if stats is not None else {}
    n_sites = n_rows = 0
    if sharded:
        for name, text, seq in items:
            rows_text, ns, nr = call_contig(model, text, name, seq, min_af, min_coverage, batch_size, score_mode, chunk_bytes, stats)
            if out is not None:
                out.write(rows_text)
            n_sites += ns; n_rows += nr
        return n_sites, n_rows
    dev = torch.device("cuda", model.ctx.device)
    side = getattr(model, "_rows_stream", None)
    if side is None:
        from . import _lib
        side = model._rows_stream = torch.cuda.Stream(dev)
        model._rows_ctx = _lib.Context(model.ctx.device)     # the writer thread's OWN context (a context serves one host thread at a time)
    wctx = model._rows_ctx
    few = max(1, host.lib().nsnp_host_threads() // 4)

    def finish(rows, name, seq, last, done=None):
        if done is not None:
            done.synchronize()                              # (the contig's last kernels: its stream call returned when they were issued)
        t0 = time.perf_counter()
        with torch.cuda.stream(side):                       # (the rows are complete: behind `done`, or behind stream_contig's own synchronize)
            text, nr = _format_rows(rows, name, seq, batch_size, score_mode, as_view=True, nthreads=0 if last else few, ctx=wctx)
        t1 = time.perf_counter()
        if out is not None:
            out.write(text)
        st["vcf_s"] = st.get("vcf_s", 0.0) + t1 - t0
        st["write_s"] = st.get("write_s", 0.0) + time.perf_counter() - t1
        return nr

    it = iter(items)
    nxt = next(it, None)
    pending = deque()
    on_device = tokenise_mode() == "device"
    finals = []
    with host.gc_paused(), ThreadPoolExecutor(max_workers=1) as writer:
        try:
            while nxt is not None:
                name, text, seq = nxt
                done = None
                if on_device:
                    # returns when the contig's last chunk is issued: the next contig's text is on its way while this one's tail computes
                    rows, done, fin = _stream_contig_dev(model, text, name, seq, 0, None, chunk_bytes, min_af, min_coverage, stats, None, defer=True)
                    finals.append(fin)
                else:
                    rows = stream_contig(model, text, name, seq, 0, None, chunk_bytes, min_af, min_coverage, stats)
                nxt = next(it, None)
                n_sites += int(rows.shape[0])
                st["sites"] = st.get("sites", 0) + int(rows.shape[0])
                if rows.shape[0]:
                    pending.append(writer.submit(finish, rows, name, seq, nxt is None, done))
                while len(pending) > 1 or (nxt is None and pending):          # at most one contig's rows behind the streaming one
                    t_w = time.perf_counter()
                    n_rows += pending.popleft().result()
                    st["wait_rows_s"] = st.get("wait_rows_s", 0.0) + time.perf_counter() - t_w
        finally:
            for fin in finals:                                                 # (per-stage times of the deferred contigs; waits for their last kernels)
                fin()
    st["vcf_rows"] = st.get("vcf_rows", 0) + n_rows
    return n_sites, n_rows


def call_variants(model, contigs, fasta_path, fai_text, output_file, **kw):
    """contigs: iterable of (name, path to <name>.mpileup).  Writes pileup.vcf (rank 0 only under torch.distributed: see
    call_contig); returns total rows.  The contigs are one run (call_contigs): the rows of one are written while the next streams."""
    import torch.distributed as tdist
    root = not (tdist.is_available() and tdist.is_initialized()) or tdist.get_rank() == 0
    maps = []

    def items():
        for name, path in contigs:
            seq = host.fasta_load_contig(fasta_path, name)
            g = open(path, "rb")
            size = os.fstat(g.fileno()).st_size
            text = mmap.mmap(g.fileno(), 0, access=mmap.ACCESS_READ) if size else b""       # parsed in place, never copied
            maps.append((g, text if size else None))
            yield name, text, seq

    f = open(output_file, "wb") if root else None
    try:
        if root:
            f.write(host.vcf_header(fai_text).encode())
        return call_contigs(model, items(), f, **kw)[1]
    finally:
        if f:
            f.close()
        for g, text in maps:
            if text is not None:
                try:
                    text.close()
                except BufferError:                  # (an exception on its way up still holds views of the mapping)
                    pass
            g.close()


# ---- .pd.bin site files -> pileup.vcf, streamed (PileupModel/predict.py:37-195 over PredictDataset files) ------------------------------
class _SiteSet:
    """buffers of one pass of a site file in flight: the [P,33,18] window matrices (as bytes: int16 or int32 views per pass)"""
    def __init__(self, P, dev=None):
        import torch
        kw = dict(pin_memory=True) if dev is None else dict(device=dev)
        self.P = P
        self.x = torch.empty(P * 33 * 18 * 4, dtype=torch.uint8, **kw)
        self.h2d_done = None
        self.free = None


def predict_pileup_bins(model, testing_paths, fai_text, output_file, batch_size=1000, score_mode=host.SCORE_FLOAT64, pass_sites=65536,
                        narrow=True, stats=None, distributed=True):
    with host.gc_paused():                             # (a generation-2 pass of the interpreter's collector sat in the set-up of a run: 38 ms)
        return _predict_pileup_bins(model, testing_paths, fai_text, output_file, batch_size, score_mode, pass_sites, narrow, stats, distributed)


def _predict_pileup_bins(model, testing_paths, fai_text, output_file, batch_size, score_mode, pass_sites, narrow, stats, distributed):
    """The reference's ``predict(model, testing_paths, reference_index_file, batch_size, output_file, device)`` (PileupModel/predict.py:
    37-195) over this repository's ``.pd.bin`` site files (sitefile.write_pileup_bin / pd_to_bin: the arrays of make_bin_predict_data.py:
    90-100): every file's windows are STREAMED - a worker thread `pread`s passes of `pass_sites` windows from the page cache into one of
    three pinned sets on all host cores (int16 counts as they are; the int32 counts of a reference-layout file narrowed to int16 on the
    way when they fit - they do: half the bytes over PCIe - else int32) and parses the ``ctg:pos:ref33`` fields natively (dataset.py:127-132), a copy stream sends the pass, the compute stream
    runs the PileupModel forward whose heads kernel writes argmax / max (10 bytes per site) straight into pinned host memory (the coverage
    slice of predict.py:63 is taken from the staged pass on the host: the H2D copy of a pass is the only copy-engine work); the
    files are one pipeline, and the rows of a file are formatted (one native call over the reference's batches of `batch_size` sites,
    which restart with every file as its DataLoader does) and appended on a writer thread while the next file computes.
    testing_paths: a list of paths, or a directory (its ``*.bin`` files in os.listdir order: predict.py:215).  Returns rows written.
    Under torch.distributed (one process per GPU) every rank works on its shard_range of every file's windows, cut at multiples of
    batch_size: the reference's batches run over the whole file, so a rank owns whole batches and formats its rows beside its compute as
    the single process does; the TEXT travels to rank 0 in one rooted gather behind the last file and rank 0 writes it in file, rank
    order (the other ranks return 0).  distributed=False: this process alone does the whole job even inside a process group."""
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    import torch
    import torch.distributed as tdist
    from . import sitefile
    from .dist import gather_text, shard_range
    from .hap_pipeline import _LocalNames
    t_begin = time.perf_counter()
    sharded = bool(distributed) and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    rank, world = (tdist.get_rank(), tdist.get_world_size()) if sharded else (0, 1)
    ctx = model.ctx
    dev = torch.device("cuda", ctx.device)
    st = stats if stats is not None else {}
    for k in ("stage_s", "h2d_s", "gpu_s", "vcf_s", "bytes_h2d", "sites", "passes", "passes_int16", "wait_stage_s", "issue_s", "drain_s", "stage_values_s",
              "stage_coverage_s", "stage_fields_s", "gpu_idle_s", "setup_s", "account_s"):
        st.setdefault(k, 0.0)
    if isinstance(testing_paths, (str, os.PathLike)):
        d = str(testing_paths)
        paths = [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".bin")] if os.path.isdir(d) else [d]
    else:
        paths = [str(p) for p in testing_paths]
    files = []
    for p in paths:                                    # every header is checked before a descriptor is opened or a row is written
        idx = sitefile.array_index(p)
        if ("position_matrix" not in idx or "position" not in idx or idx["position_matrix"][0] not in (np.dtype(np.int32), np.dtype(np.int16))
                or idx["position_matrix"][1][1:] != (33, 18)):
            raise sitefile.SiteFileError(f"{p}: not a pileup site file (position_matrix int16 / int32 [N,33,18] + position)")
        n_file = int(idx["position_matrix"][1][0])
        # this rank's windows of the file: [lo, hi) (everything without a process group), cut at multiples of batch_size: the reference's batches
        # run over the whole file, so every rank owns WHOLE batches and formats its rows without a word from the others
        lo, hi = shard_range(n_file, rank, world, align=int(batch_size))
        files.append(dict(path=p, n_file=n_file, lo=lo, n=hi - lo, x_off=idx["position_matrix"][2], fd=-1,
                          elem=idx["position_matrix"][0].itemsize,
                          position=sitefile.read_arrays(p, mmap=True)["position"]))
    P = int(max(1, pass_sites))
    seg_off = np.concatenate([[0], np.cumsum([f["n"] for f in files])]).astype(np.int64)
    n_total = int(seg_off[-1])
    passes = []                                                # (file, first window, end - ABSOLUTE indices in the file -, offset among this rank's sites)
    for fi, f in enumerate(files):
        a = f["lo"]
        while a < f["lo"] + f["n"]:
            b = min(f["lo"] + f["n"], a + (max(1, P // 4) if not passes and f["n"] > P else P))   # the very first pass: a quarter (the pipeline's fill)
            passes.append((fi, a, b, int(seg_off[fi]) + a - f["lo"]))
            a = b
    last_pass_of = {fi: k for k, (fi, _, _, _) in enumerate(passes)}
    names = _LocalNames()
    total_rows = 0
    kept = {}                                                  # sharded: the rows (text) of every file stay on this rank until the gather
    out = open(output_file, "wb") if rank == 0 else None
    try:
        for f in files:
            f["fd"] = os.open(f["path"], os.O_RDONLY)
        if out:
            out.write(host.vcf_header(fai_text).encode())
        if passes:
            n_sets = min(3, len(passes))
            hsets = getattr(model, "_site_host_sets", None)
            if not hsets or len(hsets) < n_sets or hsets[0].P < P:
                hsets = [_SiteSet(P) for _ in range(n_sets)]
                model._site_host_sets = hsets
                model._site_dev_sets = [_SiteSet(P, dev) for _ in range(n_sets)]
                model._site_copy_stream = host.copy_stream(dev)
            dsets, copy_stream = model._site_dev_sets, model._site_copy_stream
            main = torch.cuda.current_stream(dev)
            for s_ in hsets:
                s_.h2d_done = None
            for s_ in dsets:
                s_.free = None
            copy_stream.wait_stream(main)
            # pinned result arrays: one slot per file in flight (computing / being written / next), not one per run - pinning memory costs
            # about 0.5 ms per MB and a run may hold hundreds of files
            max_n = max(f["n"] for f in files)
            n_slots = 3
            res = getattr(model, "_site_results", None)
            if res is None or res["ga"].numel() < n_slots * max_n:
                mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
                res = dict(ga=mk(n_slots * max_n, torch.uint8), za=mk(n_slots * max_n, torch.uint8), gm=mk(n_slots * max_n, torch.float32),
                           zm=mk(n_slots * max_n, torch.float32))
                model._site_results = res
            pos_all = np.empty(n_total, np.int64); ctg_all = np.empty(n_total, np.int32); refb_all = np.empty(n_total, np.uint8)
            cov_all = np.empty((n_total, len(COV_CHANNELS)), np.float32)
            centers = (torch.arange(P, dtype=torch.int64, device=dev) * 33 + 16).contiguous()
            # the rows of a file have a whole file's compute time to be written: a quarter of the host threads, the staging keeps the rest busy
            writer_threads = max(1, host.lib().nsnp_host_threads() // 4)
            elem = [2 if narrow else 4]                 # int16 until a pass does not fit (then int32 for the rest of the run)
            lock = threading.Lock()

            def stage(k):
                t0 = time.perf_counter()
                fi, a, b, o = passes[k]
                f, m = files[fi], b - a
                hs = hsets[k % n_sets]
                e = 2 if f["elem"] == 2 else elem[0]
                v = hs.x.numpy()[:m * 594 * e].view(np.int16 if e == 2 else np.int32)
                if f["elem"] == 2:                      # int16 on disk (sitefile.write_pileup_bin's default): straight into the pinned set
                    host.stage_values(v, m * 594, fd=f["fd"], src_off=f["x_off"] + a * 594 * 2, src_dtype=np.int16)
                elif host.stage_values(v, m * 594, fd=f["fd"], src_off=f["x_off"] + a * 594 * 4):
                    with lock:
                        elem[0] = 4                     # a count beyond int16 (never at real coverage): this pass and the later ones go as int32
                    e = 4
                    v = hs.x.numpy()[:m * 594 * 4].view(np.int32)
                    host.stage_values(v, m * 594, fd=f["fd"], src_off=f["x_off"] + a * 594 * 4)
                t1 = time.perf_counter()
                host.window_channels(v, m, 16, COV_CHANNELS, out=cov_all[o:o + m])      # predict.py:63, from the staged pass: no D2H carries it
                t2 = time.perf_counter()
                fields = np.asarray(f["position"][a:b]).reshape(m, -1)
                p_, c_, r_ = host.parse_ctg_pos_ref(fields, names.table)
                while (c_ < 0).any():                   # a contig not seen before: one name per round (contigs are few, the parse is native)
                    names.add_names([bytes(fields[int(np.argmax(c_ < 0))]).rstrip(b"\0").strip().split(b":")[0].decode()])
                    p_, c_, r_ = host.parse_ctg_pos_ref(fields, names.table)
                pos_all[o:o + m] = p_; ctg_all[o:o + m] = c_; refb_all[o:o + m] = r_
                t3 = time.perf_counter()
                st["stage_values_s"] += t1 - t0; st["stage_coverage_s"] += t2 - t1; st["stage_fields_s"] += t3 - t2
                return e, t3 - t0

            def rows_of(fi, done):
                """writer thread: the VCF rows of file fi (the reference's batches restart with every file)"""
                nonlocal total_rows
                if done is not None:
                    done.synchronize()
                t0 = time.perf_counter()
                o0, o1 = int(seg_off[fi]), int(seg_off[fi + 1])
                r0 = (fi % n_slots) * max_n
                r1 = r0 + (o1 - o0)
                if o1 > o0:
                    text, rows = host.vcf_format_batches(names.table, ctg_all[o0:o1], pos_all[o0:o1], refb_all[o0:o1], res["ga"][r0:r1].numpy(),
                                                         res["za"][r0:r1].numpy(), res["gm"][r0:r1].numpy(), res["zm"][r0:r1].numpy(),
                                                         cov_all[o0:o1], batch_size=batch_size, score_mode=score_mode, as_view=True,
                                                         nthreads=writer_threads if fi + 1 < len(files) else 0,   # the last file: nothing else runs
                                                         first=files[fi]["lo"], n_total=files[fi]["n_file"])
                    if sharded:
                        kept[fi] = (text, rows)        # this rank's rows of the file, final: they travel as text behind the last file
                    else:
                        out.write(text)
                        total_rows += rows
                st["vcf_s"] += time.perf_counter() - t0

            tev = lambda: torch.cuda.Event(enable_timing=True)
            # h1 is what the main thread waits on before a host set is staged again: a blocking event, so that the wait sleeps (the host
            # threads are the scarce resource of this pipeline: under a 16-core quota the staging + row threads use 13-14 of them)
            ev = [dict(h0=tev(), h1=torch.cuda.Event(enable_timing=True, blocking=True), c0=tev(), c1=tev()) for _ in passes]
            seg_futs, next_seg = [], 0
            with ThreadPoolExecutor(max_workers=1) as pool, ThreadPoolExecutor(max_workers=1) as writer:
                futs = [pool.submit(stage, j) for j in range(min(2, len(passes)))]
                st["setup_s"] += time.perf_counter() - t_begin
                for k, (fi, a, b, o) in enumerate(passes):
                    m = b - a
                    t_w = time.perf_counter()
                    e, t_stage = futs[k].result()
                    t_i = time.perf_counter()
                    st["wait_stage_s"] += t_i - t_w; st["stage_s"] += t_stage
                    st["passes"] += 1; st["passes_int16"] += int(e == 2); st["sites"] += m
                    hs, ds = hsets[k % n_sets], dsets[k % n_sets]
                    nb = m * 594 * e
                    if ds.free is not None:
                        copy_stream.wait_event(ds.free)
                    with torch.cuda.stream(copy_stream):
                        ev[k]["h0"].record(copy_stream)
                        ds.x[:nb].copy_(hs.x[:nb], non_blocking=True)
                        ev[k]["h1"].record(copy_stream)
                    hs.h2d_done = ev[k]["h1"]
                    st["bytes_h2d"] += nb
                    t_c = time.perf_counter()
                    if k + 2 < len(passes):
                        nxt = hsets[(k + 2) % n_sets]
                        if nxt.h2d_done is not None:
                            nxt.h2d_done.synchronize()
                        futs.append(pool.submit(stage, k + 2))
                    t_s = time.perf_counter()
                    main.wait_event(ev[k]["h1"])
                    ev[k]["c0"].record(main)
                    x = ds.x[:nb].view(torch.int16 if e == 2 else torch.int32).view(m, 33, 18)
                    if e == 2:
                        x = x.to(torch.int32)                                         # (1.2 KB read + 2.4 KB written per site: ~1 ns of the forward's 51)
                    # argmax / max of both heads (10 bytes per site) are written by the heads kernel straight into the pinned result arrays:
                    # the H2D copy of a pass is the only copy-engine work of the run
                    if a == 0 and fi >= n_slots and fi - n_slots < len(seg_futs):
                        seg_futs[fi - n_slots].result()                                # the rows of the file that had this result slot are written
                    r = (fi % n_slots) * max_n + a - files[fi]["lo"]
                    ctx.pileup_forward_windows_calls(x.view(m * 33, 18), centers[:m],
                                                     calls_out=(res["ga"][r:r + m], res["za"][r:r + m], res["gm"][r:r + m], res["zm"][r:r + m]))
                    ev[k]["c1"].record(main)
                    ds.free = torch.cuda.Event(blocking=last_pass_of[fi] == k); ds.free.record(main)   # blocking: the writer thread sleeps on it
                    if last_pass_of[fi] == k:
                        while next_seg <= fi:
                            seg_futs.append(writer.submit(rows_of, next_seg, ds.free if next_seg == fi else None)); next_seg += 1
                    t_e = time.perf_counter()
                    st["issue_s"] += t_e - t_i
                    if "trace" in st:
                        st["trace"].append((k, m, t_w, t_i, t_c, t_s, t_e, t_stage))
                t_d = time.perf_counter()
                torch.cuda.synchronize(dev)
                for sf in seg_futs:
                    sf.result()
                st["drain_s"] += time.perf_counter() - t_d
            t_a = time.perf_counter()
            for k, e_ in enumerate(ev):
                st["h2d_s"] += e_["h0"].elapsed_time(e_["h1"]) * 1e-3
                st["gpu_s"] += e_["c0"].elapsed_time(e_["c1"]) * 1e-3
                if k:                                    # compute stream idle between two passes: waiting for a copy or for the host
                    gap = ev[k - 1]["c1"].elapsed_time(e_["c0"]) * 1e-3
                    st["gpu_idle_s"] += gap
                    if "gaps" in st and gap > 5e-4:
                        st["gaps"].append((k, round(gap * 1e3, 2), round(ev[k - 1]["c1"].elapsed_time(e_["h1"]), 2), round(e_["h0"].elapsed_time(e_["h1"]), 2)))
            st["account_s"] += time.perf_counter() - t_a
        if sharded:
            # every rank's rows are final text (whole batches of every file): the sizes travel as one small object, the text in ONE rooted
            # gather; rank 0 puts the pieces in file-major, rank-minor order
            t0 = time.perf_counter()
            mine = [(len(kept[fi][0]), kept[fi][1]) if fi in kept else (0, 0) for fi in range(len(files))]
            sizes = [None] * world
            tdist.all_gather_object(sizes, mine)
            backend_dev = torch.device("cuda", ctx.device) if tdist.get_backend() == "nccl" else "cpu"
            allt = gather_text(b"".join(kept[fi][0] for fi in range(len(files)) if fi in kept), backend_dev)
            if rank == 0:
                start = np.concatenate([[0], np.cumsum([sum(l for l, _ in sz) for sz in sizes])])
                within = [np.concatenate([[0], np.cumsum([l for l, _ in sz])]) for sz in sizes]
                for i in range(len(files)):
                    for r in range(world):
                        out.write(allt[int(start[r] + within[r][i]):int(start[r] + within[r][i + 1])])
                        total_rows += sizes[r][i][1]
            st["gather_s"] = st.get("gather_s", 0.0) + time.perf_counter() - t0
    finally:
        if out:
            out.close()
        for f in files:
            if f["fd"] >= 0:
                os.close(f["fd"])
    return total_rows


predict_pileup_bins.__doc__ = _predict_pileup_bins.__doc__
