"""ctypes binding of ``libnanosnp_host.so`` (include/nsnp_host.h): synthetic generators and
the text readers that feed the device path.  Pure host code, importable without a GPU.

Reference counterparts: the libdnasv readers (dna_sv_tensor/src/common/line_reader.cpp,
ref_reader.cpp) and make_bin_predict_data.py:48-77; see the header for file:line citations.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnanosnp_host.so")

NSNP_EINVAL = -1
NSNP_ERANGE = -5
_ERR = {-1: "invalid argument", -2: "out of memory", -3: "I/O error", -4: "malformed input",
        -5: "buffer too small"}


class HostError(RuntimeError):
    pass


class HostRangeError(HostError):
    """an output buffer was too small; n_cols / n_bytes hold what the input needs"""
    def __init__(self, msg, n_cols, n_bytes):
        super().__init__(msg)
        self.n_cols, self.n_bytes = int(n_cols), int(n_bytes)


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `make -C nanosnp_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()')")
    # The library never touches the process environment.  Its OpenMP teams are woken once per text chunk; under a container CPU
    # quota idle workers that spin between the chunks eat the quota, so APPLICATIONS (bench.py, tools/e2e_bench.py) export
    # OMP_WAIT_POLICY=passive before anything loads an OpenMP runtime - see recommend_omp_env() and INTEGRATION.md.
    lib = C.CDLL(_LIB_PATH)
    p = C.c_void_p
    lib.nsnp_host_threads.restype = C.c_int
    lib.nsnp_host_set_threads.argtypes = [C.c_int]
    # one process per GPU on a shared host (torchrun sets LOCAL_WORLD_SIZE): the ranks share the host's thread budget
    try:
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        lws = 1
    if lws > 1 and "NSNP_HOST_THREADS" not in os.environ:
        lib.nsnp_host_set_threads(max(1, lib.nsnp_host_threads() // lws))
    lib.nsnp_synth_columns.restype = C.c_int64
    lib.nsnp_synth_columns.argtypes = [C.c_uint64, C.c_int64, C.c_double, C.c_int, C.c_double,
                                       C.c_int, p, p, C.c_int64, p]
    lib.nsnp_synth_hap_planes.restype = C.c_int
    lib.nsnp_synth_hap_planes.argtypes = [C.c_uint64, C.c_int64, C.c_double, C.c_int, C.c_int,
                                          p, p, p, p, p]
    lib.nsnp_mpileup_parse.restype = C.c_int
    lib.nsnp_mpileup_parse.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64), p, p, p]
    lib.nsnp_fasta_load_contig.restype = C.c_int64
    lib.nsnp_fasta_load_contig.argtypes = [C.c_char_p, C.c_char_p, p, C.c_int64]
    lib.nsnp_stage_values.restype = C.c_int
    lib.nsnp_stage_values.argtypes = [C.c_int, p, C.c_int64, C.c_int, C.c_int64, p, C.c_int, C.POINTER(C.c_int64)]
    lib.nsnp_parse_ctg_pos.restype = C.c_int
    lib.nsnp_parse_ctg_pos.argtypes = [p, C.c_int64, C.c_int, C.c_char_p, p, C.c_int, p, p]
    lib.nsnp_parse_ctg_pos_ref.restype = C.c_int
    lib.nsnp_parse_ctg_pos_ref.argtypes = [p, C.c_int64, C.c_int, C.c_char_p, p, C.c_int, p, p, p]
    lib.nsnp_window_channels.restype = C.c_int
    lib.nsnp_window_channels.argtypes = [p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, p, C.c_int, p]
    lib.nsnp_pd_parse.restype = C.c_int64
    lib.nsnp_pd_parse.argtypes = [C.c_char_p, C.c_int64, p, p, p, p, p, C.c_int64]
    return lib


_lib = None


_COPY_STREAMS = {}


def copy_stream(dev):
    """THE host-to-device copy stream of this process on `dev` (one per device, made on first use): every streamed pipeline of the
    package sends its passes on it.  The HIP runtime binds a stream to an SDMA engine when it first copies on it, and engines differ
    (measured: after other pipelines of the process had run, a pipeline copying on a stream of its own saw 78 MB passes take 1.8 ms
    instead of 1.1 and its run 40 % longer); one stream keeps the H2D traffic on the engine the first pipeline warmed up.  The
    pipelines of one process run one after the other, so sharing it costs nothing."""
    import torch
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    s = _COPY_STREAMS.get(key)
    if s is None:
        s = _COPY_STREAMS[key] = torch.cuda.Stream(dev)
    return s


class gc_paused:
    """`with host.gc_paused():` - the interpreter's cyclic collector off for the length of a streamed run, restored on the way out.  A
    generation-2 pass in the middle of a pipeline was measured at 38 ms (twelve passes of device work, a 56 ms step among 21 ms ones in
    the text path); the pipelines allocate per chunk but make no reference cycles, reference counting frees everything as before."""
    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def recommend_omp_env(environ=None):
    """For the `if __name__ == "__main__"` part of an application, BEFORE torch / numpy / this library are imported: idle OpenMP workers
    spin briefly, then sleep (OMP_WAIT_POLICY=passive with GOMP_SPINCOUNT=5000, unless the user set them).  The host routines wake
    a team once per text chunk / staging pass: under a container CPU quota workers that always spin eat the quota (50 ms per 6 M-column
    contig on 16 cores), workers that never spin pay a wake-up per region (30-33 ms), a short spin bridges the gaps between the regions
    of one chunk (22 ms with 5,000 to 100,000 spins) - and the shortest of those leaves the quota to the threads that work: the streamed
    site-file pipeline (three regions per pass, a pass every 3 ms) used 4.0 core-seconds per 0.29 s run and was throttled by the
    16-core quota twice per run with 100,000 spins (about 1.6 ms of `pause` per idle thread), 2.5 core-seconds and never with 5,000.  A library import must not do this for the process, and it has no effect once libgomp is initialised - hence a
    function the application calls, first thing."""
    env = os.environ if environ is None else environ
    env.setdefault("OMP_WAIT_POLICY", "passive")
    env.setdefault("GOMP_SPINCOUNT", "5000")
    return env


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc < 0:
        raise HostError(f"{what}: {_ERR.get(int(rc), rc)}")
    return rc


class Columns:
    """A block of pileup columns: ``bases`` (uint8 blob), ``col_off`` (int64[M+1]),
    ``ref`` (uint8[M], reference base per column) and ``pos`` (int64[M], 1-based)."""

    def __init__(self, bases, col_off, ref, pos):
        self.bases, self.col_off, self.ref, self.pos = bases, col_off, ref, pos

    @property
    def n_cols(self):
        return int(self.ref.shape[0])

    def column(self, c):
        return self.bases[self.col_off[c]:self.col_off[c + 1]].tobytes()

    def mpileup_text_native(self, contig="chrS"):
        """the same text as mpileup_text(), written by libnanosnp_host.so (OpenMP; whole synthetic contigs for the text-to-VCF bench)"""
        l = lib()
        args = (contig.encode(), self.n_cols, _ptr(np.ascontiguousarray(self.pos, np.int64)), _ptr(self.bases), _ptr(self.col_off))
        l.nsnp_columns_to_mpileup_text.restype = C.c_int64
        l.nsnp_columns_to_mpileup_text.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        need = -int(l.nsnp_columns_to_mpileup_text(*args, None, 0)) - 16
        if need < 0:
            _check(NSNP_EINVAL, "nsnp_columns_to_mpileup_text")
        buf = np.empty(max(need, 1), np.uint8)
        n = _check(l.nsnp_columns_to_mpileup_text(*args, _ptr(buf), need), "nsnp_columns_to_mpileup_text")
        return buf[:n]

    def mpileup_text(self, contig="chrS"):
        """samtools-mpileup text of these columns (no -f: the ref column is 'N',
        make_predict_data.sh:117)."""
        out = []
        name = contig.encode()
        for c in range(self.n_cols):
            b = self.column(c)
            d = sum(1 for ch in b if ch in b"ACGTNacgtn*#")
            out.append(b"%s\t%d\tN\t%d\t%s\t%s\n" % (name, int(self.pos[c]), d, b, b"I" * max(d, 1)))
        return b"".join(out)


def synth_columns(seed, n_cols, coverage=30.0, max_depth=144, het_rate=0.02, window=0,
                  pos_start=1):
    """G1 (``window=0``) / G2 (``window=33``) generator of SURVEY.md 8(d)."""
    M = int(n_cols)
    ref = np.empty(M, np.uint8)
    col_off = np.empty(M + 1, np.int64)
    # size query (bases = NULL returns -(needed + 16)), then one exact allocation
    rc = lib().nsnp_synth_columns(int(seed), M, float(coverage), int(max_depth), float(het_rate),
                                  int(window), _ptr(ref), None, 0, _ptr(col_off))
    if rc > -16:
        _check(rc if rc < 0 else NSNP_EINVAL, "nsnp_synth_columns")
    cap = -int(rc) - 16
    bases = np.empty(max(cap, 1), np.uint8)
    rc = _check(lib().nsnp_synth_columns(int(seed), M, float(coverage), int(max_depth),
                                         float(het_rate), int(window), _ptr(ref), _ptr(bases),
                                         cap, _ptr(col_off)), "nsnp_synth_columns")
    bases = bases[:rc]
    if window:
        # stand-alone windows: leave a gap after each so that no window continues another
        idx = np.arange(M, dtype=np.int64)
        pos = pos_start + (idx // window) * (2 * window) + (idx % window)
    else:
        pos = pos_start + np.arange(M, dtype=np.int64)
    return Columns(bases, col_off, ref, pos)


def synth_hap_planes(seed, n_sites, coverage=30.0, depth=None, length=33):
    """G3 generator: returns (seq, baseq, mapq, hap) int32[N,D,L] and ref_row int32[N,L]."""
    N = int(n_sites)
    D = int(depth if depth is not None else 3 * coverage)
    L = int(length)
    planes = [np.empty((N, D, L), np.int32) for _ in range(4)]
    ref_row = np.empty((N, L), np.int32)
    _check(lib().nsnp_synth_hap_planes(int(seed), N, float(coverage), D, L,
                                       *[_ptr(a) for a in planes], _ptr(ref_row)),
           "nsnp_synth_hap_planes")
    return planes[0], planes[1], planes[2], planes[3], ref_row


def mpileup_parse(text: bytes):
    """mpileup text of one contig -> (pos int64[M], col_off int64[M+1], bases uint8[...])."""
    n_cols, n_bytes = C.c_int64(0), C.c_int64(0)
    text = bytes(text)
    tp = C.cast(C.c_char_p(text), C.c_void_p)
    _check(lib().nsnp_mpileup_parse(tp, len(text), C.byref(n_cols), C.byref(n_bytes),
                                    None, None, None), "nsnp_mpileup_parse")
    M, B = n_cols.value, n_bytes.value
    pos = np.empty(M, np.int64)
    col_off = np.empty(M + 1, np.int64)
    bases = np.empty(max(B, 1), np.uint8)
    _check(lib().nsnp_mpileup_parse(tp, len(text), C.byref(n_cols), C.byref(n_bytes),
                                    _ptr(pos), _ptr(col_off), _ptr(bases)), "nsnp_mpileup_parse")
    return pos, col_off, bases[:B]


def mpileup_parse_range(buf, lo, hi, out=None, strict_lines=False):
    """The lines of buf[lo:hi] (any buffer: bytes, mmap, numpy uint8; lo / hi on line boundaries) -> (pos, col_off, bases) without
    copying the text, every line tokenised once (nsnp_mpileup_parse_into).  out = (pos, col_off, bases) numpy arrays to fill (e.g.
    views of pinned host tensors); the returned arrays are views of their first M / M + 1 / B elements.  Without `out` the arrays
    are allocated at the always-sufficient bounds (hi - lo) / 8 columns and hi - lo bytes and trimmed.  Buffers that are too small
    raise HostRangeError (n_cols / n_bytes = what the lines need).  strict_lines: an empty / CR-only line is an error instead of being
    stepped over (nsnp_mpileup_parse_lines: for callers that count lines themselves; the reference aborts on such a line)."""
    a = np.frombuffer(buf, np.uint8) if not isinstance(buf, np.ndarray) else buf
    n = int(hi) - int(lo)
    if n <= 0:
        return np.empty(0, np.int64), np.zeros(1, np.int64), np.empty(0, np.uint8)
    fn = lib().nsnp_mpileup_parse_lines
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p,
                   C.POINTER(C.c_int64)]
    if out is None:
        out = (np.empty(n // 8 + 2, np.int64), np.empty(n // 8 + 3, np.int64), np.empty(n, np.uint8))
    pos, col_off, bases = out
    n_cols, n_bytes, n_skipped = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    rc = fn(C.c_void_p(a.ctypes.data + int(lo)), n, min(pos.size, col_off.size - 1), bases.size, C.byref(n_cols), C.byref(n_bytes),
            _ptr(pos), _ptr(col_off), _ptr(bases), C.byref(n_skipped))
    if rc == NSNP_ERANGE:
        raise HostRangeError(f"mpileup_parse_range: output buffers too small ({n_cols.value} columns, {n_bytes.value} bytes needed)",
                             n_cols.value, n_bytes.value)
    _check(rc, "nsnp_mpileup_parse_lines")
    if strict_lines and n_skipped.value:
        raise HostError(f"mpileup text holds {n_skipped.value} empty line(s): malformed input (every line must be one pileup column)")
    M, B = n_cols.value, n_bytes.value
    return pos[:M], col_off[:M + 1], bases[:B]


def stage_values(dst, n, src=None, fd=-1, src_off=0, src_dtype=np.int32):
    """n values of src_dtype (int32 / int16 / int8) from a file descriptor at byte offset src_off, or from the numpy array `src`
    (C-contiguous; src_off in bytes), into the first n elements of dst (a numpy array of int32, int16 or int8: e.g. the view of a pinned
    tensor), on all host threads (nsnp_stage_values).  int32 -> int16 / int8 narrows; returns the number of values that did not fit (0 in
    every other case)."""
    es, ed = np.dtype(src_dtype).itemsize, dst.dtype.itemsize
    if dst.size < n or not dst.flags["C_CONTIGUOUS"]:
        raise HostError("stage_values: destination too small or not contiguous")
    if src is not None:
        if not src.flags["C_CONTIGUOUS"] or src.dtype.itemsize != es or src_off + n * es > src.nbytes:
            raise HostError("stage_values: source must be a C-contiguous array of src_dtype holding n values behind src_off")
    elif fd < 0:
        raise HostError("stage_values: a file descriptor or a source array is needed")
    bad = C.c_int64(0)
    rc = lib().nsnp_stage_values(int(fd) if src is None else -1, None if src is None else C.c_void_p(src.ctypes.data), int(src_off), es, int(n),
                                 C.c_void_p(dst.ctypes.data), ed, C.byref(bad))
    _check(rc, "nsnp_stage_values")
    return bad.value


def window_channels(x, n, row, channels, out=None):
    """x: staged windows, a C-contiguous int16 / int32 array holding [n, 33, 18] values -> float32 [n, len(channels)] =
    x[:, row, channels] (the coverage slice of PileupModel/predict.py:63), on all host threads (nsnp_window_channels)."""
    ch = np.ascontiguousarray(channels, np.int32)
    if x.dtype.itemsize not in (2, 4) or not x.flags["C_CONTIGUOUS"] or x.size < n * 594:
        raise HostError("window_channels: x must be a C-contiguous int16 / int32 array of n x 33 x 18 values")
    out = np.empty((n, ch.size), np.float32) if out is None else out
    if out.dtype != np.float32 or not out.flags["C_CONTIGUOUS"] or out.size < n * ch.size:
        raise HostError("window_channels: out must be a C-contiguous float32 array of n x len(channels)")
    _check(lib().nsnp_window_channels(C.c_void_p(x.ctypes.data), x.dtype.itemsize, int(n), 33, 18, int(row), _ptr(ch), int(ch.size),
                                      C.c_void_p(out.ctypes.data)), "nsnp_window_channels")
    return out


def parse_ctg_pos(rows, table):
    """rows: uint8 [n, width] zero-padded "ctg:pos" fields (a bin's candidate_positions / haplotype_positions, any leading shape);
    table: ContigTable -> (pos int64 [...], ctg int32 [...]; -1 = a contig the table does not hold).  HostError on a field the
    reference's str.split(":") / int() would raise on (dataset_dev.py:109-110,153-154)."""
    a = np.ascontiguousarray(rows, np.uint8)
    lead, width = a.shape[:-1], a.shape[-1]
    n = int(np.prod(lead, dtype=np.int64))
    pos = np.empty(lead, np.int64); ctg = np.empty(lead, np.int32)
    if n:
        _check(lib().nsnp_parse_ctg_pos(_ptr(a), n, int(width), table.blob, _ptr(table.off), len(table.off) - 1, _ptr(pos), _ptr(ctg)),
               "nsnp_parse_ctg_pos (a position field is not 'ctg:pos')")
    return pos, ctg


def parse_ctg_pos_ref(rows, table):
    """rows: uint8 [n, width] zero-padded "ctg:pos:ref33" fields (the `position` array of a .pd.bin) -> (pos int64 [n], ctg int32 [n]
    (-1 = a contig the table does not hold), ref_base uint8 [n] = ord(seq[16])): PileupModel/dataset.py:127-132.  HostError where
    the reference raises."""
    a = np.ascontiguousarray(rows, np.uint8)
    n, width = a.shape
    pos = np.empty(n, np.int64); ctg = np.empty(n, np.int32); refb = np.empty(n, np.uint8)
    if n:
        _check(lib().nsnp_parse_ctg_pos_ref(_ptr(a), n, int(width), table.blob, _ptr(table.off), len(table.off) - 1, _ptr(pos), _ptr(ctg), _ptr(refb)),
               "nsnp_parse_ctg_pos_ref (a position field is not 'ctg:pos:ref33')")
    return pos, ctg, refb


def fasta_load_contig(path, contig):
    n = _check(lib().nsnp_fasta_load_contig(os.fsencode(path), contig.encode(), None, 0),
               f"nsnp_fasta_load_contig({contig})")
    seq = np.empty(max(n, 1), np.uint8)
    n2 = _check(lib().nsnp_fasta_load_contig(os.fsencode(path), contig.encode(), _ptr(seq), n),
                f"nsnp_fasta_load_contig({contig})")
    return seq[:n2]


def load_reference_file(ref_fn):
    """get_truth.load_reference_file (HaplotypeModel/get_truth.py:88-104), what predict_dev.py:28 hands to TestDataset: {contig:
    sequence} with the contig name = the header line up to its first BLANK (not tab), every '>' of that token removed; the stripped
    sequence lines joined as they are (case kept: BASE2INT then maps lower case to 0); a header without sequence lines leaves no entry;
    the later of two equal names wins; lines in front of the first header land under the name "".  Values are bytes (the reference's
    are str: same characters)."""
    references, ctg, parts = {}, "", []
    with open(ref_fn, "rb") as fin:
        for line in fin.read().replace(b"\r\n", b"\n").replace(b"\r", b"\n").split(b"\n"):      # (text mode there: universal newlines)
            if line.startswith(b">"):
                if parts and any(parts):
                    references[ctg] = b"".join(parts)
                parts = []
                ctg = line.decode().strip().split(" ")[0].replace(">", "")
            else:
                t = line.decode().strip().encode() if line else b""
                if t:
                    parts.append(t)
    if parts:
        references[ctg] = b"".join(parts)
    return references


def pd_parse(text: bytes):
    """.pd text -> (x int32[N,33,18], contig names list[str], pos int64[N], ref_base uint8[N])."""
    n = _check(lib().nsnp_pd_parse(text, len(text), None, None, None, None, None, 0),
               "nsnp_pd_parse")
    x = np.empty((n, 33, 18), np.int32)
    pos = np.empty(n, np.int64)
    refb = np.empty(n, np.uint8)
    cb = np.empty(n, np.int64)
    ce = np.empty(n, np.int64)
    _check(lib().nsnp_pd_parse(text, len(text), _ptr(x), _ptr(pos), _ptr(refb), _ptr(cb),
                               _ptr(ce), n), "nsnp_pd_parse")
    names = [text[b:e].decode() for b, e in zip(cb.tolist(), ce.tolist())]
    return x, names, pos, refb


def write_fasta(path, contig, seq: np.ndarray, line=60):
    """Writes a one-contig FASTA and its .fai (5 columns, ref_reader.cpp:21)."""
    s = bytes(seq)
    with open(path, "wb") as f:
        header = b">" + contig.encode() + b"\n"
        f.write(header)
        for i in range(0, len(s), line):
            f.write(s[i:i + line] + b"\n")
    with open(path + ".fai", "w") as f:
        f.write(f"{contig}\t{len(s)}\t{len(header)}\t{line}\t{line + 1}\n")


# ---- text writers of the predict loops (nsnp_vcf.c) -------------------------------------------------
def _bind_vcf():
    l = lib()
    if getattr(l, "_vcf_bound", False):
        return l
    p = C.c_void_p
    l.nsnp_vcf_format_batch.restype = C.c_int64
    l.nsnp_vcf_format_batch.argtypes = [C.c_int64, C.c_char_p, p, p, p, p, p, p, p, p, p, C.c_int, p, C.c_int64,
                                        C.POINTER(C.c_int64)]
    l.nsnp_vcf_format_batches_part.restype = C.c_int64
    l.nsnp_vcf_format_batches_part.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, p, C.c_char_p, p, p, p, p, p, p, p, p, p, C.c_int, p,
                                               C.c_int64, C.POINTER(C.c_int64), C.c_int]
    l.nsnp_vcf_format_batches.restype = C.c_int64
    l.nsnp_vcf_format_batches.argtypes = [C.c_int64, C.c_int64, C.c_char_p, p, p, p, p, p, p, p, p, p, C.c_int, p, C.c_int64,
                                          C.POINTER(C.c_int64), C.c_int]
    l.nsnp_hap_csv_format.restype = C.c_int64
    l.nsnp_hap_csv_format.argtypes = [C.c_int64, C.c_char_p, p, p, p, p, p, C.c_int, p, C.c_int64]
    l.nsnp_calculate_score.restype = C.c_double
    l.nsnp_calculate_score.argtypes = [C.c_float, C.c_int, C.POINTER(C.c_int)]
    l._vcf_bound = True
    return l


SCORE_FLOAT32, SCORE_FLOAT64 = 0, 1


class ContigTable:
    """contig names -> ids, as a blob + offsets for the C writers"""

    def __init__(self, names):
        uniq = list(dict.fromkeys(names))
        self.index = {n: i for i, n in enumerate(uniq)}
        self.blob = "".join(uniq).encode()
        self.off = np.concatenate([[0], np.cumsum([len(n.encode()) for n in uniq])]).astype(np.int64)
        self.ids = np.array([self.index[n] for n in names], np.int32)


def vcf_format_batch(table, contig_id, pos, ref_base, gt_arg, zy_arg, gt_prob, zy_prob, cov, score_mode=SCORE_FLOAT64):
    """One batch of PileupModel/predict.py:66-194 -> (bytes, n_rows)."""
    l = _bind_vcf()
    B = len(pos)
    args = [np.ascontiguousarray(contig_id, np.int32), np.ascontiguousarray(pos, np.int64),
            np.ascontiguousarray(ref_base, np.uint8), np.ascontiguousarray(gt_arg, np.uint8),
            np.ascontiguousarray(zy_arg, np.uint8), np.ascontiguousarray(gt_prob, np.float32),
            np.ascontiguousarray(zy_prob, np.float32), np.ascontiguousarray(cov, np.float32)]
    cap = 160 * B + 256
    rows = C.c_int64(0)
    while True:
        buf = np.empty(cap, np.uint8)
        n = l.nsnp_vcf_format_batch(B, table.blob, _ptr(table.off), *[_ptr(a) for a in args], int(score_mode),
                                    _ptr(buf), cap, C.byref(rows))
        if n >= 0:
            return buf[:n].tobytes(), rows.value
        if n > -16:
            _check(n, "nsnp_vcf_format_batch")
        cap = -int(n)


def vcf_format_batches(table, contig_id, pos, ref_base, gt_arg, zy_arg, gt_prob, zy_prob, cov, batch_size=1000,
                       score_mode=SCORE_FLOAT64, nthreads=None, as_view=False, first=0, n_total=None, heads=None):
    """All batches of the predict loop in one native call (OpenMP over batches) -> (bytes, n_rows); byte-identical to
    concatenating vcf_format_batch over consecutive slices of batch_size sites.  as_view: a memoryview of the output buffer instead
    of a bytes copy of it (12 MB per 200 k rows).
    first / n_total / heads: the arrays are the rows [first, first + N) of a list of n_total sites whose batches run over the whole
    list (one rank's share; nsnp_vcf_format_batches_part): heads = uint8 [ceil(n_total / batch_size), 10], the first ten argmax values
    of every batch (dist.batch_heads), or None when first is a multiple of batch_size.  The parts' texts concatenate to the whole."""
    l = _bind_vcf()
    N = len(pos)
    n_total = first + N if n_total is None else int(n_total)
    if heads is not None:
        heads = np.ascontiguousarray(heads, np.uint8)
        if heads.size < 10 * (-(-n_total // int(batch_size))):
            raise ValueError("heads holds fewer than ten values per batch of the whole list")
    args = [np.ascontiguousarray(contig_id, np.int32), np.ascontiguousarray(pos, np.int64),
            np.ascontiguousarray(ref_base, np.uint8), np.ascontiguousarray(gt_arg, np.uint8),
            np.ascontiguousarray(zy_arg, np.uint8), np.ascontiguousarray(gt_prob, np.float32),
            np.ascontiguousarray(zy_prob, np.float32), np.ascontiguousarray(cov, np.float32)]
    nthreads = int(nthreads or 0)                      # 0: nsnp_host_threads() - affinity mask and cgroup quota
    cap = 80 * N + 256
    rows = C.c_int64(0)
    while True:
        buf = np.empty(cap, np.uint8)
        n = l.nsnp_vcf_format_batches_part(N, int(batch_size), int(first), n_total, _ptr(heads) if heads is not None else None,
                                           table.blob, _ptr(table.off), *[_ptr(a) for a in args], int(score_mode),
                                           _ptr(buf), cap, C.byref(rows), nthreads)
        if n >= 0:
            return (memoryview(buf)[:n] if as_view else buf[:n].tobytes()), rows.value
        if n > -16:
            _check(n, "nsnp_vcf_format_batches")
        cap = -int(n)


def hap_csv_format(table, contig_id, pos, gt_arg, gt_prob, score_mode=SCORE_FLOAT64):
    l = _bind_vcf()
    N = len(pos)
    args = [np.ascontiguousarray(contig_id, np.int32), np.ascontiguousarray(pos, np.int64),
            np.ascontiguousarray(gt_arg, np.uint8), np.ascontiguousarray(gt_prob, np.float32)]
    cap = 96 * N + 256
    while True:
        buf = np.empty(cap, np.uint8)
        n = l.nsnp_hap_csv_format(N, table.blob, _ptr(table.off), *[_ptr(a) for a in args], int(score_mode),
                                  _ptr(buf), cap)
        if n >= 0:
            return buf[:n].tobytes()
        if n > -16:
            _check(n, "nsnp_hap_csv_format")
        cap = -int(n)


def calculate_score(p, score_mode=SCORE_FLOAT64):
    """(score, ok): ok is False where the reference's calculate_score raises"""
    ok = C.c_int(0)
    v = _bind_vcf().nsnp_calculate_score(float(np.float32(p)), int(score_mode), C.byref(ok))
    return v, bool(ok.value)


VCF_HEADER = ('##fileformat=VCFv4.3\n'
              '##FILTER=<ID=PASS,Description="All filters passed">\n'
              '##FILTER=<ID=RefCall,Description="Reference call">\n'
              '{contigs}'
              '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n'
              '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">\n'
              '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read Depth">\n'
              '##FORMAT=<ID=AF,Number=A,Type=Float,Description="Allele Frequency">\n'
              '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSample\n')


def vcf_header(fai_text: str):
    """write_head of PileupModel/predict.py:13-27 from the text of a .fai (or its path).  As there, a line that does not hold two fields - a blank
    line among them - raises IndexError (`line.strip().split()[1]`): a .fai written by samtools has neither."""
    if isinstance(fai_text, os.PathLike) or (isinstance(fai_text, str) and "\n" not in fai_text and "\t" not in fai_text and os.path.isfile(fai_text)):
        with open(fai_text) as f:                                    # the reference's argument is the PATH of the index (predict.py:37,216)
            fai_text = f.read()
    lines = fai_text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                                                  # (iterating a file yields no line behind the last newline)
    contigs = []
    for line in lines:
        f = line.strip().split()
        if len(f) < 2:
            raise IndexError(f"fai line without a length field: {line!r} (PileupModel/predict.py:19-20 indexes [1])")
        contigs.append('##contig=<ID={},length={}>\n'.format(f[0], f[1]))
    return VCF_HEADER.format(contigs="".join(contigs))


# ---- reference rows of the haplotype features (H3) ---------------------------------------------------
_BASE2INT = {65: 1, 67: 2, 71: 3, 84: 4, 78: 0}      # dataset_dev.py:9 {'A':1,'C':2,'G':3,'T':4,'N':0}


def haplotype_ref_rows(references, candidate_positions, length=33, position_lists=None):
    """PileupFeature / HaplotypeFeature reference rows (HaplotypeModel/dataset_dev.py:106-120,150-162).

    references: {contig: uint8 array or bytes}.  candidate_positions: "ctg:pos" strings.  With
    position_lists=None the row covers pos-length//2 .. pos+length//2 (the 33-wide pileup window);
    otherwise position_lists[i] holds the "ctg:pos" strings of the group (the 11 haplotype columns).
    Any failure -- unknown contig, lower-case or IUPAC base, position past the end -- gives 0 exactly
    like the reference's bare except; a NEGATIVE 0-based index wraps around like Python indexing does."""
    lut = np.zeros(256, np.int32)
    for k, v in _BASE2INT.items():
        lut[k] = v
    n = len(candidate_positions)
    width = length if position_lists is None else len(position_lists[0]) if n else length
    out = np.zeros((n, width), np.int32)
    for i in range(n):
        if position_lists is None:
            ctg, pos = candidate_positions[i].split(":")
            cols = [(ctg, j - 1) for j in range(int(pos) - length // 2, int(pos) + length // 2 + 1)]
        else:
            cols = []
            for item in position_lists[i]:
                ctg, pos = item.split(":")
                cols.append((ctg, int(pos) - 1))
        for k, (ctg, rp) in enumerate(cols):
            seq = references.get(ctg)
            if seq is None:
                continue
            L = len(seq)
            if rp >= L or rp < -L:
                continue
            out[i, k] = lut[seq[rp]]
    return out
