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
_ERR = {-1: "invalid argument", -2: "out of memory", -3: "I/O error", -4: "malformed input",
        -5: "buffer too small"}


class HostError(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `make -C nanosnp_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(_LIB_PATH)
    p = C.c_void_p
    lib.nsnp_synth_columns.restype = C.c_int64
    lib.nsnp_synth_columns.argtypes = [C.c_uint64, C.c_int64, C.c_double, C.c_int, C.c_double,
                                       C.c_int, p, p, C.c_int64, p]
    lib.nsnp_synth_hap_planes.restype = C.c_int
    lib.nsnp_synth_hap_planes.argtypes = [C.c_uint64, C.c_int64, C.c_double, C.c_int, C.c_int,
                                          p, p, p, p, p]
    lib.nsnp_mpileup_parse.restype = C.c_int
    lib.nsnp_mpileup_parse.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64), p, p, p]
    lib.nsnp_fasta_load_contig.restype = C.c_int64
    lib.nsnp_fasta_load_contig.argtypes = [C.c_char_p, C.c_char_p, p, C.c_int64]
    lib.nsnp_pd_parse.restype = C.c_int64
    lib.nsnp_pd_parse.argtypes = [C.c_char_p, C.c_int64, p, p, p, p, p, C.c_int64]
    return lib


_lib = None


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
    _check(lib().nsnp_mpileup_parse(text, len(text), C.byref(n_cols), C.byref(n_bytes),
                                    None, None, None), "nsnp_mpileup_parse")
    M, B = n_cols.value, n_bytes.value
    pos = np.empty(M, np.int64)
    col_off = np.empty(M + 1, np.int64)
    bases = np.empty(max(B, 1), np.uint8)
    _check(lib().nsnp_mpileup_parse(text, len(text), C.byref(n_cols), C.byref(n_bytes),
                                    _ptr(pos), _ptr(col_off), _ptr(bases)), "nsnp_mpileup_parse")
    return pos, col_off, bases[:B]


def fasta_load_contig(path, contig):
    n = _check(lib().nsnp_fasta_load_contig(os.fsencode(path), contig.encode(), None, 0),
               f"nsnp_fasta_load_contig({contig})")
    seq = np.empty(max(n, 1), np.uint8)
    n2 = _check(lib().nsnp_fasta_load_contig(os.fsencode(path), contig.encode(), _ptr(seq), n),
                f"nsnp_fasta_load_contig({contig})")
    return seq[:n2]


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
