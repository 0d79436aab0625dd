"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this
module (see oracle/oracle.h).  The product package ``nanosnp_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
NCH = 18


def build(force=False):
    """(Re)build liboracle.so (and oracle/_ref when the reference tree is mounted)."""
    if force or not os.path.exists(_LIB):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def _load():
    build()
    lib = C.CDLL(_LIB)
    p = C.c_void_p
    lib.orc_encode_columns.restype = None
    lib.orc_encode_columns.argtypes = [p, p, p, C.c_int64, C.c_double, C.c_int, p, p, p]
    lib.orc_encode_columns2.restype = None
    lib.orc_encode_columns2.argtypes = [p, p, p, C.c_int64, C.c_double, C.c_double, C.c_int, p, p, p]
    lib.orc_mpileup_to_pd2.restype = C.c_int64
    lib.orc_mpileup_to_pd2.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int, C.c_char_p]
    lib.orc_select_sites.restype = C.c_int64
    lib.orc_select_sites.argtypes = [p, p, C.c_int64, C.c_int, p, C.c_int64]
    lib.orc_gather_windows.restype = None
    lib.orc_gather_windows.argtypes = [p, p, C.c_int64, C.c_int, p]
    lib.orc_mpileup_to_pd.restype = C.c_int64
    lib.orc_mpileup_to_pd.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_double, C.c_int,
                                      C.c_int, C.c_char_p]
    lib.orc_mpileup_tokenise.restype = C.c_int64
    lib.orc_mpileup_tokenise.argtypes = [p, C.c_int64, C.c_int64, p, p, p]
    lib.orc_pileup_forward.restype = None
    lib.orc_pileup_forward.argtypes = [p, p, C.c_int64, p, p, C.c_int]
    lib.orc_pileup_forward_blocked.restype = None
    lib.orc_pileup_forward_blocked.argtypes = [p, p, C.c_int64, p, p, C.c_int]
    lib.orc_hap_features_batch.restype = None
    lib.orc_hap_features_batch.argtypes = [p, p, p, p, p, C.c_int64, C.c_int, C.c_int, p, C.c_int]
    lib.orc_hap_features.restype = None
    lib.orc_hap_features.argtypes = [p, p, p, p, p, C.c_int, C.c_int, p]
    lib.orc_hap_arrange.restype = None
    lib.orc_hap_arrange.argtypes = [p, p, p, p, C.c_int, C.c_int, C.c_int, C.c_int, p, p, p, p, p]
    lib.orc_hap_forward.restype = None
    lib.orc_hap_forward.argtypes = [p, p, p, C.c_int64] + [C.c_int] * 7 + [p, p, C.c_int]
    lib.orc_cat_forward.restype = None
    lib.orc_cat_forward.argtypes = [p, p, p, C.c_int64, p, C.c_int]
    lib.orc_cat_groups.restype = None
    lib.orc_cat_groups.argtypes = [p, p, p, C.c_int, p, p, p, C.c_int, C.c_int64, C.c_int, p]
    lib.orc_calculate_score.restype = C.c_double
    lib.orc_calculate_score.argtypes = [C.c_double]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def encode_columns(bases, col_off, ref, min_af=0.12, min_coverage=6, indel_min_af=None):
    bases = _c(bases, np.uint8); col_off = _c(col_off, np.int64); ref = _c(ref, np.uint8)
    M = ref.shape[0]
    counts = np.empty((M, NCH), np.int32); depth = np.empty(M, np.int32); flags = np.empty(M, np.uint8)
    if bases.size == 0:
        bases = np.zeros(1, np.uint8)
    lib().orc_encode_columns2(_p(bases), _p(col_off), _p(ref), M, min_af, min_af if indel_min_af is None else indel_min_af, min_coverage,
                              _p(counts), _p(depth), _p(flags))
    return counts, depth, flags


def select_sites(pos, flags, flank=16):
    pos = _c(pos, np.int64); flags = _c(flags, np.uint8)
    M = pos.shape[0]
    out = np.empty(max(M, 1), np.int64)
    n = lib().orc_select_sites(_p(pos), _p(flags), M, flank, _p(out), M)
    return out[:n].copy()


def gather_windows(counts, center_idx, flank=16):
    counts = _c(counts, np.int32); center_idx = _c(center_idx, np.int64)
    N = center_idx.shape[0]
    x = np.empty((N, 2 * flank + 1, NCH), np.int32)
    if N:
        lib().orc_gather_windows(_p(counts), _p(center_idx), N, flank, _p(x))
    return x


def mpileup_to_pd(mpileup_path, chr_seq: bytes, pd_path, min_af=0.12, min_coverage=6, flank=16, indel_min_af=None):
    n = lib().orc_mpileup_to_pd2(os.fsencode(mpileup_path), chr_seq, len(chr_seq), min_af, min_af if indel_min_af is None else indel_min_af,
                                min_coverage, flank, os.fsencode(pd_path))
    if n < 0:
        raise RuntimeError(f"orc_mpileup_to_pd failed: {n}")
    return n


def mpileup_tokenise(text):
    """mpileup text (bytes / uint8 array) -> (pos [M] int64, col_off [M + 1] int64, bases uint8): what the reference's reader makes of every
    line (position, column 5), the column-5 strings laid end to end.  ValueError when a line has fewer than five fields."""
    t = np.frombuffer(text, np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, np.uint8)
    cap = int((t == 10).sum()) + 1
    pos, beg, end = (np.zeros(cap, np.int64) for _ in range(3))
    m = lib().orc_mpileup_tokenise(_p(t), t.size, cap, _p(pos), _p(beg), _p(end))
    if m < 0:
        raise ValueError(f"line {-m - 1}: fewer than five fields")
    pos, beg, end = pos[:m], beg[:m], end[:m]
    off = np.zeros(m + 1, np.int64)
    np.cumsum(end - beg, out=off[1:])
    bases = np.concatenate([t[b:e] for b, e in zip(beg, end)]) if m else np.zeros(0, np.uint8)
    return pos, off, bases


def _wptrs(weights):
    ws = [_c(w, np.float32) for w in weights]
    arr = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    return ws, arr


def pileup_forward(weights, x, nthreads=1, blocked=False):
    """weights: the 24 fp32 arrays of ont_pileup.chkpt in state-dict order.  blocked=True runs the cache-blocked AVX2
    arrangement of the same schedule that bench.py times as its CPU baseline (never used as the checker)."""
    ws, arr = _wptrs(weights)
    x = _c(x, np.int32)
    N = x.shape[0]
    gt = np.empty((N, 21), np.float32); zy = np.empty((N, 3), np.float32)
    (lib().orc_pileup_forward_blocked if blocked else lib().orc_pileup_forward)(arr, _p(x), N, _p(gt), _p(zy), nthreads)
    return gt, zy


def hap_features(seq, bq, mq, hap, ref_row):
    """One site, float64 [105, L] -- the exact counterpart of get_frequency_feature + ref row."""
    seq = _c(seq, np.int32); bq = _c(bq, np.int32); mq = _c(mq, np.int32); hap = _c(hap, np.int32)
    ref_row = _c(ref_row, np.int32)
    D, L = seq.shape
    out = np.empty((105, L), np.float64)
    lib().orc_hap_features(_p(seq), _p(bq), _p(mq), _p(hap), _p(ref_row), D, L, _p(out))
    return out


def hap_features_batch(seq, bq, mq, hap, ref_row, nthreads=1):
    seq = _c(seq, np.int32); bq = _c(bq, np.int32); mq = _c(mq, np.int32); hap = _c(hap, np.int32)
    ref_row = _c(ref_row, np.int32)
    N, D, L = seq.shape
    out = np.empty((N, 105, L), np.float32)
    lib().orc_hap_features_batch(_p(seq), _p(bq), _p(mq), _p(hap), _p(ref_row), N, D, L, _p(out),
                                 nthreads)
    return out


def hap_forward(weights, xp, xh, H=256, n_layers=3, n_gt=10, n_zy=3, nthreads=1):
    ws, arr = _wptrs(weights)
    xp = _c(xp, np.float32); xh = _c(xh, np.float32)
    N, F, Lp = xp.shape
    Lh = xh.shape[2]
    gt = np.empty((N, n_gt), np.float32); zy = np.empty((N, n_zy), np.float32)
    lib().orc_hap_forward(arr, _p(xp), _p(xh), N, F, H, n_layers, Lp, Lh, n_gt, n_zy,
                          _p(gt), _p(zy), nthreads)
    return gt, zy


def cat_forward(weights, g0, g1, nthreads=1):
    """weights: the 132 float tensors of CatModel.state_dict() in order; g0, g1 [N,40,11,5]."""
    ws, arr = _wptrs(weights)
    assert len(ws) == 132
    g0 = _c(g0, np.float32); g1 = _c(g1, np.float32)
    N = g0.shape[0]
    gt = np.empty((N, 10), np.float32)
    lib().orc_cat_forward(arr, _p(g0), _p(g1), N, _p(gt), nthreads)
    return gt


def cat_groups(tag1, tag2):
    """tag1 / tag2: (read, baseq, mapq) int32 [N,D,L] -> [N,40,L,5] float32 (dataset.py:862-915)."""
    a = [_c(t, np.int32) for t in tag1]; b = [_c(t, np.int32) for t in tag2]
    N, D1, L = a[0].shape
    g = np.empty((N, 40, L, 5), np.float32)
    lib().orc_cat_groups(_p(a[0]), _p(a[1]), _p(a[2]), D1, _p(b[0]), _p(b[1]), _p(b[2]), b[0].shape[1], N, L, _p(g))
    return g


def calculate_score(p):
    return lib().orc_calculate_score(float(p))


def hap_arrange(seq, bq, mq, hap, d_out, rows=None):
    """One site: read matrices [R, L] -> padded / cut planes [d_out, L] and the kept depth."""
    seq = _c(seq, np.int32); bq = _c(bq, np.int32); mq = _c(mq, np.int32); hap = _c(hap, np.int32)
    R, L = seq.shape
    outs = [np.empty((d_out, L), np.int32) for _ in range(4)]
    depth = np.zeros(1, np.int32)
    lib().orc_hap_arrange(_p(seq), _p(bq), _p(mq), _p(hap), R if rows is None else int(rows), R, L, int(d_out),
                          *[_p(o) for o in outs], _p(depth))
    return outs[0], outs[1], outs[2], outs[3], int(depth[0])


def pileup_forward_f64(weights, x):
    """LSTMNetwork.predict of the PileupModel (PileupModel/model.py:31-39,66-73,114-119) evaluated in FLOAT64 with numpy: the same
    published equations as orc_pileup_forward (lstm_oracle.c), every product and sum in double.  Test infrastructure: the yardstick
    that separates arithmetic error from summation-order noise - |path - f64| of the fp32 MFMA kernels, of the bf16x3 / f16x3 modes
    and of the fp32 oracle itself are all measured against it (tests/test_gpu_pileup_forward.py, bench.py "error_vs_float64")."""
    w = [np.asarray(t, np.float64) for t in weights]
    x = np.asarray(x, np.float64)
    N, T, H = x.shape[0], 33, 64
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))

    def layer(inp, base):
        out = np.zeros((N, T, 2 * H))
        for d in range(2):
            wih, whh, b = w[base + 4 * d], w[base + 4 * d + 1], w[base + 4 * d + 2] + w[base + 4 * d + 3]
            h = np.zeros((N, H)); c = np.zeros((N, H))
            for s in range(T):
                t = T - 1 - s if d else s
                g = inp[:, t] @ wih.T + h @ whh.T + b
                i, f, gg, o = sig(g[:, :H]), sig(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), sig(g[:, 3 * H:])
                c = f * c + i * gg
                h = o * np.tanh(c)
                out[:, t, d * H:(d + 1) * H] = h
        return out

    h1 = layer(layer(x, 0), 8)
    enc = h1[:, 16] @ w[16].T + w[17]
    mid = np.tanh(enc @ w[18].T + w[19])

    def softmax(z):
        e = np.exp(z - z.max(1, keepdims=True))
        return e / e.sum(1, keepdims=True)
    return softmax(mid @ w[20].T + w[21]), softmax(mid @ w[22].T + w[23])
