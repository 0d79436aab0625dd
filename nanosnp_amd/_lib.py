"""ctypes loader of ``libnanosnp_hip.so`` -- the C ABI declared in include/nanosnp.h.

There is no CPU fallback: importing works everywhere (so the symbol table can be checked on a
machine without a GPU), but creating a :class:`Context` requires the built library and a
gfx950 device and fails loudly otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product always loads the in-tree build.  A/B builds of the same library (tools/build_variant.sh) are loaded only when BOTH
# NANOSNP_DEV_LIB_OVERRIDE=1 and NANOSNP_HIP_LIB=<path> are set: an environment variable alone cannot point the package at an
# arbitrary shared object, and the override says so on stderr.
_OVERRIDE = os.environ.get("NANOSNP_HIP_LIB") if os.environ.get("NANOSNP_DEV_LIB_OVERRIDE") == "1" else None
LIB_PATH = _OVERRIDE or os.path.join(_HERE, "libnanosnp_hip.so")


class NanoSNPError(RuntimeError):
    pass


_SIGNATURES = {
    # name: (restype, argtypes)
    "nsnp_version": (C.c_int, []),
    "nsnp_strerror": (C.c_char_p, [C.c_int]),
    "nsnp_last_hip_error": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p)]),
    "nsnp_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nsnp_ctx_destroy": (C.c_int, [C.c_void_p]),
    "nsnp_ctx_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "nsnp_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "nsnp_ctx_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "nsnp_ctx_read_timing": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "nsnp_ctx_shader_clock": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "nsnp_pileup_load_weights": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]),
    "nsnp_pileup_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsnp_pileup_forward_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "nsnp_pileup_forward_windows_calls": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 7),
    "nsnp_pileup_rows_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 7),
    "nsnp_pileup_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "nsnp_pileup_encode_columns": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                             C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "nsnp_pileup_encode_columns2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                              C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "nsnp_pileup_select_sites": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p]),
    "nsnp_pileup_gather_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p]),
    "nsnp_pileup_select_sites_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                                 C.c_void_p]),
    "nsnp_pileup_call_rows": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_void_p, C.c_void_p]),
    "nsnp_mpileup_tokenise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 6),
    "nsnp_hap_features": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "nsnp_hap_features_i8": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "nsnp_hap_arrange_reads": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6),
    "nsnp_hap_load_weights": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int] + [C.c_int] * 5),
    "nsnp_hap_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "nsnp_cat_load_weights": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]),
    "nsnp_cat_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "nsnp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "nsnp_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "nsnp_comm_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "nsnp_comm_destroy": (C.c_int, [C.c_void_p]),
    "nsnp_gather_check": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int]),
    "nsnp_gather_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "nsnp_cat_groups": (C.c_int, [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                        [C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)


def tokenise_status_check(status, what="mpileup text"):
    """status word of nsnp_mpileup_tokenise -> ValueError for text the reference's reader aborts on"""
    if status & 1:
        raise ValueError(f"{what}: a line with fewer than five tab-separated fields")
    if status & 2:
        raise ValueError(f"{what}: an empty line")
    if status & 4:
        raise ValueError(f"{what}: position outside the reference sequence")
    if status & 8:
        raise NanoSNPError(f"{what}: tokeniser output buffers too small")

_lib = None


def load():
    """Loads the shared library (no GPU needed for this) and binds every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NanoSNPError(
            f"{LIB_PATH} not found. The HIP extension is required (there is no CPU fallback): "
            "build it with `make -C nanosnp_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    # PyTorch-ROCm wheels bundle their own libamdhip64 (soname libamdhip64.so.7).  It has to be the
    # HIP runtime of this process, so it is loaded first; libnanosnp_hip.so's NEEDED entry then
    # resolves to it by soname.  Loading /opt/rocm's copy first would leave two HIP runtimes in one
    # process and every call from the second one fails.
    import torch  # noqa: F401
    if _OVERRIDE:
        import sys
        print(f"nanosnp_amd: DEVELOPMENT OVERRIDE - loading {LIB_PATH} instead of the in-tree library", file=sys.stderr)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, ctx=None, what=""):
    if rc == 0:
        return
    lib = load()
    msg = lib.nsnp_strerror(rc).decode()
    if rc == -3:
        txt = C.c_char_p()
        code = lib.nsnp_last_hip_error(ctx, C.byref(txt))
        msg += f" (hipError {code}: {txt.value.decode() if txt.value else '?'})"
    raise NanoSNPError(f"{what or 'nanosnp'} failed: {msg}")


def _stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Context:
    """One nsnp_ctx bound to one device (one per stream when batches overlap)."""

    def __init__(self, device=0, chunk_sites=None):
        import torch
        if not torch.cuda.is_available():
            raise NanoSNPError("no GPU visible: nanosnp_amd has no CPU fallback (the CPU restatement "
                               "under oracle/ is test infrastructure only)")
        self.lib = load()
        self.device = int(device)
        h = C.c_void_p()
        check(self.lib.nsnp_ctx_create(self.device, C.byref(h)), None, "nsnp_ctx_create")
        self.handle = h
        if chunk_sites:
            self.reserve(chunk_sites)

    def reserve(self, max_sites):
        check(self.lib.nsnp_ctx_reserve(self.handle, int(max_sites)), self.handle, "nsnp_ctx_reserve")

    def set_option(self, name, value):
        check(self.lib.nsnp_ctx_set_option(self.handle, name.encode(), int(value)), self.handle, f"nsnp_ctx_set_option({name})")

    # ids of nsnp_ctx_read_timing; the last three are ONE event pair around a chain of launches of a pass (include/nanosnp.h)
    KERNELS = ("pileup_l0", "pileup_proj1", "pileup_l1", "pileup_head", "encode_columns", "hap_features",
               "hap_lstm_chain", "cat_conv_chain", "cat_forward_pass")

    def enable_timing(self, enable=True):
        check(self.lib.nsnp_ctx_enable_timing(self.handle, int(bool(enable))), self.handle, "nsnp_ctx_enable_timing")

    def read_timing(self):
        """{kernel name: (total_ms, launches)} of the launches recorded since the last read."""
        out = {}
        for k, name in enumerate(self.KERNELS):
            ms, n = C.c_double(0), C.c_int64(0)
            check(self.lib.nsnp_ctx_read_timing(self.handle, k, C.byref(ms), C.byref(n)), self.handle,
                  "nsnp_ctx_read_timing")
            out[name] = (ms.value, n.value)
        return out

    def shader_clock_mhz(self, stream=None):
        """shader clock under a ~2 ms full-chip fp32 MFMA load (diagnostic; bench lines record it)"""
        mhz = C.c_double(0)
        check(self.lib.nsnp_ctx_shader_clock(self.handle, C.byref(mhz), _stream_ptr(stream)), self.handle, "nsnp_ctx_shader_clock")
        return mhz.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nsnp_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- optional RCCL gather of the C ABI (PyTorch ranks normally use nanosnp_amd.dist instead) ----
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        check(load().nsnp_comm_unique_id(buf), None, "nsnp_comm_unique_id")
        return bytes(buf)

    def comm_init(self, id128: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(id128)
        check(self.lib.nsnp_comm_init(self.handle, buf, int(rank), int(world)), self.handle, "nsnp_comm_init")
        self._comm = (int(rank), int(world))

    def gather_bytes(self, local, counts_bytes, root=0, stream=None):
        """local: contiguous cuda tensor of this rank; counts_bytes: bytes of every rank's block -> uint8 cuda tensor on root, else None"""
        import numpy as np
        import torch
        rank, world = self._comm
        off = np.concatenate([[0], np.cumsum(np.asarray(counts_bytes, np.int64))]).astype(np.int64)
        out = torch.empty(int(off[-1]), dtype=torch.uint8, device=local.device) if rank == root else None
        check(self.lib.nsnp_gather_results(self.handle, _dptr(local), int(local.numel() * local.element_size()), _dptr(out),
                                           off.ctypes.data_as(C.c_void_p), int(root), _stream_ptr(stream)), self.handle, "nsnp_gather_results")
        return out

    # ---- PileupModel -------------------------------------------------------------------------
    def pileup_load_weights(self, tensors):
        """tensors: 24 contiguous fp32 CPU arrays/tensors in state-dict order."""
        import numpy as np
        arrs = [np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
                for t in tensors]
        if len(arrs) < 24:
            raise NanoSNPError(f"expected 24 weight tensors, got {len(arrs)}")
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        check(self.lib.nsnp_pileup_load_weights(self.handle, ptrs, len(arrs)), self.handle,
              "nsnp_pileup_load_weights")

    def pileup_forward(self, x, gt=None, zy=None, stream=None):
        import torch
        assert x.is_cuda and x.dtype == torch.int32 and x.is_contiguous() and tuple(x.shape[1:]) == (33, 18)
        n = x.shape[0]
        gt = gt if gt is not None else torch.empty((n, 21), dtype=torch.float32, device=x.device)
        zy = zy if zy is not None else torch.empty((n, 3), dtype=torch.float32, device=x.device)
        check(self.lib.nsnp_pileup_forward(self.handle, _dptr(x), n, _dptr(gt), _dptr(zy), _stream_ptr(stream)),
              self.handle, "nsnp_pileup_forward")
        return gt, zy

    def pileup_forward_windows(self, counts, center_idx, gt=None, zy=None, stream=None):
        import torch
        assert counts.is_cuda and counts.dtype == torch.int32 and counts.is_contiguous()
        assert center_idx.is_cuda and center_idx.dtype == torch.int64 and center_idx.is_contiguous()
        n = center_idx.shape[0]
        gt = gt if gt is not None else torch.empty((n, 21), dtype=torch.float32, device=counts.device)
        zy = zy if zy is not None else torch.empty((n, 3), dtype=torch.float32, device=counts.device)
        check(self.lib.nsnp_pileup_forward_windows(self.handle, _dptr(counts), _dptr(center_idx), n, _dptr(gt),
                                                   _dptr(zy), _stream_ptr(stream)),
              self.handle, "nsnp_pileup_forward_windows")
        return gt, zy

    def pileup_forward_windows_calls(self, counts, center_idx, stream=None, calls_out=None):
        """forward + argmax / max of both heads in one call -> (gt, zy, gt_arg, zy_arg, gt_max, zy_max).  calls_out = (gt_arg uint8 [n],
        zy_arg uint8 [n], gt_max float32 [n], zy_max float32 [n]): device tensors, or PINNED host tensors - the heads kernel then writes
        the 10 bytes per site straight into host memory (hipHostMalloc memory is mapped on the device) and no D2H copy is needed; they
        are valid on the host once an event recorded behind this call has completed."""
        import torch
        n = center_idx.shape[0]
        dev = counts.device
        gt = torch.empty((n, 21), dtype=torch.float32, device=dev); zy = torch.empty((n, 3), dtype=torch.float32, device=dev)
        if calls_out is None:
            ga = torch.empty(n, dtype=torch.uint8, device=dev); za = torch.empty(n, dtype=torch.uint8, device=dev)
            gm = torch.empty(n, dtype=torch.float32, device=dev); zm = torch.empty(n, dtype=torch.float32, device=dev)
        else:
            ga, za, gm, zm = calls_out
            for t, dt in ((ga, torch.uint8), (za, torch.uint8), (gm, torch.float32), (zm, torch.float32)):
                if t.dtype != dt or t.numel() < n or not t.is_contiguous() or not (t.is_cuda or t.is_pinned()):
                    raise NanoSNPError("calls_out: contiguous uint8 / uint8 / float32 / float32 tensors of n elements, on the device or pinned")
        check(self.lib.nsnp_pileup_forward_windows_calls(self.handle, _dptr(counts), _dptr(center_idx), n, _dptr(gt), _dptr(zy), _dptr(ga),
                                                         _dptr(za), _dptr(gm), _dptr(zm), _stream_ptr(stream)),
              self.handle, "nsnp_pileup_forward_windows_calls")
        return gt, zy, ga, za, gm, zm

    def pileup_rows_unpack(self, rows, outs, stream=None):
        """rows: device float64 [n,13] (pipeline.stream_contig); outs = (pos int64 [n], gt_arg uint8, zy_arg uint8, gt_max float32, zy_max
        float32, cov float32 [n,8]): device or PINNED host tensors of at least n elements - written by the kernel itself, no D2H copy; valid
        on the host once the stream has been synchronised."""
        import torch
        n = int(rows.shape[0])
        if rows.dtype != torch.float64 or rows.dim() != 2 or rows.shape[1] != 13 or not rows.is_contiguous() or not rows.is_cuda:
            raise NanoSNPError("rows: a contiguous device float64 [n,13] tensor")
        for t, dt, k in zip(outs, (torch.int64, torch.uint8, torch.uint8, torch.float32, torch.float32, torch.float32), (1, 1, 1, 1, 1, 8)):
            if t.dtype != dt or t.numel() < n * k or not t.is_contiguous() or not (t.is_cuda or t.is_pinned()):
                raise NanoSNPError("outs: contiguous int64 / uint8 / uint8 / float32 / float32 / float32 [n,8] tensors, on the device or pinned")
        check(self.lib.nsnp_pileup_rows_unpack(self.handle, _dptr(rows), n, *[_dptr(t) for t in outs], _stream_ptr(stream)),
              self.handle, "nsnp_pileup_rows_unpack")

    def pileup_postprocess(self, gt, zy, x=None, stream=None):
        import torch
        n = gt.shape[0]
        dev = gt.device
        gt_arg = torch.empty(n, dtype=torch.uint8, device=dev)
        zy_arg = torch.empty(n, dtype=torch.uint8, device=dev)
        gt_max = torch.empty(n, dtype=torch.float32, device=dev)
        zy_max = torch.empty(n, dtype=torch.float32, device=dev)
        depth = torch.empty(n, dtype=torch.int32, device=dev) if x is not None else None
        check(self.lib.nsnp_pileup_postprocess(self.handle, _dptr(gt), _dptr(zy), _dptr(x), n, _dptr(gt_arg),
                                               _dptr(zy_arg), _dptr(gt_max), _dptr(zy_max), _dptr(depth),
                                               _stream_ptr(stream)),
              self.handle, "nsnp_pileup_postprocess")
        return gt_arg, zy_arg, gt_max, zy_max, depth

    # ---- pileup encode -----------------------------------------------------------------------
    def pileup_encode_columns(self, bases, col_off, ref, min_af=0.12, min_coverage=6, stream=None, indel_min_af=None):
        """min_af: the SNP threshold and, unless indel_min_af is given, the indel threshold too (DNA_CreateCanSnpTensor -snp_min_af /
        -indel_min_af; make_predict_data.sh passes 0.12 for both)"""
        import torch
        assert bases.is_cuda and bases.dtype == torch.uint8 and col_off.dtype == torch.int64 and ref.dtype == torch.uint8
        m = ref.shape[0]
        dev = ref.device
        counts = torch.empty((m, 18), dtype=torch.int32, device=dev)
        depth = torch.empty(m, dtype=torch.int32, device=dev)
        flags = torch.empty(m, dtype=torch.uint8, device=dev)
        check(self.lib.nsnp_pileup_encode_columns2(self.handle, _dptr(bases), _dptr(col_off), _dptr(ref), m,
                                                   float(min_af), float(min_af if indel_min_af is None else indel_min_af), int(min_coverage),
                                                   _dptr(counts), _dptr(depth), _dptr(flags), _stream_ptr(stream)),
              self.handle, "nsnp_pileup_encode_columns2")
        return counts, depth, flags

    def pileup_select_sites(self, pos, flags, cap=None, stream=None):
        import torch
        m = pos.shape[0]
        cap = int(cap if cap is not None else m)
        center = torch.empty(max(cap, 1), dtype=torch.int64, device=pos.device)
        n_sites = torch.zeros(1, dtype=torch.int64, device=pos.device)
        check(self.lib.nsnp_pileup_select_sites(self.handle, _dptr(pos), _dptr(flags), m, _dptr(center), cap,
                                                _dptr(n_sites), _stream_ptr(stream)),
              self.handle, "nsnp_pileup_select_sites")
        n = int(n_sites.item())
        return center[:min(n, cap)], n

    def pileup_select_sites_async(self, pos, flags, stream=None):
        """select_sites without the host round trip: -> (center_idx int64 [M], entries behind the selected ones = 2^62; n_sites: device
        int64 [1]).  The streamed pipeline reads the count through a pinned buffer one chunk later (nanosnp_amd/pipeline.py)."""
        import torch
        m = pos.shape[0]
        center = torch.full((max(m, 1),), 1 << 62, dtype=torch.int64, device=pos.device)
        n_sites = torch.zeros(1, dtype=torch.int64, device=pos.device)
        check(self.lib.nsnp_pileup_select_sites(self.handle, _dptr(pos), _dptr(flags), m, _dptr(center), m,
                                                _dptr(n_sites), _stream_ptr(stream)),
              self.handle, "nsnp_pileup_select_sites")
        return center, n_sites

    TOK_EFORMAT, TOK_BLANK, TOK_EPOS, TOK_ERANGE = 1, 2, 4, 8

    def mpileup_tokenise_into(self, text, chr_seq, pos, col_off, bases, ref, meta, stream=None):
        """nsnp_mpileup_tokenise into the caller's buffers, asynchronously: text uint8 [T] on the device; chr_seq uint8 on the device (or
        None, with ref None); pos int64 [cap], col_off int64 [cap + 1], bases uint8 [cap_bytes], ref uint8 [cap] on the device; meta int64 [4]
        on the device or in pinned host memory ({lines, bytes, status, 0} once the stream has passed the call)."""
        check(self.lib.nsnp_mpileup_tokenise(self.handle, _dptr(text), int(text.numel()), _dptr(chr_seq) if chr_seq is not None else None,
                                             int(chr_seq.numel()) if chr_seq is not None else 0, int(pos.numel()), int(bases.numel()),
                                             _dptr(pos), _dptr(col_off), _dptr(bases), _dptr(ref) if ref is not None else None,
                                             meta.data_ptr(), _stream_ptr(stream)),
              self.handle, "nsnp_mpileup_tokenise")

    def mpileup_tokenise(self, text, chr_seq=None, stream=None):
        """mpileup text (uint8 device tensor) -> (pos [M] int64, col_off [M + 1] int64, bases uint8, ref uint8 [M] or None) on the device.
        Synchronous (the sizes come back from the device); raises on text the reference's reader could not read."""
        import torch
        t = int(text.numel())
        cap, cap_b = t // 10 + 2, max(t, 1)
        while True:
            pos = torch.empty(cap, dtype=torch.int64, device=text.device)
            off = torch.empty(cap + 1, dtype=torch.int64, device=text.device)
            bases = torch.empty(cap_b, dtype=torch.uint8, device=text.device)
            ref = torch.empty(cap, dtype=torch.uint8, device=text.device) if chr_seq is not None else None
            meta = torch.zeros(4, dtype=torch.int64, device=text.device)
            self.mpileup_tokenise_into(text, chr_seq, pos, off, bases, ref, meta, stream)
            m, nb, status, _ = meta.tolist()
            if status & self.TOK_ERANGE and not status & (self.TOK_EFORMAT | self.TOK_BLANK):
                cap, cap_b = max(cap, m + 1), max(cap_b, nb)
                continue
            break
        tokenise_status_check(status)
        return pos[:m], off[:m + 1], bases[:nb], (ref[:m] if ref is not None else None)

    def pileup_select_sites_range(self, pos, flags, own_lo, own_hi, meta, stream=None):
        """select_sites for one chunk of a streamed text, without a host round trip: -> center_idx int64 [M] (the first meta[0] entries are the
        selected centres, ascending); meta (int64 [4], on the device or pinned) receives {n, c_lo, c_hi, n}: the chunk's own sites - centres in
        [own_lo, own_hi) - are center_idx[c_lo:c_hi]."""
        import torch
        m = pos.shape[0]
        center = torch.empty(max(m, 1), dtype=torch.int64, device=pos.device)
        check(self.lib.nsnp_pileup_select_sites_range(self.handle, _dptr(pos), _dptr(flags), m, int(own_lo), int(own_hi), _dptr(center), m,
                                                      meta.data_ptr(), _stream_ptr(stream)),
              self.handle, "nsnp_pileup_select_sites_range")
        return center

    def pileup_call_rows(self, counts, center_idx, pos, gt_arg, zy_arg, gt_max, zy_max, stream=None):
        """-> the call rows [n,13] float64 of the text pipeline (position, argmax / max of both heads, the eight coverage channels of
        predict.py:63 at the centre column) in one launch"""
        import torch
        n = int(center_idx.shape[0])
        rows = torch.empty((n, 13), dtype=torch.float64, device=counts.device)
        check(self.lib.nsnp_pileup_call_rows(self.handle, _dptr(counts), _dptr(center_idx), _dptr(pos), _dptr(gt_arg), _dptr(zy_arg), _dptr(gt_max),
                                             _dptr(zy_max), n, _dptr(rows), _stream_ptr(stream)),
              self.handle, "nsnp_pileup_call_rows")
        return rows

    def pileup_gather_windows(self, counts, center_idx, stream=None):
        import torch
        n = center_idx.shape[0]
        x = torch.empty((n, 33, 18), dtype=torch.int32, device=counts.device)
        check(self.lib.nsnp_pileup_gather_windows(self.handle, _dptr(counts), _dptr(center_idx), n, _dptr(x),
                                                  _stream_ptr(stream)),
              self.handle, "nsnp_pileup_gather_windows")
        return x

    # ---- HaplotypeModel ----------------------------------------------------------------------
    def hap_features(self, seq, bq, mq, hap, ref_row, stream=None):
        """int32 planes (the reference's bins) or int8 planes (nsnp_hap_features_i8: a quarter of the bytes); ref_row int32"""
        import torch
        n, d, l = seq.shape
        out = torch.empty((n, 105, l), dtype=torch.float32, device=seq.device)
        if seq.dtype == torch.int8:
            if not (bq.dtype == mq.dtype == hap.dtype == torch.int8):
                raise NanoSNPError("hap_features: the four read planes must share one dtype (int32 or int8)")
            if ref_row.dtype != torch.int32:
                raise NanoSNPError("hap_features: ref_row must be int32")
            check(self.lib.nsnp_hap_features_i8(self.handle, _dptr(seq), _dptr(bq), _dptr(mq), _dptr(hap), _dptr(ref_row),
                                                n, d, l, _dptr(out), _stream_ptr(stream)),
                  self.handle, "nsnp_hap_features_i8")
            return out
        check(self.lib.nsnp_hap_features(self.handle, _dptr(seq), _dptr(bq), _dptr(mq), _dptr(hap), _dptr(ref_row),
                                         n, d, l, _dptr(out), _stream_ptr(stream)),
              self.handle, "nsnp_hap_features")
        return out

    def hap_arrange_reads(self, seq, bq, mq, hap, d_out, n_reads=None, stream=None):
        import torch
        n, r, l = seq.shape
        outs = [torch.empty((n, d_out, l), dtype=torch.int32, device=seq.device) for _ in range(4)]
        depth = torch.empty(n, dtype=torch.int32, device=seq.device)
        check(self.lib.nsnp_hap_arrange_reads(self.handle, _dptr(seq), _dptr(bq), _dptr(mq), _dptr(hap), _dptr(n_reads),
                                              n, r, l, int(d_out), *[_dptr(o) for o in outs], _dptr(depth),
                                              _stream_ptr(stream)),
              self.handle, "nsnp_hap_arrange_reads")
        return outs[0], outs[1], outs[2], outs[3], depth

    def hap_load_weights(self, tensors, n_features=105, hidden=256, n_layers=3, n_gt=10, n_zy=3):
        import numpy as np
        arrs = [np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
                for t in tensors]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        check(self.lib.nsnp_hap_load_weights(self.handle, ptrs, len(arrs), n_features, hidden, n_layers, n_gt, n_zy),
              self.handle, "nsnp_hap_load_weights")
        self._hap_dims = (n_gt, n_zy)

    # ---- legacy CatModel (HaplotypeModel/predict.py -> model.CatModel) ----
    def cat_load_weights(self, tensors):
        """tensors: the 132 floating-point tensors of CatModel.state_dict() in order (num_batches_tracked skipped)."""
        import numpy as np
        arrs = [np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
                for t in tensors]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        check(self.lib.nsnp_cat_load_weights(self.handle, ptrs, len(arrs)), self.handle, "nsnp_cat_load_weights")

    def cat_forward(self, g0, g1, stream=None):
        """g0, g1: float32 cuda [N,40,11,5] -> softmax probabilities [N,10] (model.py:332-358)."""
        import torch
        if tuple(g0.shape[1:]) != (40, 11, 5) or g0.shape != g1.shape:
            raise NanoSNPError(f"cat_forward: g0/g1 must be [N,40,11,5], got {tuple(g0.shape)} / {tuple(g1.shape)}")
        g0 = g0.contiguous().float(); g1 = g1.contiguous().float()
        n = g0.shape[0]
        gt = torch.empty((n, 10), dtype=torch.float32, device=g0.device)
        check(self.lib.nsnp_cat_forward(self.handle, _dptr(g0), _dptr(g1), n, _dptr(gt), _stream_ptr(stream)),
              self.handle, "nsnp_cat_forward")
        return gt

    def cat_groups(self, tag1, tag2, stream=None):
        """tag1 / tag2: (read, baseq, mapq) int32 cuda [N,depth,L] per tag -> [N,40,L,5] float32 (dataset.py:862-915)."""
        import torch
        r1, q1, m1 = [t.contiguous() for t in tag1]
        r2, q2, m2 = [t.contiguous() for t in tag2]
        n, d1, l = r1.shape
        d2 = r2.shape[1]
        g = torch.empty((n, 40, l, 5), dtype=torch.float32, device=r1.device)
        check(self.lib.nsnp_cat_groups(self.handle, _dptr(r1), _dptr(q1), _dptr(m1), d1, _dptr(r2), _dptr(q2), _dptr(m2), d2,
                                       n, l, _dptr(g), _stream_ptr(stream)), self.handle, "nsnp_cat_groups")
        return g

    def hap_forward(self, xp, xh, stream=None):
        import torch
        n = xp.shape[0]
        n_gt, n_zy = self._hap_dims
        gt = torch.empty((n, n_gt), dtype=torch.float32, device=xp.device)
        zy = torch.empty((n, n_zy), dtype=torch.float32, device=xp.device)
        check(self.lib.nsnp_hap_forward(self.handle, _dptr(xp), _dptr(xh), n, _dptr(gt), _dptr(zy),
                                        _stream_ptr(stream)),
              self.handle, "nsnp_hap_forward")
        return gt, zy
