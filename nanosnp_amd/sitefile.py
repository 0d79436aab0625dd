"""Flat binary containers that stand in for the reference's PyTables/HDF5 ``.bin`` files.

The reference moves sites between stages as HDF5 EArrays:

* ``<chr>.pd.bin``  (``dna_sv_tensor/src/make_bin_data/make_bin_predict_data.py:48-100``): ``position_matrix int32 [N,33,18]``,
  ``position S83 [N,1]`` (``"ctg:pos:ref33"``), ``alt_info S5000 [N,1]``; read by ``PileupModel/dataset.py:118-139``.
* ``haplotype_bins/<ctg>_<s>_<e>.bin`` (``HaplotypeModel/write_to_bins.py:4-64``): eight ``int32 [N,D,L]`` read planes and two
  string arrays; read by ``HaplotypeModel/dataset_dev.py:95-105,135-147``.

PyTables is not part of this stack and a compressed chunked store is the wrong shape for a device pipeline anyway, so the same
arrays (same names, same dtypes, same row order) are kept in one uncompressed file whose arrays start on 64-byte boundaries:
``numpy.memmap`` views go to the GPU with one copy, no parse.  Strings are fixed-width byte rows exactly like the HDF5 atoms
(``alt_info`` is a ragged blob + offsets instead of 5000 bytes per site).

Layout: ``b"NSNPBIN1"``, ``u32 n_arrays``, ``u32 0``; per array a 96-byte record ``name[32] dtype[8] ndim(u32) pad(u32)
shape[4](u64) offset(u64) nbytes(u64)``; then the data.
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"NSNPBIN1"
_REC = struct.Struct("<32s8sII4QQQ")
POSITION_WIDTH = 33 + 50          # make_bin_predict_data.py:94  StringAtom(itemsize = no_of_positions + 50)


class SiteFileError(ValueError):
    pass


def write_arrays(path, arrays: dict):
    """Writes named numpy arrays (<= 4 dimensions) into one flat file."""
    items = []
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        if a.ndim > 4 or len(name.encode()) > 32:
            raise SiteFileError(f"array {name!r}: at most 4 dimensions and 32-byte names")
        items.append((name, a))
    off = len(MAGIC) + 8 + _REC.size * len(items)
    recs = []
    for name, a in items:
        off = (off + 63) & ~63
        shape = list(a.shape) + [0] * (4 - a.ndim)
        recs.append(_REC.pack(name.encode(), a.dtype.str.encode(), a.ndim, 0, *shape, off, a.nbytes))
        off += a.nbytes
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<II", len(items), 0))
        for r in recs:
            f.write(r)
        for (name, a), r in zip(items, recs):
            o = _REC.unpack(r)[8]
            f.write(b"\0" * (o - f.tell()))
            f.write(a.tobytes())


def create_arrays(path, specs: dict) -> dict:
    """Lays out a site file for arrays that are filled piece by piece (the reference appends to its EArrays chunk by chunk:
    write_to_bins.py:53-63): specs = {name: (dtype, shape)}; returns {name: writable numpy.memmap}.  Flush / delete the maps to finish."""
    items = []
    for name, (dt, shape) in specs.items():
        dt = np.dtype(dt)
        if len(shape) > 4 or len(name.encode()) > 32:
            raise SiteFileError(f"array {name!r}: at most 4 dimensions and 32-byte names")
        items.append((name, dt, tuple(int(v) for v in shape)))
    off = len(MAGIC) + 8 + _REC.size * len(items)
    recs, offs = [], []
    for name, dt, shape in items:
        off = (off + 63) & ~63
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        recs.append(_REC.pack(name.encode(), dt.str.encode(), len(shape), 0, *(list(shape) + [0] * (4 - len(shape))), off, nbytes))
        offs.append(off)
        off += nbytes
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<II", len(items), 0))
        for r in recs:
            f.write(r)
        f.truncate(max(off, f.tell()))
    return {name: (np.memmap(path, dtype=dt, mode="r+", offset=o, shape=shape) if int(np.prod(shape, dtype=np.int64)) else np.empty(shape, dt))
            for (name, dt, shape), o in zip(items, offs)}


def array_index(path) -> dict:
    """{name: (dtype, shape, byte offset, nbytes)} of a site file, without touching the data (the streamed stage-5 reader preads the
    planes straight into pinned buffers: nanosnp_amd/pipeline.py)."""
    with open(path, "rb") as f:
        head = f.read(len(MAGIC) + 8)
        if len(head) < len(MAGIC) + 8 or head[:8] != MAGIC:
            raise SiteFileError(f"{path}: not an NSNPBIN1 file")
        n, = struct.unpack_from("<I", head, 8)
        recs = [_REC.unpack(f.read(_REC.size)) for _ in range(n)]
    out = {}
    for name, dt, ndim, _, s0, s1, s2, s3, off, nbytes in recs:
        shape = (s0, s1, s2, s3)[:ndim]
        dtype = np.dtype(dt.rstrip(b"\0").decode())
        if int(np.prod(shape, dtype=np.int64)) * dtype.itemsize != nbytes:
            raise SiteFileError(f"{path}: array {name!r} has inconsistent size")
        out[name.rstrip(b"\0").decode()] = (dtype, tuple(int(v) for v in shape), int(off), int(nbytes))
    return out


def read_arrays(path, mmap=True) -> dict:
    """Returns {name: array}; with ``mmap`` the arrays are read-only views of the file."""
    with open(path, "rb") as f:
        head = f.read(len(MAGIC) + 8)
        if len(head) < len(MAGIC) + 8 or head[:8] != MAGIC:
            raise SiteFileError(f"{path}: not an NSNPBIN1 file")
        n, = struct.unpack_from("<I", head, 8)
        recs = [_REC.unpack(f.read(_REC.size)) for _ in range(n)]
    out = {}
    for name, dt, ndim, _, s0, s1, s2, s3, off, nbytes in recs:
        shape = (s0, s1, s2, s3)[:ndim]
        dtype = np.dtype(dt.rstrip(b"\0").decode())
        if int(np.prod(shape, dtype=np.int64)) * dtype.itemsize != nbytes:
            raise SiteFileError(f"{path}: array {name!r} has inconsistent size")
        key = name.rstrip(b"\0").decode()
        if nbytes == 0:
            out[key] = np.empty(shape, dtype)
        elif mmap:
            out[key] = np.memmap(path, dtype=dtype, mode="r", offset=off, shape=shape)
        else:
            out[key] = np.fromfile(path, dtype=dtype, count=int(np.prod(shape)), offset=off).reshape(shape)
    return out


# ---- <chr>.pd.bin ------------------------------------------------------------------------------------------------
def write_pileup_bin(path, position_matrix, position, alt_info=None, matrix_dtype="int16"):
    """position_matrix integer [N,33,18]; position: N strings ``ctg:pos:ref33``; alt_info: N strings or None.
    Row order is the caller's (the reference appends in input order: make_bin_predict_data.py:59-77).
    matrix_dtype "int16" (default): the counts are stored as int16 when every one of them fits (a count is at most the read depth, and
    the column encoder caps the depth at 144 reads: they always do), else as int32, the reference's Int32Atom (make_bin_predict_data.py:
    92) - half the file, half the page-cache bytes the streamed predict loop stages and no narrowing on the way; "int32" keeps int32."""
    x = np.ascontiguousarray(position_matrix)
    if x.dtype.kind not in "iu":
        raise SiteFileError("position_matrix must hold integers")
    if matrix_dtype not in ("int16", "int32"):
        raise SiteFileError("matrix_dtype: 'int16' or 'int32'")
    n = x.shape[0]
    if x.shape[1:] != (33, 18) or len(position) != n:
        raise SiteFileError("position_matrix must be [N,33,18] with one position string per row")
    fits16 = x.dtype.itemsize <= 2 and x.dtype != np.uint16 or (x.size == 0 or (int(x.min()) >= -32768 and int(x.max()) <= 32767))
    x = np.ascontiguousarray(x, dtype=np.int16 if (matrix_dtype == "int16" and fits16) else np.int32)
    pos = np.zeros((n, POSITION_WIDTH), np.uint8)
    for i, p in enumerate(position):
        b = p.encode() if isinstance(p, str) else bytes(p)
        if len(b) > POSITION_WIDTH:
            raise SiteFileError(f"position string longer than {POSITION_WIDTH} bytes: {b[:40]!r}...")
        pos[i, :len(b)] = np.frombuffer(b, np.uint8)
    arrays = {"position_matrix": x, "position": pos}
    if alt_info is not None:
        blobs = [a.encode() if isinstance(a, str) else bytes(a) for a in alt_info]
        if len(blobs) != n:
            raise SiteFileError("alt_info needs one entry per site")
        offs = np.zeros(n + 1, np.int64)
        np.cumsum([len(b) for b in blobs], out=offs[1:])
        arrays["alt_info"] = np.frombuffer(b"".join(blobs), np.uint8) if offs[-1] else np.empty(0, np.uint8)
        arrays["alt_info_offsets"] = offs
    write_arrays(path, arrays)


def read_pileup_bin(path, mmap=True):
    """-> (contig_names list[str], positions int64[N], reference_bases uint8[N], position_matrix int32[N,33,18]):
    what ``PileupModel/dataset.py:118-139`` extracts (``reference_bases = ord(seq[16])``)."""
    a = read_arrays(path, mmap)
    if "position_matrix" not in a or "position" not in a:
        raise SiteFileError(f"{path}: not a pileup site file")
    names, pos, refb = [], [], []
    for row in np.asarray(a["position"]):
        s = bytes(row).rstrip(b"\0").decode().strip()
        try:
            ctg, p, seq = s.split(":")
            names.append(ctg); pos.append(int(p)); refb.append(ord(seq[16]))
        except (ValueError, IndexError) as e:
            raise SiteFileError(f"{path}: bad position string {s!r}") from e
    x = a["position_matrix"]
    return names, np.asarray(pos, np.int64), np.asarray(refb, np.uint8), (x if x.dtype == np.int32 else np.asarray(x, np.int32))


def read_alt_info(path):
    a = read_arrays(path, mmap=False)
    if "alt_info" not in a:
        return None
    blob, offs = a["alt_info"].tobytes(), a["alt_info_offsets"]
    return [blob[offs[i]:offs[i + 1]].decode() for i in range(len(offs) - 1)]


def pd_to_bin(pd_text: bytes, path, matrix_dtype="int16"):
    """``.pd`` text -> site file: ``transform_one_input`` of make_bin_predict_data.py:48-77 (594 ints, position string,
    alt_info per line).  Lines whose tensor field does not hold 594 integers are rejected, as ``np.array(...).reshape`` would."""
    xs, positions, alts = [], [], []
    for ln, line in enumerate(pd_text.split(b"\n")):
        if not line.strip():
            continue
        fields = line.split(b"\t")
        if len(fields) < 3:
            raise SiteFileError(f".pd line {ln + 1}: expected tensor, position and alt_info fields")
        vals = np.array(fields[0].split(), dtype=np.int64)
        if vals.size != 33 * 18:
            raise SiteFileError(f".pd line {ln + 1}: {vals.size} tensor values, expected 594")
        xs.append(vals.astype(np.int32).reshape(33, 18))
        positions.append(fields[1].strip())
        alts.append(fields[2].strip())
    x = np.stack(xs) if xs else np.empty((0, 33, 18), np.int32)
    write_pileup_bin(path, x, positions, alts, matrix_dtype=matrix_dtype)
    return len(xs)


# ---- haplotype bins ---------------------------------------------------------------------------------------------------
HAP_PLANES = ("haplotype_sequences", "haplotype_hap", "haplotype_baseq", "haplotype_mapq",
              "pileup_sequences", "pileup_hap", "pileup_baseq", "pileup_mapq")


def write_haplotype_bin(path, candidate_positions, haplotype_positions, planes: dict, max_haplotype_depth=None,
                        max_pileup_depth=None, plane_dtype="int8"):
    """planes: the eight padded integer arrays of write_to_bins.py (``haplotype_*`` [N,Dh,11], ``pileup_*`` [N,Dp,33], padding
    -2).  Sites are sorted by the integer position of ``ctg:pos`` (write_to_bins.py:5-8; stable here) and depth is cut to
    ``max_*_depth`` (:39-42,54-61).
    plane_dtype: "int8" (default) stores the read planes as int8 when EVERY value of all eight fits (base codes -2..4, HP -2..3, base
    qualities <= 93, mapping qualities <= 60 from minimap2: they do; a plane holding e.g. mapq 255 makes the whole file int32) - a
    quarter of the file, of the bytes over PCIe and of the feature kernel's HBM reads, same features bit for bit
    (nsnp_hap_features_i8); "int32" writes the reference's dtype (write_to_bins.py:44-47).  Readers accept both."""
    cand = [c if isinstance(c, str) else bytes(c).decode() for c in candidate_positions]
    n = len(cand)
    order = np.argsort(np.array([int(c.split(":")[1]) for c in cand], np.int64), kind="stable") if n else np.empty(0, np.int64)
    if plane_dtype not in ("int8", "int32"):
        raise SiteFileError("plane_dtype must be 'int8' or 'int32'")
    arrays = {}
    for name in HAP_PLANES:
        a = np.asarray(planes[name])
        if a.ndim != 3 or a.shape[0] != n or a.dtype.kind != "i":
            raise SiteFileError(f"{name}: expected an integer array [N,D,L] with N = {n}")
        cut = max_haplotype_depth if name.startswith("haplotype") else max_pileup_depth
        if cut is not None and cut < a.shape[1]:
            a = a[:, :cut]
        arrays[name] = a[order]
    fits = all(a.size == 0 or (int(a.min()) >= -128 and int(a.max()) <= 127) for a in arrays.values())
    dt = np.int8 if (plane_dtype == "int8" and fits) else np.int32
    arrays = {k: np.ascontiguousarray(a, dtype=dt) for k, a in arrays.items()}
    width_c = max([len(c) for c in cand] + [1])
    cp = np.zeros((n, width_c), np.uint8)
    for i, j in enumerate(order):
        b = cand[j].encode(); cp[i, :len(b)] = np.frombuffer(b, np.uint8)
    arrays["candidate_positions"] = cp
    hp = [[p if isinstance(p, str) else bytes(p).decode() for p in row] for row in haplotype_positions]
    width_h = max([len(p) for row in hp for p in row] + [1])
    L = len(hp[0]) if hp else 0
    hpa = np.zeros((n, L, width_h), np.uint8)
    for i, j in enumerate(order):
        for k, p in enumerate(hp[j]):
            b = p.encode(); hpa[i, k, :len(b)] = np.frombuffer(b, np.uint8)
    arrays["haplotype_positions"] = hpa
    write_arrays(path, arrays)


def read_haplotype_bin(path, mmap=True):
    """-> (candidate_positions list[str], haplotype_positions list[list[str]], planes dict) as dataset_dev.py:95-105,135-147 reads them."""
    a = read_arrays(path, mmap)
    missing = [k for k in HAP_PLANES + ("candidate_positions", "haplotype_positions") if k not in a]
    if missing:
        raise SiteFileError(f"{path}: not a haplotype bin (missing {missing})")
    dec = lambda row: bytes(row).rstrip(b"\0").decode()
    cand = [dec(r) for r in np.asarray(a["candidate_positions"])]
    hpos = [[dec(p) for p in row] for row in np.asarray(a["haplotype_positions"])]
    return cand, hpos, {k: a[k] for k in HAP_PLANES}
