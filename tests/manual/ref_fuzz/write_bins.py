#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): HaplotypeModel/write_to_bins.py with a recording stand-in for PyTables (what is
appended to which EArray) against sitefile.write_haplotype_bin + read_haplotype_bin on random chunks - ragged depths padded to the
chunk maximum, depth limits below / at / above it, unsorted (distinct) positions, one site, int8 and int32 storage.
    python tests/manual/ref_fuzz/write_bins.py FIRST_SEED END_SEED"""
import os, sys, types, tempfile, argparse, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
rec = {}
class _EA:
    def __init__(self, name): self.name = name
    def append(self, a): rec[self.name] = np.array(a)
class _Root: pass
class _File:
    def __init__(self): self.root = _Root()
    def create_earray(self, where, name, atom, shape, filters=None): setattr(self.root, name, _EA(name)); rec[name + "_shape"] = tuple(shape)
    def close(self): pass
tb = types.ModuleType("tables")
tb.Filters = lambda **k: None
tb.open_file = lambda path, mode="r": (rec.__setitem__("path", path), _File())[1]
tb.Atom = types.SimpleNamespace(from_dtype=lambda d: d)
tb.StringAtom = lambda itemsize: ("S", itemsize)
sys.modules["tables"] = tb
sys.path.insert(0, "/root/reference/HaplotypeModel")
import write_to_bins as wtb
from nanosnp_amd import sitefile
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([1, 2, 17, 60]))
    pos = rng.choice(np.arange(1, 100000), N, replace=False)
    cand = [f"chr7:{p}" for p in pos]
    hpos = [[f"chr7:{int(q)}" for q in np.sort(rng.integers(1, 100000, 11))] for _ in range(N)]
    dh = rng.integers(1, 40, N); dp = rng.integers(1, 60, N)
    big = seed % 4 == 0
    def mat(d, L): return rng.integers(-1, 5, (d, L)).astype(np.int32), rng.integers(0, 4, (d, L)).astype(np.int32), rng.integers(0, 300 if big else 94, (d, L)).astype(np.int32), rng.integers(0, 61, (d, L)).astype(np.int32)
    H = [mat(int(d), 11) for d in dh]; P = [mat(int(d), 33) for d in dp]
    maxh, maxp = int(dh.max()), int(dp.max())
    lim_h = [None, maxh - 3, maxh, maxh + 5][seed % 4]; lim_p = [None, maxp, maxp - 7, maxp + 1][(seed // 2) % 4]
    if lim_h is not None and lim_h < 1: lim_h = 1
    if lim_p is not None and lim_p < 1: lim_p = 1
    rec.clear()
    args = argparse.Namespace(output="/nowhere", max_pileup_depth=lim_p, max_haplotype_depth=lim_h)
    with contextlib.redirect_stdout(io.StringIO()):
        wtb.write_to_bins(args, "chr7", 5, 16, list(cand), [list(r) for r in hpos], [h[0] for h in H], [h[1] for h in H], [h[2] for h in H], [h[3] for h in H],
                          [p[0] for p in P], [p[1] for p in P], [p[2] for p in P], [p[3] for p in P], maxh, maxp)
    # ours: pad to the chunk maximum with -2 (readmatrix.group_planes does this on the device), then the writer
    def pad(ms, k, D): return np.stack([np.pad(m[k], ((0, D - m[k].shape[0]), (0, 0)), constant_values=-2) for m in ms])
    planes = {"haplotype_sequences": pad(H, 0, maxh), "haplotype_hap": pad(H, 1, maxh), "haplotype_baseq": pad(H, 2, maxh), "haplotype_mapq": pad(H, 3, maxh),
              "pileup_sequences": pad(P, 0, maxp), "pileup_hap": pad(P, 1, maxp), "pileup_baseq": pad(P, 2, maxp), "pileup_mapq": pad(P, 3, maxp)}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "x.bin")
        sitefile.write_haplotype_bin(path, cand, hpos, planes, max_haplotype_depth=lim_h, max_pileup_depth=lim_p, plane_dtype="int8" if seed % 2 else "int32")
        c2, h2, pl2 = sitefile.read_haplotype_bin(path, mmap=False)
    ok = [x.decode() if isinstance(x, bytes) else str(x) for x in rec["candidate_positions"].reshape(-1)] == list(c2)
    ok = ok and [[(x.decode() if isinstance(x, bytes) else str(x)) for x in row] for row in rec["haplotype_positions"]] == [list(r) for r in h2]
    for k in sitefile.HAP_PLANES:
        ok = ok and rec[k].shape == pl2[k].shape and np.array_equal(rec[k], np.asarray(pl2[k], np.int64))
    want_name = os.path.basename(rec["path"])
    bad += not ok
    print(seed, "N", N, "limits", lim_h, lim_p, "file", want_name, "stored", pl2["pileup_baseq"].dtype, "identical" if ok else "DIFFER", flush=True)
print("bad", bad)
