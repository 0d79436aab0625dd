#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): the legacy HaplotypeModel/dataset.py PredictDataset.__getitem__ (:809-925) over
a stand-in for the two tag files it opens, with every site present in both and no depth filter - its group tensors g0 (surrounding
columns) and g1 (adjacent sites) [N, 40, L, 5] against oracle.cat_groups (what nsnp_cat_groups is held to) on the same matrices:
first max_depth = 20 rows per tag, (base, baseq, mapq, mask = base != -2, phase = 1 / 2).  The position merge of the two files and
the min_depth filter of :826-848 are outside the hot path and not rebuilt.
    python tests/manual/ref_fuzz/cat_dataset.py [N_SEEDS]"""
import os, sys, types, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
files = {}
tb = types.ModuleType("tables"); tb.Filters = lambda **k: None; tb.set_blosc_max_threads = lambda n: None
tb.open_file = lambda path, mode="r": types.SimpleNamespace(root=files[os.path.basename(os.path.dirname(path))], close=lambda: None)
sys.modules["tables"] = tb
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
sys.path.insert(0, "/root/reference/HaplotypeModel")
import dataset as ref_dataset
from oracle import oracle
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([1, 7, 60])); A = 11
    pos = np.array([[f"chr1:{p}".encode()] for p in np.sort(rng.choice(np.arange(1, 100000), N, replace=False))], dtype="S40")
    def tag():
        D = int(rng.choice([20, 21, 33, 60]))
        def planes(L):
            r = rng.integers(-2, 5, (N, D, L)).astype(np.int32)
            depth = rng.integers(0, D + 1, N)
            for i in range(N): r[i, depth[i]:] = -2
            return r, np.where(r > 0, rng.integers(0, 94, (N, D, L)), 0).astype(np.int32), np.where(r != -2, rng.integers(0, 61, (N, D, L)), 0).astype(np.int32)
        s = planes(A); rd = planes(A)
        root = types.SimpleNamespace(position=pos, surrounding_read_matrix=s[0], surrounding_base_quality_matrix=s[1], surrounding_mapping_quality_matrix=s[2],
                                     read_matrix=rd[0], base_quality_matrix=rd[1], mapping_quality_matrix=rd[2],
                                     edge_matrix=np.zeros((N, 25, A - 1), np.int32), pair_route=np.zeros((N, 25, A - 1), np.int32))
        return root, s, rd
    files["t1"], s1, r1 = tag(); files["t2"], s2, r2 = tag()
    with tempfile.TemporaryDirectory() as d:
        for t in ("t1", "t2"):
            os.makedirs(os.path.join(d, t)); open(os.path.join(d, t, "a.bin"), "w").close()
        ds = ref_dataset.PredictDataset(os.path.join(d, "t1"), os.path.join(d, "t2"), max_depth=20, min_depth=0)
        position, g0, g1, g2, g3 = ds[0]
    o0 = oracle.cat_groups(s1, s2); o1 = oracle.cat_groups(r1, r2)
    ok = len(position) == N and g0.shape == o0.shape and np.array_equal(g0.astype(np.float32), o0) and np.array_equal(g1.astype(np.float32), o1)
    bad += not ok
    print(seed, "N", N, "depths", s1[0].shape[1], s2[0].shape[1], "g0 / g1", "identical" if ok else "DIFFER", flush=True)
print("bad", bad)
