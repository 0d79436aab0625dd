#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): PileupModel/dataset.py PredictDataset (a stand-in for the PyTables file it opens:
root.position_matrix, root.position [N, 1] S83) on window files with ordinary and odd position strings - blanks around the fields,
signs, leading zeros, underscores, lower-case / N centre bases, long contig names - against sitefile.read_pileup_bin and the native
nsnp_parse_ctg_pos_ref (what pipeline.predict_pileup_bins uses); and the strings that make the reference raise, against HostError /
SiteFileError.
    python tests/manual/ref_fuzz/predict_dataset.py [N_SEEDS]"""
import os, sys, types, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
cur = {}
tb = types.ModuleType("tables"); tb.Filters = lambda **k: None
tb.open_file = lambda path, mode="r": types.SimpleNamespace(root=types.SimpleNamespace(position_matrix=cur["x"], position=cur["p"]), close=lambda: None)
sys.modules["tables"] = tb
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
sys.path.insert(0, "/root/reference/PileupModel")
import dataset as ref_dataset
from nanosnp_amd import host, sitefile
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(seed)
    n = 400
    x = rng.integers(-50, 50, (n, 33, 18)).astype(np.int32)
    strs = []
    for i in range(n):
        ctg = str(rng.choice(["chr1", "chrUn_KI270742v1", "HLA-DRB1*15", "c", "scaffold_" + "x" * 30]))
        p = int(rng.integers(1, 10 ** int(rng.integers(1, 9))))
        ps = str(rng.choice([str(p), f" {p}", f"{p} ", f"+{p}", f"000{p}", f"{p:,}".replace(",", "_")]))
        seq = "".join(rng.choice(list("ACGTacgtN"), 33))
        strs.append(f"{ctg}:{ps}:{seq}" if rng.random() > 0.1 else f"  {ctg}:{ps}:{seq} ")
    strs = [s for s in strs if len(s) <= 83][:n]
    x = x[:len(strs)]
    cur["x"], cur["p"] = x, np.array([[s.encode()] for s in strs], dtype="S83")
    ds = ref_dataset.PredictDataset("x.bin")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.pd.bin")
        sitefile.write_pileup_bin(path, x, strs)
        names, pos, refb, xm = sitefile.read_pileup_bin(path)
        fields = sitefile.read_arrays(path)["position"]
        uniq = list(dict.fromkeys(ds.contig_names))
        p2, c2, r2 = host.parse_ctg_pos_ref(np.asarray(fields), host.ContigTable(uniq))
    ok = (list(ds.contig_names) == names == [uniq[i] for i in c2] and np.array_equal(ds.positions, pos) and np.array_equal(pos, p2)
          and np.array_equal(ds.reference_bases, refb) and np.array_equal(refb, r2) and np.array_equal(ds.position_matrix, xm))
    # strings on which the reference raises
    raised_same = True
    for s in ("chr1:5", "chr1:5:" + "A" * 16, "chr1:x:" + "A" * 33, "chr1:5:" + "A" * 33 + ":9", "chr1::" + "A" * 33, "HLA:01:01:" + "A" * 33, "chr1:1.5:" + "A" * 33):
        cur["x"], cur["p"] = x[:1], np.array([[s.encode()]], dtype="S83")
        try:
            ref_dataset.PredictDataset("x.bin"); r_ref = "ok"
        except Exception as e:
            r_ref = type(e).__name__
        f1 = np.zeros((1, 83), np.uint8); f1[0, :len(s)] = np.frombuffer(s.encode(), np.uint8)
        try:
            host.parse_ctg_pos_ref(f1, host.ContigTable(["chr1"])); r_ours = "ok"
        except host.HostError:
            r_ours = "HostError"
        raised_same = raised_same and ((r_ref == "ok") == (r_ours == "ok"))
        if (r_ref == "ok") != (r_ours == "ok"): print("   ", repr(s), "reference:", r_ref, "ours:", r_ours)
    bad += not (ok and raised_same)
    print(seed, len(strs), "sites: fields", "identical" if ok else "DIFFER", "| raising strings agree:", raised_same, flush=True)
print("bad", bad)
