#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): dna_sv_tensor/src/make_bin_data/make_bin_predict_data.py main() with a
recording stand-in for PyTables (what is appended to which EArray, flushed every 1000 lines) on .pd text written by the reference's
own compiled programs (tests/golden/encode_*.pd.gz) and on the same text with spaces around the fields, against sitefile.pd_to_bin +
read_pileup_bin / read_alt_info: the three arrays, in order.  One input per output, as make_predict_data.sh:234 calls it (with
several inputs the reference appends what it has already flushed again - its table_dict is rebound inside transform_one_input only:
138 / 570 / 731 rows for files of 138 / 432 / 161 - not rebuilt; the last lines of this program show it).
    python tests/manual/ref_fuzz/make_bin.py"""
import os, sys, types, tempfile, gzip, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
rec = {}
class _EA:
    def __init__(self, name): self.name = name
    def append(self, a): rec.setdefault(self.name, []).append(np.array(a))
class _File:
    def __init__(self): self.root = types.SimpleNamespace()
    def create_earray(self, where, name, atom, shape, filters=None): setattr(self.root, name, _EA(name)); rec[name + "_atom"] = atom
    def close(self): pass
tb = types.ModuleType("tables")
tb.Filters = lambda **k: None; tb.set_blosc_max_threads = lambda n: None
tb.open_file = lambda path, mode="r", filters=None: _File()
tb.Atom = types.SimpleNamespace(from_dtype=lambda d: d)
tb.StringAtom = lambda itemsize: ("S", itemsize)
sys.modules["tables"] = tb
sys.path.insert(0, "/root/reference/dna_sv_tensor/src/make_bin_data")
import make_bin_predict_data as mb
from nanosnp_amd import sitefile
bad = 0
texts = {t: gzip.open(os.path.join(ROOT, "tests", "golden", f"encode_{t}.pd.gz")).read() for t in ("g1", "adv", "end", "cut", "pos")}
cases = [("g1",), ("adv",), ("end",), ("cut",), ("pos",)]
for case in cases:
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for t in case:
            p = os.path.join(d, f"{t}.pd"); open(p, "wb").write(texts[t]); paths.append(p)
        rec.clear()
        with contextlib.redirect_stdout(io.StringIO()):
            mb.main(["prog", os.path.join(d, "out.bin")] + paths)
        want_x = np.concatenate([a for a in rec["position_matrix"] if a.size]) if rec.get("position_matrix") else np.zeros((0, 33, 18))
        want_p = [v.decode() if isinstance(v, bytes) else str(v) for a in rec["position"] for v in a.reshape(-1)]
        want_a = [v.decode() if isinstance(v, bytes) else str(v) for a in rec["alt_info"] for v in a.reshape(-1)]
        ob = os.path.join(d, "ours.pd.bin")
        n = sitefile.pd_to_bin(b"".join(texts[t] for t in case), ob)
        names, pos, refb, x = sitefile.read_pileup_bin(ob)
        fields = [bytes(r).rstrip(b"\0").decode() for r in np.asarray(sitefile.read_arrays(ob)["position"])]
        alts = sitefile.read_alt_info(ob)
        ok = n == len(want_p) and np.array_equal(np.asarray(x), want_x) and fields == want_p and alts == want_a
        bad += not ok
        print(case, n, "sites:", "identical" if ok else "DIFFER", "| the reference flushed", len(rec["position_matrix"]), "times", flush=True)
with tempfile.TemporaryDirectory() as d:                      # the several-inputs quirk, for the record
    paths = []
    for t in ("g1", "adv", "pos"):
        p = os.path.join(d, f"{t}.pd"); open(p, "wb").write(texts[t]); paths.append(p)
    rec.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        mb.main(["prog", os.path.join(d, "out.bin")] + paths)
    print("three inputs of", [texts[t].count(b"\n") for t in ("g1", "adv", "pos")], "lines -> the reference appends", [len(a) for a in rec["position"]], "rows")
print("bad", bad)
