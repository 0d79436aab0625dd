#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): HaplotypeModel/predict_dev.py predict() itself - TestDataset over a stand-in
table file, its DataLoader, model_dev.LSTMNetwork with seeded weights on the CPU, the csv rows it writes - on random haplotype bins
(depths 1-120, unknown contigs, windows over both contig ends, lower-case / N reference), against the oracle chain the GPU path is held
to: host reference rows -> oracle features -> oracle forward -> argmax / max -> nsnp_hap_csv_format (NumPy-1.x promotion emulated as in
tests/golden/make_golden.py).
    python tests/manual/ref_fuzz/predict_dev.py [N_SEEDS]"""
import os, sys, types, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
cur = {}
tb = types.ModuleType("tables"); tb.Filters = lambda **k: None
tb.open_file = lambda path, mode="r": types.SimpleNamespace(root=cur["root"], close=lambda: None)
sys.modules["tables"] = tb
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
sys.path.insert(0, "/root/reference/HaplotypeModel")
import predict_dev
from model_dev import LSTMNetwork
from utils import AttrDict
from nanosnp_amd import host
from nanosnp_amd.fixtures import hap_weight_names, seeded_hap_weights
from oracle import oracle
torch.set_num_threads(8)
cfg = AttrDict({"model": {"pileup_dim": 105, "haplotype_dim": 105, "pileup_length": 33, "haplotype_length": 11, "hidden_size": 256,
                          "lstm_layers": 3, "gt_num_class": 10, "zy_num_class": 3, "dropout": 0.1}})
real_loader = torch.utils.data.DataLoader
predict_dev.torch.utils.data.DataLoader = lambda ds, batch_size, shuffle, num_workers: real_loader(ds, batch_size=batch_size, shuffle=False, num_workers=0)
orig_numpy = torch.Tensor.numpy
def widened(self, *a, **k):
    r = orig_numpy(self, *a, **k)
    return r.astype(np.float64) if r.dtype == np.float32 else r
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    rng = np.random.default_rng(300 + seed)
    ws = seeded_hap_weights(60 + seed, H=256, ih_scale=float(rng.choice([0.002, 0.03])), head_scale=float(rng.choice([8.0, 120.0])))
    m = LSTMNetwork(cfg)
    res = m.load_state_dict({k: torch.from_numpy(w) for k, w in zip(hap_weight_names(), ws)}, strict=False)
    m.eval()
    n = int(rng.choice([1, 9, 40]))
    Dp, Dh = int(rng.choice([1, 30, 90, 120])), int(rng.choice([1, 25, 90]))
    refs = {c: "".join(rng.choice(list("ACGTacgtN"), int(rng.integers(200, 3000)), p=[.23, .23, .23, .23, .02, .02, .01, .01, .02])) for c in ("ctgA", "ctgB")}
    contig = str(rng.choice(["ctgA", "ctgB", "ctgMissing"], p=[.5, .4, .1]))
    L = len(refs.get(contig, "x" * 800))
    pp = host.synth_hap_planes(7000 + seed, n, 30, Dp, 33); ph = host.synth_hap_planes(7100 + seed, n, 30, Dh, 11)
    posn = np.sort(rng.choice(np.arange(-20, L + 40), n, replace=False))
    cands = [f"{contig}:{p}" for p in posn]
    hpos = [[f"{contig}:{p + 37 * (k - 5)}" for k in range(11)] for p in posn]
    root = types.SimpleNamespace()
    root.pileup_sequences, root.pileup_baseq, root.pileup_mapq, root.pileup_hap = pp[0], pp[1], pp[2], pp[3]
    root.haplotype_sequences, root.haplotype_baseq, root.haplotype_mapq, root.haplotype_hap = ph[0], ph[1], ph[2], ph[3]
    root.candidate_positions = np.array([[c.encode()] for c in cands], dtype="S300")
    root.haplotype_positions = np.array([[p.encode() for p in row] for row in hpos], dtype="S300")
    cur["root"] = root
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "ref.fa")
        with open(fa, "w") as f:
            for c, sq in refs.items():
                f.write(f">{c}\n"); f.write("\n".join(sq[i:i + 60] for i in range(0, len(sq), 60)) + "\n")
        bins = os.path.join(d, "bins"); os.makedirs(bins); open(os.path.join(bins, "x.bin"), "w").close()
        csv_path = os.path.join(d, "h.csv")
        torch.Tensor.numpy = widened
        try:
            with torch.no_grad():
                predict_dev.predict(m, bins, fa, int(rng.choice([7, 1000])), 33, 11, csv_path, torch.device("cpu"))
        finally:
            torch.Tensor.numpy = orig_numpy
        want = open(csv_path).read()
    refs_b = {k: v.encode() for k, v in refs.items()}
    rp = host.haplotype_ref_rows(refs_b, cands, 33); rh = host.haplotype_ref_rows(refs_b, cands, 11, position_lists=hpos)
    xp = oracle.hap_features_batch(*pp[:4], rp, nthreads=8); xh = oracle.hap_features_batch(*ph[:4], rh, nthreads=8)
    ogt, _ = oracle.hap_forward(ws, xp, xh, nthreads=8)
    tbl = host.ContigTable([contig])
    got = host.hap_csv_format(tbl, np.zeros(n, np.int32), posn.astype(np.int64), ogt.argmax(1).astype(np.uint8), ogt.max(1), host.SCORE_FLOAT64).decode()
    a, b = want.splitlines(), got.splitlines()
    same_sites = len(a) == len(b) and all(x.split("\t")[:2] == y.split("\t")[:2] for x, y in zip(a, b))
    flips = sum(x.split("\t")[2] != y.split("\t")[2] for x, y in zip(a, b)) if same_sites else -1
    dq = max([abs(float(x.split("\t")[3]) - float(y.split("\t")[3])) for x, y in zip(a, b)] + [0.0]) if same_sites else -1
    ok = same_sites and flips == 0 and dq <= 0.0101
    bad += not ok
    print(f"seed {seed}: {n} sites, depths {Dp}/{Dh}, contig {contig}: rows {'match' if same_sites else 'DIFFER'}, genotype flips {flips}, worst |QUAL difference| {dq:.3f} "
          f"({'byte-identical' if want == got else 'QUAL digits differ within one unit'}) -> {'ok' if ok else 'LOOK'}", flush=True)
print("bad", bad)
