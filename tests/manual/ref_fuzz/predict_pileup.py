#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): PileupModel/predict.py predict() itself - the real PredictDataset over a stand-in
table file, its DataLoader, LSTMNetwork with a shipped checkpoint on the CPU, the rows it writes - on fresh random window files
(G2 windows at 8x / 30x / 60x, several contigs, N / lower-case centre bases among them), against the chain the GPU path is held to:
oracle forward -> argmax / max / coverage slice -> nsnp_vcf_format_batches.  Rows must be equal except a QUAL / GQ that a probability
at most 1e-6 away explains (the two sides then sit on either side of a rounding boundary of the two-decimal QUAL).
    python tests/manual/ref_fuzz/predict_pileup.py [N_SEEDS]"""
import os, sys, types, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch, yaml
cur = {}
tb = types.ModuleType("tables"); tb.Filters = lambda **k: None
tb.open_file = lambda path, mode="r": types.SimpleNamespace(root=types.SimpleNamespace(position_matrix=cur["x"], position=cur["p"]), close=lambda: None)
sys.modules["tables"] = tb
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
sys.path.insert(0, "/root/reference/PileupModel")
import predict as ref_predict
from model import LSTMNetwork
from utils import AttrDict
from nanosnp_amd import host
from oracle import oracle
from tests.helpers import qual_reachable
torch.set_num_threads(8)
REF = "/root/reference"
real_loader = ref_predict.DataLoader
ref_predict.DataLoader = lambda ds, batch_size, shuffle, num_workers: real_loader(ds, batch_size=batch_size, shuffle=False, num_workers=0)
orig_numpy = torch.Tensor.numpy
def widened(self, *a, **k):
    r = orig_numpy(self, *a, **k)
    return r.astype(np.float64) if r.dtype == np.float32 else r
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    rng = np.random.default_rng(4400 + seed)
    ck_name = ["ont_pileup.chkpt", "hg001_mix_without_balance.epoch13.chkpt", "hg001_mix_without_balance.epoch186.chkpt"][seed % 3]
    cfg = AttrDict(yaml.load(open(os.path.join(REF, "PileupModel/config/ont_pileup.yaml")), Loader=yaml.FullLoader))
    m = LSTMNetwork(cfg.model)
    path = os.path.join(REF, "PileupModel/models", ck_name)
    if not os.path.exists(path): path = os.path.join(REF, "PileupModel/models/ont_pileup.chkpt")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    m.encoder.load_state_dict(ck["encoder"]); m.forward_layer.load_state_dict(ck["forward_layer"]); m.eval()
    ws = [v.numpy().astype(np.float32) for v in list(ck["encoder"].values())[:18] + list(ck["forward_layer"].values())[:6]]
    n = int(rng.choice([300, 1100, 2500]))
    cols = host.synth_columns(8800 + seed, n * 33, coverage=float(rng.choice([8, 30, 60])), window=33)
    oc, _, _ = oracle.encode_columns(cols.bases, cols.col_off, cols.ref)
    x = oc.reshape(n, 33, 18).astype(np.int32)
    names = [str(c) for c in rng.choice(["chrS", "chrT"], n)]
    order = np.argsort(np.array(names), kind="stable"); names = [names[i] for i in order]
    pos = np.concatenate([np.sort(rng.choice(np.arange(1, 5000), int((np.array(names) == c).sum()), replace=False)) for c in ("chrS", "chrT")]).astype(np.int64)
    centre = cols.ref.reshape(n, 33)[:, 16].copy()
    centre = np.where(np.isin(centre & 0xDF, np.frombuffer(b"ACGT", np.uint8)), centre & 0xDF, ord("A")).astype(np.uint8)
    strs = [f"{c}:{int(p)}:{'N' * 16}{chr(int(b))}{'N' * 16}" for c, p, b in zip(names, pos, centre)]
    cur["x"], cur["p"] = x, np.array([[s.encode()] for s in strs], dtype="S83")
    fai_text = "chrS\t6100\t6\t60\t61\nchrT\t1600\t6\t60\t61\n"
    bs = int(rng.choice([1000, 64]))
    with tempfile.TemporaryDirectory() as d:
        fai = os.path.join(d, "ref.fa.fai"); open(fai, "w").write(fai_text)
        vcf = os.path.join(d, "p.vcf")
        torch.Tensor.numpy = widened
        try:
            with torch.no_grad():
                ref_predict.predict(m, ["x.bin"], fai, bs, vcf, torch.device("cpu"))
        finally:
            torch.Tensor.numpy = orig_numpy
        want = [l for l in open(vcf).read().splitlines() if not l.startswith("#")]
    ogt, ozy = oracle.pileup_forward(ws, x, nthreads=8)
    uniq = list(dict.fromkeys(names)); tbl = host.ContigTable(uniq)
    cov = x[:, 16, [0, 1, 2, 3, 9, 10, 11, 12]].astype(np.float32)
    text, _ = host.vcf_format_batches(tbl, np.array([uniq.index(c) for c in names], np.int32), pos, centre, ogt.argmax(1).astype(np.uint8), ozy.argmax(1).astype(np.uint8),
                                      ogt.max(1), ozy.max(1), cov, batch_size=bs, score_mode=host.SCORE_FLOAT64)
    got = text.decode().splitlines()
    site = {(c, int(p)): j for j, (c, p) in enumerate(zip(names, pos))}
    ok = len(got) == len(want); n_q = 0
    if ok:
        for g, w in zip(got, want):
            if g == w: continue
            gf, wf = g.split("\t"), w.split("\t")
            same = gf[:5] == wf[:5] and gf[6:9] == wf[6:9] and gf[9].split(":")[0] == wf[9].split(":")[0] and gf[9].split(":")[2:] == wf[9].split(":")[2:]
            j = site[(gf[0], int(gf[1]))]
            same = same and qual_reachable(float(wf[5]), ozy[j].max(), ogt[j].max(), refcall=gf[6] == "RefCall", score_mode=host.SCORE_FLOAT64)
            n_q += 1
            if not same:
                ok = False; print("   ", g, "|", w); break
    bad += not ok
    print(f"seed {seed}: {ck_name[:28]}, {n} windows, batch {bs}: {len(want)} rows, {'equal' if ok else 'DIFFER'} ({n_q} rows differ in a QUAL digit a 1e-6 probability step explains)", flush=True)
print("bad", bad)
