#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): the reference's own predict() loop (PileupModel/predict.py:37-195) driven by a
stand-in model that returns prepared probabilities - every genotype / zygosity class, ties, exact ones, depth 0, odd reference bases -
against nsnp_vcf_format_batches, byte for byte, batch sizes 1000 / 64 / 7, both NumPy promotion generations.
    python tests/manual/ref_fuzz/vcf_rows.py FIRST_SEED END_SEED"""
import os, sys, tempfile, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from torch.utils.data import Dataset
from nanosnp_amd import host
REF="/root/reference"
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object}), ("tables", {"Filters": lambda **k: None})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
sys.path.insert(0, os.path.join(REF, "PileupModel"))
import predict as ref_predict

def make_sites(seed, N):
    rng = np.random.default_rng(seed)
    names = [("chrS" if rng.random() < 0.6 else "chrT") for _ in range(N)]
    pos = np.sort(rng.integers(1, 5000, N)).astype(np.int64)
    refb = rng.choice(np.frombuffer(b"ACGT", np.uint8), N).copy()
    odd = rng.random(N) < 0.03
    refb[odd] = rng.choice(np.frombuffer(b"Nacgtn*", np.uint8), int(odd.sum()))
    x = np.zeros((N, 33, 18), np.int32)
    cov = rng.integers(0, 40, (N, 8))
    ridx = np.array([{65: 0, 67: 1, 71: 2, 84: 3}.get(int(b), int(rng.integers(0, 4))) for b in refb])
    for i in range(N):
        u = rng.random()
        if u < 0.85:
            cov[i, ridx[i]] = -int(rng.integers(1, 60)); cov[i, ridx[i] + 4] = -int(rng.integers(0, 60))
        elif u < 0.9:
            pass                                  # no negative entry: depth 0
        elif u < 0.95:
            cov[i] = -cov[i]                      # everything negative
        else:
            cov[i] = 0
    x[:, 16, [0, 1, 2, 3, 9, 10, 11, 12]] = cov
    x[:, 0, 17] = np.arange(N) % 4096; x[:, 1, 17] = np.arange(N) // 4096
    # probabilities: peaked at a class drawn from ALL classes, varied sharpness; ties; exact ones
    gt = np.zeros((N, 21), np.float32); zy = np.zeros((N, 3), np.float32)
    for i in range(N):
        for arr, C in ((gt, 21), (zy, 3)):
            k = rng.integers(0, C) if arr is zy or rng.random() < 0.25 else rng.integers(0, 10)
            sharp = rng.choice([0.5, 2.0, 6.0, 15.0, 40.0, 120.0])
            logit = rng.normal(0, 1, C); logit[k] += sharp
            p = np.exp(logit - logit.max()); p = (p / p.sum()).astype(np.float32)
            u = rng.random()
            if u < 0.03: p[:] = 0; p[k] = 1.0
            elif u < 0.06: p[(k + 1) % C] = p[k]                 # a tie
            elif u < 0.08: p[:] = np.float32(1.0 / C)
            arr[i] = p
    return names, pos, refb, x, gt, zy

class FakeModel:
    def __init__(self, gt, zy): self.gt, self.zy = torch.from_numpy(gt), torch.from_numpy(zy)
    def eval(self): pass
    def predict(self, ft):
        idx = (ft[:, 0, 17] + 4096 * ft[:, 1, 17]).long()
        return self.gt[idx], self.zy[idx]

def run_reference(names, pos, refb, x, gt, zy, bs, fai_text, np1):
    class FakeDataset(Dataset):
        def __init__(self, datapath): pass
        def __getitem__(self, i): return names[i], pos[i], refb[i], x[i]
        def __len__(self): return len(x)
    ref_predict.PredictDataset = FakeDataset
    from torch.utils.data import DataLoader as DL
    ref_predict.DataLoader = lambda ds, batch_size, shuffle, num_workers: DL(ds, batch_size=batch_size, shuffle=False, num_workers=0)
    orig = torch.Tensor.numpy
    def widened(self, *a, **k):
        r = orig(self, *a, **k)
        return r.astype(np.float64) if r.dtype == np.float32 else r
    with tempfile.TemporaryDirectory() as d:
        fai = os.path.join(d, "ref.fa.fai"); open(fai, "w").write(fai_text)
        vcf = os.path.join(d, "o.vcf")
        torch.Tensor.numpy = widened if np1 else orig
        try:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref_predict.predict(FakeModel(gt, zy), ["x.bin"], fai, bs, vcf, torch.device("cpu"))
        finally:
            torch.Tensor.numpy = orig
        return open(vcf, "rb").read()

def run_ours(names, pos, refb, x, gt, zy, bs, fai_text, np1):
    uniq = list(dict.fromkeys(names)); table = host.ContigTable(uniq)
    cid = np.array([uniq.index(n) for n in names], np.int32)
    cov = x[:, 16, [0, 1, 2, 3, 9, 10, 11, 12]].astype(np.float32)
    text, rows = host.vcf_format_batches(table, cid, pos, refb, gt.argmax(1).astype(np.uint8), zy.argmax(1).astype(np.uint8), gt.max(1), zy.max(1), cov,
                                         batch_size=bs, score_mode=host.SCORE_FLOAT64 if np1 else host.SCORE_FLOAT32)
    return host.vcf_header(fai_text).encode() + text

if __name__ == "__main__":
    fai_text = "chrS\t6100\t6\t60\t61\nchrT\t1600\t6\t60\t61\n"
    bad = 0
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        N = 1500
        s = make_sites(seed, N)
        for bs in (1000, 64, 7):
            for np1 in (False, True):
                want = run_reference(*s, bs, fai_text, np1); got = run_ours(*s, bs, fai_text, np1)
                ok = want == got
                bad += not ok
                print(seed, bs, "np1" if np1 else "np2", "rows", want.count(b"\n"), got.count(b"\n"), "identical" if ok else "DIFFER", flush=True)
                if not ok:
                    a, b = want.split(b"\n"), got.split(b"\n")
                    shown = 0
                    import difflib
                    for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, a, b, autojunk=False).get_opcodes():
                        if tag != "equal" and shown < 4:
                            print("  ", tag, "ref:", a[i1:i2][:2], "ours:", b[j1:j2][:2]); shown += 1
    print("bad", bad)
