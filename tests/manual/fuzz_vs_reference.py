#!/usr/bin/env python3
"""Development-container fuzz of the CPU restatements against the reference's own code (needs /root/reference; never
runs on the GPU box): random and extreme inputs beyond the committed goldens.

    python tests/manual/fuzz_vs_reference.py hapfeat    oracle.hap_features        vs dataset_dev.get_frequency_feature   (bit-exact)
    python tests/manual/fuzz_vs_reference.py pileup     oracle.pileup_forward      vs PileupModel/model.py + ont_pileup.chkpt
    python tests/manual/fuzz_vs_reference.py vcf        host.vcf_format_batches    vs PileupModel/predict.py predict()    (byte-exact)
    python tests/manual/fuzz_vs_reference.py encode     oracle.mpileup_to_pd       vs oracle/_ref programs                (byte-exact)

One group per process (the reference's PileupModel and HaplotypeModel module names collide)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg                      # noqa: E402


def fuzz_hapfeat():
    mg._stub_modules()
    sys.path.insert(0, "/root/reference/HaplotypeModel")
    from dataset_dev import get_frequency_feature
    from oracle import oracle
    rng = np.random.default_rng(99)
    worst = 0; n = 0
    for it in range(400):
        D = int(rng.choice([1, 2, 3, 10, 40, 90, 180])); L = int(rng.choice([11, 33]))
        kind = it % 4
        if kind == 0:
            seq = rng.integers(-2, 5, (D, L)); hap = rng.integers(-2, 4, (D, L))
        elif kind == 1:
            seq = rng.integers(1, 5, (D, L)); hap = np.tile(rng.integers(1, 4, (D, 1)), (1, L))
            pad = rng.integers(0, D + 1); seq[D - pad:] = -2; hap[D - pad:] = -2
        elif kind == 2:
            seq = rng.integers(-3, 7, (D, L)); hap = rng.integers(-3, 6, (D, L))      # codes outside the documented sets
        else:
            seq = np.where(rng.random((D, L)) < 0.1, 0, rng.integers(-1, 5, (D, L))); hap = np.where(seq == 0, 0, rng.integers(1, 4, (D, 1)))
        bq = rng.integers(-2, 94, (D, L)); mq = rng.integers(-2, 61, (D, L))
        want = get_frequency_feature(seq, bq, mq, hap)
        got = oracle.hap_features(seq, bq, mq, hap, np.zeros(L, np.int32))[:104]
        if not np.array_equal(got, want):
            d = np.abs(got - want); print("MISMATCH", it, kind, D, L, d.max(), np.argwhere(d > 0)[:5]); worst += 1
        n += 1
    print(n, "sites, mismatches:", worst)


def fuzz_pileup():
    import torch, yaml
    mg._stub_modules()
    sys.path.insert(0, "/root/reference/PileupModel")
    from model import LSTMNetwork
    from utils import AttrDict
    from oracle import oracle
    from tests.helpers import load_pileup_weights
    cfg = AttrDict(yaml.load(open("/root/reference/PileupModel/config/ont_pileup.yaml"), Loader=yaml.FullLoader))
    m = LSTMNetwork(cfg.model)
    ck = torch.load("/root/reference/PileupModel/models/ont_pileup.chkpt", map_location="cpu", weights_only=False)
    m.encoder.load_state_dict(ck["encoder"]); m.forward_layer.load_state_dict(ck["forward_layer"]); m.eval()
    w = load_pileup_weights()
    rng = np.random.default_rng(5)
    worst = 0
    for kind in range(5):
        n = 200
        if kind == 0: x = rng.integers(0, 50, (n, 33, 18)) - 10
        elif kind == 1: x = np.zeros((n, 33, 18), np.int64)
        elif kind == 2: x = rng.integers(-144, 145, (n, 33, 18))
        elif kind == 3: x = rng.poisson(2, (n, 33, 18)) * (rng.random((n, 33, 18)) < 0.3)
        else: x = rng.integers(-2000, 2000, (n, 33, 18))
        with torch.no_grad():
            gt, zy = m.predict(torch.from_numpy(x.astype(np.int32)).type(torch.FloatTensor))
        og, oz = oracle.pileup_forward(w, x.astype(np.int32), nthreads=8)
        d = max(np.abs(og - gt.numpy()).max(), np.abs(oz - zy.numpy()).max())
        print(kind, d); worst = max(worst, d)
    print("worst", worst)


def fuzz_vcf():
    import torch
    from torch.utils.data import Dataset
    from nanosnp_amd import host
    mg._stub_modules()
    sys.path.insert(0, "/root/reference/PileupModel")
    import predict as ref_predict
    rng = np.random.default_rng(2024)
    real_loader = ref_predict.DataLoader
    ref_predict.DataLoader = lambda ds, batch_size, shuffle, num_workers: real_loader(ds, batch_size=batch_size, shuffle=False, num_workers=0)
    COV = [0, 1, 2, 3, 9, 10, 11, 12]
    bad = 0; n_cmp = 0
    for trial in range(12):
        N = int(rng.choice([1, 5, 9, 10, 11, 63, 500, 1500]))
        x = rng.integers(-40, 40, (N, 33, 18)).astype(np.int32)
        if trial % 3 == 0: x[:, 16, :] = np.abs(x[:, 16, :])            # no negative coverage -> depth 0 -> inf / nan AF
        if trial % 4 == 1: x[rng.random(N) < 0.3, 16] = 0
        names = [("chrA" if rng.random() < 0.5 else "chr_B.2") for _ in range(N)]
        pos = np.sort(rng.integers(1, 10**9, N)).astype(np.int64)
        refb = rng.choice(np.frombuffer(b"ACGTNacgt", np.uint8), N).astype(np.uint8)
        gt = rng.dirichlet(np.full(21, 0.3), N).astype(np.float32); zy = rng.dirichlet(np.full(3, 0.5), N).astype(np.float32)
        sel = rng.random(N) < 0.1
        gt[sel] = 0; gt[sel, rng.integers(0, 21, sel.sum())] = 1.0       # p == 1 exactly: log domain error path
        zy[rng.random(N) < 0.05] = np.array([0, 1, 0], np.float32)
        class FakeDataset(Dataset):
            def __init__(self, datapath): pass
            def __getitem__(self, i): return names[i], pos[i], refb[i], x[i]
            def __len__(self): return N
        class FakeModel:
            def __init__(self): self.i = 0
            def eval(self): pass
            def predict(self, inputs):
                b = inputs.shape[0]; g = torch.from_numpy(gt[self.i:self.i + b]); z = torch.from_numpy(zy[self.i:self.i + b]); self.i += b
                return g, z
        ref_predict.PredictDataset = FakeDataset
        for bs in (1000, 64, 7):
            with tempfile.TemporaryDirectory() as d:
                fai = os.path.join(d, "r.fai"); open(fai, "w").write("chrA\t1000000000\t6\t60\t61\nchr_B.2\t1000000000\t6\t60\t61\n")
                for mode in (host.SCORE_FLOAT32, host.SCORE_FLOAT64):
                    # mode 0: the reference as this container's NumPy 2 runs it; mode 1: under NumPy 1.x scalar promotion, emulated by
                    # widening the float32 arrays where they enter NumPy (tests/golden/make_golden.py vcf)
                    vcf = os.path.join(d, "p.vcf")
                    orig_numpy = torch.Tensor.numpy
                    if mode == host.SCORE_FLOAT64:
                        torch.Tensor.numpy = lambda self, *a, **k: (lambda r: r.astype(np.float64) if r.dtype == np.float32 else r)(orig_numpy(self, *a, **k))
                    try:
                        ref_predict.predict(FakeModel(), ["x.bin"], fai, bs, vcf, torch.device("cpu"))
                        want = open(vcf, "rb").read(); ref_err = None
                    except Exception as e:
                        want = None; ref_err = repr(e)
                    finally:
                        torch.Tensor.numpy = orig_numpy
                    table = host.ContigTable(names)
                    cov = x[:, 16, COV].astype(np.float32)
                    try:
                        text, rows = host.vcf_format_batches(table, table.ids, pos, refb, gt.argmax(1), zy.argmax(1), gt.max(1), zy.max(1), cov, batch_size=bs,
                                                             score_mode=mode)
                        got = host.vcf_header(open(fai).read()).encode() + text; my_err = None
                    except Exception as e:
                        got = None; my_err = repr(e)
                    ok = (got == want) if want is not None and got is not None else (want is None and got is None)
                    n_cmp += 1
                    if not ok:
                        bad += 1
                        print("MISMATCH trial", trial, "mode", mode, "N", N, "bs", bs, "ref_err", ref_err, "my_err", my_err)
                        if want is not None and got is not None:
                            gl, wl = got.split(b"\n"), want.split(b"\n")
                            print(len(gl), len(wl))
                            for a, b in zip(gl, wl):
                                if a != b: print(a, b"|", b); break
    print("comparisons:", n_cmp, "mismatches:", bad)


def fuzz_encode():
    from nanosnp_amd import host
    from oracle import oracle
    for seed in (1, 3, 5, 7):
        rng = np.random.default_rng(700 + seed)
        n = 6000
        seq = rng.choice(np.frombuffer(b"ACGTACGTACGTacgtN", np.uint8), n + 40).astype(np.uint8)
        d = tempfile.mkdtemp()
        fa = os.path.join(d, "r.fa"); host.write_fasta(fa, "ctgF", seq)
        lines = []
        for p in range(1, n + 1):
            if rng.random() < 0.002:
                continue                                            # position gaps
            L = int(rng.integers(1, 160))
            col = mg.adversarial_columns(rng, 1, seq[p - 1:p])[0].encode() or b"*"
            lines.append(b"ctgF\t%d\tN\t%d\t%s\t%s\n" % (p, L, col, b"~" * max(1, L)))
        mp = b"".join(lines)
        want = mg.run_ref_encode(d, "ctgF", fa, mp)
        mpf = os.path.join(d, "x.mpileup"); open(mpf, "wb").write(mp)
        pdf = os.path.join(d, "o.pd"); oracle.mpileup_to_pd(mpf, bytes(seq), pdf)
        print(seed, want.count(b"\n"), "sites, byte-identical:", open(pdf, "rb").read() == want)




def fuzz_merge(rounds=200, seed=4242):
    """nanosnp_amd.merge.merge_calls against the reference's scripts/merge.py Run on random pileup.vcf / haplotype.csv pairs: every
    genotype-letter pair incl. D / I, qualities on both sides of 13 and of the threshold, RefCall rows, sites without a haplotype call"""
    import argparse, importlib.util, os, random, tempfile
    from nanosnp_amd.merge import merge_calls
    spec = importlib.util.spec_from_file_location("refmerge", "/root/reference/scripts/merge.py")      # development container only
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    rng = random.Random(seed)
    letters = "ACGTDI"
    n_rows = 0
    with tempfile.TemporaryDirectory() as tmp:
        for r in range(rounds):
            thr = rng.choice([15.0, 19.0, 0.0, 13.0, 30.0])
            vcf = ["##fileformat=VCFv4.3\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSample\n"]
            csv = []
            pos = 0
            for _ in range(rng.randint(1, 60)):
                pos += rng.randint(1, 50)
                ctg = rng.choice(["chr1", "chr2", "7"])
                refb = rng.choice("ACGT")
                q = rng.choice([0.0, 12.99, 13.0, 13.01, thr, thr + 0.01, round(rng.uniform(0, 60), 2)])
                filt = rng.choice(["PASS", "PASS", "RefCall"])
                vcf.append(f"{ctg}\t{pos}\t.\t{refb}\t{rng.choice('ACGT')}\t{q}\t{filt}\t.\tGT:GQ:DP:AF\t{rng.choice(['0/1', '1/1', '0/0'])}:{int(q)}:{rng.randint(1, 90)}:{rng.random():f}\n")
                if rng.random() < 0.7:
                    csv.append(f"{ctg}\t{pos}\t{rng.choice(letters)}{rng.choice(letters)}\t{rng.choice([12.99, 13.0, round(rng.uniform(0, 60), 2)])}\n")
            vp, cp, op = (os.path.join(tmp, n) for n in ("p.vcf", "h.csv", "o.vcf"))
            open(vp, "w").write("".join(vcf)); open(cp, "w").write("".join(csv))
            if not csv:
                continue                                   # (the reference returns early on an empty csv: merge.py:30-33)
            ref.Run(argparse.Namespace(cat_predict=cp, output=op, pileup_vcf=vp, quality=thr))
            want = open(op).read()
            got = merge_calls("".join(vcf), "".join(csv), thr)
            assert got == want, (r, thr)
            n_rows += len(vcf) - 2
    print(f"merge: {rounds} random file pairs ({n_rows} rows) byte-identical to scripts/merge.py Run")


if __name__ == "__main__":
    {"hapfeat": fuzz_hapfeat, "pileup": fuzz_pileup, "vcf": fuzz_vcf, "encode": fuzz_encode, "merge": fuzz_merge}[sys.argv[1]]()
