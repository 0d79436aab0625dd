#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE ITSELF.

Runs only in the development container, where the upstream tree is mounted read-only at
/root/reference; the fixtures it writes are data (inputs + expected outputs) and are what the
tests on the GPU box compare against -- the reference does not travel.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py pileup     # one group: encode | pileup | hapfeat | hapfwd

Groups (each runs in its own interpreter: the reference's two model packages both define
top-level modules named model/optim/utils/options):
  encode   oracle/_ref (the reference's dna_sv_tensor programs compiled from the reference
           tree) on synthetic + adversarial mpileup text  -> encode_*.{mpileup,fa,pd}.gz
  pileup   PileupModel/model.py LSTMNetwork.predict with the shipped ont_pileup.chkpt
           (CPU torch)                                    -> nanosnp_amd/data/ont_pileup_weights.npz, pileup_fwd.npz
  hapfeat  HaplotypeModel/dataset_dev.get_frequency_feature -> hap_features.npz
  hapfwd   HaplotypeModel/model_dev.LSTMNetwork.predict with seeded weights
           (trained weights are absent upstream)          -> hap_fwd_h32.npz, hap_fwd_h256.npz
"""
from __future__ import annotations

import gzip
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("NANOSNP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _stub_modules():
    for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object}),
                        ("tables", {"Filters": lambda **k: None}), ("pysam", {})):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m


# ------------------------------------------------------------------------------------------
def run_ref_encode(workdir, contig, fasta, mpileup_bytes):
    """DNA_CreateCanSnpTensor + DNA_CreatePredictData with the flags of
    make_predict_data.sh:184-194,210-215; returns the .pd bytes."""
    refdir = os.path.join(ROOT, "oracle", "_ref")
    pile = os.path.join(workdir, "pile"); os.makedirs(pile, exist_ok=True)
    with open(os.path.join(pile, contig + ".mpileup"), "wb") as f:
        f.write(mpileup_bytes)
    subprocess.run([os.path.join(refdir, "DNA_CreateCanSnpTensor"), "-reference", fasta,
                    "-chr_pileup_dir", pile, "-output_dir", os.path.join(workdir, "tensor"),
                    "-min_af", "0.12", "-snp_min_af", "0.12", "-indel_min_af", "0.12",
                    "-min_coverage", "6", "-flanking_base", "16", "-num_threads", "1", contig],
                   check=True, capture_output=True)
    subprocess.run([os.path.join(refdir, "DNA_CreatePredictData"), "-chr_tensor_dir",
                    os.path.join(workdir, "tensor"), "-reference", fasta, "-output_dir",
                    os.path.join(workdir, "pd"), "-num_threads", "1", contig],
                   check=True, capture_output=True)
    with open(os.path.join(workdir, "pd", contig + ".pd"), "rb") as f:
        return f.read()


def cut_allele_columns(rng, n_cols, ref_seq):
    """bytes per column: (1) read stacks over a two-letter allele alphabet that end in an indel whose declared length exceeds what is
    left of the column, beside complete alleles with the same visible characters, in both cases and signs; (2) printable bytes drawn
    at random: every symbol class of the scanner, +n / -n with allele text shorter than n, digits, punctuation"""
    sets = [b"ACGTNacgtn", b"*#", b"^", b"$", b"+-", b"0123456789", b".,<>!?@~;:=/\\|%&()[]{}'`_", bytes(range(33, 127))]
    cols = []
    for c in range(n_cols):
        r = chr(ref_seq[c]).upper()
        r = r if r in "ACGT" else "A"
        out = bytearray()
        if c % 2 == 0:
            depth = int(rng.choice([6, 8, 12, 30]))
            alph = rng.choice(["AC", "ac", "AG", "aC", "N*", "#a"])
            for _ in range(depth):
                out += (r if rng.random() < 0.5 else r.lower()).encode()
                if rng.random() < 0.45:
                    n = int(rng.choice([1, 2, 2, 3]))
                    out += (rng.choice(["+", "-"]) + str(n) + "".join(rng.choice(list(alph), n))).encode()
            n = int(rng.choice([1, 2, 3, 4, 9, 59, 60, 61]))
            vis = int(rng.integers(0, n))                                     # 0 .. n - 1 characters of the allele are left
            out += (rng.choice(["+", "-"]) + str(n) + "".join(rng.choice(list(alph), vis))).encode()
        else:
            w = rng.dirichlet(np.ones(len(sets)) * 0.7); w[0] += 1.0; w /= w.sum()
            for _ in range(int(rng.integers(1, 90))):
                a = sets[rng.choice(len(sets), p=w)]
                ch = a[rng.integers(0, len(a))]
                out.append(ch)
                if ch in b"+-" and rng.random() < 0.9:
                    n = int(rng.choice([0, 1, 2, 3, 5, 9, 10, 30, 59, 60, 61, 99, 100]))
                    out += str(n).encode() if rng.random() < 0.95 else b""
                    ln = n if rng.random() < 0.75 else int(rng.integers(0, n + 3))
                    out += (bytes(rng.choice(np.frombuffer(b"ACGTNacgtn*#", np.uint8), size=ln)) if rng.random() < 0.85
                            else bytes(rng.integers(33, 127, size=ln, dtype=np.uint8)))
        cols.append(bytes(out))
    return cols


def adversarial_columns(rng, n_cols, ref_seq):
    """Columns that exercise every branch of tensor_maker.cpp:83-114,127-188: long indels
    (> 60 are skipped but not counted), multi-digit lengths, '^' swallowing a base-like
    quality char, '$', N/n reads, '*'/'#', mixed-case duplicates of one allele, very deep
    and very shallow columns, digits-free '+'."""
    cols = []
    up, lo = "ACGT", "acgt"
    for c in range(n_cols):
        kind = rng.integers(0, 10)
        depth = int(rng.choice([1, 2, 5, 6, 7, 30, 60, 144, 200]))
        r = chr(ref_seq[c]).upper()
        ri = up.find(r) if r in up else 0
        parts = []
        for _ in range(depth):
            rev = rng.random() < 0.5
            tab = lo if rev else up
            s = ""
            if rng.random() < 0.05:
                s += "^" + rng.choice(list("ACGT+-*#I!~0"))   # the char after ^ is a quality
            u = rng.random()
            if kind == 0 and u < 0.5:
                s += tab[(ri + 1) % 4]
            elif kind == 1 and u < 0.9:
                s += tab[(ri + 2) % 4]
            elif u < 0.06:
                s += "#" if rev else "*"
            elif u < 0.10:
                s += "n" if rev else "N"
            elif u < 0.16:
                s += tab[int(rng.integers(0, 4))]
            else:
                s += tab[ri]
            v = rng.random()
            p_indel = 0.5 if kind in (2, 3) else 0.05
            if v < p_indel:
                sign = "+" if (kind == 2 or (kind != 3 and rng.random() < 0.5)) else "-"
                L = int(rng.choice([1, 1, 1, 2, 2, 3, 9, 10, 11, 59, 60, 61, 75]))
                if kind in (2, 3) and rng.random() < 0.7:
                    seq = ("AC" * 40)[:L] if rng.random() < 0.6 else ("AG" * 40)[:L]
                else:
                    seq = "".join(rng.choice(list(up + "N"), L))
                seq = seq.lower() if rev else seq
                if rng.random() < 0.1:   # mixed-case duplicate of the same allele
                    seq = seq.swapcase() if len(seq) > 1 else seq
                s += f"{sign}{L}{seq}"
            if rng.random() < 0.05:
                s += "$"
            parts.append(s)
        cols.append("".join(parts))
    return cols


def group_encode():
    from nanosnp_amd import host
    os.makedirs(GOLD, exist_ok=True)
    # (a) G1 synthetic contig with lower-case / N reference bases and position gaps
    M = 6000
    cols = host.synth_columns(20260000, M, coverage=30)
    rng = np.random.default_rng(7)
    seq = np.concatenate([cols.ref, np.frombuffer(b"ACGT" * 25, np.uint8)]).copy()
    low = rng.random(seq.size) < 0.05
    seq[low] |= 0x20
    seq[rng.random(seq.size) < 0.003] = ord("N")
    seq[rng.random(seq.size) < 0.001] = ord("n")
    keep = np.ones(M, bool)
    for g in rng.integers(100, M - 100, 8):
        keep[g:g + int(rng.integers(1, 4))] = False
    lines = cols.mpileup_text("chrS").split(b"\n")[:-1]
    text_a = b"\n".join(l for i, l in enumerate(lines) if keep[i]) + b"\n"
    # (b) adversarial contig
    M2 = 1500
    seq2 = rng.choice(list(b"ACGTacgtN"), M2 + 100,
                      p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]).astype(np.uint8)
    acols = adversarial_columns(rng, M2, seq2)
    text_b = b"".join(b"chrT\t%d\tN\t%d\t%s\t%s\n" % (i + 1, 1, c.encode(), b"I")
                      for i, c in enumerate(acols) if c)
    # (c) a short contig whose long deletions run past its end: the reference reads the NUL behind its sequence
    #     buffer into the alt_dict key and its "%s" output of alt_info stops there (tensor_maker.cpp:150-155)
    M3 = 240
    seq3 = rng.choice(list(b"ACGT"), M3).astype(np.uint8)
    ecols = adversarial_columns(rng, M3, seq3)
    for i in range(120, M3):
        if i % 3 == 0:
            ecols[i] += "".join(f"-{L}{'ACGTN' * 12}"[:len(str(L)) + 1 + L] for L in (int(rng.choice([30, 59, 60])), 45)) * 3
    text_c = b"".join(b"chrE\t%d\tN\t%d\t%s\t%s\n" % (i + 1, 1, c.encode(), b"I") for i, c in enumerate(ecols) if c)
    # (d) alleles the END OF THE COLUMN cuts short (declared length > the characters left): tensor_maker.cpp:101 appends `advance`
    #     characters from c_str() regardless, so the key holds the string's NUL - an allele of its own, never merged with a complete
    #     allele that shows the same characters (channels I1 / i1 / D1 / d1), a deletion's length is the declared one, and the alt_info
    #     text stops behind the visible part of a cut insertion; plus columns of printable bytes drawn at random under a loose grammar
    M4 = 1400
    seq4 = rng.choice(list(b"ACGTacgtN"), M4 + 100, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]).astype(np.uint8)
    dcols = cut_allele_columns(rng, M4, seq4)
    text_d = b"".join(b"chrC\t%d\tN\t%d\t%s\t%s\n" % (i + 1, 1, c, b"I") for i, c in enumerate(dcols) if c)
    # (e) positions that repeat or step back, zero-depth placeholder lines, extra columns (concatenated / damaged text): main.cpp:174-178
    #     resets the window at every position that is not the previous one + 1 - a gap of two and a repeated position must not cancel
    M5 = 3000
    cols5 = host.synth_columns(20260005, M5, coverage=20, het_rate=0.2)
    seq5 = np.concatenate([cols5.ref, np.frombuffer(b"ACGT" * 25, np.uint8)]).copy()
    seq5[rng.random(seq5.size) < 0.04] |= 0x20
    lines5 = cols5.mpileup_text("chrP").split(b"\n")[:-1]
    out5, i = [], 0
    while i < M5:
        u = rng.random()
        if u < 0.01: i += int(rng.integers(1, 40)); continue
        if u < 0.02 and i > 50: i -= int(rng.integers(1, 40)); continue
        if u < 0.035: out5.append(lines5[i])                                       # the same line twice
        if u < 0.045:
            f = lines5[i].split(b"\t"); out5.append(b"\t".join([f[0], f[1], b"N", b"0", b"*", b"*"])); i += 1; continue
        if u < 0.055: out5.append(lines5[i] + b"\textra\tcolumns"); i += 1; continue
        if u < 0.065 and i + 3 < M5: out5.append(lines5[i]); out5.append(lines5[i + 2]); i += 3; continue   # a gap of two shortly before / behind a repeat
        out5.append(lines5[i]); i += 1
    text_e = b"\n".join(out5) + b"\n"
    for tag, contig, sq, text in (("g1", "chrS", seq, text_a), ("adv", "chrT", seq2, text_b), ("end", "chrE", seq3, text_c), ("cut", "chrC", seq4, text_d),
                                  ("pos", "chrP", seq5, text_e)):
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "ref.fa")
            host.write_fasta(fa, contig, sq)
            pd = run_ref_encode(d, contig, fa, text)
            fa_bytes = open(fa, "rb").read()
        for name, data in ((f"encode_{tag}.mpileup.gz", text), (f"encode_{tag}.fa.gz", fa_bytes),
                           (f"encode_{tag}.pd.gz", pd)):
            with gzip.GzipFile(os.path.join(GOLD, name), "wb", mtime=0) as f:
                f.write(data)
        print(f"encode_{tag}: {text.count(10)} columns -> {pd.count(10)} sites")


# ------------------------------------------------------------------------------------------
def group_encode_reader():
    """encode_rdr.*: the corner cases of the reference's READER on otherwise ordinary columns - runs of tabs and a leading / trailing tab
    (split_line collapses them: cpp_aux.cpp:43-59), "\\r\\n" line ends (line_reader.cpp:95-127), a missing quality column and extra columns
    (only tokens 0, 1, 4 are read: main.cpp:162-172), positions atoll reads through (white space, a sign, leading zeros, trailing junk),
    another name in column 0 (the program's contig argument names the output), a last line without its newline.  The device tokeniser
    (mpileup_tokenise.hip), the host one (nsnp_textio.c) and the oracle's reader must all give the tensors the reference wrote."""
    from nanosnp_amd import host
    M = 5000
    cols = host.synth_columns(20260006, M, coverage=25, het_rate=0.1)
    rng = np.random.default_rng(11)
    seq = np.concatenate([cols.ref, np.frombuffer(b"ACGT" * 25, np.uint8)]).copy()
    seq[rng.random(seq.size) < 0.04] |= 0x20
    lines = cols.mpileup_text("chrR").split(b"\n")[:-1]
    out = []
    for i, l in enumerate(lines):
        f = l.split(b"\t")
        u = rng.random()
        if u < 0.04: f[1] = b"+000" + f[1] + b"xyz"
        elif u < 0.07: f[1] = b"  " + f[1]
        elif u < 0.09: f[1] = b"0" + f[1] + b" 7"
        elif u < 0.11: f[0] = b"some_other_name"
        l = b"\t".join(f)
        u = rng.random()
        if u < 0.05: l = l.replace(b"\t", b"\t\t\t", int(rng.integers(1, 6)))
        elif u < 0.08: l = b"\t" + l + b"\t"
        elif u < 0.12: l = b"\t".join(f[:5])
        elif u < 0.20: l = l + b"\r"
        elif u < 0.23: l = l + b"\textra\tcolumns"
        elif u < 0.25: l = b"\t".join(f[:5]) + b"\r"
        out.append(l)
    text = b"\n".join(out)                              # (no newline behind the last line)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "ref.fa")
        host.write_fasta(fa, "chrR", seq)
        pd = run_ref_encode(d, "chrR", fa, text)
        fa_bytes = open(fa, "rb").read()
    for name, data in (("encode_rdr.mpileup.gz", text), ("encode_rdr.fa.gz", fa_bytes), ("encode_rdr.pd.gz", pd)):
        with gzip.GzipFile(os.path.join(GOLD, name), "wb", mtime=0) as f:
            f.write(data)
    print(f"encode_rdr: {text.count(10) + 1} columns -> {pd.count(10)} sites")


def group_pileup():
    import torch
    import yaml
    from oracle import oracle
    from nanosnp_amd import host
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "PileupModel"))
    from model import LSTMNetwork          # noqa: E402  (reference module)
    from utils import AttrDict             # noqa: E402
    cfg = AttrDict(yaml.load(open(os.path.join(REF, "PileupModel/config/ont_pileup.yaml")),
                             Loader=yaml.FullLoader))
    m = LSTMNetwork(cfg.model)
    ck = torch.load(os.path.join(REF, "PileupModel/models/ont_pileup.chkpt"), map_location="cpu",
                    weights_only=False)
    m.encoder.load_state_dict(ck["encoder"])
    m.forward_layer.load_state_dict(ck["forward_layer"])
    m.eval()
    w = {}
    for k, v in ck["encoder"].items():
        w["encoder." + k] = v.numpy().astype(np.float32)
    for k, v in ck["forward_layer"].items():
        w["forward_layer." + k] = v.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, "nanosnp_amd", "data", "ont_pileup_weights.npz"), **w)

    # inputs: G2 windows (seed 20260001) encoded by the oracle + hand-made edge windows
    N = 224
    cols = host.synth_columns(20260001, N * 33, coverage=30, window=33)
    counts, depth, flags = oracle.encode_columns(cols.bases, cols.col_off, cols.ref)
    x = counts.reshape(N, 33, 18)
    rng = np.random.default_rng(3)
    edge = np.zeros((32, 33, 18), np.int32)
    edge[1] = 144
    edge[2] = -144
    edge[3, 16, :] = rng.integers(-144, 145, 18)
    edge[4:16] = rng.integers(-60, 61, (12, 33, 18))
    edge[16:32] = rng.integers(0, 8, (16, 33, 18)) * (rng.random((16, 33, 18)) < 0.2)
    x = np.concatenate([x, edge]).astype(np.int32)
    with torch.no_grad():
        gt, zy = m.predict(torch.from_numpy(x).type(torch.FloatTensor))   # predict.py:49-51
    np.savez_compressed(os.path.join(GOLD, "pileup_fwd.npz"), x=x.astype(np.int16),
                        gt=gt.numpy(), zy=zy.numpy())
    print("pileup_fwd:", x.shape, "gt argmax histogram", np.bincount(gt.numpy().argmax(1), minlength=21))


def group_pileup_ckpts():
    """The other two checkpoints PileupModel/models/ ships (same architecture: config/hg001_mix_without_balance.yaml:6-20): their
    weights + the reference's LSTMNetwork.predict on the SAME 256 inputs as pileup_fwd.npz -> pileup_fwd_hg001_e13.npz / _e186.npz"""
    import torch
    import yaml
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "PileupModel"))
    from model import LSTMNetwork          # noqa: E402  (reference module)
    from utils import AttrDict             # noqa: E402
    cfg = AttrDict(yaml.load(open(os.path.join(REF, "PileupModel/config/hg001_mix_without_balance.yaml")), Loader=yaml.FullLoader))
    x = np.load(os.path.join(GOLD, "pileup_fwd.npz"))["x"].astype(np.int32)
    for tag, epoch in (("e13", 13), ("e186", 186)):
        m = LSTMNetwork(cfg.model)
        ck = torch.load(os.path.join(REF, f"PileupModel/models/hg001_mix_without_balance.epoch{epoch}.chkpt"), map_location="cpu",
                        weights_only=False)
        m.encoder.load_state_dict(ck["encoder"])
        m.forward_layer.load_state_dict(ck["forward_layer"])
        m.eval()
        out = {}
        for part in ("encoder", "forward_layer"):
            for k, v in ck[part].items():
                out[f"{part}.{k}"] = v.numpy().astype(np.float32)
        with torch.no_grad():
            gt, zy = m.predict(torch.from_numpy(x).type(torch.FloatTensor))   # predict.py:49-51
        np.savez_compressed(os.path.join(GOLD, f"pileup_fwd_hg001_{tag}.npz"), gt=gt.numpy(), zy=zy.numpy(),
                            epoch=np.int64(ck.get("epoch", -1)), **out)
        wmax = max(float(np.abs(v).max()) for v in out.values())
        print(f"pileup_fwd_hg001_{tag}: epoch {ck.get('epoch')}, max |W| {wmax:.3f}, gt argmax histogram", np.bincount(gt.numpy().argmax(1), minlength=21))


# ------------------------------------------------------------------------------------------
def group_hapfeat():
    from nanosnp_amd import host
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    import dataset_dev                       # noqa: E402  (reference module)
    out = {}
    for tag, L in (("p", 33), ("h", 11)):
        seq, bq, mq, hap, ref_row = host.synth_hap_planes(20260002 + L, 12, coverage=30, depth=90, length=L)
        # edge sites: empty site, HP1 missing, single read, padded rows only at the end, all deletions
        seq[0], bq[0], mq[0], hap[0] = -2, -2, -2, -2
        hap[1][hap[1] == 1] = 3
        seq[2, 1:], bq[2, 1:], mq[2, 1:], hap[2, 1:] = -2, -2, -2, -2
        seq[3][seq[3] > 0] = -1; bq[3][seq[3] == -1] = 0
        feats = np.stack([dataset_dev.get_frequency_feature(seq[i], bq[i], mq[i], hap[i])
                          for i in range(seq.shape[0])])
        assert feats.shape == (12, 104, L) and feats.dtype == np.float64
        out.update({f"{tag}_seq": seq.astype(np.int8), f"{tag}_bq": bq.astype(np.int8),
                    f"{tag}_mq": mq.astype(np.int8), f"{tag}_hap": hap.astype(np.int8),
                    f"{tag}_ref": ref_row.astype(np.int8), f"{tag}_feat": feats})
    np.savez_compressed(os.path.join(GOLD, "hap_features.npz"), **out)
    print("hap_features: ok")


def group_hapfwd():
    import torch
    from nanosnp_amd import host
    from oracle import oracle
    from tests.helpers import hap_weight_names, seeded_hap_weights
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    from model_dev import LSTMNetwork       # noqa: E402  (reference module)
    from utils import AttrDict              # noqa: E402
    # (the third set uses the scaled seeded weights of the two-stage fixture: genotypes differ from site to site and the
    #  probabilities span 0.4 .. 0.9, so a wrong feature or weight layout cannot hide behind a constant output)
    for H, N, seed, kw, name in ((32, 24, 11, {}, "hap_fwd_h32"), (256, 40, 12, {}, "hap_fwd_h256"),
                                 (256, 48, 13, {"ih_scale": 0.03, "head_scale": 120.0}, "hap_fwd_h256x")):
        cfg = AttrDict({"model": {"pileup_dim": 105, "haplotype_dim": 105, "pileup_length": 33,
                                  "haplotype_length": 11, "hidden_size": H, "lstm_layers": 3,
                                  "gt_num_class": 10, "zy_num_class": 3, "dropout": 0.1}})
        m = LSTMNetwork(cfg)
        ws = seeded_hap_weights(seed, H=H, **kw)
        sd = {k: torch.from_numpy(w) for k, w in zip(hap_weight_names(), ws)}
        missing = m.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys and all("crit" in k for k in missing.missing_keys), missing
        m.eval()
        planes_p = host.synth_hap_planes(100 + seed, N, 30, 90, 33)
        planes_h = host.synth_hap_planes(200 + seed, N, 30, 90, 11)
        xp = oracle.hap_features_batch(*planes_p)     # oracle features are pinned by hapfeat
        xh = oracle.hap_features_batch(*planes_h)
        with torch.no_grad():
            gt, zy = m.predict(torch.from_numpy(xp), torch.from_numpy(xh))   # predict_dev.py:35-39
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), xp=xp, xh=xh,
                            gt=gt.numpy(), zy=zy.numpy(), seed=seed)
        print(name, ": gt argmax", gt.numpy().argmax(1), "max p %.2f .. %.2f" % (gt.numpy().max(1).min(), gt.numpy().max(1).max()))


def group_cat():
    """legacy CatModel.predict (HaplotypeModel/model.py:332-358) with seeded weights -> tests/golden/cat_fwd.npz"""
    import torch
    from tests.helpers import cat_weight_names, seeded_cat_weights, synth_cat_groups
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    from model import CatModel              # noqa: E402  (reference module)
    seed, N = 21, 64
    m = CatModel(nc0=5, nc1=5, nc2=2, nclass=10, nh=256)     # predict.py:78-86 / config_prev/cat45.yaml
    ws = seeded_cat_weights(seed)
    sd = {k: torch.from_numpy(w) for k, w in zip(cat_weight_names(), ws)}
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("num_batches_tracked" in k or "crit" in k for k in res.missing_keys), res
    m.eval()                                                  # predict.py:28
    g0, g1 = synth_cat_groups(300 + seed, N)
    with torch.no_grad():
        gt = m.predict(torch.from_numpy(g0), torch.from_numpy(g1), None, None).numpy()
    np.savez_compressed(os.path.join(GOLD, "cat_fwd.npz"), g0=g0.astype(np.int8), g1=g1.astype(np.int8), gt=gt, seed=seed)
    print("cat_fwd: argmax", gt.argmax(1), "max p", gt.max(1).round(3))


def group_vcf():
    """pileup.vcf exactly as PileupModel/predict.py:37-195 writes it: the reference's own predict()
    is run on CPU with the shipped weights; only PredictDataset (PyTables reader) is replaced by an
    in-memory stand-in with the same four fields (dataset.py:118-149).  Also haplotype.csv rows
    as HaplotypeModel/predict_dev.py:40-47 formats them."""
    import gzip as _gz
    import torch
    import yaml
    from torch.utils.data import Dataset
    from nanosnp_amd import host
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "PileupModel"))
    import predict as ref_predict          # noqa: E402  (reference module)
    from model import LSTMNetwork          # noqa: E402
    from utils import AttrDict             # noqa: E402
    cfg = AttrDict(yaml.load(open(os.path.join(REF, "PileupModel/config/ont_pileup.yaml")), Loader=yaml.FullLoader))
    m = LSTMNetwork(cfg.model)
    ck = torch.load(os.path.join(REF, "PileupModel/models/ont_pileup.chkpt"), map_location="cpu", weights_only=False)
    m.encoder.load_state_dict(ck["encoder"]); m.forward_layer.load_state_dict(ck["forward_layer"]); m.eval()

    # sites: the two encode fixtures' .pd files (real window tensors with positions / ref bases)
    xs, names, poss, refs = [], [], [], []
    for tag in ("g1", "adv"):
        pd = _gz.open(os.path.join(GOLD, f"encode_{tag}.pd.gz")).read()
        x, nm, pos, refb = host.pd_parse(pd)
        xs.append(x); names += nm; poss.append(pos); refs.append(refb)
    x = np.concatenate(xs); pos = np.concatenate(poss); refb = np.concatenate(refs)

    class FakeDataset(Dataset):             # stands in for PredictDataset(datapath) (dataset.py:118-149)
        def __init__(self, datapath):
            pass
        def __getitem__(self, i):
            return names[i], pos[i], refb[i], x[i]
        def __len__(self):
            return len(x)

    ref_predict.PredictDataset = FakeDataset
    real_loader = ref_predict.DataLoader
    ref_predict.DataLoader = lambda ds, batch_size, shuffle, num_workers: real_loader(ds, batch_size=batch_size, shuffle=False, num_workers=0)
    with tempfile.TemporaryDirectory() as d:
        fai = os.path.join(d, "ref.fa.fai")
        open(fai, "w").write("chrS\t6100\t6\t60\t61\nchrT\t1600\t6\t60\t61\n")
        out = {}
        for bs in (1000, 64, 7):
            vcf = os.path.join(d, f"p{bs}.vcf")
            ref_predict.predict(m, ["x.bin"], fai, bs, vcf, torch.device("cpu"))
            out[f"vcf_bs{bs}"] = np.frombuffer(open(vcf, "rb").read(), np.uint8)
    with torch.no_grad():
        gt, zy = m.predict(torch.from_numpy(x).type(torch.FloatTensor))
    np.savez_compressed(os.path.join(GOLD, "pileup_vcf.npz"), x=x.astype(np.int16), pos=pos, refb=refb,
                        names=np.array(names), gt=gt.numpy(), zy=zy.numpy(), fai=np.frombuffer(open(fai, "rb").read() if False else b"chrS\t6100\t6\t60\t61\nchrT\t1600\t6\t60\t61\n", np.uint8), **out)
    for k, v in out.items():
        print(k, len(bytes(v).splitlines()), "lines")

    # ---- the same loop under NumPy 1.x scalar promotion (the reference's own environment: Dockerfile:13-29, py38) ----
    # NumPy >= 2.2 cannot be switched back to legacy promotion, so the float32 arrays the loop reads are widened to
    # float64 where they enter NumPy (Tensor.numpy()): every scalar expression of predict.py:66-194 then evaluates in
    # float64 on the exactly widened float32 values, which is what NumPy 1.x's value-based promotion does with
    # `1.0 - p`, `0 + cov[i]`, `-1 * cov.sum()` and `support / depth` (array-level max / argmax / integer-valued sums are
    # exact in either width).  Extra sites with saturated windows make the softmax maximum exactly 1.0f, the case the
    # two modes treat differently (QUAL 3010.3 vs a skipped site).
    xe = np.zeros((12, 33, 18), np.int32)
    rng = np.random.default_rng(5)
    for i in range(12):
        xe[i, :, 1] = -(80 + 5 * i); xe[i, :, 10] = -(60 + i); xe[i, 16, 0] = 70 + i; xe[i, 16, 9] = 75     # deep, clean het-like columns
        xe[i, 16, 1] = -(150 + 40 * i)
    x1 = np.concatenate([x, xe]); pos1 = np.concatenate([pos, 7000 + np.arange(12)]); refb1 = np.concatenate([refb, np.full(12, ord("C"), refb.dtype)])
    names1 = names + ["chrT"] * 12

    class FakeDataset1(Dataset):
        def __init__(self, datapath):
            pass
        def __getitem__(self, i):
            return names1[i], pos1[i], refb1[i], x1[i]
        def __len__(self):
            return len(x1)

    ref_predict.PredictDataset = FakeDataset1
    orig_numpy = torch.Tensor.numpy
    def widened(self, *a, **k):
        r = orig_numpy(self, *a, **k)
        return r.astype(np.float64) if r.dtype == np.float32 else r
    out1 = {}
    with tempfile.TemporaryDirectory() as d:
        fai = os.path.join(d, "ref.fa.fai")
        open(fai, "w").write("chrS\t6100\t6\t60\t61\nchrT\t1600\t6\t60\t61\n")
        for mode, patch in (("np2", False), ("np1", True)):
            torch.Tensor.numpy = widened if patch else orig_numpy
            try:
                for bs in (1000, 64, 7):
                    vcf = os.path.join(d, f"q{mode}{bs}.vcf")
                    ref_predict.predict(m, ["x.bin"], fai, bs, vcf, torch.device("cpu"))
                    out1[f"vcf_{mode}_bs{bs}"] = np.frombuffer(open(vcf, "rb").read(), np.uint8)
            finally:
                torch.Tensor.numpy = orig_numpy
    with torch.no_grad():
        gt1, zy1 = m.predict(torch.from_numpy(x1).type(torch.FloatTensor))
    print("sites with a softmax maximum of exactly 1.0f:", int((gt1.numpy().max(1) == 1.0).sum()), int((zy1.numpy().max(1) == 1.0).sum()))
    np.savez_compressed(os.path.join(GOLD, "pileup_vcf_modes.npz"), x=x1.astype(np.int16), pos=pos1, refb=refb1, names=np.array(names1),
                        gt=gt1.numpy(), zy=zy1.numpy(), fai=np.frombuffer(b"chrS\t6100\t6\t60\t61\nchrT\t1600\t6\t60\t61\n", np.uint8), **out1)
    for k, v in out1.items():
        print(k, len(bytes(v).splitlines()), "lines")


def group_next():
    """merge.py and select_hetesnp_homosnp.find_adjacent_sites run on the golden pileup.vcf plus a
    synthetic haplotype.csv"""
    import argparse, json
    z = np.load(os.path.join(GOLD, "pileup_vcf.npz"))
    vcf = bytes(z["vcf_bs1000"]).decode()
    rng = np.random.default_rng(17)
    labels = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT", "DD", "AD", "CD", "GD", "TD", "II", "AI", "CI", "GI", "TI", "ID"]
    rows = []
    for line in vcf.splitlines():
        if line.startswith("#"):
            continue
        f = line.split("\t")
        if rng.random() < 0.7:
            gt = labels[int(rng.integers(0, 21))] if rng.random() < 0.25 else labels[int(rng.integers(0, 10))]
            rows.append(f"{f[0]}\t{f[1]}\t{gt}\t{round(float(rng.uniform(5, 30)), 2)}")
    csv = "\n".join(rows) + "\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.vcf"), "w").write(vcf); open(os.path.join(d, "h.csv"), "w").write(csv)
        sys.path.insert(0, os.path.join(REF, "scripts"))
        import merge as ref_merge            # noqa: E402  (reference module)
        merged = {}
        for q in (15.0, 19.0):
            ref_merge.Run(argparse.Namespace(cat_predict=os.path.join(d, "h.csv"), output=os.path.join(d, "m.vcf"),
                                             pileup_vcf=os.path.join(d, "p.vcf"), quality=q))
            merged[str(q)] = open(os.path.join(d, "m.vcf")).read()
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    import select_hetesnp_homosnp as sel     # noqa: E402
    # contig_dict exactly as select_snp_multiprocess builds it (:82-104), then the per-chunk worker
    contig_dict = {}
    for row in vcf.splitlines():
        if row[0] == "#": continue
        c = row.strip().split(); g = c[9].split(":")[0].replace("|", "/"); ql = float(c[5])
        if (g == "0/0" and ql >= 19) or (g == "1/1" and ql >= 19): continue
        contig_dict.setdefault(c[0], {})[int(c[1])] = (g, ql)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        one = sel.find_adjacent_sites(contig_dict, ["chrS", "chrT"], 5, 19, 14)      # one chunk: the :226 bug drops chrS
        each = {}
        for ctg in ("chrS", "chrT"):
            each.update(sel.find_adjacent_sites(contig_dict, [ctg], 5, 19, 14))
    ser = lambda gd: {k: [[(it.position, it.homo_hete, it.info) for it in g] for g in v] for k, v in gd.items()}
    json.dump({"csv": csv, "merged": merged, "groups_one_chunk": ser(one), "groups_each": ser(each)},
              open(os.path.join(GOLD, "next_rows.json"), "w"))
    print("next_rows:", {k: len(v.splitlines()) for k, v in merged.items()}, {k: len(v) for k, v in ser(each).items()}, list(ser(one)))


def group_twostage_s2():
    """stage 2 of the two-stage fixture (own interpreter: PileupModel's modules collide with HaplotypeModel's): the
    reference's predict() on the chrS sites of encode_g1 alone, NumPy 1.x scalar promotion -> two_stage_s2.npz"""
    import gzip as _gz
    import torch
    import yaml
    from torch.utils.data import Dataset
    from nanosnp_amd import host
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "PileupModel"))
    import predict as ref_predict          # noqa: E402  (reference module)
    from model import LSTMNetwork          # noqa: E402
    from utils import AttrDict             # noqa: E402
    cfg = AttrDict(yaml.load(open(os.path.join(REF, "PileupModel/config/ont_pileup.yaml")), Loader=yaml.FullLoader))
    m = LSTMNetwork(cfg.model)
    ck = torch.load(os.path.join(REF, "PileupModel/models/ont_pileup.chkpt"), map_location="cpu", weights_only=False)
    m.encoder.load_state_dict(ck["encoder"]); m.forward_layer.load_state_dict(ck["forward_layer"]); m.eval()
    x, names, pos, refb = host.pd_parse(_gz.open(os.path.join(GOLD, "encode_g1.pd.gz")).read())

    class FakeDataset(Dataset):
        def __init__(self, datapath):
            pass
        def __getitem__(self, i):
            return names[i], pos[i], refb[i], x[i]
        def __len__(self):
            return len(x)

    ref_predict.PredictDataset = FakeDataset
    real_loader = ref_predict.DataLoader
    ref_predict.DataLoader = lambda ds, batch_size, shuffle, num_workers: real_loader(ds, batch_size=batch_size, shuffle=False, num_workers=0)
    orig_numpy = torch.Tensor.numpy
    def widened(self, *a, **k):              # NumPy 1.x scalar promotion (see group_vcf)
        r = orig_numpy(self, *a, **k)
        return r.astype(np.float64) if r.dtype == np.float32 else r
    with tempfile.TemporaryDirectory() as d:
        fai = os.path.join(d, "ref.fa.fai")
        open(fai, "w").write("chrS\t6100\t6\t60\t61\n")
        vcf = os.path.join(d, "p.vcf")
        torch.Tensor.numpy = widened
        try:
            ref_predict.predict(m, ["x.bin"], fai, 1000, vcf, torch.device("cpu"))
        finally:
            torch.Tensor.numpy = orig_numpy
        text = open(vcf, "rb").read()
    np.savez_compressed(os.path.join(GOLD, "two_stage_s2.npz"), vcf=np.frombuffer(text, np.uint8))
    print("two_stage_s2:", len(text.splitlines()), "lines")


def group_twostage():
    """BASELINE configs[3] in miniature, every stage by the reference's own code where it can run here
    (run_caller.sh:109-141):
      s2  PileupModel/predict.py predict()                               (group twostage_s2, shipped weights)
      s4  select_hetesnp_homosnp.py: contig_dict as :82-104, find_adjacent_sites(5, 19, 14)
          read matrices: the BAM pass of create_pileup_haplotype.py needs htslib -> generator G3 planes instead (inputs
          of the fixture), in the arrays write_to_bins.py stores
      s5  HaplotypeModel/predict_dev.py predict(): TestDataset + PileupFeature / HaplotypeFeature (reference rows, H3) +
          get_frequency_feature + model_dev.LSTMNetwork (seeded weights: trained ones are absent upstream) + the csv rows (H7);
          only tables.open_file is served from memory
      s6  scripts/merge.py Run
    -> two_stage.npz (planes, groups, haplotype.csv, final VCF)"""
    import gzip as _gz
    subprocess.run([sys.executable, os.path.abspath(__file__), "twostage_s2"], check=True)
    vcf = bytes(np.load(os.path.join(GOLD, "two_stage_s2.npz"))["vcf"]).decode()
    import argparse, io, contextlib
    import torch
    from nanosnp_amd import host
    from tests.helpers import hap_weight_names, seeded_hap_weights
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    import select_hetesnp_homosnp as sel     # noqa: E402  (reference module)
    contig_dict = {}
    for row in vcf.splitlines():
        if row[0] == "#": continue
        c = row.strip().split(); g = c[9].split(":")[0].replace("|", "/"); ql = float(c[5])
        if (g == "0/0" and ql >= 19) or (g == "1/1" and ql >= 19): continue
        contig_dict.setdefault(c[0], {})[int(c[1])] = (g, ql)
    with contextlib.redirect_stdout(io.StringIO()):
        groups = sel.find_adjacent_sites(contig_dict, ["chrS"], 5, 19, 14)["chrS"]
    G = len(groups)
    gpos = np.array([[it.position for it in g] for g in groups], np.int64)            # [G, 11], candidate in the middle
    cand = gpos[:, 5]
    # read planes of every group (generator G3), as write_to_bins.py stores them
    pp = host.synth_hap_planes(4100, G, 30, 90, 33)
    ph = host.synth_hap_planes(4200, G, 30, 90, 11)

    class Root:
        pass
    root = Root()
    root.pileup_sequences, root.pileup_baseq, root.pileup_mapq, root.pileup_hap = pp[0], pp[1], pp[2], pp[3]
    root.haplotype_sequences, root.haplotype_baseq, root.haplotype_mapq, root.haplotype_hap = ph[0], ph[1], ph[2], ph[3]
    root.candidate_positions = np.array([[f"chrS:{p}".encode()] for p in cand], dtype="S300")
    root.haplotype_positions = np.array([[f"chrS:{p}".encode() for p in row] for row in gpos], dtype="S300")

    class FakeFile:
        def __init__(self):
            self.root = root
        def close(self):
            pass
    sys.modules["tables"].open_file = lambda path, mode="r": FakeFile()
    import predict_dev                        # noqa: E402  (reference module)
    from model_dev import LSTMNetwork         # noqa: E402
    from utils import AttrDict                # noqa: E402
    cfg = AttrDict({"model": {"pileup_dim": 105, "haplotype_dim": 105, "pileup_length": 33, "haplotype_length": 11,
                              "hidden_size": 256, "lstm_layers": 3, "gt_num_class": 10, "zy_num_class": 3, "dropout": 0.1}})
    m = LSTMNetwork(cfg)
    # seeded weights scaled so that the 19 sites get different genotypes and half of them a confident call (QUAL >= 13)
    sd = {k: torch.from_numpy(w) for k, w in zip(hap_weight_names(), seeded_hap_weights(13, H=256, ih_scale=0.03, head_scale=120.0))}
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("crit" in k for k in res.missing_keys), res
    real_loader = torch.utils.data.DataLoader
    predict_dev.torch.utils.data.DataLoader = lambda ds, batch_size, shuffle, num_workers: real_loader(ds, batch_size=batch_size, shuffle=False, num_workers=0)
    orig_numpy = torch.Tensor.numpy
    def widened(self, *a, **k):
        r = orig_numpy(self, *a, **k)
        return r.astype(np.float64) if r.dtype == np.float32 else r
    fa_text = _gz.open(os.path.join(GOLD, "encode_g1.fa.gz")).read()
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "ref.fa"); open(fa, "wb").write(fa_text)
        bins = os.path.join(d, "bins"); os.makedirs(bins); open(os.path.join(bins, "chrS_0_6100.bin"), "w").close()
        csv_path = os.path.join(d, "haplotype.csv")
        torch.Tensor.numpy = widened
        try:
            with torch.no_grad():
                predict_dev.predict(m, bins, fa, 7, 33, 11, csv_path, torch.device("cpu"))       # a batch size that does not divide G
        finally:
            torch.Tensor.numpy = orig_numpy
            predict_dev.torch.utils.data.DataLoader = real_loader
        csv = open(csv_path).read()
        # reference rows exactly as PileupFeature / HaplotypeFeature build them (dataset_dev.py:106-120,150-162)
        import dataset_dev                    # noqa: E402
        from get_truth import load_reference_file      # noqa: E402
        refs = load_reference_file(fa)
        pf = dataset_dev.PileupFeature(FakeFile(), refs, 33)
        hf = dataset_dev.HaplotypeFeature(FakeFile(), refs, 11)
        open(os.path.join(d, "p.vcf"), "w").write(vcf)
        sys.path.insert(0, os.path.join(REF, "scripts"))
        import merge as ref_merge             # noqa: E402  (reference module)
        merged = {}
        for ql in (15.0, 19.0):            # s6 default threshold (scripts/merge.py:151) and the candidate threshold of s4
            ref_merge.Run(argparse.Namespace(cat_predict=csv_path, output=os.path.join(d, "m.vcf"), pileup_vcf=os.path.join(d, "p.vcf"), quality=ql))
            merged[ql] = open(os.path.join(d, "m.vcf")).read()
    np.savez_compressed(os.path.join(GOLD, "two_stage.npz"), vcf_s2=np.frombuffer(vcf.encode(), np.uint8), group_pos=gpos,
                        **{f"p_{n}": a.astype(np.int8) for n, a in zip(("seq", "bq", "mq", "hap"), pp[:4])},
                        **{f"h_{n}": a.astype(np.int8) for n, a in zip(("seq", "bq", "mq", "hap"), ph[:4])},
                        ref_rows_pileup=np.asarray(pf.candidate_reference_sequences, np.int32),
                        ref_rows_haplotype=np.asarray(hf.candidate_reference_sequences, np.int32),
                        csv=np.frombuffer(csv.encode(), np.uint8), merged_q15=np.frombuffer(merged[15.0].encode(), np.uint8),
                        merged_q19=np.frombuffer(merged[19.0].encode(), np.uint8))
    print("two_stage:", G, "groups;", len(csv.splitlines()), "csv rows;", {q: (len(t.splitlines()), sum(1 for l in t.splitlines() if "\tH\t" in l)) for q, t in merged.items()},
          "(lines, rows from the haplotype model)")



def group_haparrange():
    """H1: create_pileup_haplotype.single_group_pileup_haplotype_feature (:22-214) itself, driven by a stand-in for the
    pysam.AlignmentFile it iterates (only .pileup() columns with .pos / .n / .pileups[*].alignment.{query_name, has_tag,
    get_tag, query_sequence, query_qualities, mapping_quality}, .is_del, .is_refskip, .query_position are used, :39-47,90-134).
    The reads and groups come from tests/helpers.py (synth_reads(77), synth_groups(78)); the fixture holds, per group, the read x
    position matrices the function builds internally (re-derived here from the same reads) and the filtered, HP-sorted matrices
    it returns, plus its position lists -> hap_arrange.npz"""
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    import create_pileup_haplotype as cph        # noqa: E402  (reference module)
    from select_hetesnp_homosnp import SNPItem   # noqa: E402
    from tests.helpers import FakeSamfile, synth_groups, synth_reads
    reads = synth_reads(77)
    groups = [[SNPItem(c, p, "0/1", 10.0 if k == 5 else 20.0) for k, (c, p) in enumerate(g)] for g in synth_groups(78)]
    FakeSam = lambda: FakeSamfile(reads)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        out = cph.single_group_pileup_haplotype_feature(FakeSam(), groups, 10000, 5, 16)
    cand, hpos, hseq, hbq, hmq, hhap, maxh, pseq, pbq, pmq, phap, maxp = out
    assert len(cand) == len(groups) and maxh > 10 and maxp > 10, (len(cand), maxh, maxp)
    # the read x position matrices of :86-134 for the columns of each group, from the same reads
    b2i = {"A": 1, "C": 2, "G": 3, "T": 4}
    def matrices(cols):
        R = len(reads)
        m = [np.zeros((R, len(cols)), np.int32) for _ in range(4)]                  # seq, bq, mq, hap
        for r, rd in enumerate(reads):
            tag = rd["hp"] if rd["hp"] is not None else 3
            for j, p in enumerate(cols):
                if rd["a"] <= p <= rd["b"]:
                    o = rd["ops"][p - rd["a"]]
                    if o == "D":
                        m[0][r, j] = -1; m[3][r, j] = tag; m[2][r, j] = rd["mapq"]
                    else:
                        m[0][r, j] = b2i[o.upper()]; m[3][r, j] = tag; m[1][r, j] = rd["quals"][p - rd["a"]]; m[2][r, j] = rd["mapq"]
        # the reference's dictionaries only hold reads seen in some column of extend_positions: rows of reads that cover none
        # of them do not exist there; they are all-zero here and are dropped by the centre filter either way
        return m
    fx = {"n_groups": len(groups)}
    for g, grp in enumerate(groups):
        gp = [int(it.position) for it in grp]
        wp = list(range(gp[5] - 16, gp[5] + 17))
        for tag, cols, outs in (("h", gp, (hseq[g], hbq[g], hmq[g], hhap[g])), ("p", wp, (pseq[g], pbq[g], pmq[g], phap[g]))):
            ms = matrices(cols)
            for nm, a, o in zip(("seq", "bq", "mq", "hap"), ms, outs):
                fx[f"g{g}_{tag}_in_{nm}"] = a.astype(np.int16)
                fx[f"g{g}_{tag}_out_{nm}"] = np.asarray(o, np.int16)
    fx["candidates"] = np.array(cand)
    fx["haplotype_positions"] = np.array(hpos)
    fx["max_depths"] = np.array([maxh, maxp])
    np.savez_compressed(os.path.join(GOLD, "hap_arrange.npz"), **fx)
    print("hap_arrange:", len(groups), "groups, depths", [np.asarray(a).shape[0] for a in hseq], [np.asarray(a).shape[0] for a in pseq])



def _edge_planes(planes, base):
    """overwrites sites base .. base+31 of G3 planes with the edge cases: all padding, depth 1, saturated, all deletions"""
    seq, bq, mq, hap, ref_row = planes
    for i in range(base, base + 8):                                   # all-padding planes: every statistic is zero
        seq[i], bq[i], mq[i], hap[i] = -2, -2, -2, -2
    for i in range(base + 8, base + 16):                              # depth-1 sites
        seq[i, 1:], bq[i, 1:], mq[i, 1:], hap[i, 1:] = -2, -2, -2, -2
    for k, i in enumerate(range(base + 16, base + 24)):               # saturated features: every read the same base, top qualities
        seq[i] = 1 + k % 4; hap[i] = 1 + k % 3; bq[i] = 93; mq[i] = 60
    for i in range(base + 24, base + 32):                             # deletions only
        live = seq[i] != -2
        seq[i][live] = -1; bq[i][live] = 0


def group_hapfwd_large():
    """hap_fwd_large.npz: 256 sites through the reference's OWN chain dataset_dev.get_frequency_feature (+ reference row, as
    TestDataset.__getitem__ :337-349) -> model_dev.LSTMNetwork.predict (:133-143) with the scaled seeded weights (genotypes differ
    from site to site); the fixture holds the int8 read planes and the probabilities, so a test runs features AND forward"""
    import torch
    from nanosnp_amd import host
    from tests.helpers import hap_weight_names, seeded_hap_weights
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    import dataset_dev                      # noqa: E402  (reference modules)
    from model_dev import LSTMNetwork       # noqa: E402
    from utils import AttrDict              # noqa: E402
    N, seed = 256, 14
    cfg = AttrDict({"model": {"pileup_dim": 105, "haplotype_dim": 105, "pileup_length": 33, "haplotype_length": 11, "hidden_size": 256,
                              "lstm_layers": 3, "gt_num_class": 10, "zy_num_class": 3, "dropout": 0.1}})
    m = LSTMNetwork(cfg)
    ws = seeded_hap_weights(seed, H=256, ih_scale=0.03, head_scale=120.0)
    res = m.load_state_dict({k: torch.from_numpy(w) for k, w in zip(hap_weight_names(), ws)}, strict=False)
    assert not res.unexpected_keys and all("crit" in k for k in res.missing_keys), res
    m.eval()
    fx = {"seed": seed}
    xs = []
    for tag, L, sd in (("p", 33, 300), ("h", 11, 400)):
        planes = [a.copy() for a in host.synth_hap_planes(sd + seed, N, 30, 90, L)]
        _edge_planes(planes, 0)
        seq, bq, mq, hap, ref_row = planes
        feat = np.stack([np.concatenate([dataset_dev.get_frequency_feature(seq[i], bq[i], mq[i], hap[i]),
                                         ref_row[i][None].astype(np.float64)], 0) for i in range(N)])      # [N,105,L] float64
        xs.append(torch.from_numpy(feat).type(torch.FloatTensor))                                            # predict_dev.py:35-36
        for k, a in zip(("seq", "bq", "mq", "hap", "ref"), planes):
            assert a.min() >= -128 and a.max() <= 127
            fx[f"{tag}_{k}"] = a.astype(np.int8)
    with torch.no_grad():
        gt, zy = m.predict(xs[0], xs[1])
    fx["gt"] = gt.numpy(); fx["zy"] = zy.numpy()
    np.savez_compressed(os.path.join(GOLD, "hap_fwd_large.npz"), **fx)
    g = gt.numpy()
    print("hap_fwd_large: argmax histogram", np.bincount(g.argmax(1), minlength=10), "max p %.2f .. %.2f" % (g.max(1).min(), g.max(1).max()),
          "finite", np.isfinite(g).all())


def group_cat_large():
    """cat_fwd_large.npz: 256 sites through the reference's CatModel.predict (seeded weights), incl. empty tags, one-read tags and
    saturated group tensors"""
    import torch
    from tests.helpers import cat_weight_names, seeded_cat_weights, synth_cat_groups
    _stub_modules()
    sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
    from model import CatModel              # noqa: E402  (reference module)
    seed, N = 22, 256
    m = CatModel(nc0=5, nc1=5, nc2=2, nclass=10, nh=256)
    res = m.load_state_dict({k: torch.from_numpy(w) for k, w in zip(cat_weight_names(), seeded_cat_weights(seed))}, strict=False)
    assert not res.unexpected_keys and all("num_batches_tracked" in k or "crit" in k for k in res.missing_keys), res
    m.eval()
    g0, g1 = synth_cat_groups(400 + seed, N)
    for g in (g0, g1):
        g[0:4] = 0; g[0:4, :, :, 0] = -2; g[0:4, :20, :, 4] = 1; g[0:4, 20:, :, 4] = 2               # all padding
        for i in range(4, 8):                                                                        # one read per tag
            g[i, 1:20] = 0; g[i, 1:20, :, 0] = -2; g[i, 1:20, :, 4] = 1
            g[i, 21:40] = 0; g[i, 21:40, :, 0] = -2; g[i, 21:40, :, 4] = 2
        for k, i in enumerate(range(8, 12)):                                                         # saturated: one base, top qualities
            g[i, :, :, 0] = 1 + k; g[i, :, :, 1] = 60; g[i, :, :, 2] = 60; g[i, :, :, 3] = 1
    with torch.no_grad():
        gt = m.predict(torch.from_numpy(g0), torch.from_numpy(g1), None, None).numpy()
    np.savez_compressed(os.path.join(GOLD, "cat_fwd_large.npz"), g0=g0.astype(np.int8), g1=g1.astype(np.int8), gt=gt, seed=seed)
    print("cat_fwd_large: argmax histogram", np.bincount(gt.argmax(1), minlength=10), "max p %.3f .. %.3f" % (gt.max(1).min(), gt.max(1).max()))


GROUPS = {"haparrange": group_haparrange, "twostage": group_twostage, "twostage_s2": group_twostage_s2, "next": group_next, "vcf": group_vcf, "encode": group_encode, "encode_reader": group_encode_reader, "pileup": group_pileup, "pileup_ckpts": group_pileup_ckpts, "hapfeat": group_hapfeat,
          "hapfwd": group_hapfwd, "cat": group_cat, "hapfwd_large": group_hapfwd_large, "cat_large": group_cat_large}

if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit(f"{REF} is not mounted: goldens can only be generated in the development container")
    which = sys.argv[1:] or list(GROUPS)
    if len(which) == 1:
        GROUPS[which[0]]()
    else:
        for g in which:
            subprocess.run([sys.executable, os.path.abspath(__file__), g], check=True)
