"""End-to-end predict loops on the GPU vs what the reference's predict() wrote."""
import re

import numpy as np
import pytest

from nanosnp_amd import host
from tests.helpers import golden

pytestmark = pytest.mark.gpu


def test_pileup_vcf_end_to_end(tmp_path, pileup_weights):
    """position_matrix -> pileup.vcf.  Probabilities differ from CPU torch by ~1e-7, which can move a
    QUAL by one unit in its second decimal; everything else must be identical."""
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.predict import predict_pileup
    z = np.load(golden("pileup_vcf_modes.npz"))          # the wrappers default to NumPy 1.x scalar promotion (score_mode 1)
    m = LSTMNetwork().load_weight_list(pileup_weights)
    out = tmp_path / "pileup.vcf"
    rows = predict_pileup(m, z["x"].astype(np.int32), list(z["names"]), z["pos"], z["refb"],
                          bytes(z["fai"]).decode(), str(out), batch_size=1000)
    got = out.read_bytes().decode().splitlines()
    want = bytes(z["vcf_np1_bs1000"]).decode().splitlines()
    assert len(got) == len(want) and rows == sum(1 for l in want if not l.startswith("#"))
    n_qual_diff = 0
    for g, w in zip(got, want):
        if g == w:
            continue
        gf, wf = g.split("\t"), w.split("\t")
        assert gf[:5] == wf[:5] and gf[6:9] == wf[6:9], (g, w)
        assert abs(float(gf[5]) - float(wf[5])) <= 0.0101, (g, w)
        gs, ws = gf[9].split(":"), wf[9].split(":")
        assert gs[0] == ws[0] and gs[2:] == ws[2:] and abs(int(gs[1]) - int(ws[1])) <= 1
        n_qual_diff += 1
    assert n_qual_diff <= len(want) // 20


def test_haplotype_csv_end_to_end(tmp_path, gpu_ctx):
    from nanosnp_amd.predict import predict_haplotype
    from oracle import oracle
    from tests.helpers import seeded_hap_weights
    ws = seeded_hap_weights(12, H=256)
    gpu_ctx.hap_load_weights(ws)
    n = 70
    pp = host.synth_hap_planes(31, n, 30, 90, 33)
    ph = host.synth_hap_planes(32, n, 30, 90, 11)
    cands = [f"chr{1 + i % 2}:{1000 + 7 * i}" for i in range(n)]
    out = tmp_path / "haplotype.csv"
    predict_haplotype(gpu_ctx, pp, ph, cands, str(out), batch_size=32)
    rows = out.read_text().splitlines()
    assert len(rows) == n
    ogt, _ = oracle.hap_forward(ws, oracle.hap_features_batch(*pp), oracle.hap_features_batch(*ph), nthreads=8)
    labels = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT"]
    for j, r in enumerate(rows):
        ctg, pos, gt, q = r.split("\t")
        assert f"{ctg}:{pos}" == cands[j] and re.fullmatch(r"\d+\.\d+", q)
        top2 = np.sort(ogt[j])[-2:]
        if top2[1] - top2[0] > 1e-3:                      # unambiguous argmax
            assert gt == labels[int(ogt[j].argmax())]
        want_q, ok = host.calculate_score(ogt[j].max())
        assert ok and abs(float(q) - want_q) <= 0.0101
    # the same planes handed over as int8: identical csv
    out8 = tmp_path / "haplotype8.csv"
    predict_haplotype(gpu_ctx, [a.astype(np.int8) for a in pp[:4]] + [pp[4]], [a.astype(np.int8) for a in ph[:4]] + [ph[4]],
                      cands, str(out8), batch_size=32)
    assert out8.read_bytes() == out.read_bytes()


def test_mpileup_to_vcf_pipeline(tmp_path, pileup_weights):
    """s1+s2 in one pass: mpileup text + FASTA -> VCF, against the rows the reference's predict() wrote
    for the same contig's sites (golden pileup_vcf.npz holds the chrS sites first, batch 1000)."""
    import gzip
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_variants
    z = np.load(golden("pileup_vcf.npz"))
    m = LSTMNetwork().load_weight_list(pileup_weights)
    fa = tmp_path / "ref.fa"
    fa.write_bytes(gzip.open(golden("encode_g1.fa.gz")).read())
    mp = tmp_path / "chrS.mpileup"
    mp.write_bytes(gzip.open(golden("encode_g1.mpileup.gz")).read())
    out = tmp_path / "pileup.vcf"
    rows = call_variants(m, [("chrS", str(mp))], str(fa), "chrS\t6100\t6\t60\t61\n", str(out))
    got = [l for l in out.read_text().splitlines() if not l.startswith("#")]
    want = [l for l in bytes(z["vcf_bs1000"]).decode().splitlines() if l.startswith("chrS\t")]
    # the golden batch also held chrT sites, so the batch-dependent fallback rows (ALT taken from other
    # sites' classes, predict.py:102-109) may differ; every other row must agree up to QUAL rounding
    assert rows == len(got) == len(want)
    diff = 0
    for g, w in zip(got, want):
        gf, wf = g.split("\t"), w.split("\t")
        assert gf[:4] == wf[:4] and gf[6] == wf[6]
        if gf[4] != wf[4]:
            diff += 1
            continue
        assert abs(float(gf[5]) - float(wf[5])) <= 0.0101
    assert diff <= 3


def test_streamed_pipeline_is_independent_of_the_chunk_size(tmp_path, pileup_weights):
    """pipeline.call_contig works the text off in chunks of whole lines (parse of chunk k + 1 on the host beside the device work of
    chunk k, 16 lines of halo re-parsed): the VCF is byte-identical whatever the chunk size - one chunk, a few, or chunks shorter
    than a window - on a contig with position gaps and lower-case / N reference bases (encode_g1) and on a 40 k-column G1 contig"""
    import gzip
    from nanosnp_amd import host
    from nanosnp_amd.pileup_model import LSTMNetwork
    from nanosnp_amd.pipeline import call_contig
    m = LSTMNetwork().load_weight_list(pileup_weights)
    text = gzip.open(golden("encode_g1.mpileup.gz")).read()
    fa = tmp_path / "ref.fa"
    fa.write_bytes(gzip.open(golden("encode_g1.fa.gz")).read())
    seq = host.fasta_load_contig(str(fa), "chrS")
    ref_rows, n_sites, n_rows = call_contig(m, text, "chrS", seq, chunk_bytes=1 << 30)
    assert n_sites > 50 and n_rows > 0
    for cb in (100_000, 20_000, 3_000, 700):
        st = {}
        rows, ns, nr = call_contig(m, text, "chrS", seq, chunk_bytes=cb, stats=st)
        assert (ns, nr) == (n_sites, n_rows) and rows == ref_rows, cb
        assert st["chunks"] >= len(text) // cb and st["columns"] == text.count(b"\n") and st["sites"] == n_sites
    cols = host.synth_columns(20261111, 40_000, coverage=30, het_rate=0.03)
    big = cols.mpileup_text_native("chrB")
    seqb = cols.ref.copy()
    want = call_contig(m, big, "chrB", seqb, chunk_bytes=1 << 30)
    got = call_contig(m, big, "chrB", seqb, chunk_bytes=256 << 10)
    assert got == want and want[1] > 500
