"""The CPU oracle against the golden vectors generated from the reference itself
(tests/golden/make_golden.py), and against the reference's own compiled programs when oracle/_ref is
present.  Integer results bit-exact; probabilities within 1e-4 (measured ~1e-6)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from nanosnp_amd import host
from oracle import oracle
from tests.helpers import GOLDEN, PROB_ATOL, ROOT, golden, load_pileup_weights, seeded_hap_weights


def _load_encode_fixture(tag):
    text = gzip.open(golden(f"encode_{tag}.mpileup.gz")).read()
    fa = gzip.open(golden(f"encode_{tag}.fa.gz")).read()
    pd = gzip.open(golden(f"encode_{tag}.pd.gz")).read()
    seq = b"".join(fa.split(b"\n")[1:])
    return text, seq, pd


@pytest.mark.parametrize("tag", ["g1", "adv", "end", "cut", "pos", "rdr"])
def test_mpileup_to_pd_is_byte_identical_to_the_reference_output(tag, tmp_path):
    text, seq, pd = _load_encode_fixture(tag)
    mp = tmp_path / "x.mpileup"
    mp.write_bytes(text)
    n = oracle.mpileup_to_pd(str(mp), seq, str(tmp_path / "o.pd"))
    assert n == pd.count(b"\n")
    assert (tmp_path / "o.pd").read_bytes() == pd


@pytest.mark.parametrize("tag", ["g1", "adv", "end", "cut", "pos", "rdr"])
def test_array_path_matches_the_reference_tensors(tag):
    """encode_columns -> select_sites -> gather_windows == the [N,33,18] matrices in the .pd"""
    text, seq, pd = _load_encode_fixture(tag)
    pos, col_off, bases = oracle.mpileup_tokenise(text)           # the oracle's restatement of the reference's reader ...
    hpos, hoff, hbases = host.mpileup_parse(text)                 # ... and the host library's tokeniser: the same columns
    assert np.array_equal(hpos, pos) and np.array_equal(hoff, col_off) and np.array_equal(hbases, bases)
    ref = np.frombuffer(seq, np.uint8)[pos - 1]
    counts, depth, flags = oracle.encode_columns(bases, col_off, ref)
    centers = oracle.select_sites(pos, flags)
    x = oracle.gather_windows(counts, centers)
    gx, names, gpos, gref = host.pd_parse(pd)
    assert np.array_equal(x, gx)
    assert np.array_equal(pos[centers], gpos)
    assert np.array_equal(ref[centers] & 0xDF, gref)
    # the depth the .pd carries in its alt_info column ("depth-ALT cnt ...")
    pd_depth = np.array([int(l.split(b"\t")[2].split(b"-")[0]) for l in pd.splitlines()])
    assert np.array_equal(depth[centers], pd_depth)


@pytest.mark.parametrize("out_file,w_file", [("pileup_fwd.npz", None),
                                             ("pileup_fwd_hg001_e13.npz", "pileup_fwd_hg001_e13.npz"),
                                             ("pileup_fwd_hg001_e186.npz", "pileup_fwd_hg001_e186.npz")])
def test_pileup_forward_matches_reference_model_outputs(out_file, w_file):
    """all three checkpoints PileupModel/models/ ships, the same 256 inputs"""
    w = load_pileup_weights(golden(w_file) if w_file else None)
    z = np.load(golden(out_file))
    x = np.load(golden("pileup_fwd.npz"))["x"]
    gt, zy = oracle.pileup_forward(w, x.astype(np.int32), nthreads=4)
    assert np.abs(gt - z["gt"]).max() < PROB_ATOL
    assert np.abs(zy - z["zy"]).max() < PROB_ATOL
    assert np.array_equal(gt.argmax(1), z["gt"].argmax(1))
    assert np.abs(gt - z["gt"]).max() < 5e-6     # what a plain fp32 restatement achieves


@pytest.mark.parametrize("tag", ["p", "h"])
def test_hap_features_bit_exact_float64(tag):
    z = np.load(golden("hap_features.npz"))
    seq, bq, mq, hap, ref = [z[f"{tag}_{k}"].astype(np.int32) for k in ("seq", "bq", "mq", "hap", "ref")]
    for i in range(seq.shape[0]):
        f = oracle.hap_features(seq[i], bq[i], mq[i], hap[i], ref[i])
        assert np.array_equal(f[:104], z[f"{tag}_feat"][i]), i
        assert np.array_equal(f[104], ref[i])


@pytest.mark.parametrize("H", [32, 256])
def test_hap_forward_matches_reference_module_with_seeded_weights(H):
    z = np.load(golden(f"hap_fwd_h{H}.npz"))
    ws = seeded_hap_weights(int(z["seed"]), H=H)
    gt, zy = oracle.hap_forward(ws, z["xp"], z["xh"], H=H, nthreads=4)
    assert np.abs(gt - z["gt"]).max() < PROB_ATOL and np.abs(zy - z["zy"]).max() < PROB_ATOL
    assert np.abs(gt - z["gt"]).max() < 5e-6


def test_hap_forward_golden_with_site_dependent_outputs():
    """hap_fwd_h256x.npz: model_dev.LSTMNetwork.predict with the scaled seeded weights of the two-stage fixture on 48 sites whose
    genotypes differ (three classes, p_max 0.37 .. 0.90): a constant-output golden cannot hide a layout error, this one cannot"""
    z = np.load(golden("hap_fwd_h256x.npz"))
    ws = seeded_hap_weights(int(z["seed"]), H=256, ih_scale=0.03, head_scale=120.0)
    gt, zy = oracle.hap_forward(ws, z["xp"], z["xh"], H=256, nthreads=8)
    assert np.abs(gt - z["gt"]).max() < 2e-5 and np.abs(zy - z["zy"]).max() < 2e-5
    assert np.array_equal(gt.argmax(1), z["gt"].argmax(1)) and len(set(z["gt"].argmax(1).tolist())) >= 3


def test_hap_features_and_forward_large_golden_incl_edge_sites():
    """hap_fwd_large.npz: 256 sites through the reference's own get_frequency_feature + ref row + LSTMNetwork.predict (all-padding
    planes, depth-1 sites, saturated features, deletion-only sites among them); the oracle runs features AND forward"""
    z = np.load(golden("hap_fwd_large.npz"))
    ws = seeded_hap_weights(int(z["seed"]), H=256, ih_scale=0.03, head_scale=120.0)
    xs = [oracle.hap_features_batch(*[z[f"{t}_{k}"].astype(np.int32) for k in ("seq", "bq", "mq", "hap", "ref")], nthreads=8) for t in ("p", "h")]
    assert np.all(xs[0][:8, :104] == 0) and np.isfinite(xs[0]).all()              # all-padding planes: zero statistics
    gt, zy = oracle.hap_forward(ws, xs[0], xs[1], H=256, nthreads=8)
    assert np.abs(gt - z["gt"]).max() < 2e-5 and np.abs(zy - z["zy"]).max() < 2e-5
    agree = gt.argmax(1) == z["gt"].argmax(1)
    top2 = np.sort(z["gt"], 1)[:, -2:]
    assert np.all(agree | (top2[:, 1] - top2[:, 0] < 1e-4)) and len(set(z["gt"].argmax(1).tolist())) >= 4


def test_cat_forward_large_golden_incl_edge_sites():
    """cat_fwd_large.npz: 256 sites of CatModel.predict (empty tags, one-read tags, saturated tensors among them)"""
    from tests.helpers import seeded_cat_weights
    z = np.load(golden("cat_fwd_large.npz"))
    gt = oracle.cat_forward(seeded_cat_weights(int(z["seed"])), z["g0"], z["g1"], nthreads=8)
    assert np.abs(gt - z["gt"]).max() < 2e-5 and np.isfinite(gt).all()
    assert len(set(z["gt"].argmax(1).tolist())) >= 5


def test_cat_forward_matches_reference_module_with_seeded_weights():
    """legacy CatModel.predict (HaplotypeModel/model.py:332-358): golden from the reference module (tests/golden/make_golden.py cat)"""
    from tests.helpers import seeded_cat_weights
    z = np.load(golden("cat_fwd.npz"))
    ws = seeded_cat_weights(int(z["seed"]))
    gt = oracle.cat_forward(ws, z["g0"], z["g1"], nthreads=4)
    assert np.abs(gt - z["gt"]).max() < 1e-5
    assert len(set(z["gt"].argmax(1))) > 1            # the fixture discriminates between sites


def test_cat_groups_layout():
    """dataset.py:862-915 restated with numpy concatenation"""
    rng = np.random.default_rng(4)
    N, L = 5, 11
    t1 = [rng.integers(-2, 5, (N, 25, L)).astype(np.int32), rng.integers(0, 60, (N, 25, L)).astype(np.int32), rng.integers(0, 60, (N, 25, L)).astype(np.int32)]
    t2 = [rng.integers(-2, 5, (N, 31, L)).astype(np.int32), rng.integers(0, 60, (N, 31, L)).astype(np.int32), rng.integers(0, 60, (N, 31, L)).astype(np.int32)]
    g = oracle.cat_groups(t1, t2)
    r = np.concatenate([t1[0][:, :20], t2[0][:, :20]], 1)
    want = np.stack([r, np.concatenate([t1[1][:, :20], t2[1][:, :20]], 1), np.concatenate([t1[2][:, :20], t2[2][:, :20]], 1),
                     (r != -2).astype(int), np.concatenate([np.ones_like(t1[0][:, :20]), np.ones_like(t2[0][:, :20]) + 1], 1)], 3)
    assert np.array_equal(g, want.astype(np.float32))


def test_calculate_score_matches_python_formula():
    from math import e, log
    def ref(p):   # PileupModel/predict.py:31-34 verbatim formula
        tmp = max((-10 * log(e, 10)) * log(((1.0 - p) + 1e-300) / (p + 1e-300)) + 10, 0)
        return float(round(tmp, 2))
    rng = np.random.default_rng(1)
    ps = np.concatenate([rng.random(2000), [0.0, 1.0, 0.5, 1e-9, 1 - 1e-9, 0.0909090909, 0.999]])
    ps = np.concatenate([ps, ps.astype(np.float32).astype(np.float64)])
    for p in ps:
        assert oracle.calculate_score(p) == ref(p), p


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "DNA_CreateCanSnpTensor")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref not built (no reference tree)")
def test_oracle_vs_reference_binaries_on_a_fresh_contig(tmp_path):
    """Re-pins the oracle against the compiled reference on data that is NOT in the fixtures."""
    M = 40000
    cols = host.synth_columns(977, M, coverage=30)
    rng = np.random.default_rng(5)
    seq = np.concatenate([cols.ref, np.frombuffer(b"ACGT" * 25, np.uint8)]).copy()
    seq[rng.random(seq.size) < 0.04] |= 0x20
    seq[rng.random(seq.size) < 0.002] = ord("N")
    fa = str(tmp_path / "ref.fa")
    host.write_fasta(fa, "chrQ", seq)
    pile = tmp_path / "pile"; pile.mkdir()
    keep = np.ones(M, bool)
    for g in rng.integers(100, M - 100, 20):
        keep[g:g + int(rng.integers(1, 4))] = False
    lines = cols.mpileup_text("chrQ").split(b"\n")[:-1]
    (pile / "chrQ.mpileup").write_bytes(b"\n".join(l for i, l in enumerate(lines) if keep[i]) + b"\n")
    refdir = os.path.join(ROOT, "oracle", "_ref")
    subprocess.run([os.path.join(refdir, "DNA_CreateCanSnpTensor"), "-reference", fa, "-chr_pileup_dir", str(pile),
                    "-output_dir", str(tmp_path / "tensor"), "-min_af", "0.12", "-snp_min_af", "0.12",
                    "-indel_min_af", "0.12", "-min_coverage", "6", "-flanking_base", "16", "-num_threads", "1", "chrQ"],
                   check=True, capture_output=True)
    subprocess.run([os.path.join(refdir, "DNA_CreatePredictData"), "-chr_tensor_dir", str(tmp_path / "tensor"),
                    "-reference", fa, "-output_dir", str(tmp_path / "pd"), "-num_threads", "1", "chrQ"],
                   check=True, capture_output=True)
    want = (tmp_path / "pd" / "chrQ.pd").read_bytes()
    n = oracle.mpileup_to_pd(str(pile / "chrQ.mpileup"), bytes(seq), str(tmp_path / "o.pd"))
    assert n == want.count(b"\n") and n > 500
    assert (tmp_path / "o.pd").read_bytes() == want


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref not built (no reference tree)")
@pytest.mark.parametrize("seed", [31, 32])
def test_oracle_vs_reference_binaries_on_cut_alleles_and_random_bytes(tmp_path, seed):
    """Fresh columns of the kind behind tests/golden/encode_cut.*: indels the end of the column cuts short beside complete alleles with
    the same visible characters (tensor_maker.cpp:101 reads `advance` characters whatever the string holds), and printable bytes drawn
    at random under a loose grammar - the .pd of the compiled reference against the oracle's, byte for byte (alt_info included)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", golden("make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    rng = np.random.default_rng(seed)
    M = 3000
    seq = rng.choice(list(b"ACGTacgtN"), M + 100, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]).astype(np.uint8)
    cols = mg.cut_allele_columns(rng, M, seq)
    fa = str(tmp_path / "ref.fa")
    host.write_fasta(fa, "chrC", seq)
    pile = tmp_path / "pile"; pile.mkdir()
    (pile / "chrC.mpileup").write_bytes(b"".join(b"chrC\t%d\tN\t%d\t%s\t%s\n" % (i + 1, 1, c, b"I") for i, c in enumerate(cols) if c))
    refdir = os.path.join(ROOT, "oracle", "_ref")
    for snp, ind, mc in (("0.12", "0.12", 6), ("0.3", "0.05", 6), ("0.05", "0.4", 3)):     # (-snp_min_af and -indel_min_af apart: main.cpp:79-88)
        out_t, out_p = tmp_path / f"tensor{snp}{ind}", tmp_path / f"pd{snp}{ind}"
        subprocess.run([os.path.join(refdir, "DNA_CreateCanSnpTensor"), "-reference", fa, "-chr_pileup_dir", str(pile),
                        "-output_dir", str(out_t), "-min_af", "0.12", "-snp_min_af", snp,
                        "-indel_min_af", ind, "-min_coverage", str(mc), "-flanking_base", "16", "-num_threads", "1", "chrC"],
                       check=True, capture_output=True)
        subprocess.run([os.path.join(refdir, "DNA_CreatePredictData"), "-chr_tensor_dir", str(out_t),
                        "-reference", fa, "-output_dir", str(out_p), "-num_threads", "1", "chrC"],
                       check=True, capture_output=True)
        want = (out_p / "chrC.pd").read_bytes()
        n = oracle.mpileup_to_pd(str(pile / "chrC.mpileup"), bytes(seq), str(tmp_path / "o.pd"), min_af=float(snp), indel_min_af=float(ind), min_coverage=mc)
        assert n == want.count(b"\n") and n > 1000, (snp, ind)
        assert (tmp_path / "o.pd").read_bytes() == want, (snp, ind)


def _check_arranged_against_reference(arrange, z):
    """arrange(seq, bq, mq, hap, D) -> (oseq, obq, omq, ohap, depth) against the matrices the reference's
    single_group_pileup_haplotype_feature returned (tests/golden/hap_arrange.npz).  pandas' default sort leaves ties in an
    unspecified order, so what is compared is what is specified: the depth, the sorted centre-HP column, and the multiset of
    (seq, bq, mq, hap) rows inside each HP group; padding rows are -2; a cut keeps a prefix of the sorted rows."""
    for g in range(int(z["n_groups"])):
        for tag in ("h", "p"):
            ins = [z[f"g{g}_{tag}_in_{n}"].astype(np.int32) for n in ("seq", "bq", "mq", "hap")]
            want = [z[f"g{g}_{tag}_out_{n}"].astype(np.int32) for n in ("seq", "bq", "mq", "hap")]
            L = ins[0].shape[1]
            depth_ref = want[0].shape[0]
            D = depth_ref + 4
            oseq, obq, omq, ohap, depth = arrange(*ins, D)
            assert depth == depth_ref, (g, tag)
            assert np.array_equal(ohap[:depth, L // 2], want[3][:, L // 2])                   # sorted by the centre HP
            for o in (oseq, obq, omq, ohap):
                assert (o[depth:] == -2).all()
            got_rows = np.concatenate([oseq[:depth], obq[:depth], omq[:depth], ohap[:depth]], axis=1)
            want_rows = np.concatenate(want, axis=1)
            for hp in (1, 2, 3):
                a = got_rows[ohap[:depth, L // 2] == hp]; b = want_rows[want[3][:, L // 2] == hp]
                assert sorted(map(bytes, a)) == sorted(map(bytes, b)), (g, tag, hp)
            cseq, _, _, _, cdepth = arrange(*ins, max(1, depth // 2))                          # write_to_bins.py:54-61
            assert cdepth == max(1, depth // 2) and np.array_equal(cseq, oseq[:cdepth])


def test_hap_arrange_matches_the_reference_function():
    """H1 pinned by create_pileup_haplotype.single_group_pileup_haplotype_feature itself (run with a stand-in for the pysam
    file it iterates: tests/golden/make_golden.py haparrange): centre filter (:145-149,181-185) and HP sort (:158-165,193-200)"""
    _check_arranged_against_reference(oracle.hap_arrange, np.load(golden("hap_arrange.npz")))


def test_blocked_cpu_baseline_equals_the_plain_restatement(pileup_weights=None):
    """the cache-blocked AVX2 arrangement bench.py times as its CPU baseline computes the same function as the checker"""
    from tests.helpers import load_pileup_weights
    w = load_pileup_weights()
    z = np.load(golden("pileup_fwd.npz"))
    g0, z0 = oracle.pileup_forward(w, z["x"], nthreads=4)
    g1, z1 = oracle.pileup_forward(w, z["x"], nthreads=4, blocked=True)
    assert np.abs(g0 - g1).max() < 5e-6 and np.abs(z0 - z1).max() < 5e-6
    assert np.abs(g1 - z["gt"]).max() < 1e-5
    for n in (1, 31, 33, 65):                                   # ragged block sizes
        ga, _ = oracle.pileup_forward(w, z["x"][:n], nthreads=3, blocked=True)
        assert np.abs(ga - g0[:n]).max() < 5e-6
