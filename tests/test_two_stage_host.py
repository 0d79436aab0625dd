"""Host pieces of the two-stage chain (BASELINE configs[3]) against tests/golden/two_stage.npz, which the reference's own
code produced stage by stage (tests/golden/make_golden.py twostage): select_hetesnp_homosnp.find_adjacent_sites,
dataset_dev.PileupFeature / HaplotypeFeature (reference rows), predict_dev.predict (haplotype.csv rows), scripts/merge.py."""
import gzip

import numpy as np

from nanosnp_amd import host, merge
from tests.helpers import PROB_ATOL, golden, seeded_hap_weights

from nanosnp_amd.fixtures import TWO_STAGE_HAP_WEIGHTS      # noqa: E402  what make_golden.py twostage loads


def _fixture():
    z = np.load(golden("two_stage.npz"))
    vcf = bytes(z["vcf_s2"]).decode()
    csv = bytes(z["csv"]).decode()
    return z, vcf, csv


def test_group_selection_equals_find_adjacent_sites():
    z, vcf, _ = _fixture()
    groups = merge.select_groups(vcf, quality_threshold=19.0, adjacent_size=5, support_quality=14.0)
    got = np.array([[p for p, _, _ in g] for g in groups["chrS"]], np.int64)
    assert np.array_equal(got, z["group_pos"])
    assert got.shape[1] == 11 and np.all(np.diff(got, axis=1) > 0)


def test_reference_rows_equal_pileupfeature_and_haplotypefeature():
    """H3 pinned by the reference classes themselves (dataset_dev.py:92-172 run on the fixture's FASTA, which holds
    lower-case and N bases): 33-wide rows around the candidate and the 11 group positions"""
    z, _, _ = _fixture()
    fa = gzip.open(golden("encode_g1.fa.gz")).read()
    seq = np.frombuffer(b"".join(fa.splitlines()[1:]), np.uint8)
    refs = {"chrS": seq}
    gpos = z["group_pos"]
    cands = [f"chrS:{p}" for p in gpos[:, 5]]
    rp = host.haplotype_ref_rows(refs, cands, 33)
    rh = host.haplotype_ref_rows(refs, cands, 11, position_lists=[[f"chrS:{p}" for p in row] for row in gpos])
    assert np.array_equal(rp, z["ref_rows_pileup"]) and np.array_equal(rh, z["ref_rows_haplotype"])
    assert (z["ref_rows_pileup"] == 0).any()            # the fixture does exercise the "anything else -> 0" branch


def test_oracle_reproduces_the_reference_csv_rows():
    """H4-H7 through the oracle: features, forward and csv formatting against predict_dev.predict's output (GT exact,
    QUAL within one unit of its second decimal: CPU torch and the oracle differ by ~1e-7 in the probabilities)"""
    from oracle import oracle
    z, _, csv = _fixture()
    pp = [z[f"p_{n}"].astype(np.int32) for n in ("seq", "bq", "mq", "hap")] + [z["ref_rows_pileup"]]
    ph = [z[f"h_{n}"].astype(np.int32) for n in ("seq", "bq", "mq", "hap")] + [z["ref_rows_haplotype"]]
    gt, _ = oracle.hap_forward(seeded_hap_weights(**TWO_STAGE_HAP_WEIGHTS), oracle.hap_features_batch(*pp),
                               oracle.hap_features_batch(*ph), nthreads=8)
    labels = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT"]
    rows = csv.splitlines()
    assert len(rows) == gt.shape[0]
    for j, r in enumerate(rows):
        ctg, pos, g, q = r.split("\t")
        assert ctg == "chrS" and int(pos) == z["group_pos"][j, 5] and g == labels[int(gt[j].argmax())]
        want, ok = host.calculate_score(gt[j].max())
        assert ok and abs(float(q) - want) <= 0.0101


def test_merge_equals_merge_py_at_both_thresholds():
    z, vcf, csv = _fixture()
    for q in (15.0, 19.0):
        assert merge.merge_calls(vcf, csv, q) == bytes(z[f"merged_q{int(q)}"]).decode()
    assert sum("\tH\t" in l for l in bytes(z["merged_q19"]).decode().splitlines()) == 9
