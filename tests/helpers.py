"""Shared test helpers: fixture paths, tolerance constants, the stand-in alignment file of the stage-4 tests.

The weight loaders / seeded weight generators live in nanosnp_amd/fixtures.py (bench.py, smoke() and tools/ use them
too and must not import the test package); they are re-exported here for the tests."""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# north_star: float outputs within 1e-4 abs of the reference PyTorch CPU path
PROB_ATOL = 1e-4

from nanosnp_amd.fixtures import (CAT_CHANNELS, PILEUP_WEIGHT_KEYS, cat_weight_names, cat_weight_shapes,  # noqa: F401,E402
                                  hap_weight_names, hap_weight_shapes, load_pileup_weights, seeded_cat_weights,
                                  seeded_hap_weights, synth_cat_groups)


def golden(name):
    return os.path.join(GOLDEN, name)


# ---- stand-in for the pysam.AlignmentFile that create_pileup_haplotype.single_group_pileup_haplotype_feature iterates ----------
# Only what the function touches (create_pileup_haplotype.py:39-47,90-134): .pileup() yielding columns with .pos / .n /
# .pileups[*].alignment.{query_name, has_tag, get_tag, query_sequence, query_qualities, mapping_quality}, .is_del, .is_refskip,
# .query_position.  A read is a dict {name, a, b (1-based reference span), ops (one of ACGTacgt / 'D' per reference position), hp (1, 2 or
# None), quals (one per op), mapq}.
def synth_reads(seed=77, n_reads=70, span=(1, 900)):
    rng = np.random.default_rng(seed)
    reads = []
    for r in range(n_reads):
        a = int(rng.integers(span[0], span[1] - 200)); b = int(min(span[1], a + rng.integers(150, 700)))
        ops = [("D" if rng.random() < 0.03 else "ACGT"[int(rng.integers(0, 4))]) for _ in range(a, b + 1)]
        if rng.random() < 0.1:
            ops = [o.lower() if o != "D" else o for o in ops]          # str.upper() at create_pileup_haplotype.py:121
        hp = [1, 2, None][int(rng.integers(0, 3))]
        quals = rng.integers(1, 60, len(ops)).tolist()
        reads.append(dict(name=f"r{r}", a=a, b=b, ops=ops, hp=hp, quals=quals, mapq=int(rng.integers(0, 61))))
    return reads


class _FakeAlignment:
    def __init__(self, rd):
        self.query_name = rd["name"]; self.rd = rd
        self.query_sequence = "".join(o for o in rd["ops"] if o != "D")
        self.query_qualities = [q for o, q in zip(rd["ops"], rd["quals"]) if o != "D"]
        self.mapping_quality = rd["mapq"]

    def has_tag(self, t):
        return t == "HP" and self.rd["hp"] is not None

    def get_tag(self, t):
        return self.rd["hp"]


class _FakePileupRead:
    def __init__(self, aln, k):
        self.alignment = aln
        self.is_del = aln.rd["ops"][k] == "D"
        self.is_refskip = False
        self.query_position = None if self.is_del else sum(1 for o in aln.rd["ops"][:k] if o != "D")


class _FakeColumn:
    def __init__(self, pos0, prs):
        self.pos = pos0; self.pileups = prs; self.n = len(prs)


class FakeSamfile:
    def __init__(self, reads):
        self.alns = [_FakeAlignment(rd) for rd in reads]

    def pileup(self, contig, start, end, min_base_quality=0, min_mapping_quality=0):
        for p in range(max(start - 3, 1), end + 4):                   # 1-based p; pysam yields columns around the region too
            prs = [_FakePileupRead(a, p - a.rd["a"]) for a in self.alns if a.rd["a"] <= p <= a.rd["b"]]
            if prs:
                yield _FakeColumn(p - 1, prs)


def synth_groups(seed=78, centres=(260, 300, 455, 610), ctg="c"):
    """[[(contig, position)] x 11] per centre: five support positions each side"""
    rng = np.random.default_rng(seed)
    groups = []
    for c in centres:
        left = sorted(rng.choice(np.arange(c - 120, c - 2), 5, replace=False).tolist())
        right = sorted(rng.choice(np.arange(c + 2, c + 120), 5, replace=False).tolist())
        groups.append([(ctg, int(p)) for p in left] + [(ctg, int(c))] + [(ctg, int(p)) for p in right])
    return groups


# ---- a QUAL / GQ that differs from the reference's row must be explained by a probability difference of at most dp ------------------
def qual_reachable(q_ref, p_zy, p_gt=None, refcall=False, score_mode=1, dp=1e-6):
    """PileupModel/predict.py:86-88: QUAL = calculate_score(zy_prob) on PASS rows, min(calculate_score(gt_prob), calculate_score(zy_prob)) on
    RefCall rows; calculate_score is monotone in the probability.  True when the reference's QUAL q_ref lies between the QUALs of
    probabilities dp below and dp above the GPU's own: the row differs only because the float32 probability differs by <= dp (the
    two sides then sit on either side of a rounding boundary of the two-decimal QUAL)."""
    from nanosnp_amd import host

    def q(pz, pg):
        s = host.calculate_score(np.float32(min(max(pz, 0.0), 1.0)), score_mode)[0]
        if refcall and pg is not None:
            s = min(s, host.calculate_score(np.float32(min(max(pg, 0.0), 1.0)), score_mode)[0])
        return s
    lo = q(float(p_zy) - dp, None if p_gt is None else float(p_gt) - dp)
    hi = q(float(p_zy) + dp, None if p_gt is None else float(p_gt) + dp)
    return lo - 1e-9 <= float(q_ref) <= hi + 1e-9
