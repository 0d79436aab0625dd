"""merge.py and stage-4 group selection vs the reference functions' own output (tests/golden/make_golden.py next)."""
import json

import numpy as np

from nanosnp_amd.merge import merge_calls, select_groups
from tests.helpers import golden


def _load():
    z = np.load(golden("pileup_vcf.npz"))
    return bytes(z["vcf_bs1000"]).decode(), json.load(open(golden("next_rows.json")))


def test_merge_matches_reference_merge_py():
    vcf, g = _load()
    for q, want in g["merged"].items():
        assert merge_calls(vcf, g["csv"], float(q)) == want, q


def test_group_selection_matches_find_adjacent_sites():
    vcf, g = _load()
    norm = lambda d: {k: [[tuple(it) for it in grp] for grp in v] for k, v in d.items()}
    assert norm(select_groups(vcf, 19, 5, 14, nthreads=1, reference_bug=True)) == norm(g["groups_one_chunk"])
    assert norm(select_groups(vcf, 19, 5, 14, nthreads=1, reference_bug=False)) == norm(g["groups_each"])
    assert norm(select_groups(vcf, 19, 5, 14, nthreads=10, reference_bug=True)) == norm(g["groups_each"])   # one contig per chunk
    assert sum(len(v) for v in g["groups_each"].values()) > 10
