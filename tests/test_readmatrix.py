"""Stage-4 read-matrix builder (nanosnp_amd/readmatrix.py) against create_pileup_haplotype.single_group_pileup_haplotype_feature
run on the same stand-in alignment file (tests/golden/hap_arrange.npz, tests/golden/make_golden.py haparrange)."""
import numpy as np
import pytest

from nanosnp_amd import readmatrix
from tests.helpers import FakeSamfile, golden, synth_groups, synth_reads


def _compare_with_reference(z, g, tag, planes, depth):
    """planes: (seq, bq, mq, hap) [D, L] padded with -2; the reference's rows as multisets inside each centre-HP group"""
    want = [z[f"g{g}_{tag}_out_{n}"].astype(np.int32) for n in ("seq", "bq", "mq", "hap")]
    L = want[0].shape[1]
    assert depth == want[0].shape[0]
    seq, bq, mq, hap = planes
    assert np.array_equal(hap[:depth, L // 2], want[3][:, L // 2])
    for o in planes:
        assert (o[depth:] == -2).all()
    got_rows = np.concatenate([seq[:depth], bq[:depth], mq[:depth], hap[:depth]], axis=1)
    want_rows = np.concatenate(want, axis=1)
    for hp in (1, 2, 3):
        a = got_rows[hap[:depth, L // 2] == hp]; b = want_rows[want[3][:, L // 2] == hp]
        assert sorted(map(bytes, a)) == sorted(map(bytes, b)), (g, tag, hp)


def test_read_matrices_and_oracle_arrangement_equal_the_reference_function():
    from oracle import oracle
    z = np.load(golden("hap_arrange.npz"))
    rm = readmatrix.read_matrices(FakeSamfile(synth_reads(77)), synth_groups(78), max_coverage=10000)
    sl = readmatrix.group_slices(rm)
    assert [s["candidate"] for s in sl] == z["candidates"].tolist()
    assert [s["haplotype_positions"] for s in sl] == z["haplotype_positions"].tolist()
    depths = {"h": [], "p": []}
    for g, s in enumerate(sl):
        for tag, key in (("h", "hap_cols"), ("p", "pile_cols")):
            ins = [m[:, s[key]] for m in (rm.seq, rm.baseq, rm.mapq, rm.hap)]
            want_depth = z[f"g{g}_{tag}_out_seq"].shape[0]
            oseq, obq, omq, ohap, depth = oracle.hap_arrange(*ins, want_depth + 3)
            _compare_with_reference(z, g, tag, (oseq, obq, omq, ohap), depth)
            depths[tag].append(depth)
    assert [max(depths["h"]), max(depths["p"])] == z["max_depths"].tolist()


def test_coverage_filter_and_foreign_bases():
    reads = synth_reads(77)
    groups = synth_groups(78)
    # create_pileup_haplotype.py:39-60: a group with a position deeper than max_coverage is dropped
    cov = {}
    for col in FakeSamfile(reads).pileup("c", 1, 900):
        cov[col.pos + 1] = col.n
    limit = max(cov[p] for _, p in groups[0]) - 1
    rm = readmatrix.read_matrices(FakeSamfile(reads), groups, max_coverage=limit)
    kept = [g for g in groups if all(cov[p] <= limit for _, p in g)]
    assert (rm is None and not kept) or [g for g in rm.groups] == kept
    assert groups[0] not in (rm.groups if rm else [])
    assert readmatrix.read_matrices(FakeSamfile(reads), groups, max_coverage=0) is None
    # a base outside ACGT at a wanted column: the reference's KeyError lands in its bare except and nothing is returned (:209-214)
    bad = [dict(r) for r in reads]
    k = next(i for i, r in enumerate(bad) if r["a"] <= 260 <= r["b"] and r["ops"][260 - r["a"]] != "D")
    bad[k]["ops"] = list(bad[k]["ops"]); bad[k]["ops"][260 - bad[k]["a"]] = "N"
    assert readmatrix.read_matrices(FakeSamfile(bad), groups, max_coverage=10000) is None
    # an HP tag other than 1 / 2 (3 = untagged): the reference's assert (:104) lands in the same bare except - nothing, not an exception
    odd = [dict(r) for r in reads]
    odd[0]["hp"] = 7
    assert readmatrix.read_matrices(FakeSamfile(odd), groups, max_coverage=10000) is None
    # a WINDOW column (not a group position) deeper than max_coverage: the first pass keeps the group, the assert of :98 then ends the chunk
    gpos = {p for g in groups for _, p in g}
    centre = groups[0][5][1]
    wcol = next(p for p in range(centre - 16, centre + 17) if p not in gpos)
    lim = max(cov.values())                                         # no column of the plain reads is deeper
    extra = [dict(name=f"x{k}", a=wcol, b=wcol, ops=["A"], hp=1, quals=[30], mapq=60) for k in range(lim - cov[wcol] + 1)]   # reads that cover only that column
    assert readmatrix.read_matrices(FakeSamfile(reads + extra), groups, max_coverage=lim) is None
    assert readmatrix.read_matrices(FakeSamfile(reads + extra[:-1]), groups, max_coverage=lim) is not None


@pytest.mark.gpu
def test_group_planes_on_the_device_equal_the_reference_function(gpu_ctx):
    import torch
    z = np.load(golden("hap_arrange.npz"))
    rm = readmatrix.read_matrices(FakeSamfile(synth_reads(77)), synth_groups(78), max_coverage=10000)
    Dh, Dp = (int(v) + 2 for v in z["max_depths"])
    cand, hpos, hplanes, pplanes, dh, dp = readmatrix.group_planes(gpu_ctx, rm, Dh, Dp)
    torch.cuda.synchronize()
    assert cand == z["candidates"].tolist() and hpos == z["haplotype_positions"].tolist()
    for g in range(len(cand)):
        _compare_with_reference(z, g, "h", tuple(p[g].cpu().numpy() for p in hplanes), int(dh[g].item()))
        _compare_with_reference(z, g, "p", tuple(p[g].cpu().numpy() for p in pplanes), int(dp[g].item()))
    # the planes feed the feature kernel as they are (write_to_bins.py layout)
    ref_row = torch.zeros((len(cand), 33), dtype=torch.int32, device="cuda")
    feat = gpu_ctx.hap_features(pplanes[0], pplanes[1], pplanes[2], pplanes[3], ref_row)
    assert feat.shape == (len(cand), 105, 33) and torch.isfinite(feat).all()
