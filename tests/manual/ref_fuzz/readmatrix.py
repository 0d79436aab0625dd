#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): create_pileup_haplotype.single_group_pileup_haplotype_feature driven by the
stand-in alignment file of tests/helpers.py on random read sets (few / many reads, short reads that leave group columns uncovered,
other HP values, refskips, every read untagged, groups at the contig's start) against nanosnp_amd.readmatrix + the oracle's arrangement:
candidates, position lists, depths, the HP-sorted centre column, the rows of every HP group as multisets.
    python tests/manual/ref_fuzz/readmatrix.py FIRST_SEED END_SEED"""
import os, sys, types, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object}), ("tables", {"Filters": lambda **k: None}), ("pysam", {})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
sys.path.insert(0, "/root/reference/HaplotypeModel")
import create_pileup_haplotype as cph
from select_hetesnp_homosnp import SNPItem
from nanosnp_amd import readmatrix
from oracle import oracle
from tests.helpers import FakeSamfile, synth_groups, synth_reads

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    n_reads = int(rng.choice([3, 12, 70, 200]))
    reads = synth_reads(seed, n_reads=n_reads)
    mode = seed % 5
    if mode == 1:
        for r in reads: r["hp"] = None
    elif mode == 2:
        for r in reads: r["hp"] = int(rng.choice([1, 2, 3, 7, 0]))
    elif mode == 3:                      # short reads
        for r in reads:
            k = int(rng.integers(20, 120)); r["b"] = min(r["b"], r["a"] + k); r["ops"] = r["ops"][:r["b"] - r["a"] + 1]; r["quals"] = r["quals"][:len(r["ops"])]
    centres = tuple(sorted(int(c) for c in rng.choice(np.arange(130, 760), int(rng.integers(1, 7)), replace=False)))
    groups = synth_groups(seed + 1, centres=centres)
    ref_groups = [[SNPItem(c, p, "0/1", 10.0 if k == 5 else 20.0) for k, (c, p) in enumerate(g)] for g in groups]
    maxcov = int(rng.choice([10000, 10000, n_reads // 2 + 1]))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            out = cph.single_group_pileup_haplotype_feature(FakeSamfile(reads), ref_groups, maxcov, 5, 16)
    except Exception as e:
        out = ("EXC", repr(e)[:80])
    try:
        rm = readmatrix.read_matrices(FakeSamfile(reads), groups, max_coverage=maxcov)
    except Exception as e:
        rm = ("EXC", repr(e)[:80])
    msg = "identical"
    if out is None or (isinstance(out, tuple) and len(out) and out[0] == "EXC") or (not isinstance(out, tuple)) or len(out[0]) == 0:
        empty_ref = True
    else:
        empty_ref = False
    if empty_ref:
        if not (rm is None or (not isinstance(rm, tuple) and len(readmatrix.group_slices(rm)) == 0)):
            msg = f"reference returned nothing ({out if isinstance(out, tuple) and out and out[0] == 'EXC' else 'empty'}), ours {type(rm).__name__}"
    elif rm is None or isinstance(rm, tuple):
        msg = f"reference returned {len(out[0])} groups, ours nothing {rm}"
    else:
        cand, hpos, hseq, hbq, hmq, hhap, maxh, pseq, pbq, pmq, phap, maxp = out
        sl = readmatrix.group_slices(rm)
        if [s["candidate"] for s in sl] != list(cand) or [s["haplotype_positions"] for s in sl] != [list(h) for h in hpos]:
            msg = "candidates / position lists differ"
        else:
            dmax = {"h": 0, "p": 0}
            for g, s in enumerate(sl):
                for tag, key, wants in (("h", "hap_cols", (hseq[g], hbq[g], hmq[g], hhap[g])), ("p", "pile_cols", (pseq[g], pbq[g], pmq[g], phap[g]))):
                    want = [np.asarray(w, np.int32) for w in wants]
                    L = want[0].shape[1]
                    ins = [m[:, s[key]] for m in (rm.seq, rm.baseq, rm.mapq, rm.hap)]
                    o = oracle.hap_arrange(*ins, want[0].shape[0] + 3)
                    depth = o[4]; dmax[tag] = max(dmax[tag], depth)
                    if depth != want[0].shape[0] or not np.array_equal(o[3][:depth, L // 2], want[3][:, L // 2]):
                        msg = f"group {g} {tag}: depth / sorted centre column differ ({depth} vs {want[0].shape[0]})"; break
                    got_rows = np.concatenate([x[:depth] for x in o[:4]], axis=1); want_rows = np.concatenate(want, axis=1)
                    for hp in np.unique(want[3][:, L // 2]):
                        a = got_rows[o[3][:depth, L // 2] == hp]; b = want_rows[want[3][:, L // 2] == hp]
                        if sorted(map(bytes, a)) != sorted(map(bytes, b)):
                            msg = f"group {g} {tag}: rows of HP {hp} differ"; break
                if msg != "identical": break
            if msg == "identical" and [dmax["h"], dmax["p"]] != [maxh, maxp]:
                msg = f"max depths differ {dmax} vs {maxh, maxp}"
    bad += msg != "identical"
    print(seed, "reads", n_reads, "mode", mode, "groups", len(groups), "maxcov", maxcov, "reference groups", "-" if empty_ref else len(out[0]), msg, flush=True)
print("bad", bad)
