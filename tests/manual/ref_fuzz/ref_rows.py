#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): dataset_dev.PileupFeature / HaplotypeFeature reference rows on a stand-in table
file - positions that are negative, zero, past the contig's end, contigs the reference dictionary lacks, lower-case / N / IUPAC
sequence - against host.haplotype_ref_rows and hap_pipeline.DeviceReference.rows (torch on the CPU).
    python tests/manual/ref_fuzz/ref_rows.py FIRST_SEED END_SEED"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
for name, attrs in (("ranger", {"Ranger": object}), ("ranger21", {"Ranger21": object}), ("tables", {"Filters": lambda **k: None})):
    m = types.ModuleType(name); [setattr(m, k, v) for k, v in attrs.items()]; sys.modules[name] = m
sys.path.insert(0, "/root/reference/HaplotypeModel")
import dataset_dev
from nanosnp_amd import host
from nanosnp_amd.hap_pipeline import DeviceReference
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    refs = {}
    for c in ("chr1", "chr2", "tiny"):
        n = 9 if c == "tiny" else int(rng.integers(200, 900))
        refs[c] = "".join(rng.choice(list("ACGTacgtNnRYKM"), n, p=[.2, .2, .2, .2, .03, .03, .03, .03, .02, .02, .01, .01, .01, .01]))
    N = 300
    def rand_pos():
        c = rng.choice(["chr1", "chr2", "tiny", "chrMissing"], p=[.45, .35, .1, .1])
        L = len(refs.get(c, "x" * 500))
        u = rng.random()
        p = int(rng.integers(1, L + 1)) if u < 0.6 else int(rng.integers(-40, 20)) if u < 0.8 else int(rng.integers(L - 20, L + 40))
        return f"{c}:{p}"
    cands = [rand_pos() for _ in range(N)]
    hpos = [[rand_pos() for _ in range(11)] for _ in range(N)]
    root = types.SimpleNamespace()
    z = np.zeros((N, 2, 33), np.int32); zh = np.zeros((N, 2, 11), np.int32)
    root.pileup_sequences = root.pileup_hap = root.pileup_baseq = root.pileup_mapq = z
    root.haplotype_sequences = root.haplotype_hap = root.haplotype_baseq = root.haplotype_mapq = zh
    root.candidate_positions = np.array([[c.encode()] for c in cands], dtype="S300")
    root.haplotype_positions = np.array([[p.encode() for p in row] for row in hpos], dtype="S300")
    ff = types.SimpleNamespace(root=root, close=lambda: None)
    want_p = np.asarray(dataset_dev.PileupFeature(ff, refs, 33).candidate_reference_sequences, np.int32)
    want_h = np.asarray(dataset_dev.HaplotypeFeature(ff, refs, 11).candidate_reference_sequences, np.int32)
    refs_b = {k: v.encode() for k, v in refs.items()}
    got_p = host.haplotype_ref_rows(refs_b, cands, 33)
    got_h = host.haplotype_ref_rows(refs_b, cands, 11, position_lists=hpos)
    ok = np.array_equal(got_p, want_p) and np.array_equal(got_h, want_h)
    # the device-side gather (torch on the CPU here): contig ids + positions -> rows
    dr = DeviceReference(refs_b, torch.device("cpu"))
    names = list(refs_b) + ["chrMissing"]
    def ids(strs):
        c = np.array([dr.names.index(s.split(":")[0]) if s.split(":")[0] in dr.names else -1 for s in strs], np.int32)
        p = np.array([int(s.split(":")[1]) for s in strs], np.int64)
        return torch.from_numpy(c), torch.from_numpy(p)
    cc, pp = ids(cands)
    win = (pp[:, None] + torch.arange(-16, 17)[None, :]) - 1
    dev_p = dr.rows(cc[:, None].expand(-1, 33), win).numpy()
    hc, hp = ids([s for row in hpos for s in row])
    dev_h = dr.rows(hc.view(N, 11), hp.view(N, 11) - 1).numpy()
    ok2 = np.array_equal(dev_p, want_p) and np.array_equal(dev_h, want_h)
    bad += (not ok) or (not ok2)
    print(seed, "host rows", "identical" if ok else "DIFFER", "| DeviceReference.rows", "identical" if ok2 else "DIFFER", flush=True)
print("bad", bad)
