#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (imports /root/reference): scripts/merge.py Run and select_hetesnp_homosnp.find_adjacent_sites on random
call sets (VCF rows from tests/manual/ref_fuzz/vcf_rows.py's sites, haplotype rows over all 21 labels, positions the VCF lacks) against
nanosnp_amd.merge.merge_calls / select_groups.
    python tests/manual/ref_fuzz/merge_select.py FIRST_SEED END_SEED"""
import os, sys, tempfile, types, argparse, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from nanosnp_amd import host
from nanosnp_amd.merge import merge_calls, select_groups
import vcf_rows as fzv
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "scripts"))
import merge as ref_merge
sys.path.insert(0, os.path.join(REF, "HaplotypeModel"))
import select_hetesnp_homosnp as sel
labels = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT", "DD", "AD", "CD", "GD", "TD", "II", "AI", "CI", "GI", "TI", "ID"]
fai_text = "chrS\t6100\t6\t60\t61\nchrT\t1600\t6\t60\t61\n"
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(1000 + seed)
    names, pos, refb, x, gt, zy = fzv.make_sites(seed, 1200)
    # unique positions per contig (a VCF of one caller)
    key = sorted({(n, int(p)): i for i, (n, p) in enumerate(zip(names, pos))}.values())
    names = [names[i] for i in key]; pos, refb, x, gt, zy = pos[key], refb[key], x[key], gt[key], zy[key]
    order = np.lexsort((pos, np.array(names)))
    names = [names[i] for i in order]; pos, refb, x, gt, zy = pos[order], refb[order], x[order], gt[order], zy[order]
    x[:, 0, 17] = np.arange(len(pos)) % 4096; x[:, 1, 17] = np.arange(len(pos)) // 4096
    vcf = fzv.run_ours(names, pos, refb, x, gt, zy, 1000, fai_text, True).decode()
    rows = []
    for line in vcf.splitlines():
        if line.startswith("#"): continue
        f = line.split("\t")
        if rng.random() < 0.7:
            g = labels[int(rng.integers(0, 21))] if rng.random() < 0.3 else labels[int(rng.integers(0, 10))]
            rows.append(f"{f[0]}\t{f[1]}\t{g}\t{round(float(rng.uniform(0, 40)), 2)}")
    for _ in range(30):                          # haplotype calls at positions the VCF does not hold
        rows.append(f"chrS\t{int(rng.integers(5001, 6000))}\t{labels[int(rng.integers(0, 21))]}\t{round(float(rng.uniform(0, 40)), 2)}")
    csv = "\n".join(rows) + "\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.vcf"), "w").write(vcf); open(os.path.join(d, "h.csv"), "w").write(csv)
        for q in (13.0, 15.0, 19.0):
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    ref_merge.Run(argparse.Namespace(cat_predict=os.path.join(d, "h.csv"), output=os.path.join(d, "m.vcf"), pileup_vcf=os.path.join(d, "p.vcf"), quality=q))
                want = open(os.path.join(d, "m.vcf")).read()
            except Exception as e:
                want = "EXC " + type(e).__name__
            try:
                got = merge_calls(vcf, csv, q)
            except Exception as e:
                got = "EXC " + type(e).__name__
            ok = want == got; bad += not ok
            print(seed, "merge q", q, len(want.splitlines()) if not want.startswith("EXC") else want, "identical" if ok else "DIFFER", flush=True)
            if not ok and not want.startswith("EXC") and not got.startswith("EXC"):
                a, b = want.splitlines(), got.splitlines()
                import difflib
                n = 0
                for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, a, b, autojunk=False).get_opcodes():
                    if tag != "equal" and n < 3: print("   ", tag, a[i1:i2][:2], b[j1:j2][:2]); n += 1
            elif not ok: print("   want", want[:80], "got", got[:80])
    for (qt, adj, sq) in ((19, 5, 14), (25, 3, 10), (19, 1, 0)):
        contig_dict = {}
        for row in vcf.splitlines():
            if row[0] == "#": continue
            c = row.strip().split(); g = c[9].split(":")[0].replace("|", "/"); ql = float(c[5])
            if (g == "0/0" and ql >= qt) or (g == "1/1" and ql >= qt): continue
            contig_dict.setdefault(c[0], {})[int(c[1])] = (g, ql)
        with contextlib.redirect_stdout(io.StringIO()):
            each = {}
            for ctg in ("chrS", "chrT"):
                if ctg in contig_dict: each.update(sel.find_adjacent_sites(contig_dict, [ctg], adj, qt, sq))
        ser = {k: [[(it.position, it.homo_hete, it.info) for it in g] for g in v] for k, v in each.items()}
        mine = select_groups(vcf, qt, adj, sq, nthreads=1, reference_bug=False)
        mine = {k: [[tuple(it) for it in g] for g in v] for k, v in mine.items()}
        ok = ser == mine; bad += not ok
        print(seed, "groups", qt, adj, sq, {k: len(v) for k, v in ser.items()}, "identical" if ok else "DIFFER", flush=True)
print("bad", bad)
