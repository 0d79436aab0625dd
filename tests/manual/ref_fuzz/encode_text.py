#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (runs oracle/_ref): mutations of the mpileup TEXT that the reference's line reader and split_line
(cpp_aux.cpp:44-59: runs of tabs collapse; line_reader.cpp:95-127: \\r\\n, a lone \\r stays in the line) take without failing - doubled
tabs, leading / trailing tabs, an empty field in front of the bases (every later field moves up), no quality column, \\r\\n, a lone \\r
inside a field, "+12" / "12abc" / " 12" positions (atoll), no final newline - the compiled reference against oracle.mpileup_to_pd and
against the product's parsers (nsnp_mpileup_parse and nsnp_mpileup_parse_into, the chunked AVX2 one) + the oracle's array path.
    python tests/manual/ref_fuzz/encode_text.py FIRST_SEED END_SEED"""
import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
from nanosnp_amd import host
from oracle import oracle
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    M = 3000
    cols = host.synth_columns(900 + seed, M, coverage=20, het_rate=0.2)
    seq = np.concatenate([cols.ref, np.frombuffer(b"ACGT" * 25, np.uint8)]).copy()
    lines = cols.mpileup_text("chrQ").split(b"\n")[:-1]
    out = []
    for i, l in enumerate(lines):
        f = l.split(b"\t")
        u = rng.random()
        if u < 0.02: l = l.replace(b"\t", b"\t\t", int(rng.integers(1, 6)))
        elif u < 0.03: l = b"\t" + l
        elif u < 0.04: l = l + b"\t\t"
        elif u < 0.05: l = b"\t".join(f[:5])                                    # no quality column
        elif u < 0.06: l = l + b"\r"                                            # \r\n
        elif u < 0.07: f[5] = f[5][:1] + b"\r" + f[5][1:]; l = b"\t".join(f)    # a lone \r inside the qualities
        elif u < 0.08: f[1] = b"+" + f[1]; l = b"\t".join(f)
        elif u < 0.09: f[1] = f[1] + b"abc"; l = b"\t".join(f)
        elif u < 0.10: f[1] = b" " + f[1]; l = b"\t".join(f)
        elif u < 0.11: f[1] = b"000" + f[1]; l = b"\t".join(f)
        elif u < 0.12: f[2] = b""; l = b"\t".join(f)                            # empty ref column: depth is read as ref, bases as depth, QUALITIES as bases
        elif u < 0.13: f[0] = b"other_name"; l = b"\t".join(f)
        out.append(l)
    text = b"\n".join(out) + (b"\n" if seed % 2 else b"")
    tmp = tempfile.mkdtemp()
    fa = os.path.join(tmp, "ref.fa"); host.write_fasta(fa, "chrQ", seq)
    pile = os.path.join(tmp, "pile"); os.mkdir(pile)
    open(os.path.join(pile, "chrQ.mpileup"), "wb").write(text)
    refdir = os.path.join(ROOT, "oracle", "_ref")
    r1 = subprocess.run([os.path.join(refdir, "DNA_CreateCanSnpTensor"), "-reference", fa, "-chr_pileup_dir", pile, "-output_dir", os.path.join(tmp, "tensor"), "-min_af", "0.12", "-snp_min_af", "0.12",
                         "-indel_min_af", "0.12", "-min_coverage", "6", "-flanking_base", "16", "-num_threads", "1", "chrQ"], capture_output=True)
    r2 = subprocess.run([os.path.join(refdir, "DNA_CreatePredictData"), "-chr_tensor_dir", os.path.join(tmp, "tensor"), "-reference", fa, "-output_dir", os.path.join(tmp, "pd"), "-num_threads", "1", "chrQ"], capture_output=True)
    if r1.returncode or r2.returncode:
        print(seed, "reference failed", r1.returncode, r1.stderr[-200:], r2.returncode); continue
    want = open(os.path.join(tmp, "pd", "chrQ.pd"), "rb").read()
    n = oracle.mpileup_to_pd(os.path.join(pile, "chrQ.mpileup"), bytes(seq), os.path.join(tmp, "o.pd"))
    ok = open(os.path.join(tmp, "o.pd"), "rb").read() == want
    res = {}
    gx, names, gpos, gref = host.pd_parse(want)
    for name, parse in (("parse", lambda: host.mpileup_parse(text)), ("parse_range", lambda: host.mpileup_parse_range(np.frombuffer(text, np.uint8), 0, len(text)))):
        try:
            got = parse()
            pos, col_off, bases = got[0], got[1], got[2]
            pos = np.asarray(pos); col_off = np.asarray(col_off); bases = np.asarray(bases)
            ref = seq[pos - 1]
            counts, depth, flags = oracle.encode_columns(bases, col_off, ref)
            centers = oracle.select_sites(pos, flags)
            x = oracle.gather_windows(counts, centers)
            res[name] = bool(np.array_equal(x, gx) and np.array_equal(pos[centers], gpos))
        except Exception as e:
            res[name] = repr(e)[:120]
    bad += (not ok) or any(v is not True for v in res.values())
    print(seed, "sites", n, want.count(b"\n"), "oracle text path", "identical" if ok else "DIFFER", "| host parsers + array path:", res, flush=True)
print("bad", bad)
