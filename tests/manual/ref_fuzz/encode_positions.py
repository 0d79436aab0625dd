#!/usr/bin/env python3
"""DEVELOPMENT CONTAINER ONLY (runs oracle/_ref): position gaps, backward steps, duplicate lines, zero-depth lines, extra columns in the
mpileup text - the compiled reference against oracle.mpileup_to_pd and against the product's host parser + the oracle's array path.
    python tests/manual/ref_fuzz/encode_positions.py FIRST_SEED END_SEED"""
import os, sys, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from nanosnp_amd import host
from oracle import oracle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    M = 3000
    cols = host.synth_columns(500 + seed, M, coverage=20, het_rate=0.2)
    seq = np.concatenate([cols.ref, np.frombuffer(b"ACGT" * 25, np.uint8)]).copy()
    seq[rng.random(seq.size) < 0.04] |= 0x20
    lines = cols.mpileup_text("chrQ").split(b"\n")[:-1]
    out = []
    i = 0
    while i < M:
        u = rng.random()
        if u < 0.01: i += int(rng.integers(1, 40)); continue            # gap
        if u < 0.02 and i > 50: i -= int(rng.integers(1, 40)); continue       # going backwards
        if u < 0.03: out.append(lines[i])                               # duplicate line
        if u < 0.04:                                                    # zero-depth placeholder line
            f = lines[i].split(b"\t"); out.append(b"\t".join([f[0], f[1], b"N", b"0", b"*", b"*"])); i += 1; continue
        if u < 0.05:                                                    # extra columns behind the sixth
            out.append(lines[i] + b"\textra\tcolumns"); i += 1; continue
        out.append(lines[i]); i += 1
    tmp = tempfile.mkdtemp()
    fa = os.path.join(tmp, "ref.fa"); host.write_fasta(fa, "chrQ", seq)
    pile = os.path.join(tmp, "pile"); os.mkdir(pile)
    text = b"\n".join(out) + b"\n"
    open(os.path.join(pile, "chrQ.mpileup"), "wb").write(text)
    refdir = os.path.join(ROOT, "oracle", "_ref")
    r1 = subprocess.run([os.path.join(refdir, "DNA_CreateCanSnpTensor"), "-reference", fa, "-chr_pileup_dir", pile, "-output_dir", os.path.join(tmp, "tensor"), "-min_af", "0.12", "-snp_min_af", "0.12",
                    "-indel_min_af", "0.12", "-min_coverage", "6", "-flanking_base", "16", "-num_threads", "1", "chrQ"], capture_output=True)
    r2 = subprocess.run([os.path.join(refdir, "DNA_CreatePredictData"), "-chr_tensor_dir", os.path.join(tmp, "tensor"), "-reference", fa, "-output_dir", os.path.join(tmp, "pd"), "-num_threads", "1", "chrQ"], capture_output=True)
    if r1.returncode or r2.returncode:
        print(seed, "reference failed", r1.returncode, r1.stderr[-200:], r2.returncode); continue
    want = open(os.path.join(tmp, "pd", "chrQ.pd"), "rb").read()
    n = oracle.mpileup_to_pd(os.path.join(pile, "chrQ.mpileup"), bytes(seq), os.path.join(tmp, "o.pd"))
    got = open(os.path.join(tmp, "o.pd"), "rb").read()
    ok = got == want
    # the product's host parser + oracle array path
    try:
        pos, col_off, bases = host.mpileup_parse(text)
        ref = seq[pos - 1]
        counts, depth, flags = oracle.encode_columns(bases, col_off, ref)
        centers = oracle.select_sites(pos, flags)
        x = oracle.gather_windows(counts, centers)
        gx, names, gpos, gref = host.pd_parse(want)
        ok2 = np.array_equal(x, gx) and np.array_equal(pos[centers], gpos)
    except Exception as e:
        ok2 = repr(e)
    bad += (not ok) or (ok2 is not True)
    print(seed, "sites", n, want.count(b"\n"), "oracle text path", "identical" if ok else "DIFFER", "| host parse + array path:", ok2, flush=True)
print("bad", bad)
