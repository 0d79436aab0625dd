#!/usr/bin/env python3
"""Stress of the streamed text-to-VCF pipeline on adversarial contigs (the generator of the encode goldens: long indels, '^' swallowing
base-like characters, N reads, very deep and very shallow columns, position gaps, lower-case / N reference): the VCF of every chunk size
must equal, byte for byte, the one-chunk run; the one-chunk run itself is checked against the oracle chain (mpileup parse -> oracle encode
-> oracle site selection -> oracle forward on the selected windows: calls within 1e-4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import make_golden as mg
from nanosnp_amd import host
from nanosnp_amd.fixtures import load_pileup_weights
from nanosnp_amd.pileup_model import LSTMNetwork
from nanosnp_amd.pipeline import call_contig, stream_contig
from oracle import oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_cols = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
w = load_pileup_weights()
m = LSTMNetwork().load_weight_list(w)
bad = 0
for r in range(rounds):
    rng = np.random.default_rng(7700 + r + int(os.environ.get("NSNP_STRESS_SEED", "0")))
    seq = rng.choice(np.frombuffer(b"ACGTacgtNn", np.uint8), n_cols + 5000, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02]).astype(np.uint8)
    step = np.where(rng.random(n_cols) < 0.01, rng.integers(2, 40, n_cols), 1)
    if r % 2:                                      # odd rounds: positions that step BACK or repeat (main.cpp:174-178 resets its window at
        u = rng.random(n_cols)                     # every position that is not the previous one + 1), and indel alleles cut short by the
        step = np.where(u < 0.004, -rng.integers(1, 30, n_cols), np.where(u < 0.008, 0, step))     # end of their column
    pos = np.cumsum(step); pos = pos[(pos < seq.size) & (pos >= 1)]
    cols = [c.decode("latin-1") for c in mg.cut_allele_columns(rng, len(pos), seq[pos - 1])] if r % 2 else mg.adversarial_columns(rng, len(pos), seq[pos - 1])
    lines = []
    for p, c in zip(pos, cols):
        if not c:
            c = "*"
        lines.append(b"chrA\t%d\t%c\t%d\t%s\t%s\n" % (p, seq[p - 1], len(c), c.encode("latin-1"), b"I" * max(1, len(c) // 2)))
    if r % 3 == 2:                                 # every third round: the line reader's and split_line's corner cases (cpp_aux.cpp:44-59,
        for i in range(len(lines)):                # line_reader.cpp:95-127): runs of tabs, \r\n, no quality column, atoll-style positions
            u = rng.random()
            l = lines[i][:-1]
            f = l.split(b"\t")
            if u < 0.02: l = l.replace(b"\t", b"\t\t", int(rng.integers(1, 6)))
            elif u < 0.03: l = b"\t" + l + b"\t"
            elif u < 0.04: l = b"\t".join(f[:5])
            elif u < 0.05: l = l + b"\r"
            elif u < 0.06: f[1] = b"+000" + f[1] + b"xyz"; l = b"\t".join(f)
            elif u < 0.07: f[0] = b"another_name"; l = b"\t".join(f)
            lines[i] = l + b"\n"
    text = b"".join(lines)
    want = call_contig(m, text, "chrA", seq, chunk_bytes=1 << 40)
    for cb in (len(text) // 3, len(text) // 7, 50_000, 9_000):
        got = call_contig(m, text, "chrA", seq, chunk_bytes=cb)
        if not (bytes(got[0]) == bytes(want[0]) and got[1:] == want[1:]):
            bad += 1; print("round", r, "chunk", cb, "DIFFERS from the one-chunk run")
    # the text cut into columns on the host cores (nsnp_mpileup_parse_into) instead of on the device (nsnp_mpileup_tokenise): the same bytes
    for cb in (1 << 40, len(text) // 5):
        got = call_contig(m, text, "chrA", seq, chunk_bytes=cb, tokenise="host")
        if not (bytes(got[0]) == bytes(want[0]) and got[1:] == want[1:]):
            bad += 1; print("round", r, "chunk", cb, "host-parsed run DIFFERS from the device-tokenised one")
    # the one-chunk rows against the oracle chain
    rows = stream_contig(m, text, "chrA", seq, chunk_bytes=1 << 40).cpu().numpy()
    ppos, poff, pbases = oracle.mpileup_tokenise(text)            # the oracle's restatement of the reference's reader
    hp = host.mpileup_parse(text)
    if not (np.array_equal(hp[0], ppos) and np.array_equal(hp[1], poff) and np.array_equal(hp[2], pbases)):
        bad += 1; print("round", r, "the host tokeniser differs from the oracle's reader")
    oc, od, of = oracle.encode_columns(pbases, poff, seq[ppos - 1])
    centers = oracle.select_sites(ppos, of)
    if not (len(centers) == rows.shape[0] and np.array_equal(ppos[centers], rows[:, 0].astype(np.int64))):
        bad += 1; print("round", r, "selected sites differ from the oracle:", len(centers), rows.shape[0])
    elif len(centers):
        x = np.stack([oc[c - 16:c + 17] for c in centers])
        ogt, ozy = oracle.pileup_forward(w, x, nthreads=8)
        d = max(np.abs(ogt.max(1) - rows[:, 3]).max(), np.abs(ozy.max(1) - rows[:, 4]).max())
        near = np.sort(ogt, 1)[:, -1] - np.sort(ogt, 1)[:, -2] < 1e-4
        wrong = (ogt.argmax(1) != rows[:, 1]) & ~near
        if d > 1e-4 or wrong.any():
            bad += 1; print("round", r, "calls differ from the oracle: max |dp|", d, "wrong argmax", int(wrong.sum()))
    print(f"round {r}: {len(pos)} columns, {want[1]} sites, {want[2]} rows, {len(text)} bytes of text: ok" if not bad else f"round {r}: FAILURES so far {bad}")
sys.exit(1 if bad else 0)
