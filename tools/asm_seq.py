#!/usr/bin/env python3
"""Instruction-class SEQUENCE of the loops of one kernel in a hipcc -S listing (run-length encoded):  asm_seq.py file.s <kernel substring> [min loop lines]
M = MFMA, T = transcendental, V = other VALU, L = LDS read, W = LDS write, G = global/scratch, S = scalar, w = s_waitcnt, B = barrier, n = s_nop"""
import re, sys

def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "T"
    if op.startswith("v_"): return "V"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "L"
    if op.startswith("ds_"): return "W"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "G"
    if op.startswith("s_waitcnt"): return "w"
    if op.startswith("s_barrier"): return "B"
    if op.startswith("s_nop"): return "n"
    if op.startswith("s_"): return "S"
    return "?"

path, key = sys.argv[1], sys.argv[2]
minlines = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\w+):", l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\w+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] >= minlines:
        seq = []
        for l2 in body[labels[m.group(1)]:i + 1]:
            m2 = re.match(r"^\s+([a-z_0-9]+)", l2)
            if m2 and not l2.strip().startswith((";", ".")):
                seq.append(cls(m2.group(1)))
        out, j = [], 0
        while j < len(seq):
            k = j
            while k < len(seq) and seq[k] == seq[j]: k += 1
            out.append(seq[j] + (str(k - j) if k - j > 1 else ""))
            j = k
        print(f"loop {m.group(1)} ({i - labels[m.group(1)]} lines): " + " ".join(out))
