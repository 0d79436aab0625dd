#!/usr/bin/env python3
"""Instruction mix of the innermost loops of one kernel in a hipcc -S listing:  asm_loop_mix.py file.s <substring of the kernel symbol>
Prints, per backward-branch loop (label .. s_cbranch to that label), the instruction count by class."""
import collections, re, sys

def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path, key = sys.argv[1], sys.argv[2]
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
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lo = labels[m.group(1)]
            mix = collections.Counter()
            ops = collections.Counter()
            for l2 in body[lo:i + 1]:
                m2 = re.match(r"^\s+([a-z_0-9]+)", l2)
                if m2 and not l2.strip().startswith((";", ".")):
                    mix[cls(m2.group(1))] += 1
                    if cls(m2.group(1)) in ("valu", "valu_pk", "trans"): ops[m2.group(1)] += 1
            print(f"loop {m.group(1)}: {i - lo} lines  " + "  ".join(f"{k}={v}" for k, v in sorted(mix.items())))
            print("   valu ops: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(30)))

main()
