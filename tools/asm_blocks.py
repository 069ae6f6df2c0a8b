#!/usr/bin/env python3
"""Instruction-class counts per basic block of one kernel's assembly (labels and branch targets shown),
for adding up what the hot path of a loop really issues.  usage: asm_blocks.py file.s kernel_symbol"""
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
text = open(path).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith(needle + ":"))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])


def classify(op):
    if op.startswith("v_pk_"): return "pk"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "ds"
    if op.startswith(("global", "flat", "buffer", "scratch")): return "vmem"
    if op == "s_waitcnt": return "wait"
    if op == "s_nop": return "nop"
    if op == "s_barrier": return "BARRIER"
    if op.startswith("s_"): return "salu"
    return "other"


label, d, tgt = "entry", {}, []
def flush():
    if d:
        print(f"{label:12s} " + " ".join(f"{k}={v}" for k, v in sorted(d.items())) + ("   -> " + ",".join(tgt) if tgt else ""))
for l in text[start + 1:end + 1]:
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        flush()
        label, d, tgt = m.group(1), {}, []
        continue
    m = re.match(r"\s+([a-z_0-9]+)(\s+(\S+))?", l)
    if m and not l.strip().startswith(";"):
        k = classify(m.group(1))
        d[k] = d.get(k, 0) + 1
        if m.group(1).startswith(("s_cbranch", "s_branch")):
            tgt.append(m.group(1)[2:] + ":" + (m.group(3) or ""))
flush()
