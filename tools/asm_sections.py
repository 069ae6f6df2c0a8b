#!/usr/bin/env python3
"""Instruction-class counts of a kernel's assembly, split at workgroup barriers."""
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
text = open(path).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith(needle) and l.rstrip().endswith(("E:", "E: ; @" + needle[:-1])) or l.startswith(needle + ":"))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
lines = text[start:end]
bars = [i for i, l in enumerate(lines) if re.match(r"\s+s_barrier", l)]


def classify(op):
    if op.startswith("v_pk_"): return "pk"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_cndmask"): return "cnd"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "ds"
    if op.startswith("scratch"): return "scratch"
    if op.startswith(("global", "flat", "buffer")): return "vmem"
    if op == "s_waitcnt": return "wait"
    if op.startswith("s_"): return "salu"
    return "other"


prev = 0
for b in bars + [len(lines)]:
    d = {}
    for l in lines[prev:b]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if m:
            k = classify(m.group(1))
            d[k] = d.get(k, 0) + 1
    print(f"{prev:5d}-{b:5d} " + " ".join(f"{k}={v}" for k, v in sorted(d.items())))
    prev = b
