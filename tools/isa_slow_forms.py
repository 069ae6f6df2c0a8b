#!/usr/bin/env python3
"""Count, in one kernel of the device assembly, the vector instructions in forms that cost more than their plain
issue time on gfx950 (tools/ubench/vgpr_bank.hip): three source registers in one register bank (register number
mod 4) double a v_fma/v_fmac; a scalar-register source in a three-operand VOP3 adds about 40 %.

usage: isa_slow_forms.py build/mfcc_amd.s <kernel-name-substring>
"""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    slow = Counter()
    total = Counter()
    examples = {}
    for l in lines[start:end]:
        l = l.strip()
        m = re.match(r"(v_[a-z0-9_]+)\s+(.*)", l)
        if not m:
            continue
        op, rest = m.groups()
        rest = rest.split(";")[0]
        ops = [o.strip() for o in rest.split(",")]
        total[op] += 1
        srcs = ops[1:]
        if op.startswith(("v_fmac", "v_mac", "v_dot")):  # the destination is also read
            srcs = srcs + [ops[0]]
        vs = [int(re.match(r"v(\d+)$", o).group(1)) for o in srcs if re.match(r"v(\d+)$", o)]
        ss = [o for o in srcs if re.match(r"s(\d+)$|s\[", o)]
        if len(set(vs)) >= 3 and len({v % 4 for v in set(vs)}) == 1:
            slow[(op, "three sources in one bank")] += 1
            examples.setdefault((op, "three sources in one bank"), l)
        if ss and len(srcs) >= 3 and not op.startswith("v_pk") and not op.startswith("v_mfma") and not op.startswith("v_cndmask"):
            slow[(op, "scalar source in a three-operand form")] += 1
            examples.setdefault((op, "scalar source in a three-operand form"), l)
    for k, n in sorted(slow.items(), key=lambda kv: -kv[1]):
        print(f"{n:5d} of {total[k[0]]:5d}  {k[0]:24s} {k[1]:40s} e.g. {examples[k]}")
    print(f"vector instructions in the kernel: {sum(total.values())}")


if __name__ == "__main__":
    main()
