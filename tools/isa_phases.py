#!/usr/bin/env python3
"""Instruction counts per phase of the fused kernel's batch loop, from the assembly of a build with phase marks.

    make asm EXTRA=-DMFCC_PHASE_MARKS          (writes build/mfcc_amd.s)
    python3 tools/isa_phases.py build/mfcc_amd.s [kernel symbol]  > profiles/rN_isa_phases.txt

The kernel source carries MFCC_PHASE("...") marks (asm comments) where a phase begins and at the head of every
branch that the steady state of the benchmark does not take ("cold: ...").  The assembly is cut into basic
blocks; a block that contains a mark belongs to that phase from the mark on, a block without one inherits the
phase of the block that falls through or branches into it (first in program order), so out-of-line cold blocks
stay with their "cold" mark and the hot path keeps its phase across them.  Counts are static: a phase that runs
several times per batch (the filterbank slots, the group ends) is listed once per code copy and summed, with the
number of copies; the dynamic totals per launch are the SQ_INSTS_* counters of the PMC summary."""
import re
import sys
from collections import OrderedDict, defaultdict

path = sys.argv[1] if len(sys.argv) > 1 else "build/mfcc_amd.s"
sym = sys.argv[2] if len(sys.argv) > 2 else "_ZN4mfcc12fused_kernelILi0ELi13EsNS_9FrameGeomILi400ELi160EEEEEvNS_11FusedParamsE"
text = open(path).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])


def classify(op):
    if op.startswith("v_pk_"): return "pk_f32"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_log", "v_sqrt", "v_rcp", "v_exp")): return "trans"
    if op.endswith("_dpp") or "dpp" in op: return "valu"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_bpermute"): return "bperm"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_rd"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "lds_wr"
    if op.startswith(("global_load", "flat_load", "buffer_load")): return "vm_rd"
    if op.startswith(("global_store", "flat_store", "buffer_store")): return "vm_wr"
    if op == "s_waitcnt": return "wait"
    if op == "s_nop": return "nop"
    if op == "s_barrier": return "barrier"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


# basic blocks: (label, [ (kind, payload) ... ], falls_through, branch targets)
blocks, cur = [], {"label": "entry", "items": [], "targets": [], "fall": True}
for l in text[start + 1:end + 1]:
    m = re.match(r"(\.LBB\d+_\d+):", l) or re.match(r"; (%bb\.\d+):", l)   # labelled blocks and fall-through-only blocks
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1), "items": [], "targets": [], "fall": True}
        continue
    m = re.search(r";\s*MFCC_PHASE\s+(.*)$", l)
    if m:
        cur["items"].append(("mark", m.group(1).strip()))
        continue
    m = re.match(r"\s+([a-z_0-9]+)(\s+(\S+))?", l)
    if m and not l.strip().startswith(";"):
        op = m.group(1)
        cur["items"].append(("op", op))
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["targets"].append(m.group(3))
            if op == "s_branch":
                cur["fall"] = False
blocks.append(cur)
index = {b["label"]: i for i, b in enumerate(blocks)}
entry_phase = [None] * len(blocks)
entry_phase[0] = "0 kernel set-up (once per tile)"
counts = OrderedDict()
copies = defaultdict(int)
# propagate in program order (two sweeps cover backward edges into unmarked blocks)
for sweep in range(2):
    for i, b in enumerate(blocks):
        ph = entry_phase[i]
        for kind, val in b["items"]:
            if kind == "mark":
                ph = val
        succ = [index[t] for t in b["targets"] if t in index] + ([i + 1] if b["fall"] and i + 1 < len(blocks) else [])
        for j in succ:
            if entry_phase[j] is None:
                entry_phase[j] = ph
for i, b in enumerate(blocks):
    ph = entry_phase[i] or "unreached"
    for kind, val in b["items"]:
        if kind == "mark":
            ph = val
            copies[ph] += 1
        else:
            counts.setdefault(ph, defaultdict(int))[classify(val)] += 1


def key(name):
    m = re.match(r"(\d+)", name)
    return (0, int(m.group(1))) if m else (1, name)


cols = ["pk_f32", "valu", "trans", "mfma", "lds_rd", "lds_wr", "bperm", "vm_rd", "vm_wr", "salu", "smem", "wait", "nop", "barrier"]
print(f"# static instruction counts per phase, kernel {sym}")
print(f"# {'phase':78s} copies " + " ".join(f"{c:>7s}" for c in cols))
tot = defaultdict(int)
for ph in sorted(counts, key=key):
    d = counts[ph]
    print(f"{ph:80s} {max(copies.get(ph, 1), 1):6d} " + " ".join(f"{d.get(c, 0):7d}" for c in cols))
    if re.match(r"\d+ ", ph) and not ph.startswith("0 "):
        for c in cols:
            tot[c] += d.get(c, 0)
print(f"{'hot path of one batch, all code copies of phases 7 and 8 (a wave runs 5-6 of 6 and 1-2 of 6)':80s}        " + " ".join(f"{tot[c]:7d}" for c in cols))
