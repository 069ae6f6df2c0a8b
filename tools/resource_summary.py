#!/usr/bin/env python3
"""Per-kernel registers / scratch / occupancy from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(build/resource_usage.txt, written by `make asm`)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "build/resource_usage.txt"
cur, rows = None, []
for line in open(path):
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    print(f'{n[:110]:110s} vgpr {r.get("VGPRs", "?"):>3} sgpr {r.get("TotalSGPRs", "?"):>3} scratch {r.get("ScratchSize [bytes/lane]", "?"):>3} '
          f'occ {r.get("Occupancy [waves/SIMD]", "?")} spill s{r.get("SGPRs Spill", "?")}/v{r.get("VGPRs Spill", "?")}')
