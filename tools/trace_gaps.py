#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run: per kernel name count / total / mean, and the idle
time between consecutive kernels.  usage: trace_gaps.py <dir with *kernel_trace.csv> [last N]"""
import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
agg = {}
for s, e, k in rows:
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:12]:
    print(f"  {c:5d} x {t / c / 1e3:10.1f} us = {t / 1e6:8.3f} ms  {k}")
gaps = sorted(((rows[i + 1][0] - rows[i][1]) / 1e3, rows[i][2][:40], rows[i + 1][2][:40]) for i in range(len(rows) - 1))
print("largest gaps (us):")
for g in gaps[-8:]:
    print("  %.1f  after %s  before %s" % g)
