#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per-counter mean over the dispatches of a kernel."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
needle = sys.argv[2] if len(sys.argv) > 2 else "fused_kernel"
acc = defaultdict(list)
for path in sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if needle in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, vals in sorted(acc.items()):
    print(f"{name:36s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")
