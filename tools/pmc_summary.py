#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per (kernel, grid size): pmc_summary.py <dir> [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_match"
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if want not in r["Kernel_Name"]:
                continue
            k = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
for (kn, grid, cn), (s, n) in sorted(acc.items()):
    print(f"{kn:60s} grid {grid:>9s} {cn:28s} mean {s / n:16.1f}  n {n}")
