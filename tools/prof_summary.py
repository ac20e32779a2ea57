#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel stats + PMC) into one small text file for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    out.append(f"== kernel stats ({os.path.relpath(f, root)})")
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")[:70]
        out.append(f"{name:70s} calls={row.get('Calls')} total_ns={row.get('TotalDurationNs')} "
                   f"avg_ns={row.get('AverageNs')} pct={row.get('Percentage')}")
for p in sorted(glob.glob(os.path.join(root, "pmc*"))):
    if not os.path.isdir(p):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")[:50]
            acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    out.append(f"== PMC {os.path.basename(p)} (mean per dispatch)")
    for k in sorted(acc):
        if "hdp" not in k and "kernel" not in k:
            continue
        vals = "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(acc[k].items()))
        out.append(f"{k:50s} n={len(next(iter(acc[k].values())))}  {vals}")
print("\n".join(out))
