#!/usr/bin/env python
"""Timeline of the time-major path from a rocprofv3 kernel trace: start/end (ms) of every transpose / thresholds /
exceedance / state-machine dispatch after the last generate kernel.  usage: tools/tm_trace.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
out = []
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void hdp::", "")[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None:
        t0 = s
    out.append((short, (s - t0) / 1e6, (e - t0) / 1e6))
# keep the last 60 dispatches (the timed tm passes are at the end of the run)
for short, s, e in out[-int(sys.argv[2]) if len(sys.argv) > 2 else -60:]:
    print(f"{short:42s} {s:10.3f} -> {e:10.3f}  ({e - s:7.3f} ms)")
