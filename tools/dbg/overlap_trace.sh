#!/bin/bash
# Do the thresholds kernel (stream A) and the metrics kernels (stream B) of tools/dbg/overlap_time.py overlap in time?
# ON THE GPU BOX: tools/dbg/overlap_trace.sh <tag> [n_cells]
TAG=$1; N=${2:-65536}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ov_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/dbg/overlap_time.py $N > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("exceed", "cells16", "thresholds_lane")):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][-40:]))
rows.sort()
t0 = rows[0][0]
for s, e, n in rows[-40:]:
    print(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} {(e - s) / 1e6:8.3f} ms  {n}")
PY
