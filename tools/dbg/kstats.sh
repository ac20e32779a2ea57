#!/bin/bash
# kernel-trace stats of a tools/dbg script ON THE GPU BOX: tools/dbg/kstats.sh <tag> <script.py> [args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/dbg/$@ > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("exceed", "cells16", "thresholds", "transpose")):
            print("$TAG", r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
PY
