#!/bin/bash
# A/B of a library switch on the bench's kernels: tools/ab.sh <cells> VAR=val [VAR=val ...]; one bench line per setting
cells=$1; shift
for kv in "" "$@"; do
  out=$(env $kv python bench.py --cells $cells --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$kv" <<PY
import json,sys
l=json.loads('''$out''')
k=l["kernels"]
print("%-28s thr %.3f ms  met %.3f ms  step %.3f ms" % (sys.argv[1] or "default", k["thresholds_kernel"]["ms_per_launch"], k["metrics_kernel"]["ms_per_launch"], l["ms_per_step"]))
PY
done
