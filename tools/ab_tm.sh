#!/bin/bash
# A/B of the time-major path on the GPU box: tools/ab_tm.sh <cells> VAR=val [VAR=val ...]
cells=$1; shift
for kv in "" "$@"; do
  out=$(env $kv python bench.py --layout tm --cells $cells --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$kv" <<PY
import json,sys
l=json.loads('''$out''')
t=l["layout_tm"]
if "error" in t: print(sys.argv[1], t); sys.exit()
print("%-22s cells %d  thresholds tm %.2f ms (cm %.2f: x%.2f)  metrics tm %.2f ms (cm %.2f: x%.2f)  identical %s" % (
  sys.argv[1] or "default", t["cells"], t["thresholds_ms"], t["series_major_thresholds_ms_same_cells"], t["thresholds_ms"]/t["series_major_thresholds_ms_same_cells"],
  t["metrics_ms"], t["series_major_metrics_ms_same_cells"], t["metrics_ms"]/t["series_major_metrics_ms_same_cells"], t["identical_to_series_major"]))
PY
done
