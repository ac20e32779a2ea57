#!/bin/bash
for lib in libhdp_dbg libhdp_bias3 libhdp_bias4; do
  echo "== $lib"
  HDP_LIB_PATH=$PWD/tools/dbg/$lib.so HDP_THR_DEBUG=1024 python bench.py --cells 65536 --steps 2 --warmup 1 --no-cpu-baseline --no-tm 2>&1 | grep -E "hdp thresholds lane" | cut -c60-260
  HDP_LIB_PATH=$PWD/tools/dbg/$lib.so python bench.py --cells 65536 --steps 4 --warmup 1 --no-cpu-baseline --no-tm 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']; print('thr %.3f met %.3f' % (k['thresholds_kernel']['ms_per_launch'], k['metrics_kernel']['ms_per_launch']))"
done
