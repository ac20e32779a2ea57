#!/bin/bash
# Everything profiles/rNN_vK_* is made from, ON THE GPU BOX: tools/final_profiles.sh <tag>  (e.g. r03_v2)
TAG=$1
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_c3_full.json 2> $O/bench_c3_full.err; echo "bench rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 $GRAFT_REPO_ROOT/bench.py > $O/trace_default.log 2>&1); echo "trace rc=$?"
timeout -k 10 200 python bench.py --config c2 --steps 20 --warmup 3 > $O/bench_c2.json 2>/dev/null; echo "c2 rc=$?"
timeout -k 10 300 python bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.json 2>/dev/null; echo "c5 rc=$?"
for q in tail median spread; do
  timeout -k 10 200 python bench.py --cells 65536 --steps 3 --warmup 1 --no-cpu-baseline --no-tm --qset $q 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=l['kernels']
print('qset $q: thresholds %.3f ms per 65536 cells (%.1f %% of HBM), metrics %.3f ms' % (k['thresholds_kernel']['ms_per_launch'], 100*k['thresholds_kernel']['frac_hbm'], k['metrics_kernel']['ms_per_launch']))"
done > $O/bench_c3_qsets.txt 2>&1; cat $O/bench_c3_qsets.txt
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --share-device --cells 131072 --steps 2 --warmup 1 > $O/rehearsal.json 2> $O/rehearsal.err; echo "rehearsal rc=$?"
tools/prof.sh ${TAG}_c3_65536 --cells 65536 --steps 2 --warmup 1 --no-tm > $O/prof.log 2>&1; echo "prof rc=$?"
python3 tools/prof_summary.py gpurun_out/prof_${TAG}_c3_65536 > $O/c3_65536cells_rocprofv3_summary.txt
tools/prof.sh ${TAG}_c5 --config c5 --steps 1 --warmup 1 --no-tm > $O/prof_c5.log 2>&1; echo "prof c5 rc=$?"
python3 tools/prof_summary.py gpurun_out/prof_${TAG}_c5 > $O/c5_rocprofv3_summary.txt
