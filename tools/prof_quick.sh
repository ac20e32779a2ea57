#!/bin/bash
# Quick A/B profile ON THE GPU BOX: kernel trace + one PMC pass (instruction counts, wave cycles) for one bench setting.
# usage: tools/prof_quick.sh <tag> <bench args...>     (environment switches are inherited)
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc2 -- $B > $OUT/pmc2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.txt
