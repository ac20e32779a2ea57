#!/bin/bash
# PMC counters of the thresholds kernel alone (tools/dbg/thr_time.py), run ON THE GPU BOX via gpurun.
set -e
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/tools/dbg/thr_time.py"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- $B > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $B > $OUT/pmc2.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt
grep thresholds $OUT/summary.txt | cut -c1-420
