#!/bin/bash
# Two PMC passes (issue and wait counters) over a workload script:  gpurun -- 'bash tools/pmc_quick.sh <tag> <script>'  -> gpurun_out/<tag>/summary.json
TAG=$1; SCRIPT=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $SCRIPT > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log | cut -c1-200
done
python3 tools/pmc_summary.py $OUT
