#!/bin/bash
# rocprofv3 PMC passes over one GPU's share of BASELINE config 5 (252 pairs of 1200 x 1200, the wide layout: k_pair_wide):
#   gpurun -- 'bash tools/pmc_share.sh [tag [script]]'      -> gpurun_out/<tag>/summary.json
# (script: the workload, default tools/config5_share_time.py; tools/c3_share_time.py = the headline's one-of-8 share)
# Separate passes, --kernel-trace only (MI355X_MICROARCH.md).
TAG=${1:-r04_pmc_c5share}
SCRIPT=${2:-tools/config5_share_time.py}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $SCRIPT > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log | cut -c1-200
done
python3 tools/pmc_summary.py $OUT
