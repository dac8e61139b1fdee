#!/bin/bash
# rocprofv3 PMC passes over the progressive alignment of 128 x 300 (the staged kernels of cr_staged.h):
#   gpurun -- 'bash tools/pmc_tree.sh [tag]'      -> gpurun_out/<tag>/summary.json
# Separate passes, --kernel-trace only (MI355X_MICROARCH.md).
TAG=${1:-r03_pmc_tree}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 tools/bench_msa.py 128 300 > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log | cut -c1-200
done
python3 tools/pmc_summary.py $OUT
