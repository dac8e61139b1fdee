#!/bin/bash
# rocprofv3 PMC passes over bench.py (run on the GPU box: gpurun -- 'bash tools/pmc_profile.sh [tag] [bench args]').
# Counters are collected in separate passes (<= 8 SQ counters / pass; FETCH_SIZE and WRITE_SIZE apart),
# with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
TAG=${1:-pmc}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python3 bench.py $ARGS > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log | cut -c1-200
done
python3 tools/pmc_summary.py $OUT
