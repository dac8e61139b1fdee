#!/bin/bash
# time tools/c3_share_time.py on the main library and on calibration builds:  bash tools/trio_compare.sh REPS name ...   ("main" = the tree's library)
reps=$1; shift
for i in $(seq $reps); do for v in "$@"; do
  if [ "$v" = main ]; then unset CARETTA_HIP_LIB; else export CARETTA_HIP_LIB=$PWD/gpurun_out/variants/lib_$v.so; fi
  echo "[$v] $(python tools/c3_share_time.py 2>/dev/null | tail -1)"
done; done
