#!/bin/bash
# Calibration builds of the library with other constants of cr_trio.h:  bash tools/trio_variants.sh name "-DCR_TRIO_RING=12" [name flags ...]
# -> gpurun_out/variants/lib_<name>.so (scratch, never part of the tree: run it ON the GPU box; CARETTA_HIP_LIB=<that file> selects it at run time).
cd "$(dirname "$0")/.." || exit 1
C=caretta_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-variable -pthread"
V=gpurun_out/variants; mkdir -p $V
build_one() {
  name=$1; extra=$2; o=$V/obj_$name; mkdir -p $o
  /opt/rocm/bin/hipcc $FLAGS $extra -c $C/cr_api.hip -o $o/cr_api.o &
  /opt/rocm/bin/hipcc $FLAGS $extra -mllvm -amdgpu-sched-strategy=iterative-ilp -c $C/cr_kernels_duo.hip -o $o/cr_kernels_duo.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $o/cr_api.o $C/obj/cr_kernels_ilp.hip.o $o/cr_kernels_duo.o -o $V/lib_$name.so && echo "built $name ($extra)"
}
while [ $# -ge 2 ]; do
  build_one "$1" "$2" &
  shift 2
  while [ $(jobs -r | wc -l) -ge 3 ]; do sleep 2; done
done
wait
