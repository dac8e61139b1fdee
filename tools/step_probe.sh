#!/bin/bash
# Build the variants of tools/step_probe.hip (here, for gfx950) or run them (on the GPU box):
#   bash tools/step_probe.sh build      bash tools/step_probe.sh run > gpurun_out/step_probe.txt
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-variable -Wno-unused-value"
case "$1" in
build)
  hipcc $FLAGS tools/step_probe.hip -o tools/step_probe_lib.bin &
  hipcc $FLAGS -DCR_PROBE_NO_DECISIONS tools/step_probe.hip -o tools/step_probe_nodec.bin &
  hipcc $FLAGS -DCR_PROBE_NO_DUMP tools/step_probe.hip -o tools/step_probe_nodump.bin &
  hipcc $FLAGS -DCR_PROBE_NO_DECISIONS -DCR_PROBE_NO_DUMP tools/step_probe.hip -o tools/step_probe_neither.bin &
  hipcc $FLAGS -DCR_PROBE_MASKED_RAMPS tools/step_probe.hip -o tools/step_probe_masked.bin &
  hipcc $FLAGS -DCR_PROBE_ROWS tools/step_probe.hip -o tools/step_probe_rows.bin &
  wait ;;
run)
  for v in lib masked nodec nodump neither rows; do tools/step_probe_$v.bin; done ;;
esac
