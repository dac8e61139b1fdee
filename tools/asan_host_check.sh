#!/bin/bash
# Host side of libcaretta_hip.so (neighbor joining incl. helper threads, argument validation, pair-list layout, block
# cache) under AddressSanitizer + UBSan: builds an instrumented copy (device code untouched; GPU ASan is not available
# here) and runs the CPU tests of the C ABI against it.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/caretta_asan
mkdir -p "$OUT"
cd "$ROOT/caretta_amd/csrc"
COMMON="--offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -pthread"
hipcc $COMMON -O1 -g -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -c cr_api.hip -o "$OUT/api.o"
# (the kernel-only translation units come from the product build: obj/*.o of __graft_entry__.build())
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1 || true
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -fsanitize=address,undefined -fno-gpu-sanitize "$OUT/api.o" obj/cr_kernels_ilp.hip.o obj/cr_kernels_duo.hip.o -o "$OUT/libcaretta_hip_asan.so"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT CARETTA_HIP_LIB="$OUT/libcaretta_hip_asan.so" CARETTA_SYSTEM_HIP=1 \
    python -m pytest tests/test_capi_cpu.py -x -q

# The oracle (test infrastructure) under gcc's ASan + UBSan against the golden vectors
cp "$ROOT"/oracle/caretta_oracle.c "$ROOT"/oracle/caretta_oracle.h "$ROOT"/oracle/exp_table.inc "$OUT"/
cd "$OUT"
OFLAGS="-O1 -g -ffp-contract=off -mfma -fopenmp -fPIC -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -shared"
gcc $OFLAGS -o libcaretta_oracle.so caretta_oracle.c -lm
gcc $OFLAGS -DCRO_LIBM_EXP -o libcaretta_oracle_libm.so caretta_oracle.c -lm
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) CARETTA_ORACLE_DIR="$OUT" \
    python -m pytest tests/test_oracle_golden.py -x -q
