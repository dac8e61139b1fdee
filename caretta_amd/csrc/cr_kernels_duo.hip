// Explicit instantiations of the mid-size pair-list kernels (cr_duo.h, cr_duo_instances.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-sched-strategy=iterative-ilp -c
#include <hip/hip_runtime.h>

#include <cstdint>

#define CR_KERNELS_TEMPLATES_ONLY
#include "cr_duo.h"
#include "cr_trio.h"
#include "cr_duo_instances.h"

#define CR_X(RA, RB, D, SC) template CR_PAIR_DUO_SIGNATURE(RA, RB, D, SC)
CR_DUO_INSTANCES(CR_X)
#undef CR_X
#define CR_X(R, D, SC) template CR_PAIR_TRIO_SIGNATURE(R, D, SC)
CR_TRIO_INSTANCES(CR_X)
#undef CR_X
