// Explicit instantiations of the kernels that gain from the iterative-ILP instruction scheduler (cr_ilp_instances.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-sched-strategy=iterative-ilp -c
#include <hip/hip_runtime.h>

#include <cstdint>

#define CR_KERNELS_TEMPLATES_ONLY
#include "cr_kernels.h"
#include "cr_ilp_instances.h"

#define CR_X(R, D, ZG) template CR_SEED_SIGNATURE(R, D, ZG)
CR_ILP_SEED_INSTANCES(CR_X)
#undef CR_X
#define CR_X(R, ZG) template CR_ALIGN_SIGNATURE(R, ZG)
CR_ILP_ALIGN_INSTANCES(CR_X)
#undef CR_X
#define CR_X(R, D, ZG) template CR_SEED_TEAM_SIGNATURE(R, D, ZG)
CR_ILP_SEED_TEAM_INSTANCES(CR_X)
#undef CR_X
#define CR_X(R) template CR_NODE_TEAM_SIGNATURE(R)
CR_ILP_NODE_TEAM_INSTANCES(CR_X)
#undef CR_X
#define CR_X(RA, RB, D, ZG) template CR_SEED_WIDE_SIGNATURE(RA, RB, D, ZG)
CR_ILP_SEED_WIDE_INSTANCES(CR_X)
#undef CR_X
#define CR_X(RA, RB, D, ZG, SC) template CR_PAIR_WIDE_SIGNATURE(RA, RB, D, ZG, SC)
CR_ILP_PAIR_WIDE_INSTANCES(CR_X)
#undef CR_X
