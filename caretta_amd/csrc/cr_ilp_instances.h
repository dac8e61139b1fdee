// Kernels that are compiled in their own translation unit (cr_kernels_ilp.hip) with
// -mllvm -amdgpu-sched-strategy=iterative-ilp.  Measured on MI355X against the default scheduler: k_seed -5 %
// (128 x 300), -9 % (32 x 150); k_align<3> -13 %; k_align<5> +57 % (it loses its 4 waves per SIMD), so that one
// and everything else stay in cr_api.hip.  cr_api.hip declares these instances `extern template`.
#pragma once

#define CR_ILP_SEED_INSTANCES(X) \
    X(2, 4, true) X(2, 4, false) X(2, 8, true) X(2, 8, false) X(2, 10, true) X(2, 10, false) X(2, 16, true) X(2, 16, false) \
    X(3, 4, true) X(3, 4, false) X(3, 8, true) X(3, 8, false) X(3, 10, true) X(3, 10, false) X(3, 16, true) X(3, 16, false) \
    X(4, 4, true) X(4, 4, false) X(4, 8, true) X(4, 8, false) X(4, 10, true) X(4, 10, false) X(4, 16, true) X(4, 16, false) \
    X(5, 4, true) X(5, 4, false) X(5, 8, true) X(5, 8, false) X(5, 10, true) X(5, 10, false) X(5, 16, true) X(5, 16, false) \
    X(2, 24, true) X(2, 24, false) X(2, 32, true) X(2, 32, false) \
    X(3, 24, true) X(3, 24, false) X(3, 32, true) X(3, 32, false) \
    X(4, 24, true) X(4, 24, false) X(4, 32, true) X(4, 32, false) \
    X(5, 24, true) X(5, 24, false) X(5, 32, true) X(5, 32, false)

#define CR_ILP_ALIGN_INSTANCES(X) X(2, true) X(2, false) X(3, true) X(3, false) X(4, true) X(4, false)

#define CR_SEED_SIGNATURE(R, D, ZG)                                                                                  \
    __global__ void cr::k_seed<R, D, ZG>(const cr::PairDesc*, const double*, int, const double*, double, double, int, \
                                         uint32_t*, double*, cr::Transform*, double*);
#define CR_ALIGN_SIGNATURE(R, ZG)                                                                                      \
    __global__ void cr::k_align<R, ZG>(const cr::PairDesc*, const double*, const cr::Transform*, const double*, double, \
                                       double, double, double, int, uint32_t*, double*, int32_t*, cr::PairResult*, const cr::HostOut);

// the team kernels of the progressive alignment levels (+4 % under the same scheduler)
#define CR_ILP_SEED_TEAM_INSTANCES(X)                                                                                  \
    X(1, 4, true) X(1, 4, false) X(1, 8, true) X(1, 8, false) X(1, 10, true) X(1, 10, false) X(1, 16, true) X(1, 16, false) \
    X(2, 4, true) X(2, 4, false) X(2, 8, true) X(2, 8, false) X(2, 10, true) X(2, 10, false) X(2, 16, true) X(2, 16, false) \
    X(3, 4, true) X(3, 4, false) X(3, 8, true) X(3, 8, false) X(3, 10, true) X(3, 10, false) X(3, 16, true) X(3, 16, false) \
    X(4, 4, true) X(4, 4, false) X(4, 8, true) X(4, 8, false) X(4, 10, true) X(4, 10, false) X(4, 16, true) X(4, 16, false) \
    X(5, 4, true) X(5, 4, false) X(5, 8, true) X(5, 8, false) X(5, 10, true) X(5, 10, false) X(5, 16, true) X(5, 16, false) \
    X(1, 24, true) X(1, 24, false) X(1, 32, true) X(1, 32, false) \
    X(2, 24, true) X(2, 24, false) X(2, 32, true) X(2, 32, false) \
    X(3, 24, true) X(3, 24, false) X(3, 32, true) X(3, 32, false) \
    X(4, 24, true) X(4, 24, false) X(4, 32, true) X(4, 32, false) \
    X(5, 24, true) X(5, 24, false) X(5, 32, true) X(5, 32, false)
#define CR_ILP_NODE_TEAM_INSTANCES(X) X(1) X(2) X(3) X(4) X(5)

#define CR_SEED_TEAM_SIGNATURE(R, D, ZG)                                                                                   \
    __global__ void cr::k_seed_team<R, D, ZG>(const cr::PairDesc*, const double*, int, const double*, double, double, int, \
                                              uint32_t*, cr::Transform*, double*);
#define CR_NODE_TEAM_SIGNATURE(R)                                                                                            \
    __global__ void cr::k_node_team<R>(const cr::PairDesc*, const double*, const double*, int, const double*,                \
                                       const cr::NodeDesc*, const cr::Transform*, double, double, double, double, int,       \
                                       uint32_t*, double*, int32_t*, double*, double*, double*, cr::NodeOut*);

// wide kernels (one wave per strip, up to 16 waves per pair; RA rows per lane in the first strips, RB in the others)
#define CR_ILP_SEED_WIDE_D(X, RA, RB) \
    X(RA, RB, 4, true) X(RA, RB, 4, false) X(RA, RB, 8, true) X(RA, RB, 8, false) X(RA, RB, 10, true) X(RA, RB, 10, false) \
    X(RA, RB, 16, true) X(RA, RB, 16, false)
#define CR_ILP_SEED_WIDE_INSTANCES(X) CR_ILP_SEED_WIDE_D(X, 2, 2) CR_ILP_SEED_WIDE_D(X, 3, 3) CR_ILP_SEED_WIDE_D(X, 3, 2)
#define CR_SEED_WIDE_SIGNATURE(RA, RB, D, ZG)                                                                                   \
    __global__ void cr::k_seed_wide<RA, RB, D, ZG>(const cr::PairDesc*, const double*, int, const double*, double, double, int, \
                                                   int, int, uint32_t*, cr::Transform*, double*);
// both stages of a pair in one launch (k_pair_wide; SCORES only with gap 0)
#define CR_ILP_PAIR_WIDE_D(X, RA, RB) \
    X(RA, RB, 4, true, false) X(RA, RB, 4, false, false) X(RA, RB, 4, true, true) X(RA, RB, 8, true, false) X(RA, RB, 8, false, false) \
    X(RA, RB, 8, true, true) X(RA, RB, 10, true, false) X(RA, RB, 10, false, false) X(RA, RB, 10, true, true) \
    X(RA, RB, 16, true, false) X(RA, RB, 16, false, false) X(RA, RB, 16, true, true)
#define CR_ILP_PAIR_WIDE_INSTANCES(X) CR_ILP_PAIR_WIDE_D(X, 2, 2) CR_ILP_PAIR_WIDE_D(X, 3, 3) CR_ILP_PAIR_WIDE_D(X, 3, 2)
#define CR_PAIR_WIDE_SIGNATURE(RA, RB, D, ZG, SC)                                                                               \
    __global__ void cr::k_pair_wide<RA, RB, D, ZG, SC>(const cr::PairDesc*, const double*, int, const double*, double, double,  \
                                                       double, double, double, int, int, int, int, uint32_t*, uint32_t*,        \
                                                       cr::Transform*, double*, int32_t*, cr::PairResult*, const cr::HostOut);
