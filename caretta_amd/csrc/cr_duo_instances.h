// Instances of k_pair_duo (cr_duo.h), compiled in their own translation unit (cr_kernels_duo.hip); cr_api.hip declares them
// `extern template`.  (RA, RB): rows per lane of the first nA strips and of the others.
#pragma once

#define CR_DUO_D(X, RA, RB) \
    X(RA, RB, 4, false) X(RA, RB, 4, true) X(RA, RB, 8, false) X(RA, RB, 8, true) X(RA, RB, 10, false) X(RA, RB, 10, true) \
    X(RA, RB, 16, false) X(RA, RB, 16, true)
#define CR_DUO_INSTANCES(X) CR_DUO_D(X, 1, 1) CR_DUO_D(X, 2, 1) CR_DUO_D(X, 2, 2) CR_DUO_D(X, 3, 2) CR_DUO_D(X, 3, 3)
#define CR_PAIR_DUO_SIGNATURE(RA, RB, D, SC)                                                                                  \
    __global__ void cr::k_pair_duo<RA, RB, D, SC>(const cr::PairDesc*, const double*, int, const double*, double, double,     \
                                                  double, double, int, int, int, uint32_t*, uint32_t*, cr::Transform*, double*, \
                                                  int32_t*, cr::PairResult*, const cr::HostOut);

// k_pair_trio (cr_trio.h): one wave of recurrences + two waves of scores per pair, pairs of at most 64 R rows
// (two to five rows per lane: 65 .. 320 rows; tensor widths padded to 4, 8, 10, 12, 16 -- the width of THIS kernel's score
// waves, chosen from the stored width by trio_width(), not the batch's d_pad: a family with d = 12 pays for 12 features)
#define CR_TRIO_D(X, R) X(R, 4, false) X(R, 4, true) X(R, 8, false) X(R, 8, true) X(R, 10, false) X(R, 10, true) \
    X(R, 12, false) X(R, 12, true) X(R, 16, false) X(R, 16, true)
#define CR_TRIO_INSTANCES(X) CR_TRIO_D(X, 2) CR_TRIO_D(X, 3) CR_TRIO_D(X, 4) CR_TRIO_D(X, 5)
#define CR_PAIR_TRIO_SIGNATURE(R, D, SC)                                                                                   \
    __global__ void cr::k_pair_trio<R, D, SC>(const cr::PairDesc*, const double*, int, const double*, double, double, double, \
                                              double, int, int, int, uint32_t*, uint32_t*, cr::Transform*, double*, int32_t*, \
                                              cr::PairResult*, const cr::HostOut);
