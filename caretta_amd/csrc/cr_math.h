// Device/host FP64 math shared by every kernel of libcaretta_hip.
//
// Everything here is written for bit-reproducibility: FP64 only, explicit __builtin_fma where
// a fused operation is intended and nowhere else (the library is built with -ffp-contract=off),
// fixed operation order.  The same algorithms are restated in plain C in oracle/caretta_oracle.c
// (the test-only checker); the two are kept in step by tests/test_gpu_parity.py, which demands
// bit-identical results.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CR_HD __host__ __device__ __forceinline__
#define CR_D __device__ __forceinline__

namespace cr {

// ---------------------------------------------------------------------------------------------
// exp(x) = 2^(k/16) * exp(r), k = RN(x*16/ln2), r = x - k*ln2/16, |r| <= ln2/32.
// 16-entry hi/lo table (256 B = one LDS bank row: a ds_read_b128 per lane is conflict-free),
// degree-7 Taylor polynomial, result = th + fma(th, p, tl), scaled by two exact powers of two so
// that subnormal results round once.  < 0.52 ulp.  Constants: tools/gen_exp_constants.py.
// Stands in for libm exp at the reference's score_functions.py:11.
// ---------------------------------------------------------------------------------------------
struct ExpEntry {
    double hi, lo;
};

static __device__ const ExpEntry kExpTable[16] = {
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54},
    {0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55},
    {0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54},
    {0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55},
    {0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55},
    {0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56},
    {0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54},
    {0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54},
    {0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55},
    {0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54},
    {0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56},
    {0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54},
    {0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55},
    {0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55},
    {0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54},
};

// `tab` points at a copy of kExpTable in LDS (16-byte aligned).
// NONPOS: the caller guarantees x <= 0 (RBF with gamma >= 0), so the overflow clamp is dropped.
// The final scaling y * 2^e is one v_ldexp_f64; it rounds once into the subnormal range, which is
// bit-identical to the oracle's two exact-then-rounded power-of-two multiplies.
template <bool NONPOS>
CR_D double exp_tab(double x, const ExpEntry* tab) {
    const double INV_LN2_16 = 0x1.71547652b82fep+4;
    const double LN2_16_HI = 0x1.62e42fefa39efp-5;
    const double LN2_16_LO = 0x1.abc9e3b39803fp-60;
    const double SHIFT = 0x1.8p52;
    const double C2 = 0x1.0000000000000p-1, C3 = 0x1.5555555555555p-3, C4 = 0x1.5555555555555p-5;
    const double C5 = 0x1.1111111111111p-7, C6 = 0x1.6c16c16c16c17p-10, C7 = 0x1.a01a01a01a01ap-13;
    if constexpr (!NONPOS) x = __builtin_fmin(x, 710.0);   // -> +inf through the scaling below
    x = __builtin_fmax(x, -746.0);                         // -> 0
    double z = __builtin_fma(x, INV_LN2_16, SHIFT);
    int ki = __double2loint(z);
    double kd = z - SHIFT;
    double r = __builtin_fma(kd, -LN2_16_HI, x);
    r = __builtin_fma(kd, -LN2_16_LO, r);
    double r2 = r * r;
    double q = __builtin_fma(r, C7, C6);
    q = __builtin_fma(r, q, C5);
    q = __builtin_fma(r, q, C4);
    q = __builtin_fma(r, q, C3);
    q = __builtin_fma(r, q, C2);
    double p = __builtin_fma(r2, q, r);
    ExpEntry t = tab[ki & 15];
    double y = t.hi + __builtin_fma(t.hi, p, t.lo);
    return __builtin_ldexp(y, ki >> 4);
}

// ---------------------------------------------------------------------------------------------
// 3x3 SVD by one-sided Jacobi (stands in for LAPACK dgesdd at superposition_functions.py:28),
// singular values descending; a numerically vanishing third direction is completed by a cross
// product.  Row-major 3x3 arrays.
// ---------------------------------------------------------------------------------------------
CR_HD double det3(const double* A) {
    return (A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6])) +
           A[2] * (A[3] * A[7] - A[4] * A[6]);
}

CR_HD void svd3(const double* C, double* U, double* Sg, double* Vt) {
    double A[9], V[9];
    for (int x = 0; x < 9; x++) {
        A[x] = C[x];
        V[x] = (x % 4 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; sweep++) {
        bool rotated = false;
        for (int x = 0; x < 3; x++) {
            const int p = (x == 2) ? 1 : 0;
            const int q = (x == 0) ? 1 : 2;
            double alpha = (A[p] * A[p] + A[3 + p] * A[3 + p]) + A[6 + p] * A[6 + p];
            double beta = (A[q] * A[q] + A[3 + q] * A[3 + q]) + A[6 + q] * A[6 + q];
            double gam = (A[p] * A[q] + A[3 + p] * A[3 + q]) + A[6 + p] * A[6 + q];
            if (gam == 0.0) continue;
            if (fabs(gam) <= 1e-15 * sqrt(alpha * beta)) continue;
            double zeta = (beta - alpha) / (2.0 * gam);
            double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            double c = 1.0 / sqrt(1.0 + tt * tt);
            double s = c * tt;
            for (int r = 0; r < 3; r++) {
                double ap = A[3 * r + p], aq = A[3 * r + q];
                A[3 * r + p] = c * ap - s * aq;
                A[3 * r + q] = s * ap + c * aq;
                double vp = V[3 * r + p], vq = V[3 * r + q];
                V[3 * r + p] = c * vp - s * vq;
                V[3 * r + q] = s * vp + c * vq;
            }
            rotated = true;
        }
        if (!rotated) break;
    }
    double sg[3];
    for (int j = 0; j < 3; j++) sg[j] = sqrt((A[j] * A[j] + A[3 + j] * A[3 + j]) + A[6 + j] * A[6 + j]);
    int o0 = 0, o1 = 1, o2 = 2, tmp;
    if (sg[o1] > sg[o0]) { tmp = o0; o0 = o1; o1 = tmp; }
    if (sg[o2] > sg[o1]) { tmp = o1; o1 = o2; o2 = tmp; }
    if (sg[o1] > sg[o0]) { tmp = o0; o0 = o1; o1 = tmp; }
    const int ord[3] = {o0, o1, o2};
    double Um[9], Vm[9];
    for (int j = 0; j < 3; j++) {
        const int o = ord[j];
        Sg[j] = sg[o];
        for (int r = 0; r < 3; r++) Vm[3 * r + j] = V[3 * r + o];
        if (sg[o] > 0.0) {
            for (int r = 0; r < 3; r++) Um[3 * r + j] = A[3 * r + o] / sg[o];
        } else {
            for (int r = 0; r < 3; r++) Um[3 * r + j] = (r == j) ? 1.0 : 0.0;
        }
    }
    if (!(Sg[2] > 1e-12 * Sg[0])) {
        Um[2] = Um[3] * Um[7] - Um[6] * Um[4];
        Um[5] = Um[6] * Um[1] - Um[0] * Um[7];
        Um[8] = Um[0] * Um[4] - Um[3] * Um[1];
    }
    for (int x = 0; x < 9; x++) U[x] = Um[x];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Vt[3 * r + c] = Vm[3 * c + r];
}

// Kabsch tail (superposition_functions.py:28-34): correlation matrix C (= X2c^T X1c) and the two
// centroids -> rotation R and translation t with  X2 @ R + t ~ X1.
CR_HD void kabsch_from_correlation(const double* C, const double* c1, const double* c2, double* R, double* t) {
    double U[9], S[3], Vt[9];
    svd3(C, U, S, Vt);
    if (det3(U) * det3(Vt) < 0.0) {
        U[2] = -U[2];
        U[5] = -U[5];
        U[8] = -U[8];
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            R[3 * r + c] = (U[3 * r] * Vt[c] + U[3 * r + 1] * Vt[3 + c]) + U[3 * r + 2] * Vt[6 + c];
    for (int c = 0; c < 3; c++) t[c] = c1[c] - ((c2[0] * R[c] + c2[1] * R[3 + c]) + c2[2] * R[6 + c]);
}

// out = x @ R (no translation), row vector times row-major R
CR_HD void rot3(const double* x, const double* R, double* out) {
    for (int c = 0; c < 3; c++) out[c] = (x[0] * R[c] + x[1] * R[3 + c]) + x[2] * R[6 + c];
}

// ---------------------------------------------------------------------------------------------
// Cross-lane: shift a double one lane up the wave (lane l receives lane l-1's value; lane 0
// receives `fill`) with two DPP moves (v_mov_b32_dpp wave_shr:1) -- no LDS traffic.
// Must be executed with all 64 lanes enabled.
// ---------------------------------------------------------------------------------------------
CR_D double wave_shr1(double v, double fill) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int flo = __double2loint(fill), fhi = __double2hiint(fill);
    lo = __builtin_amdgcn_update_dpp(flo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

}  // namespace cr
