// Pair lists whose longest structure has 65 .. 320 rows (ONE strip of R = 2 .. 5 rows per lane): three or four waves per pair
// with different jobs -- one wave runs the recurrences, two or three form the scores (k_pair_trio<R, D, SCORES>).
//
// One GPU's share of the headline configuration on 8 GPUs is 1 016 pairs of 300 x 300 on 1 024 SIMDs.  Splitting a pair by
// ROWS (cr_duo.h: two waves of 3 and 2 rows per lane) leaves every wave a chain of dependent instructions per step -- score,
// then recurrence, then the cross-lane hand-off -- and a wave bound by the latency of its chain loses time to every issue
// slot its neighbour on the SIMD takes (0.60 ms where the instruction count alone would allow 0.40).  Here
// the split is by FUNCTION: 50 of the 66 instructions of a seed cell and 23 of the 50 of an alignment cell form the score
// exp(-gamma |a - b|^2), which depends on nothing the recurrence produces.  Waves 1 .. (the producers) form the scores of
// alternate columns (steps) for all R x 64 rows and park them in an LDS ring; they have no dependency from one column to
// the next, so they fill whatever issue slots the SIMD has.  Wave 0 (the consumer) reads the scores and runs dp_column /
// the column-sweep recurrence of the single-wave kernels with all R rows per lane -- a single strip: no hand-off between
// strips, no lag, the decision words of k_seed / k_align in their layout, the same walkers and ordered sums behind it.
// The consumer asks the SIMD's arbiter for priority (s_setprio): taking it away costs 10 %.
// The layout's time does not depend on the pair count while the chip is not full, so cr_batch_set_pairs hands it lists from
// 65 (R <= 3) / 111 (R = 4) / 161 (R = 5) pairs on, up to 1 300 (BASELINE config 2, 496 pairs of 150, 0.23 -> 0.17 ms).
//
// Ring: kTrioRing columns (steps) of R x 64 doubles, slot = column (step) mod kTrioRing.  Progress words in LDS (LDS executes
// one wave's instructions in order): prod[p] = columns (steps) producer p has finished, cons = columns (steps) the consumer
// has TAKEN into registers -- it takes kTrioBatch at a time and gives their slots back at once --; a producer waits before it
// reuses a slot.
// Every value bit-identical to the single-wave kernels: the providers' own score code, dp_column and the column-sweep
// recurrence on the same values in the same order.
//
// Reference: multiple_alignment.py:321-349 (score function), :158-170 (pair loop), dynamic_time_warping.py:8-144, :205-278.
#pragma once

#include "cr_duo.h"

namespace cr {

// (calibration builds override these: tools/trio_variants.sh builds, tools/trio_compare.sh times them beside the tree's library)
#ifndef CR_TRIO_RING
#define CR_TRIO_RING 8
#endif
#ifndef CR_TRIO_EARLY
#define CR_TRIO_EARLY 1
#endif
#ifndef CR_TRIO_CONS_SLEEP
#define CR_TRIO_CONS_SLEEP 2
#endif
#ifndef CR_TRIO_PROD_SLEEP
#define CR_TRIO_PROD_SLEEP 2
#endif
#ifndef CR_TRIO_PRIO_CONS
#define CR_TRIO_PRIO_CONS 3
#endif
#ifndef CR_TRIO_PRIO_PROD
#define CR_TRIO_PRIO_PROD 0
#endif
constexpr int kTrioRing = CR_TRIO_RING; // columns (steps) of scores between the producers and the consumer
constexpr int kTrioBatch = 4;           // columns (steps) per wait / publication of the consumer
constexpr int kTrioMaxWaves = 5;        // 1 consumer + up to 4 producers (the launch decides: blockDim.x / 64)

// LDS (doubles): exp table | 8 progress words (prod[0..3], cons) | ring kTrioRing x R x 64 | resident columns 3 x m (alignment stage)
template <int R>
__host__ __device__ inline size_t trio_lds_doubles(int m) {
    return (size_t)kExpDoubles + 4 + (size_t)kTrioRing * R * kWave + (size_t)3 * m;
}

template <int SLEEP>
CR_D void trio_wait(const int* word, int need, unsigned long long& waited) {
#ifdef CR_STAMPS
    const unsigned long long t0 = CR_DUO_NOW();
#endif
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(word)) < need) __builtin_amdgcn_s_sleep(SLEEP);
    asm volatile("" ::: "memory");       // (compiler: the scores are read behind the word)
#ifdef CR_STAMPS
    waited += CR_DUO_NOW() - t0;
#endif
}

// ---- seed stage -----------------------------------------------------------------------------------------------------
template <int R, int D>
CR_D void trio_seed_producer(const int p, const int np, RbfTensor<R, D>& src, const int n, const int m, const ExpEntry* tab, double* ring, int* words) {
    const int lane = threadIdx.x & (kWave - 1);
    const bool full = src.d == D;
    src.load_rows(lane * R, n);
    unsigned long long waited = 0;
    auto fetch = [&](int j) {                            // the column's features: wave-uniform scalar loads (ColSweep::prefetch_into)
        const double* __restrict__ cg = src.cols_g;
        const int d = full ? D : src.d;
#pragma unroll
        for (int k = 0; k < D; k++) {
            const double v = cg[(int64_t)j * d + k];
            src.col[k] = (full || k < d) ? v : 0.0;
        }
    };
    if (p < m) fetch(p);
    CR_DUO_STAMP(p + 1, 0, CR_DUO_NOW());
#pragma unroll 1
    for (int c = p; c < m; c += np) {
        if (c >= kTrioRing) trio_wait<CR_TRIO_PROD_SLEEP>(words + 4, c - kTrioRing + 1, waited);          // the slot's last column has been consumed
        double acc[R];
#pragma unroll
        for (int q = 0; q < R; q++) acc[q] = src.dist2_of(q, src.col);
        fetch(c + np < m ? c + np : c);                  // (the next own column while the exps run)
        double* slot = ring + (size_t)((unsigned)c % (unsigned)kTrioRing) * (R * kWave) + lane;
#pragma unroll
        for (int q = 0; q < R; q++) slot[q * kWave] = exp_tab<true>(src.neg_gamma * acc[q], tab);
        if (lane == 0) duo_publish(words + p, c + 1);
    }
    CR_DUO_STAMP(p + 1, 1, CR_DUO_NOW());
    CR_DUO_STAMP(p + 1, 2, waited);
}

// the recurrence of ColSweep::step on scores that are already there (same operations on the same values)
template <int R>
struct TrioCols {
    double hprev[R], eprev;
    int rowfirst[R];
    uint32_t bits[R];
    CR_D void reset() {
#pragma unroll
        for (int q = 0; q < R; q++) {
            hprev[q] = 0.0;
            rowfirst[q] = 0;
            bits[q] = 0;
        }
        eprev = 0.0;
    }
    CR_D void advance(const double* sc, int j) {
        double dg[R], p[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            dg[q] = (q == 0 ? eprev : hprev[q - 1]) + sc[q];
            const double b = vmax(dg[q], hprev[q]);
            p[q] = q == 0 ? b : vmax(p[q - 1], b);
        }
        const double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
        const int sh2 = (j & 15) * 2;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double h = vmax(p[q], e);
            const bool same = h == hprev[q];
            uint32_t code = (h == dg[q]) ? 1u : same ? 2u : 3u;     // (:255-277) diag, then left, else up
            code = (h > 0.0) ? code : 0u;
            bits[q] |= code << sh2;
            rowfirst[q] = same ? rowfirst[q] : j;                   // column of the row's last strict increase
            hprev[q] = h;
        }
        eprev = e;
    }
    // without decisions (smith_waterman_score alone: sweep_cols_score)
    CR_D void advance_score(const double* sc) {
        double p[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double dg = (q == 0 ? eprev : hprev[q - 1]) + sc[q];
            const double b = vmax(dg, hprev[q]);
            p[q] = q == 0 ? b : vmax(p[q - 1], b);
        }
        const double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
#pragma unroll
        for (int q = 0; q < R; q++) hprev[q] = vmax(p[q], e);
        eprev = e;
    }
};

// wait for the batch [j0, jend) of all producers (`r0` = j0 mod np, kept by the caller: no division here), take its scores
// into registers and give the slots back at once: LDS executes this wave's reads before the word's write, and a producer
// writes a slot only behind its own read of the word
template <int R>
CR_D void trio_take(const double* ring, int* words, const int np, int j0, int jend, int& r0, double (&sc)[kTrioBatch][R], unsigned long long& waited) {
    const int lane = threadIdx.x & (kWave - 1);
    const int len = jend - j0;
    for (int k = len - 1; k >= 0 && k >= len - np; k--) {        // the last column of every producer in the batch
        int p = r0 + k;
        while (p >= np) p -= np;
        trio_wait<CR_TRIO_CONS_SLEEP>(words + p, j0 + k + 1, waited);
    }
    r0 += kTrioBatch;                                            // (the next batch starts kTrioBatch columns on)
    while (r0 >= np) r0 -= np;
    static_assert(kTrioRing % kTrioBatch == 0, "a batch's slots are contiguous");
    const double* slot = ring + (size_t)((unsigned)j0 % (unsigned)kTrioRing) * (R * kWave) + lane;       // (j0 is a multiple of kTrioBatch)
#pragma unroll
    for (int k = 0; k < kTrioBatch; k++) {
#pragma unroll
        for (int q = 0; q < R; q++) sc[k][q] = slot[(k * R + q) * kWave];            // (columns past jend: stale slots, never used)
    }
#if CR_TRIO_EARLY
    if (lane == 0) duo_publish(words + 4, jend);
#endif
}

template <int R>
CR_D void trio_seed_consumer(const int np, const int n, const int m, const double* ring, int* words, uint32_t* __restrict__ sw_dirs, SeedMax& seed_out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int TB = (m + 15) >> 4;
    TrioCols<R> st;
    st.reset();
    unsigned long long waited = 0;
    const int nb = kTrioBatch;
    int r0 = 0;
    CR_DUO_STAMP(0, 0, CR_DUO_NOW());
#pragma unroll 1
    for (int j0 = 0; j0 < m; j0 += nb) {
        const int jend = j0 + nb < m ? j0 + nb : m;
        double sc[kTrioBatch][R];
        trio_take<R>(ring, words, np, j0, jend, r0, sc, waited);
#pragma unroll
        for (int k = 0; k < kTrioBatch; k++) {
            const int j = j0 + k;
            if (j < jend) {
                st.advance(sc[k], j);
                if ((j & 15) == 15 || j == m - 1) {                 // a decision word holds 16 columns (strip 0 of k_seed's layout)
                    const int64_t base = ((int64_t)(j >> 4) * R) * kWave + lane;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        sw_dirs[base + q * kWave] = st.bits[q];
                        st.bits[q] = 0;
                    }
                }
            }
        }
#if !CR_TRIO_EARLY
        if (lane == 0) duo_publish(words + 4, jend);
#endif
    }
    (void)TB;
    CR_DUO_STAMP(0, 1, CR_DUO_NOW());
    CR_DUO_STAMP(0, 2, waited);
    // the rows' maxima (= last values) and their first columns, rows ascending (ColSweep::fold + wave_first_max)
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < R; q++) {
        const bool gt = st.hprev[q] > best_v;
        best_v = gt ? st.hprev[q] : best_v;
        best_i = gt ? lane * R + q : best_i;
        best_j = gt ? st.rowfirst[q] : best_j;
    }
    wave_first_max(best_v, best_i, best_j);
    seed_out.score = best_v;
    seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
    seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
}

// ---- alignment stage (time-skewed: step t, lane l -> column t - l) -----------------------------------------------------
template <int R>
CR_D void trio_align_producer(const int p, const int np, RbfCoords<R>& src, const int n, const int m, const int T, const ExpEntry* tab, const double* cols,
                              double* ring, int* words) {
    const int lane = threadIdx.x & (kWave - 1);
    src.load_rows(lane * R, n);
    unsigned long long waited = 0;
    CR_DUO_STAMP(p + 1, 4, CR_DUO_NOW());
#pragma unroll 1
    for (int t = p; t < T; t += np) {
        if (t >= kTrioRing) trio_wait<CR_TRIO_PROD_SLEEP>(words + 4, t - kTrioRing + 1, waited);
        const int c = t - lane;
        if ((unsigned)c < (unsigned)m) {
            src.fetch_resident(cols, m, c);
            double* slot = ring + (size_t)((unsigned)t % (unsigned)kTrioRing) * (R * kWave) + lane;
#pragma unroll
            for (int q = 0; q < R; q++) slot[q * kWave] = src.score(q, tab);
        }
        if (lane == 0) duo_publish(words + p, t + 1);
    }
    CR_DUO_STAMP(p + 1, 5, CR_DUO_NOW());
    CR_DUO_STAMP(p + 1, 6, waited);
}

template <int R, int MODE>
CR_D void trio_align_consumer(const int np, const int n, const int m, const int T, const SweepParams prm, const double* ring, int* words,
                              uint32_t* __restrict__ dtw_bits, AlignEnd& end_out) {
    constexpr bool SW = (MODE & kSwScore) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    const int lane = threadIdx.x & (kWave - 1);
    const int rowbase = lane * R;
    const double col0_m2 = kMinF64 - prm.gap_open;
    RbfCoords<R> unused;                                 // (dp_column reads only the provider's traits when the scores are given)
    DpState<R> st;
    st.sw_max = 0.0;
    st.reset_column0(col0_m2);
#pragma unroll
    for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;
    unsigned long long waited = 0;
    const int nb = kTrioBatch;
    int r0 = 0;
    CR_DUO_STAMP(0, 4, CR_DUO_NOW());
#pragma unroll 1
    for (int t0 = 0; t0 < T; t0 += nb) {
        const int tend = t0 + nb < T ? t0 + nb : T;
        double sc[kTrioBatch][R];
        trio_take<R>(ring, words, np, t0, tend, r0, sc, waited);
#pragma unroll
        for (int k = 0; k < kTrioBatch; k++) {
            const int t = t0 + k;
            if (t < tend) {
                const int c = t - lane;
                double h_top = 0.0, m0_top = 0.0, m1_top = 0.0;
                if constexpr (SW) h_top = wave_shr1(st.h_left[R - 1], 0.0);
                if constexpr (DTW) {
                    m0_top = wave_shr1(st.m0_left[R - 1], col0_m2);          // M[0][j][0] = MIN - open, M[0][j][1] = 0
                    m1_top = wave_shr1(st.m1_left[R - 1], 0.0);
                }
                if ((unsigned)c < (unsigned)m) dp_column<R, MODE>(unused, st, prm, nullptr, c, rowbase, n, 0, (t & 7) * 4, h_top, m0_top, m1_top, sc[k]);
                if constexpr (DTW) {
                    if ((t & 7) == 7 || t == T - 1) {                  // strip 0 of k_align's layout
                        const int64_t base = ((int64_t)(t >> 3) * R) * kWave + lane;
#pragma unroll
                        for (int q = 0; q < R; q++) {
                            dtw_bits[base + q * kWave] = st.dtbits[q];
                            st.dtbits[q] = 0;
                        }
                    }
                }
            }
        }
#if !CR_TRIO_EARLY
        if (lane == 0) duo_publish(words + 4, tend);
#endif
    }
    CR_DUO_STAMP(0, 5, CR_DUO_NOW());
    CR_DUO_STAMP(0, 6, waited);
    double sw_max = st.sw_max;
    if constexpr (SW) {
        for (int off = 32; off > 0; off >>= 1) sw_max = __builtin_fmax(sw_max, __shfl_xor(sw_max, off));
    }
    const int owner = ((n - 1) / R) % kWave;             // lane and register slot that own row n - 1
    const int qo = (n - 1) % R;
    double fin0 = 0.0, fin1 = 0.0, fin2 = 0.0;           // M[n][m][0..2]
#pragma unroll
    for (int q = 0; q < R; q++) {
        fin0 = (q == qo) ? st.m0_left[q] : fin0;
        fin1 = (q == qo) ? st.m1_left[q] : fin1;
        fin2 = (q == qo) ? st.m2_left[q] : fin2;
    }
    fin0 = lane_value(fin0, owner);
    fin1 = lane_value(fin1, owner);
    fin2 = lane_value(fin2, owner);
    end_out.sw = sw_max;
    int idx = 0;                                         // np.argmax of the three layers at (n, m), :181-182
    double best = fin0;
    if (fin1 > best) { best = fin1; idx = 1; }
    if (fin2 > best) { best = fin2; idx = 2; }
    end_out.dtw_score = DTW ? best : 0.0;
    end_out.start_layer = idx;
    end_out.pad = 0;
}

// ---- smith_waterman_score alone (the matrix entries): the column sweep on the coordinate scores -----------------------
template <int R>
CR_D void trio_score_producer(const int p, const int np, RbfCoords<R>& src, const int n, const int m, const ExpEntry* tab, const double* cols, double* ring,
                              int* words) {
    const int lane = threadIdx.x & (kWave - 1);
    src.load_rows(lane * R, n);
    unsigned long long waited = 0;
#pragma unroll 1
    for (int c = p; c < m; c += np) {
        if (c >= kTrioRing) trio_wait<CR_TRIO_PROD_SLEEP>(words + 4, c - kTrioRing + 1, waited);
        src.fetch_resident(cols, m, c);                  // wave-uniform address: an LDS broadcast
        double* slot = ring + (size_t)((unsigned)c % (unsigned)kTrioRing) * (R * kWave) + lane;
#pragma unroll
        for (int q = 0; q < R; q++) slot[q * kWave] = src.score(q, tab);
        if (lane == 0) duo_publish(words + p, c + 1);
    }
}

template <int R>
CR_D double trio_score_consumer(const int np, const int n, const int m, const double* ring, int* words) {
    const int lane = threadIdx.x & (kWave - 1);
    TrioCols<R> st;
    st.reset();
    unsigned long long waited = 0;
    const int nb = kTrioBatch;
    int r0 = 0;
#pragma unroll 1
    for (int j0 = 0; j0 < m; j0 += nb) {
        const int jend = j0 + nb < m ? j0 + nb : m;
        double sc[kTrioBatch][R];
        trio_take<R>(ring, words, np, j0, jend, r0, sc, waited);
#pragma unroll
        for (int k = 0; k < kTrioBatch; k++)
            if (j0 + k < jend) st.advance_score(sc[k]);
#if !CR_TRIO_EARLY
        if (lane == 0) duo_publish(words + 4, jend);
#endif
    }
    // H[n][m]: row n - 1 lives in lane (n - 1) / R, slot (n - 1) % R (np.max of the matrix, by monotonicity)
    const int qo = (n - 1) % R;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) v = (q == qo) ? st.hprev[q] : v;
    return lane_value(v, ((n - 1) / R) % kWave);
}

// ---------------------------------------------------------------------------------------------
// Both stages of a pair in one launch: 192 threads, wave 0 = recurrences + walks + superpositions + metrics, waves 1, 2 =
// scores.  Pairs of at most 64 R rows (one strip); decision words, transforms and results exactly as k_seed / k_align /
// k_score leave them.  Dynamic LDS: max(trio_lds_doubles<R>(m_max), kExpDoubles + trace_lds_doubles) doubles.
// ---------------------------------------------------------------------------------------------
template <int R, int D, bool SCORES>
__global__ __launch_bounds__(kTrioMaxWaves* kWave, 2) void k_pair_trio(const PairDesc* __restrict__ pairs, const double* __restrict__ tensors, int d,
                                                                   const double* __restrict__ coords, double gamma_tensor,
                                                                   double gamma_coords, double gap_open, double gap_extend,
                                                                   int seed_entries, int align_entries, int np2, uint32_t* __restrict__ dirs,
                                                                   uint32_t* __restrict__ bits, Transform* __restrict__ xf,
                                                                   double* __restrict__ seed_score, int32_t* __restrict__ aln,
                                                                   PairResult* __restrict__ res, const HostOut hout) {
    extern __shared__ double lds[];
    __shared__ Transform s_tr;
    CR_STAMP(0);
    const PairDesc pd = pairs[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef CR_STAMPS
    // where this wave runs (HW_REG_HW_ID: wave, SIMD, CU, SH, SE; HW_REG_XCC_ID): slot 3 of the wave's stamps (tools/stamps.py, STAMPS_PLACEMENT=1)
    CR_DUO_STAMP(w, 3, ((unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf) << 32) |
                           (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));
#endif
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    int* words = reinterpret_cast<int*>(lds + kExpDoubles);
    double* ring = lds + kExpDoubles + 4;
    double* cols = ring + (size_t)kTrioRing * R * kWave;
    load_exp_table(lds, threadIdx.x);
    if (threadIdx.x < 8) words[threadIdx.x] = 0;
    __syncthreads();
    SeedMax sm;
    sm.score = 0.0;
    sm.i = sm.j = 0;
    if (w == 0) {
        __builtin_amdgcn_s_setprio(CR_TRIO_PRIO_CONS);
        trio_seed_consumer<R>((int)(blockDim.x >> 6) - 1, pd.n, pd.m, ring, words, dirs + pd.dirs_off, sm);
        drain_stores();
        CR_STAMP(1);
        Transform tr;
        seed_trace<R, 0>(pd, seed_entries, coords, dirs, sm, lds + kExpDoubles, tr);
        if (threadIdx.x == 0) {
            xf[blockIdx.x] = tr;
            seed_score[blockIdx.x] = sm.score;
            s_tr = tr;
        }
        CR_STAMP(3);
    } else {
        __builtin_amdgcn_s_setprio(CR_TRIO_PRIO_PROD);
        RbfTensor<R, D> src;
        src.rows_g = tensors + pd.off_i * d;
        src.cols_g = tensors + pd.off_j * d;
        src.d = d;
        src.neg_gamma = -gamma_tensor;
        trio_seed_producer<R, D>(w - 1, (int)(blockDim.x >> 6) - 1, src, pd.n, pd.m, tab, ring, words);
    }
    __syncthreads();                                       // the seed superposition is there; the ring is free again
    CR_STAMP(4);
    {
        RbfCoords<R> src;                                  // the columns in the seed's frame, resident (RbfCoords::load_resident)
        src.rows_g = coords + pd.off_i * 3;
        src.cols_g = coords + pd.off_j * 3;
        src.xf = &s_tr;
        src.neg_gamma = -gamma_coords;
        src.load_resident(cols, pd.m, pd.m, (int)threadIdx.x, (int)blockDim.x);
        if (threadIdx.x < 8) words[threadIdx.x] = 0;
        __syncthreads();
        // The two stages can run with different numbers of score waves (np2 <= blockDim.x / 64 - 1 in the second): the seed's
        // tensor scores are 50 of a cell's 66 instructions, the coordinate scores 23 of 50.  The waves the second stage does not
        // use have helped to load its resident columns and leave here (a finished wave no longer counts at a barrier -- there is
        // none behind this point anyway).
        if (w > np2) return;
        const int lanes_here = pd.n >= kWave * R ? kWave : (pd.n + R - 1) / R;
        const int T = pd.m + lanes_here - 1;
        const int np = np2;
        if (w != 0) {
            if constexpr (SCORES) trio_score_producer<R>(w - 1, np, src, pd.n, pd.m, tab, cols, ring, words);
            else trio_align_producer<R>(w - 1, np, src, pd.n, pd.m, T, tab, cols, ring, words);
            return;                                        // wave 0 goes on alone (wave_sync, no s_barrier from here on)
        }
        PairResult r;
        if constexpr (SCORES) {
            r.sw = trio_score_consumer<R>(np, pd.n, pd.m, ring, words);
            r.dtw_score = 0.0;
#pragma unroll
            for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
            for (int x = 0; x < 3; x++) r.t[x] = 0.0;
            r.rmsd = r.coverage = r.tm = 0.0;
            r.aln_len = r.aln_start = 0;
            r.flags = 0;
        } else {
            AlignEnd e;
            SweepParams prm{0.0, gap_open, gap_extend};
            trio_align_consumer<R, kSwScore | kDtw | kZeroGap>(np, pd.n, pd.m, T, prm, ring, words, bits + pd.bt_off, e);
            drain_stores();
            CR_STAMP(5);
            // (the producers may still be leaving their last loop iteration: they touch LDS no more -- their last writes were
            // consumed above -- so the entries and the sum scratch can take the ring's place)
            align_trace<R>(pd, align_entries, coords, bits, e, lds + kExpDoubles, aln, r, hout);
        }
        r.seed_score = sm.score;
        r.seed_len = s_tr.seed_len;
        r.flags |= s_tr.flags;
        if (threadIdx.x == 0) {
            res[blockIdx.x] = r;
            if (!SCORES && hout.res) hout.res[hout.dst(blockIdx.x)] = r;
        }
        CR_STAMP(7);
    }
}

}  // namespace cr
