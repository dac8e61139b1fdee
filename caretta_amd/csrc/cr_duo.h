// Mid-size pair lists: one small WORKGROUP per pair, one wave per strip (2 .. 4 waves), strips paced by LDS progress words.
//
// A list of a few hundred to a few thousand pairs of 200 .. 600 rows -- ONE GPU'S SHARE OF THE HEADLINE CONFIGURATION ON
// 8 GPUs is 1 016 pairs of 300 x 300 -- gives the single-wave kernels (k_seed / k_align: one wave per pair) about one wave per
// SIMD, and a wave that has its SIMD to itself issues one FP64-rate instruction per ~6.5 cycles instead of one per 4
// (DESIGN.md 4.1c).  The wide layout (k_pair_wide) is built for one pair per CU: its workgroup-wide ordered sums take a
// 73 KB term tile, so at most two pairs share a CU, and its strips meet at a barrier every 8 steps.  Here a pair is TWO
// waves (3 rows per lane in strip 0, 2 in strip 1: 320 rows -- the five row slots of the single-wave kernel -- up to four
// waves for longer rows), the workgroup needs ~20 KB of LDS, so four or more pairs share a CU and every SIMD holds two or
// more waves of DIFFERENT instruction streams.  No barrier inside a fill:
//   * the last row of strip s goes to strip s + 1 through a FULL-LENGTH LDS array (NB values per column, never
//     overwritten), so strip s never waits for anybody;
//   * strip s publishes the number of steps it has completed in an LDS word every kDuoPublish steps (LDS executes one
//     wave's instructions in order: the word is written behind the edge values it covers); strip s + 1 polls that word
//     once per kDuoPublish steps and otherwise runs at its own pace.  Waves of one pair drift freely, a wave that waits
//     sleeps (s_sleep) and leaves its SIMD to the other pairs' waves.
// Both stages in one launch as k_pair_wide (seed fill -> walk + Kabsch by wave 0 -> align fill -> walk + Kabsch + metrics by
// wave 0); decision words in the wide layout (StripGeom / WidePlan), so the walkers are the shared ones.  Gap 0 only
// (sw_gap != 0 runs the same layout on k_pair_wide).  Every value bit-identical to the single-wave kernels: same providers,
// dp_column, ColSweep::step, walkers and ordered sums.
//
// Reference: multiple_alignment.py:321-349 (score function), :158-170 (pair loop), dynamic_time_warping.py:8-144, :205-278.
#pragma once

#include "cr_kernels.h"

namespace cr {

// Diagnostic build only (-DCR_STAMPS): per wave w of the first 4096 blocks, slots w * 8 + {0: seed loop start, 1: seed loop
// end, 2: cycles waited in the seed loop, 4: align loop start, 5: align loop end, 6: cycles waited in the align loop}.
#ifdef CR_STAMPS
static __device__ unsigned long long g_duo_stamps[4096 * 32];
#define CR_DUO_STAMP(w, k, v)                                                                                    \
    do {                                                                                                         \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) g_duo_stamps[blockIdx.x * 32 + (w) * 8 + (k)] = (v);   \
    } while (0)
#define CR_DUO_NOW() ((unsigned long long)__builtin_amdgcn_s_memtime())
#else
#define CR_DUO_STAMP(w, k, v) \
    do {                      \
    } while (0)
#define CR_DUO_NOW() 0ull
#endif

constexpr int kDuoMaxWaves = 8;
constexpr int kDuoPublish = 8;          // steps (columns) between two publications of a strip's progress
constexpr int kDuoColRing = 64;         // column sweeps: columns of the row above a strip kept in its ring
constexpr int kDuoEdge = 128;           // skewed sweep: columns per value kept in a ring (the reader is 64 .. 71 behind the writer)

// The strips of a pair form a chain: strip w + 1 can never be ahead of strip w, so the pair is as fast as its FIRST strip.
// Every SIMD holds waves of several pairs; the earlier strips ask the SIMD's arbiter for priority (s_setprio 3 - w), so a
// first strip is not held up by a later strip of another pair that would only run into its own wait.
CR_D void duo_priority(int w) {
    switch (w) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
    }
}

CR_D void duo_publish(int* word, int steps_done) {
    asm volatile("" ::: "memory");       // (compiler: the edge values of these steps are written first)
    *reinterpret_cast<volatile int*>(word) = steps_done;
}

CR_D void duo_wait(const int* word, int need, unsigned long long& waited) {
#ifdef CR_STAMPS
    const unsigned long long t0 = CR_DUO_NOW();
#endif
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(word)) < need) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");       // (compiler: edge values are read behind the word)
#ifdef CR_STAMPS
    waited += CR_DUO_NOW() - t0;
#endif
}

// LDS (doubles) of the three fills.  `m`: columns of the pair list's longest structure.
__host__ __device__ inline size_t duo_cols_lds_doubles(int waves, int /*m*/) {
    return kExpDoubles + (size_t)waves * kDuoColRing + (size_t)waves * 4 + kDuoMaxWaves / 2;
}
template <int MODE, class Src>
__host__ __device__ inline size_t duo_sweep_lds_doubles(int waves, int m) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    return kExpDoubles + (size_t)Src::kColDoubles * m + (size_t)waves * NB * kDuoEdge + (size_t)waves * 8 + kDuoMaxWaves / 2;
}
template <class Src>
__host__ __device__ inline size_t duo_score_lds_doubles(int waves, int m) {
    return kExpDoubles + (size_t)Src::kColDoubles * m + (size_t)waves * kDuoColRing + 8 + kDuoMaxWaves / 2;
}

// ---------------------------------------------------------------------------------------------
// Seed fill: the column sweep (sweep_cols_team's arithmetic), strip w follows strip w - 1 by progress word.
// LDS (doubles): exp table | (NW - 1) edge rows of m | NW * 4 reduction slots | progress words.
// ---------------------------------------------------------------------------------------------
template <int R, int D>
CR_D void sweep_cols_duo(RbfTensor<R, D>& src, const int n, const int m, double* lds, uint32_t* __restrict__ sw_dirs,
                         SeedMax& seed_out, const StripGeom geom) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* edges = lds + kExpDoubles;
    double* edge_out = edges + (size_t)w * kDuoColRing;
    const double* edge_in = edges + (size_t)(w > 0 ? w - 1 : 0) * kDuoColRing;
    double* red = edges + (size_t)NW * kDuoColRing;
    int* prog = reinterpret_cast<int*>(red + NW * 4);
    load_exp_table(lds, threadIdx.x);
    if (threadIdx.x < kDuoMaxWaves) prog[threadIdx.x] = 0;

    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const int TB = (m + 15) >> 4;
    const bool mine = w < nstrips;
    const bool full = src.d == D;
    const int rowbase = geom.rowbase0 + lane * R;
    const bool hand_out = w + 1 < nstrips;
    ColSweep<R, D> st;
    st.reset();
    if (mine) src.load_rows(rowbase, n);
    const int chunks = (m + kDuoPublish - 1) / kDuoPublish;
    __syncthreads();                                     // exp table and progress words
    duo_priority(w);

    unsigned long long waited = 0;
    CR_DUO_STAMP(w, 0, CR_DUO_NOW());
    auto run = [&](auto full_tag, auto top_tag) {
        constexpr bool FULL = decltype(full_tag)::value, TOP = decltype(top_tag)::value;
        st.template prefetch<FULL>(src, 0);
#pragma unroll 1
        for (int c = 0; c < chunks; c++) {
            const int j0 = c * kDuoPublish;
            const int jend = j0 + kDuoPublish < m ? j0 + kDuoPublish : m;
            double top_vec = 0.0;                        // the row above the strip for this chunk: lane x holds column j0 + x
            if constexpr (TOP) {
                duo_wait(prog + w - 1, jend, waited);
                if (lane < jend - j0) top_vec = edge_in[(j0 + lane) & (kDuoColRing - 1)];
            }
            // (the strip below has taken the ring slots these columns go to: it reads a chunk when it starts it)
            if (hand_out && jend > kDuoColRing) duo_wait(prog + w + 1, jend - kDuoColRing, waited);
#pragma unroll 1
            for (int j = j0; j < jend; j++) {
                st.template step<FULL, TOP>(src, tab, j, j + 1 < m ? j + 1 : j, TOP ? lane_value(top_vec, j - j0) : 0.0);
                if (hand_out && lane == kWave - 1) edge_out[j & (kDuoColRing - 1)] = st.hprev[R - 1];
            }
            if (lane == 0) duo_publish(prog + w, jend);
            if (((jend - 1) & 15) == 15 || jend == m)                            // a decision word holds 16 columns
                st.flush(sw_dirs, ((int64_t)geom.slot0 * TB + (int64_t)((jend - 1) >> 4) * R) * kWave + lane);
        }
    };
    if (mine) {
        if (w == 0) {
            if (full) run(std::true_type{}, std::false_type{});
            else run(std::false_type{}, std::false_type{});
        } else {
            if (full) run(std::true_type{}, std::true_type{});
            else run(std::false_type{}, std::true_type{});
        }
    }
    CR_DUO_STAMP(w, 1, CR_DUO_NOW());
    CR_DUO_STAMP(w, 2, waited);

    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    if (mine) st.fold(rowbase, best_v, best_i, best_j);
    wave_first_max(best_v, best_i, best_j);
    if (lane == 0) {
        red[w * 4 + 0] = best_v;
        red[w * 4 + 1] = (double)best_i;
        red[w * 4 + 2] = (double)best_j;
    }
    __threadfence();                                   // decision words of every wave visible to wave 0's walk
    __syncthreads();
    best_v = 0.0;
    best_i = best_j = 0x7fffffff;
    for (int x = 0; x < nstrips; x++) {
        const double ov = red[x * 4 + 0];
        const int oi = (int)red[x * 4 + 1], oj = (int)red[x * 4 + 2];
        const bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
        best_v = take ? ov : best_v;
        best_i = take ? oi : best_i;
        best_j = take ? oj : best_j;
    }
    seed_out.score = best_v;
    seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
    seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Alignment fill: the time-skewed sweep (sweep_wide's arithmetic: all m columns resident in LDS), strip w follows strip
// w - 1 by progress word.  Lane 0 of strip w needs column t of the row above in its step t; lane 63 of strip w - 1 forms
// it in ITS step t + 63; progress is published every kDuoPublish steps, so strip w runs 64 .. 71 steps behind.
// LDS (doubles): exp table | Src::kColDoubles planes of m | (NW - 1) * NB edge rows of m | NW * 8 | progress words.
// ---------------------------------------------------------------------------------------------
template <int R, int MODE, class Src>
CR_D void sweep_duo(Src& src, const int n, const int m, const SweepParams prm, double* lds, uint32_t* __restrict__ sw_dirs,
                    uint32_t* __restrict__ dtw_bits, SeedMax& seed_out, AlignEnd& end_out, const StripGeom geom) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const int stride = m;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* res = lds + kExpDoubles;
    double* edges = res + (size_t)Src::kColDoubles * stride;
    double* edge_out = edges + (size_t)w * (NB * kDuoEdge);
    const double* edge_in = edges + (size_t)(w > 0 ? w - 1 : 0) * (NB * kDuoEdge);
    double* red = edges + (size_t)NW * (NB * kDuoEdge);
    int* prog = reinterpret_cast<int*>(red + NW * 8);

    load_exp_table(lds, threadIdx.x);
    src.load_resident(res, stride, m, (int)threadIdx.x, (int)blockDim.x);
    if (threadIdx.x < kDuoMaxWaves) prog[threadIdx.x] = 0;

    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;
    const bool mine = w < nstrips;
    const bool hand_out = w + 1 < nstrips;
    const int rowbase = geom.rowbase0 + lane * R;
    const int rows_here = n - geom.rowbase0;
    const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
    const int T = mine ? m + lanes_here - 1 : 0;
    const int T_above = m + kWave - 1;                   // (a strip with a strip below it is full)

    DpState<R> st;
    st.sw_max = 0.0;
    if (mine) src.load_rows(rowbase, n);
    st.reset_column0(col0_m2);
#pragma unroll
    for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;

    constexpr bool AHEAD = R <= 2;                         // scores one column ahead (sweep_wide)
    double sc_cur[R];
    __syncthreads();                                       // resident columns, exp table, progress words
    duo_priority(w);
    if constexpr (AHEAD) {
        src.fetch_resident(res, stride, 0);
#pragma unroll
        for (int q = 0; q < R; q++) sc_cur[q] = src.score(q, tab);
    }
    unsigned long long waited = 0;
    CR_DUO_STAMP(w, 4, CR_DUO_NOW());
#pragma unroll 1
    for (int t = 0; t < T; t++) {
        if ((t & (kDuoPublish - 1)) == 0) {
            if (w > 0) {
                const int need = t + kDuoPublish + kWave - 1;
                duo_wait(prog + w - 1, need < T_above ? need : T_above, waited);
            }
            // lane 63 writes columns t - 63 .. t - 56 in the next steps: the strip below (lane 0: column = step) is past the
            // columns kDuoEdge before them
            const int taken = t - (kWave - kDuoPublish) - kDuoEdge + 1;
            if (hand_out && taken > 0) duo_wait(prog + w + 1, taken, waited);
        }
        const int c = t - lane;
        const bool active = (unsigned)c < (unsigned)m;

        double h_top0 = 0.0, m0_top0 = col0_m2, m1_top0 = 0.0;
        if (w > 0 && lane == 0 && active) {
            if constexpr (SW) h_top0 = edge_in[c & (kDuoEdge - 1)];
            if constexpr (DTW) {
                m0_top0 = edge_in[(NB - 2) * kDuoEdge + (c & (kDuoEdge - 1))];
                m1_top0 = edge_in[(NB - 1) * kDuoEdge + (c & (kDuoEdge - 1))];
            }
        }
        double h_top = 0.0, m0_top = 0.0, m1_top = 0.0;
        if constexpr (SW) h_top = wave_shr1(st.h_left[R - 1], h_top0);
        if constexpr (DTW) {
            m0_top = wave_shr1(st.m0_left[R - 1], m0_top0);
            m1_top = wave_shr1(st.m1_left[R - 1], m1_top0);
        }
        const int sh2 = (t & 15) * 2, sh4 = (t & 7) * 4;

        if (active) {
            if constexpr (AHEAD) {
                double sc_next[R];
                src.fetch_resident(res, stride, c + 1 < m ? c + 1 : c);
#pragma unroll
                for (int q = 0; q < R; q++) sc_next[q] = src.score(q, tab);
                dp_column<R, MODE>(src, st, prm, tab, c, rowbase, n, sh2, sh4, h_top, m0_top, m1_top, sc_cur);
#pragma unroll
                for (int q = 0; q < R; q++) sc_cur[q] = sc_next[q];
            } else {
                src.fetch_resident(res, stride, c);
                dp_column<R, MODE>(src, st, prm, tab, c, rowbase, n, sh2, sh4, h_top, m0_top, m1_top);
            }
            if (hand_out && lane == kWave - 1) {
                if constexpr (SW) edge_out[c & (kDuoEdge - 1)] = st.h_left[R - 1];
                if constexpr (DTW) {
                    edge_out[(NB - 2) * kDuoEdge + (c & (kDuoEdge - 1))] = st.m0_left[R - 1];
                    edge_out[(NB - 1) * kDuoEdge + (c & (kDuoEdge - 1))] = st.m1_left[R - 1];
                }
            }
        }
        const bool word_end = (t & (kDuoPublish - 1)) == kDuoPublish - 1 || t == T - 1;
        if (word_end && lane == 0) duo_publish(prog + w, t + 1);
        if constexpr (TRACE) {
            if ((t & 15) == 15 || t == T - 1) {
                const int64_t base = ((int64_t)geom.slot0 * TB_SW + (int64_t)(t >> 4) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    sw_dirs[base + q * kWave] = st.swbits[q];
                    st.swbits[q] = 0;
                }
            }
        }
        if constexpr (DTW) {
            if (word_end) {
                const int64_t base = ((int64_t)geom.slot0 * TB_DTW + (int64_t)(t >> 3) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    dtw_bits[base + q * kWave] = st.dtbits[q];
                    st.dtbits[q] = 0;
                }
            }
        }
    }
    CR_DUO_STAMP(w, 5, CR_DUO_NOW());
    CR_DUO_STAMP(w, 6, waited);
    wide_finish<R, MODE>(st, mine, w, lane, rowbase, geom, red, seed_out, end_out);
}

// ---------------------------------------------------------------------------------------------
// smith_waterman_score (gap 0) alone, for the matrix entries (sweep_cols_score_team's arithmetic; columns resident, read
// with wave-uniform addresses).  LDS (doubles): exp table | Src::kColDoubles planes of m | (NW - 1) edge rows of m | 8 | words.
// ---------------------------------------------------------------------------------------------
template <int R, class Src>
CR_D double sweep_cols_score_duo(Src& src, const int n, const int m, double* lds, const StripGeom geom) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const int stride = m;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* res = lds + kExpDoubles;
    double* edges = res + (size_t)Src::kColDoubles * stride;
    double* edge_out = edges + (size_t)w * kDuoColRing;
    const double* edge_in = edges + (size_t)(w > 0 ? w - 1 : 0) * kDuoColRing;
    double* red = edges + (size_t)NW * kDuoColRing;
    int* prog = reinterpret_cast<int*>(red + 8);
    load_exp_table(lds, threadIdx.x);
    src.load_resident(res, stride, m, (int)threadIdx.x, (int)blockDim.x);
    if (threadIdx.x < kDuoMaxWaves) prog[threadIdx.x] = 0;
    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const bool mine = w < nstrips;
    const int rowbase = geom.rowbase0 + lane * R;
    const bool hand_out = w + 1 < nstrips;
    double hprev[R], eprev = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) hprev[q] = 0.0;
    if (mine) src.load_rows(rowbase, n);
    const int chunks = mine ? (m + kDuoPublish - 1) / kDuoPublish : 0;
    __syncthreads();
    duo_priority(w);
#pragma unroll 1
    for (int c = 0; c < chunks; c++) {
        const int j0 = c * kDuoPublish;
        const int jend = j0 + kDuoPublish < m ? j0 + kDuoPublish : m;
        double top_vec = 0.0;
        unsigned long long waited = 0;
        if (w > 0) {
            duo_wait(prog + w - 1, jend, waited);
            if (lane < jend - j0) top_vec = edge_in[(j0 + lane) & (kDuoColRing - 1)];
        }
        if (hand_out && jend > kDuoColRing) duo_wait(prog + w + 1, jend - kDuoColRing, waited);
#pragma unroll 1
        for (int j = j0; j < jend; j++) {
            src.fetch_resident(res, stride, j);
            double p[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const double sc = src.score(q, tab);
                const double dg = (q == 0 ? eprev : hprev[q - 1]) + sc;
                const double b = vmax(dg, hprev[q]);
                p[q] = q == 0 ? b : vmax(p[q - 1], b);
            }
            double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
            if (w > 0) e = vmax(e, lane_value(top_vec, j - j0));
#pragma unroll
            for (int q = 0; q < R; q++) hprev[q] = vmax(p[q], e);
            eprev = e;
            if (hand_out && lane == kWave - 1) edge_out[j & (kDuoColRing - 1)] = hprev[R - 1];
        }
        if (lane == 0) duo_publish(prog + w, jend);
    }
    const int qo = geom.owner_q;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) v = (q == qo) ? hprev[q] : v;
    if (w == geom.owner_wave && lane == geom.owner_lane) red[0] = v;
    __syncthreads();
    const double out = red[0];
    __syncthreads();
    return out;
}

// ---------------------------------------------------------------------------------------------
// Both stages of a pair in one launch.  Dynamic LDS (doubles): the largest of the three fills and of
// kExpDoubles + trace_lds_doubles (the walks and ordered sums are wave 0's, as in k_seed / k_align).
// ---------------------------------------------------------------------------------------------
template <int RA, int RB, int D, bool SCORES>
__global__ __launch_bounds__(kDuoMaxWaves* kWave, 2) void k_pair_duo(const PairDesc* __restrict__ pairs,
                                                                    const double* __restrict__ tensors, int d,
                                                                    const double* __restrict__ coords, double gamma_tensor,
                                                                    double gamma_coords, double gap_open, double gap_extend,
                                                                    int seed_entries, int align_entries, int nA,
                                                                    uint32_t* __restrict__ dirs, uint32_t* __restrict__ bits,
                                                                    Transform* __restrict__ xf, double* __restrict__ seed_score,
                                                                    int32_t* __restrict__ aln, PairResult* __restrict__ res,
                                                                    const HostOut hout) {
    extern __shared__ double lds[];
    __shared__ Transform s_tr;
    CR_STAMP(0);
    const PairDesc pd = pairs[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const WidePlan<RA, RB> plan{nA};
    const StripGeom geom = plan.geom(w, pd.n);
    SeedMax sm;
    {
        auto fill = [&](auto rtag) {
            constexpr int R = decltype(rtag)::value;
            RbfTensor<R, D> src;
            src.rows_g = tensors + pd.off_i * d;
            src.cols_g = tensors + pd.off_j * d;
            src.d = d;
            src.neg_gamma = -gamma_tensor;
            sweep_cols_duo<R, D>(src, pd.n, pd.m, lds, dirs + pd.dirs_off, sm, geom);
        };
        if (plan.wave_in_a(w)) fill(std::integral_constant<int, RA>{});
        else if constexpr (RA != RB) fill(std::integral_constant<int, RB>{});
    }
    if (threadIdx.x < kWave) {                             // wave 0 walks and superposes; the others wait at the barrier
        CR_STAMP(1);
        Transform tr;
        seed_trace<RA, 0, RB>(pd, seed_entries, coords, dirs, sm, lds + kExpDoubles, tr, nA);
        if (threadIdx.x == 0) {
            xf[blockIdx.x] = tr;
            seed_score[blockIdx.x] = sm.score;
            s_tr = tr;
        }
        CR_STAMP(3);
    }
    __syncthreads();
    CR_STAMP(4);
    AlignEnd e;
    double sw_only = 0.0;
    {
        SeedMax unused;
        auto fill = [&](auto rtag) {
            constexpr int R = decltype(rtag)::value;
            RbfCoords<R> src;
            src.rows_g = coords + pd.off_i * 3;
            src.cols_g = coords + pd.off_j * 3;
            src.xf = &s_tr;
            src.neg_gamma = -gamma_coords;
            if constexpr (SCORES) {
                sw_only = sweep_cols_score_duo<R>(src, pd.n, pd.m, lds, geom);
            } else {
                SweepParams prm{0.0, gap_open, gap_extend};
                sweep_duo<R, kSwScore | kDtw | kZeroGap>(src, pd.n, pd.m, prm, lds, nullptr, bits + pd.bt_off, unused, e, geom);
            }
        };
        if (plan.wave_in_a(w)) fill(std::integral_constant<int, RA>{});
        else if constexpr (RA != RB) fill(std::integral_constant<int, RB>{});
    }
    if (threadIdx.x >= kWave) return;                      // wave 0 goes on alone (wave_sync, no s_barrier from here on)
    CR_STAMP(5);
    PairResult r;
    if constexpr (SCORES) {
        r.sw = sw_only;
        r.dtw_score = 0.0;
#pragma unroll
        for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
        for (int x = 0; x < 3; x++) r.t[x] = 0.0;
        r.rmsd = r.coverage = r.tm = 0.0;
        r.aln_len = r.aln_start = 0;
        r.flags = 0;
    } else {
        align_trace<RA, RB>(pd, align_entries, coords, bits, e, lds + kExpDoubles, aln, r, hout, nA);
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

}  // namespace cr
