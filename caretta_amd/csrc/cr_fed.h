// One pair per workgroup with its strips FED: every strip of the wide layout (cr_kernels.h: k_pair_wide, one wave per strip of
// 2 or 3 rows per lane) gets a second wave that forms its scores (k_pair_fed).
//
// One GPU's share of a long-chain family (BASELINE config 5 on 8 GPUs: 252 pairs of 1200 x 1200, one pair per CU) runs the
// wide layout with eight waves per CU -- two per SIMD -- and each of them carries the whole chain of a step: score, then
// recurrence, then the cross-lane hand-off.  Two such waves leave a SIMD idle in 40 % of its cycles (profiles/r04/pmc_c5share.json:
// issue fraction 0.60).  As in cr_trio.h the split is by FUNCTION: 50 of the 66 instructions of a seed cell and 23 of the 50 of an
// alignment cell form exp(-gamma |a - b|^2), which depends on nothing the recurrence produces.  Strip w keeps its CONSUMER wave
// (recurrence, decisions, hand-off to the strip below -- paced by LDS progress words as in cr_duo.h, no barrier inside a fill)
// and gets a PRODUCER wave (w + NW) that forms the strip's scores a few columns (steps) ahead into an LDS ring.  Sixteen waves
// per CU, four per SIMD: two latency-bound chains and two throughput-bound score streams.
//
// Words per strip (LDS executes one wave's instructions in order): done[w] = columns (steps) strip w has finished (read by the
// strip below and, for the edge ring, by the strip itself one ring later), made[w] = columns (steps) its producer has
// finished, taken[w] = columns (steps) whose scores the consumer has taken into registers (the producer may reuse their slots).
// Ring of strip w: `ring` columns (steps) x RA x 64 doubles, slot = column (step) mod ring; the consumer takes kFedBatch at a time.
// Decision words, walkers and the workgroup-wide ordered sums are k_pair_wide's: every value bit-identical to it.  Gap 0 only.
//
// Reference: multiple_alignment.py:321-349 (score function), :158-170 (pair loop), dynamic_time_warping.py:8-144, :205-278.
#pragma once

#include "cr_trio.h"

namespace cr {

constexpr int kFedMaxStrips = 8;        // consumers; as many producers
constexpr int kFedBatch = 4;            // columns (steps) per wait of a consumer for its producer
constexpr int kFedHeadDoubles = 16 + 2 * kFedMaxStrips * 8;      // words (3 x 8 ints, padded) | red: 8 doubles per wave

// LDS (doubles).  Seed fill: exp table | head | NW edge rings of kDuoColRing | NW score rings.
// Alignment fill: exp table | head | 3 planes of m resident columns | NW x 3 edge rings of kDuoEdge | NW score rings.
template <int RA>
__host__ __device__ inline size_t fed_seed_lds_doubles(int nw, int ring) {
    return (size_t)kExpDoubles + kFedHeadDoubles + (size_t)nw * kDuoColRing + (size_t)nw * ring * RA * kWave;
}
template <int RA>
__host__ __device__ inline size_t fed_align_lds_doubles(int nw, int ring, int m) {
    return (size_t)kExpDoubles + kFedHeadDoubles + (size_t)3 * m + (size_t)nw * 3 * kDuoEdge + (size_t)nw * ring * RA * kWave;
}

// the recurrence of ColSweep::step on scores that are already there (same operations on the same values), with the row
// above the strip
template <int R>
struct FedCols {
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
    template <bool TOP>
    CR_D void advance(const double* sc, int j, double top) {
        double dg[R], p[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            dg[q] = (q == 0 ? eprev : hprev[q - 1]) + sc[q];
            const double b = vmax(dg[q], hprev[q]);
            p[q] = q == 0 ? b : vmax(p[q - 1], b);
        }
        double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
        if constexpr (TOP) e = vmax(e, top);
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
    CR_D void flush(uint32_t* __restrict__ sw_dirs, int64_t base) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            sw_dirs[base + q * kWave] = bits[q];
            bits[q] = 0;
        }
    }
    CR_D void fold(int rowbase, double& best_v, int& best_i, int& best_j) const {
#pragma unroll
        for (int q = 0; q < R; q++) {
            const bool gt = hprev[q] > best_v;
            best_v = gt ? hprev[q] : best_v;
            best_i = gt ? rowbase + q : best_i;
            best_j = gt ? rowfirst[q] : best_j;
        }
    }
};

struct FedLds {                          // where things are (all waves compute the same)
    int* done;
    int* taken;
    int* made;
    double* red;
    double* body;                        // behind the head
};
CR_D FedLds fed_lds(double* lds) {
    FedLds f;
    int* words = reinterpret_cast<int*>(lds + kExpDoubles);
    f.done = words;
    f.taken = words + kFedMaxStrips;
    f.made = words + 2 * kFedMaxStrips;
    f.red = lds + kExpDoubles + 16;
    f.body = lds + kExpDoubles + kFedHeadDoubles;
    return f;
}

// take the scores of columns (steps) [jb, jbe) of the strip's ring into registers and give the slots back
template <int R, int RA>
CR_D void fed_take(const double* myring, int ring, int* made, int* taken, int jb, int jbe, double (&sc)[kFedBatch][R], unsigned long long& waited) {
    const int lane = threadIdx.x & (kWave - 1);
    trio_wait<CR_TRIO_CONS_SLEEP>(made, jbe, waited);
    const double* slot = myring + (size_t)((unsigned)jb & (unsigned)(ring - 1)) * (RA * kWave) + lane;     // (jb, ring: multiples of kFedBatch)
#pragma unroll
    for (int k = 0; k < kFedBatch; k++) {
#pragma unroll
        for (int q = 0; q < R; q++) sc[k][q] = slot[(k * RA + q) * kWave];       // (columns past jbe: stale slots, never used)
    }
    if (lane == 0) duo_publish(taken, jbe);
}

// ---- seed stage: column sweep ------------------------------------------------------------------------------------------
template <int R, int RA, int D>
CR_D void fed_seed_producer(const int w, RbfTensor<R, D>& src, const int rowbase, const int n, const int m, const ExpEntry* tab, double* myring,
                            const int ring, const FedLds f, unsigned long long& waited) {
    const int lane = threadIdx.x & (kWave - 1);
    const bool full = src.d == D;
    src.load_rows(rowbase, n);
    // TWO columns per iteration: six independent chains of squared distances and exps (a score wave is there to fill issue
    // slots: with one column its three chains, the table gathers and the poll of the ring word expose their latencies one after
    // the other), one poll and one publication for both.  The columns' features are wave-uniform scalar loads into two sets of
    // registers, requested for the next iteration while the exps run.
    auto fetch = [&](int j, double (&set)[D]) {
        const double* __restrict__ cg = src.cols_g;
        const int d = full ? D : src.d;
#pragma unroll
        for (int k = 0; k < D; k++) {
            const double v = cg[(int64_t)j * d + k];
            set[k] = (full || k < d) ? v : 0.0;
        }
    };
    if (m > 0) {
        fetch(0, src.col);
        fetch(1 < m ? 1 : 0, src.col2);
    }
#pragma unroll 1
    for (int c = 0; c < m; c += 2) {
        const bool two = c + 1 < m;
        if (c + 1 >= ring) trio_wait<CR_TRIO_PROD_SLEEP>(f.taken + w, c + 1 - ring + 1, waited);   // the slots' last columns have been taken
        double acc0[R], acc1[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            acc0[q] = src.dist2_of(q, src.col);
            acc1[q] = src.dist2_of(q, src.col2);
        }
        fetch(c + 2 < m ? c + 2 : c, src.col);           // (the next two columns while the exps run)
        fetch(c + 3 < m ? c + 3 : c, src.col2);
        double* slot0 = myring + (size_t)((unsigned)c & (unsigned)(ring - 1)) * (RA * kWave) + lane;
        double* slot1 = myring + (size_t)((unsigned)(c + 1) & (unsigned)(ring - 1)) * (RA * kWave) + lane;
        double e0[R], e1[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            e0[q] = exp_tab<true>(src.neg_gamma * acc0[q], tab);
            e1[q] = exp_tab<true>(src.neg_gamma * acc1[q], tab);
        }
#pragma unroll
        for (int q = 0; q < R; q++) slot0[q * kWave] = e0[q];
        if (two) {
#pragma unroll
            for (int q = 0; q < R; q++) slot1[q * kWave] = e1[q];
        }
        if (lane == 0) duo_publish(f.made + w, two ? c + 2 : c + 1);
    }
}

template <int R, int RA>
CR_D void fed_seed_consumer(const int w, const int NW, const int rowbase, const int n, const int m, const double* myring, const int ring, const FedLds f,
                            uint32_t* __restrict__ sw_dirs, const StripGeom geom, FedCols<R>& st, unsigned long long& waited) {
    const int lane = threadIdx.x & (kWave - 1);
    double* edges = f.body;
    double* edge_out = edges + (size_t)w * kDuoColRing;
    const double* edge_in = edges + (size_t)(w > 0 ? w - 1 : 0) * kDuoColRing;
    const int TB = (m + 15) >> 4;
    const bool hand_out = w + 1 < geom.nstrips;
    const int chunks = (m + kDuoPublish - 1) / kDuoPublish;
    auto run = [&](auto top_tag) {
        constexpr bool TOP = decltype(top_tag)::value;
#pragma unroll 1
        for (int c = 0; c < chunks; c++) {
            const int j0 = c * kDuoPublish;
            const int jend = j0 + kDuoPublish < m ? j0 + kDuoPublish : m;
            double top_vec = 0.0;                        // the row above the strip for this chunk: lane x holds column j0 + x
            if constexpr (TOP) {
                duo_wait(f.done + w - 1, jend, waited);
                if (lane < jend - j0) top_vec = edge_in[(j0 + lane) & (kDuoColRing - 1)];
            }
            // (the strip below has taken the ring slots these columns go to: it reads a chunk when it starts it)
            if (hand_out && jend > kDuoColRing) duo_wait(f.done + w + 1, jend - kDuoColRing, waited);
#pragma unroll 1
            for (int jb = j0; jb < jend; jb += kFedBatch) {
                const int jbe = jb + kFedBatch < jend ? jb + kFedBatch : jend;
                double sc[kFedBatch][R];
                fed_take<R, RA>(myring, ring, f.made + w, f.taken + w, jb, jbe, sc, waited);
#pragma unroll
                for (int k = 0; k < kFedBatch; k++) {
                    const int j = jb + k;
                    if (j < jbe) {
                        st.template advance<TOP>(sc[k], j, TOP ? lane_value(top_vec, j - j0) : 0.0);
                        if (hand_out && lane == kWave - 1) edge_out[j & (kDuoColRing - 1)] = st.hprev[R - 1];
                    }
                }
            }
            if (lane == 0) duo_publish(f.done + w, jend);
            if (((jend - 1) & 15) == 15 || jend == m)                            // a decision word holds 16 columns
                st.flush(sw_dirs, ((int64_t)geom.slot0 * TB + (int64_t)((jend - 1) >> 4) * R) * kWave + lane);
        }
    };
    if (w == 0) run(std::false_type{});
    else run(std::true_type{});
    (void)NW;
    (void)n;
}

// ---- alignment stage: time-skewed sweep (step t, lane l -> column t - l) ---------------------------------------------------
template <int R, int RA>
CR_D void fed_align_producer(const int w, RbfCoords<R>& src, const int rowbase, const int n, const int m, const int T, const ExpEntry* tab,
                             const double* cols, double* myring, const int ring, const FedLds f, unsigned long long& waited) {
    const int lane = threadIdx.x & (kWave - 1);
    src.load_rows(rowbase, n);
    // TWO steps per iteration (see fed_seed_producer), every lane forms both scores whatever its column: lanes outside the
    // matrix take the nearest column and their values are never read (the consumer's cells are masked by its own column)
#pragma unroll 1
    for (int t = 0; t < T; t += 2) {
        const bool two = t + 1 < T;
        if (t + 1 >= ring) trio_wait<CR_TRIO_PROD_SLEEP>(f.taken + w, t + 1 - ring + 1, waited);
        int c0 = t - lane, c1 = t + 1 - lane;
        c0 = c0 < 0 ? 0 : c0 >= m ? m - 1 : c0;
        c1 = c1 < 0 ? 0 : c1 >= m ? m - 1 : c1;
        RbfCoords<R> other = src;                        // (the same rows, the second column)
        src.fetch_resident(cols, m, c0);
        other.fetch_resident(cols, m, c1);
        double* slot0 = myring + (size_t)((unsigned)t & (unsigned)(ring - 1)) * (RA * kWave) + lane;
        double* slot1 = myring + (size_t)((unsigned)(t + 1) & (unsigned)(ring - 1)) * (RA * kWave) + lane;
        double e0[R], e1[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            e0[q] = src.score(q, tab);
            e1[q] = other.score(q, tab);
        }
#pragma unroll
        for (int q = 0; q < R; q++) slot0[q * kWave] = e0[q];
        if (two) {
#pragma unroll
            for (int q = 0; q < R; q++) slot1[q * kWave] = e1[q];
        }
        if (lane == 0) duo_publish(f.made + w, two ? t + 2 : t + 1);
    }
}

template <int R, int RA, int MODE>
CR_D void fed_align_consumer(const int w, const int rowbase, const int n, const int m, const int T, const SweepParams prm, const double* myring,
                             const int ring, const FedLds f, double* edges, uint32_t* __restrict__ dtw_bits, const StripGeom geom, DpState<R>& st, unsigned long long& waited) {
    constexpr bool SW = (MODE & kSwScore) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);
    const int lane = threadIdx.x & (kWave - 1);
    double* edge_out = edges + (size_t)w * (NB * kDuoEdge);
    const double* edge_in = edges + (size_t)(w > 0 ? w - 1 : 0) * (NB * kDuoEdge);
    const int TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;
    const bool hand_out = w + 1 < geom.nstrips;
    const int T_above = m + kWave - 1;                   // (a strip with a strip below it is full)
    RbfCoords<R> traits;                                 // (dp_column reads only the provider's traits when the scores are given)
#pragma unroll 1
    for (int tb = 0; tb < T; tb += kFedBatch) {
        if ((tb & (kDuoPublish - 1)) == 0) {
            if (w > 0) {
                const int need = tb + kDuoPublish + kWave - 1;
                duo_wait(f.done + w - 1, need < T_above ? need : T_above, waited);
            }
            // lane 63 writes columns tb - 63 .. tb - 56 in the next steps: the strip below (lane 0: column = step) is past the
            // columns kDuoEdge before them
            const int past = tb - (kWave - kDuoPublish) - kDuoEdge + 1;
            if (hand_out && past > 0) duo_wait(f.done + w + 1, past, waited);
        }
        const int tbe = tb + kFedBatch < T ? tb + kFedBatch : T;
        double sc[kFedBatch][R];
        fed_take<R, RA>(myring, ring, f.made + w, f.taken + w, tb, tbe, sc, waited);
#pragma unroll
        for (int k = 0; k < kFedBatch; k++) {
            const int t = tb + k;
            if (t < tbe) {
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
                if (active) {
                    dp_column<R, MODE>(traits, st, prm, nullptr, c, rowbase, n, (t & 15) * 2, (t & 7) * 4, h_top, m0_top, m1_top, sc[k]);
                    if (hand_out && lane == kWave - 1) {
                        if constexpr (SW) edge_out[c & (kDuoEdge - 1)] = st.h_left[R - 1];
                        if constexpr (DTW) {
                            edge_out[(NB - 2) * kDuoEdge + (c & (kDuoEdge - 1))] = st.m0_left[R - 1];
                            edge_out[(NB - 1) * kDuoEdge + (c & (kDuoEdge - 1))] = st.m1_left[R - 1];
                        }
                    }
                }
                const bool word_end = (t & (kDuoPublish - 1)) == kDuoPublish - 1 || t == T - 1;
                if (word_end && lane == 0) duo_publish(f.done + w, t + 1);
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
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Both stages of a pair in one launch: 2 NW waves (NW = strips of the list's longest structure), waves [0, NW) = the strips'
// recurrences, waves [NW, 2 NW) = their scores; wave 0 walks, everybody takes the position-ordered sums (k_pair_wide's).
// Dynamic LDS: the largest of fed_seed_lds_doubles, fed_align_lds_doubles and kExpDoubles + trace_team_lds_doubles.
// ---------------------------------------------------------------------------------------------
template <int RA, int RB, int D>
__global__ __launch_bounds__(2 * kFedMaxStrips* kWave) void k_pair_fed(const PairDesc* __restrict__ pairs, const double* __restrict__ tensors, int d,
                                                                      const double* __restrict__ coords, double gamma_tensor, double gamma_coords,
                                                                      double gap_open, double gap_extend, int seed_entries, int align_entries, int nA,
                                                                      int ring, uint32_t* __restrict__ dirs, uint32_t* __restrict__ bits,
                                                                      Transform* __restrict__ xf, double* __restrict__ seed_score,
                                                                      int32_t* __restrict__ aln, PairResult* __restrict__ res, const HostOut hout) {
    extern __shared__ double lds[];
    __shared__ Transform s_tr;
    __shared__ int s_walk[4];
    CR_STAMP(0);
    const PairDesc pd = pairs[blockIdx.x];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 7);
    const bool producer = wv >= NW;
    const int w = producer ? wv - NW : wv;                 // the strip this wave works for
    const WidePlan<RA, RB> plan{nA};
    const StripGeom geom = plan.geom(w, pd.n);
    const bool mine = w < geom.nstrips;
    // (diagnostic build: stamps of the first and the last strip's consumer and producer)
    const int sg = (w == 0 ? 0 : w == geom.nstrips - 1 ? 2 : -1) + (producer ? 1 : 0);
    const bool stamped = w == 0 || w == geom.nstrips - 1;
    unsigned long long waited = 0;
    (void)sg;
    (void)stamped;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    const FedLds f = fed_lds(lds);
    load_exp_table(lds, threadIdx.x);
    if (threadIdx.x < 3 * kFedMaxStrips) f.done[threadIdx.x] = 0;
    __syncthreads();

    // ---- seed fill -----------------------------------------------------------------------------------------------------
    SeedMax sm;
    {
        double* rings = f.body + (size_t)NW * kDuoColRing;
        double* myring = rings + (size_t)w * ring * (RA * kWave);
        double best_v = 0.0;
        int best_i = 0x7fffffff, best_j = 0x7fffffff;
        if (stamped) CR_DUO_STAMP(sg, 0, CR_DUO_NOW());
        auto fill = [&](auto rtag) {
            constexpr int R = decltype(rtag)::value;
            const int rowbase = geom.rowbase0 + lane * R;
            if (producer) {
                __builtin_amdgcn_s_setprio(CR_TRIO_PRIO_PROD);
                RbfTensor<R, D> src;
                src.rows_g = tensors + pd.off_i * d;
                src.cols_g = tensors + pd.off_j * d;
                src.d = d;
                src.neg_gamma = -gamma_tensor;
                if (mine) fed_seed_producer<R, RA, D>(w, src, rowbase, pd.n, pd.m, tab, myring, ring, f, waited);
            } else {
                duo_priority(w);
                FedCols<R> st;
                st.reset();
                if (mine) {
                    fed_seed_consumer<R, RA>(w, NW, rowbase, pd.n, pd.m, myring, ring, f, dirs + pd.dirs_off, geom, st, waited);
                    st.fold(rowbase, best_v, best_i, best_j);
                }
            }
        };
        if (plan.wave_in_a(w)) fill(std::integral_constant<int, RA>{});
        else if constexpr (RA != RB) fill(std::integral_constant<int, RB>{});
        __builtin_amdgcn_s_setprio(0);
        if (stamped) {
            CR_DUO_STAMP(sg, 1, CR_DUO_NOW());
            CR_DUO_STAMP(sg, 2, waited);
        }
        waited = 0;
        wave_first_max(best_v, best_i, best_j);
        if (!producer && lane == 0) {
            f.red[w * 8 + 0] = best_v;
            f.red[w * 8 + 1] = (double)best_i;
            f.red[w * 8 + 2] = (double)best_j;
        }
        __threadfence();                                   // decision words of every wave visible to wave 0's walk
        __syncthreads();
        best_v = 0.0;
        best_i = best_j = 0x7fffffff;
        for (int x = 0; x < geom.nstrips; x++) {
            const double ov = f.red[x * 8 + 0];
            const int oi = (int)f.red[x * 8 + 1], oj = (int)f.red[x * 8 + 2];
            const bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
        sm.score = best_v;
        sm.i = best_v > 0.0 ? best_i + 1 : 0;
        sm.j = best_v > 0.0 ? best_j + 1 : 0;
        __syncthreads();
    }
    // ---- wave 0 walks (the others wait at the barrier); the position-ordered sums behind the walk are taken by everybody
    uint32_t* const seed_list = reinterpret_cast<uint32_t*>(lds + kExpDoubles);
    double* const seed_terms = lds + kExpDoubles + ((size_t)seed_entries + 3) / 4 * 2;
    if (threadIdx.x < kWave) {
        CR_STAMP(1);
        int k, len;
        uint32_t fl;
        seed_walk<RA, 0, RB>(pd, dirs, sm, seed_list, nA, k, len, fl);
        if (threadIdx.x == 0) {
            s_walk[0] = k;
            s_walk[1] = len;
            s_walk[2] = (int)fl;
        }
        CR_STAMP(2);
    }
    __syncthreads();
    {
        const int k = s_walk[0];
        Transform tr;
#pragma unroll
        for (int x = 0; x < 3; x++) tr.c1[x] = tr.c2[x] = 0.0;
#pragma unroll
        for (int x = 0; x < 9; x++) tr.R[x] = (x % 4 == 0) ? 1.0 : 0.0;
        tr.flags = (uint32_t)s_walk[2];
        tr.seed_len = s_walk[1];
        if (k <= 3) {
            tr.flags |= kFlagSeedSkipped;
        } else {
            double t[3];
            const int cap = pd.n < pd.m ? pd.n : pd.m;
            kabsch_team(coords + pd.off_i * 3, coords + pd.off_j * 3, seed_list + (cap - k), k, k, seed_terms, seed_terms + kSumTile * kMaxAcc + kSumSlack,
                        tr.c1, tr.c2, tr.R, t);
        }
        if (threadIdx.x == 0) {
            xf[blockIdx.x] = tr;
            seed_score[blockIdx.x] = sm.score;
            s_tr = tr;
        }
        CR_STAMP(3);
    }
    __syncthreads();
    CR_STAMP(4);

    // ---- alignment fill ------------------------------------------------------------------------------------------------
    AlignEnd e;
    {
        constexpr int MODE = kSwScore | kDtw | kZeroGap;
        double* cols = f.body;
        double* edges = cols + (size_t)3 * pd.m;
        double* rings = edges + (size_t)NW * 3 * kDuoEdge;
        double* myring = rings + (size_t)w * ring * (RA * kWave);
        load_exp_table(lds, threadIdx.x);                  // (the walk's entries and the term tile lay over it)
        if (threadIdx.x < 3 * kFedMaxStrips) f.done[threadIdx.x] = 0;
        {
            RbfCoords<1> all;                              // the columns in the seed's frame, resident
            all.rows_g = coords + pd.off_i * 3;
            all.cols_g = coords + pd.off_j * 3;
            all.xf = &s_tr;
            all.neg_gamma = -gamma_coords;
            all.load_resident(cols, pd.m, pd.m, (int)threadIdx.x, (int)blockDim.x);
        }
        __syncthreads();
        SeedMax unused;
        const SweepParams prm{0.0, gap_open, gap_extend};
        if (stamped) CR_DUO_STAMP(sg, 4, CR_DUO_NOW());
        auto fill = [&](auto rtag) {
            constexpr int R = decltype(rtag)::value;
            const int rowbase = geom.rowbase0 + lane * R;
            const int rows_here = pd.n - geom.rowbase0;
            const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
            const int T = mine ? pd.m + lanes_here - 1 : 0;
            DpState<R> st;
            st.sw_max = 0.0;
            st.reset_column0(kMinF64 - gap_open);
#pragma unroll
            for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;
            if (producer) {
                __builtin_amdgcn_s_setprio(CR_TRIO_PRIO_PROD);
                RbfCoords<R> src;
                src.rows_g = coords + pd.off_i * 3;
                src.cols_g = coords + pd.off_j * 3;
                src.xf = &s_tr;
                src.neg_gamma = -gamma_coords;
                if (mine) fed_align_producer<R, RA>(w, src, rowbase, pd.n, pd.m, T, tab, cols, myring, ring, f, waited);
                __builtin_amdgcn_s_setprio(0);
                if (stamped) {
                    CR_DUO_STAMP(sg, 5, CR_DUO_NOW());
                    CR_DUO_STAMP(sg, 6, waited);
                }
                // (the barriers of wide_finish)
                __threadfence();
                __syncthreads();
                __syncthreads();
            } else {
                duo_priority(w);
                if (mine) fed_align_consumer<R, RA, MODE>(w, rowbase, pd.n, pd.m, T, prm, myring, ring, f, edges, bits + pd.bt_off, geom, st, waited);
                __builtin_amdgcn_s_setprio(0);
                if (stamped) {
                    CR_DUO_STAMP(sg, 5, CR_DUO_NOW());
                    CR_DUO_STAMP(sg, 6, waited);
                }
                wide_finish<R, MODE>(st, mine, w, lane, rowbase, geom, f.red, unused, e);
            }
        };
        if (plan.wave_in_a(w)) fill(std::integral_constant<int, RA>{});
        else if constexpr (RA != RB) fill(std::integral_constant<int, RB>{});
    }
    CR_STAMP(5);
    PairResult r;
    r.sw = e.sw;
    r.dtw_score = e.dtw_score;
#pragma unroll
    for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
    for (int x = 0; x < 3; x++) r.t[x] = 0.0;
    r.rmsd = r.coverage = r.tm = 0.0;
    r.aln_len = r.aln_start = 0;
    r.flags = 0;
    {
        uint32_t* const arow = reinterpret_cast<uint32_t*>(lds + kExpDoubles);
        double* const terms = lds + kExpDoubles + ((size_t)align_entries + 3) / 4 * 2;
        const int cap = pd.n + pd.m;
        if (threadIdx.x < kWave) {                     // wave 0 walks, the others wait at the barrier
            int idx, k;
            dtw_walk<RA, RB>(pd.n, pd.m, align_entries, bits + pd.bt_off, e.start_layer, lds + kExpDoubles, aln + pd.aln_off, idx, k, nA);
            stream_rows(hout, arow + (cap - idx), idx, (int)threadIdx.x);
            if (threadIdx.x == 0) {
                s_walk[0] = idx;
                s_walk[1] = k;
            }
            CR_STAMP(6);
        }
        __syncthreads();
        const int idx = s_walk[0], k = s_walk[1], first = cap - idx;
        r.aln_len = idx;
        r.aln_start = first;
        if (k < 3) {
            r.flags |= kFlagMetricsSkipped;
        } else {
            const double* Xi = coords + pd.off_i * 3;
            const double* Xj = coords + pd.off_j * 3;
            double c1[3], c2[3];
            kabsch_team(Xi, Xj, arow + first, idx, k, terms, terms + kSumTile * kMaxAcc + kSumSlack, c1, c2, r.R, r.t);
            rmsd_tm_team<true>(Xi, Xj, arow + first, idx, k, pd.n, pd.m, r.R, r.t, terms, terms + kSumTile * kMaxAcc + kSumSlack, r.rmsd, r.tm);
            r.coverage = (double)k / (double)idx;
        }
    }
    r.seed_score = sm.score;
    r.seed_len = s_tr.seed_len;
    r.flags |= s_tr.flags;
    if (threadIdx.x == 0) {
        res[blockIdx.x] = r;
        if (hout.res) hout.res[hout.dst(blockIdx.x)] = r;
    }
    CR_STAMP(7);
}

}  // namespace cr
