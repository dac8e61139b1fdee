// One workgroup per pair: the team sweep (four waves), the wide sweep (up to sixteen) and the sweep on staged scores.
// Part of cr_kernels.h (included there, inside namespace cr, in this order: cr_providers.h, cr_sweep.h, cr_sweep_cols.h,
// cr_sweep_wide.h, cr_trace.h, cr_pair_kernels.h); not a header of its own.

// ---------------------------------------------------------------------------------------------
// The team sweep: one WORKGROUP per pair, one wave per strip, all strips in flight at once.
// For launches with too few pairs to fill the chip (a level of the guide tree, a small pair list) the
// single-wave sweep is latency bound: one wave issues one instruction every few cycles and walks the
// strips one after the other.  Here strip s runs on wave s, kTeamDelay = 64 steps behind strip s-1 (the
// smallest lag: lane 63 of strip s-1 finishes column c one step before lane 0 of strip s needs it), and takes
// the row above it from an LDS ring that strip s-1's last lane fills; the waves meet at a barrier every step.
// (A 128-step lag needs no barrier beyond those of the column-chunk loads, but the longer pipeline costs more
// than the barriers: 15.8 vs 13.8 ms for the 17 levels of the 128 x 300 guide tree.)  Decision words use the
// same (strip, time block, row, lane) layout as the single-wave sweep, so the traceback code is shared.
// Results are returned in every lane of every wave.
// LDS (doubles): exp table | NW column rings | NW edge rings of NB * kEdgeRing | NW * 8 reduction slots.
// ---------------------------------------------------------------------------------------------
constexpr int kTeamDelay = kWave;
constexpr int kEdgeRing = 4 * kWave;
constexpr int kTeamWaves = 4;

template <int R, int MODE, class Src>
CR_D void sweep_team(Src& src, const int n, const int m, const SweepParams prm, double* lds,
                     uint32_t* __restrict__ sw_dirs, uint32_t* __restrict__ dtw_bits, SeedMax& seed_out,
                     AlignEnd& end_out) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* ring = lds + kExpDoubles + w * Src::kRingDoubles;
    double* edges = lds + kExpDoubles + NW * Src::kRingDoubles;
    double* edge_out = edges + w * (NB * kEdgeRing);
    const double* edge_in = edges + (w > 0 ? w - 1 : 0) * (NB * kEdgeRing);
    double* red = edges + NW * (NB * kEdgeRing);

    load_exp_table(lds, lane);
    src.init_ring(ring, lane);
    __syncthreads();

    const int nstrips = strips_of(n, R);                 // <= NW, guaranteed by the launcher
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;
    const bool mine = w < nstrips;
    const int rowbase = (w * kWave + lane) * R;
    const int rows_here = n - w * kWave * R;
    const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
    const int T = mine ? m + lanes_here - 1 : 0;

    DpState<R> st;
    st.sw_max = 0.0;
    if (mine) src.load_rows(rowbase, n);
    st.reset_column0(col0_m2);
#pragma unroll
    for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;

    const int G = kTeamDelay * (nstrips - 1) + m + kWave - 1;
    for (int g = 0; g < G; g++) {
        const int t = g - kTeamDelay * w;
        const bool live = mine && t >= 0 && t < T;
        const bool boundary = (g & (kWave - 1)) == 0;
        __syncthreads();                                   // edge values of step g-1 visible to the next strip
        if (boundary) {
            if (live) src.load_chunk(ring, t >> 6, m, lane);
            __syncthreads();
        }
        if (!live) continue;
        const int c = t - lane;
        const bool active = (unsigned)c < (unsigned)m;

        double h_top0 = 0.0, m0_top0 = col0_m2, m1_top0 = 0.0;
        if (w > 0 && lane == 0 && active) {
            if constexpr (SW) h_top0 = edge_in[c & (kEdgeRing - 1)];
            if constexpr (DTW) {
                m0_top0 = edge_in[(NB - 2) * kEdgeRing + (c & (kEdgeRing - 1))];
                m1_top0 = edge_in[(NB - 1) * kEdgeRing + (c & (kEdgeRing - 1))];
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
            if constexpr (Src::kRingDoubles == 0) src.set_col(c, m);
            src.fetch_col(ring, c & (kRing - 1));
            dp_column<R, MODE>(src, st, prm, tab, c, rowbase, n, sh2, sh4, h_top, m0_top, m1_top);
            if (w + 1 < nstrips && lane == kWave - 1) {
                if constexpr (SW) edge_out[c & (kEdgeRing - 1)] = st.h_left[R - 1];
                if constexpr (DTW) {
                    edge_out[(NB - 2) * kEdgeRing + (c & (kEdgeRing - 1))] = st.m0_left[R - 1];
                    edge_out[(NB - 1) * kEdgeRing + (c & (kEdgeRing - 1))] = st.m1_left[R - 1];
                }
            }
        }
        if constexpr (TRACE) {
            if ((t & 15) == 15 || t == T - 1) {
                const int64_t base = ((int64_t)(w * TB_SW + (t >> 4)) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    sw_dirs[base + q * kWave] = st.swbits[q];
                    st.swbits[q] = 0;
                }
            }
        }
        if constexpr (DTW) {
            if ((t & 7) == 7 || t == T - 1) {
                const int64_t base = ((int64_t)(w * TB_DTW + (t >> 3)) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    dtw_bits[base + q * kWave] = st.dtbits[q];
                    st.dtbits[q] = 0;
                }
            }
        }
    }

    // ---- per-wave results, then across the waves through LDS ----------------------------------------
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    if constexpr (TRACE) {
        if (mine) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const bool gt = st.rowmax[q] > best_v;
                best_v = gt ? st.rowmax[q] : best_v;
                best_i = gt ? rowbase + q : best_i;
                best_j = gt ? st.rowarg[q] : best_j;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(best_v, off);
            int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
            bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
    }
    double sw_max = mine ? st.sw_max : 0.0;
    if constexpr ((MODE & kSwScore) != 0) {
        for (int off = 32; off > 0; off >>= 1) sw_max = __builtin_fmax(sw_max, __shfl_xor(sw_max, off));
    }
    const int owner_wave = (n - 1) / (kWave * R);
    if (lane == 0) {
        red[w * 8 + 0] = best_v;
        red[w * 8 + 1] = (double)best_i;
        red[w * 8 + 2] = (double)best_j;
        red[w * 8 + 3] = sw_max;
    }
    if constexpr (DTW) {
        if (w == owner_wave) {
            const int owner = ((n - 1) / R) % kWave;
            const int qo = (n - 1) % R;
            double fin0 = 0.0, fin1 = 0.0, fin2 = 0.0;
#pragma unroll
            for (int q = 0; q < R; q++) {
                fin0 = (q == qo) ? st.m0_left[q] : fin0;
                fin1 = (q == qo) ? st.m1_left[q] : fin1;
                fin2 = (q == qo) ? st.m2_left[q] : fin2;
            }
            if (lane == owner) {
                red[w * 8 + 4] = fin0;
                red[w * 8 + 5] = fin1;
                red[w * 8 + 6] = fin2;
            }
        }
    }
    __threadfence();                                   // decision words of every wave visible to wave 0's walk
    __syncthreads();
    if constexpr (TRACE) {
        best_v = 0.0;
        best_i = best_j = 0x7fffffff;
        for (int x = 0; x < nstrips; x++) {
            const double ov = red[x * 8 + 0];
            const int oi = (int)red[x * 8 + 1], oj = (int)red[x * 8 + 2];
            const bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
        seed_out.score = best_v;
        seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
        seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    }
    if constexpr ((MODE & kSwScore) != 0 || DTW) {
        double smax = 0.0;
        for (int x = 0; x < nstrips; x++) smax = __builtin_fmax(smax, red[x * 8 + 3]);
        const double fin0 = red[owner_wave * 8 + 4], fin1 = red[owner_wave * 8 + 5], fin2 = red[owner_wave * 8 + 6];
        end_out.sw = smax;
        int idx = 0;
        double best = fin0;
        if (fin1 > best) { best = fin1; idx = 1; }
        if (fin2 > best) { best = fin2; idx = 2; }
        end_out.dtw_score = DTW ? best : 0.0;
        end_out.start_layer = idx;
        end_out.pad = 0;
    }
    __syncthreads();
}

template <int R, int MODE, class Src>
__host__ __device__ inline size_t sweep_team_lds_doubles(int waves) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    return kExpDoubles + (size_t)waves * (Src::kRingDoubles + NB * kEdgeRing + 8);
}

// ---------------------------------------------------------------------------------------------
// The wide sweep: one WORKGROUP of up to kWideMaxWaves waves per pair, one wave per strip, for pair lists that
// cannot fill the chip with one or four waves per pair (one GPU's share of a sharded long-chain family: 252 pairs
// of 1200 x 1200 on 256 CUs).  Differences from sweep_team:
//   * all m columns of the pair are RESIDENT in LDS (feature-major planes, loaded once by the whole workgroup with
//     coalesced reads): no per-wave column rings, no chunk loads, no chunk barriers, and the LDS cost does not
//     grow with the number of waves;
//   * strip s runs lag = 63 + B steps behind strip s-1 and the waves meet at a barrier every B steps only
//     (B = sync_every): a value written by strip s-1's last lane in global step g is read by strip s in step g + B,
//     and every window of B consecutive steps holds one barrier.  B = 1 is sweep_team's lock step; larger B lets the
//     waves of one SIMD drift and fill each other's issue gaps at the price of a (S - 1) * (B - 1) steps longer
//     pipeline.  Edge rings of kWideEdge entries per value: the writer is at most 2B - 1 columns ahead (B <= 32).
// Decision words use the same (strip, time block, row, lane) layout as the other sweeps (shared traceback).
// LDS (doubles): exp table | Src::kColDoubles planes of `stride` | NW edge rings of NB * kWideEdge | NW * 8.
// ---------------------------------------------------------------------------------------------
constexpr int kWideEdge = 64;
constexpr int kWideMaxWaves = 16;
constexpr int kWideMaxSync = 32;

// The end of a one-wave-per-strip sweep: per-wave results, then across the waves through LDS (as sweep_team).
// `red`: 8 doubles per wave.
template <int R, int MODE>
CR_D void wide_finish(const DpState<R>& st, const bool mine, const int w, const int lane, const int rowbase, const StripGeom geom,
                      double* red, SeedMax& seed_out, AlignEnd& end_out) {
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    const int nstrips = geom.nstrips;
    // ---- per-wave results, then across the waves through LDS (as sweep_team) -------------------------
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    if constexpr (TRACE) {
        if (mine) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const bool gt = st.rowmax[q] > best_v;
                best_v = gt ? st.rowmax[q] : best_v;
                best_i = gt ? rowbase + q : best_i;
                best_j = gt ? st.rowarg[q] : best_j;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(best_v, off);
            int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
            bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
    }
    double sw_max = mine ? st.sw_max : 0.0;
    if constexpr ((MODE & kSwScore) != 0) {
        for (int off = 32; off > 0; off >>= 1) sw_max = __builtin_fmax(sw_max, __shfl_xor(sw_max, off));
    }
    const int owner_wave = geom.owner_wave;
    if (lane == 0) {
        red[w * 8 + 0] = best_v;
        red[w * 8 + 1] = (double)best_i;
        red[w * 8 + 2] = (double)best_j;
        red[w * 8 + 3] = sw_max;
    }
    if constexpr (DTW) {
        if (w == owner_wave) {
            const int owner = geom.owner_lane;
            const int qo = geom.owner_q;
            double fin0 = 0.0, fin1 = 0.0, fin2 = 0.0;
#pragma unroll
            for (int q = 0; q < R; q++) {
                fin0 = (q == qo) ? st.m0_left[q] : fin0;
                fin1 = (q == qo) ? st.m1_left[q] : fin1;
                fin2 = (q == qo) ? st.m2_left[q] : fin2;
            }
            if (lane == owner) {
                red[w * 8 + 4] = fin0;
                red[w * 8 + 5] = fin1;
                red[w * 8 + 6] = fin2;
            }
        }
    }
    __threadfence();                                   // decision words of every wave visible to wave 0's walk
    __syncthreads();
    if constexpr (TRACE) {
        best_v = 0.0;
        best_i = best_j = 0x7fffffff;
        for (int x = 0; x < nstrips; x++) {
            const double ov = red[x * 8 + 0];
            const int oi = (int)red[x * 8 + 1], oj = (int)red[x * 8 + 2];
            const bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
        seed_out.score = best_v;
        seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
        seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    }
    if constexpr ((MODE & kSwScore) != 0 || DTW) {
        double smax = 0.0;
        for (int x = 0; x < nstrips; x++) smax = __builtin_fmax(smax, red[x * 8 + 3]);
        const double fin0 = red[owner_wave * 8 + 4], fin1 = red[owner_wave * 8 + 5], fin2 = red[owner_wave * 8 + 6];
        end_out.sw = smax;
        int idx = 0;
        double best = fin0;
        if (fin1 > best) { best = fin1; idx = 1; }
        if (fin2 > best) { best = fin2; idx = 2; }
        end_out.dtw_score = DTW ? best : 0.0;
        end_out.start_layer = idx;
        end_out.pad = 0;
    }
    __syncthreads();
}

template <int R, int MODE, class Src>
CR_D void sweep_wide(Src& src, const int n, const int m, const SweepParams prm, double* lds, const int sync_every,
                     uint32_t* __restrict__ sw_dirs, uint32_t* __restrict__ dtw_bits, SeedMax& seed_out,
                     AlignEnd& end_out, const StripGeom geom) {
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
    double* edge_out = edges + w * (NB * kWideEdge);
    const double* edge_in = edges + (w > 0 ? w - 1 : 0) * (NB * kWideEdge);
    double* red = edges + NW * (NB * kWideEdge);

    load_exp_table(lds, threadIdx.x);
    src.load_resident(res, stride, m, (int)threadIdx.x, (int)blockDim.x);

    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;
    const bool mine = w < nstrips;
    const int rowbase = geom.rowbase0 + lane * R;
    const int rows_here = n - geom.rowbase0;
    const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
    const int T = mine ? m + lanes_here - 1 : 0;
    const int lag = kWave - 1 + sync_every;

    DpState<R> st;
    st.sw_max = 0.0;
    if (mine) src.load_rows(rowbase, n);
    st.reset_column0(col0_m2);
#pragma unroll
    for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;

    // Few rows per lane: a step is one long chain of dependent FP64 instructions (squared distance -> exp -> recurrences,
    // ~10 cycles each for a wave that has its SIMD to itself) with nothing to interleave.  The scores do not depend on the
    // recurrence, so they are formed ONE COLUMN AHEAD: the chain of column c + 1's scores runs beside the recurrence of
    // column c, and the step becomes issue-bound.  (Every lane's first column is column 0: its scores are formed here.)
    constexpr bool AHEAD = R <= 2;
    double sc_cur[R];
    __syncthreads();                                       // the resident columns and the exp table are complete (every
                                                           // wave, whatever its rows per lane: barriers must pair up)
    if constexpr (AHEAD) {
        src.fetch_resident(res, stride, 0);
#pragma unroll
        for (int q = 0; q < R; q++) sc_cur[q] = src.score(q, tab);
    }
    const int G = lag * (nstrips - 1) + m + kWave - 1;
    int until_sync = 0;
    for (int g = 0; g < G; g++) {
        if (until_sync == 0) {
            lds_barrier();                                 // edge values of the last B steps visible to the next strip
            until_sync = sync_every;
        }
        until_sync--;
        const int t = g - lag * w;
        const bool live = mine && t >= 0 && t < T;
        if (!live) continue;
        const int c = t - lane;
        const bool active = (unsigned)c < (unsigned)m;

        double h_top0 = 0.0, m0_top0 = col0_m2, m1_top0 = 0.0;
        if (w > 0 && lane == 0 && active) {
            if constexpr (SW) h_top0 = edge_in[c & (kWideEdge - 1)];
            if constexpr (DTW) {
                m0_top0 = edge_in[(NB - 2) * kWideEdge + (c & (kWideEdge - 1))];
                m1_top0 = edge_in[(NB - 1) * kWideEdge + (c & (kWideEdge - 1))];
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
            if (w + 1 < nstrips && lane == kWave - 1) {
                if constexpr (SW) edge_out[c & (kWideEdge - 1)] = st.h_left[R - 1];
                if constexpr (DTW) {
                    edge_out[(NB - 2) * kWideEdge + (c & (kWideEdge - 1))] = st.m0_left[R - 1];
                    edge_out[(NB - 1) * kWideEdge + (c & (kWideEdge - 1))] = st.m1_left[R - 1];
                }
            }
        }
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
            if ((t & 7) == 7 || t == T - 1) {
                const int64_t base = ((int64_t)geom.slot0 * TB_DTW + (int64_t)(t >> 3) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    dtw_bits[base + q * kWave] = st.dtbits[q];
                    st.dtbits[q] = 0;
                }
            }
        }
    }

    wide_finish<R, MODE>(st, mine, w, lane, rowbase, geom, red, seed_out, end_out);
}

template <int MODE, class Src>
__host__ __device__ inline size_t sweep_wide_lds_doubles(int waves, int m_max) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    return kExpDoubles + (size_t)Src::kColDoubles * m_max + (size_t)waves * (NB * kWideEdge + 8);
}

// ---------------------------------------------------------------------------------------------
// The wide sweep on scores that another launch has already formed (cr_staged.h): ONE row per lane up to 320 rows (five
// strips), then two, three, four (up to 2048 rows; blocks of 8 steps from three rows on).
//
// When a launch has few workgroups -- a level of the progressive alignment, a short pair list -- the fused kernels are
// bound by the instruction issue of the few waves that hold the recurrence, and 50 of a seed step's 59 instructions (30
// of an alignment step's 49) are the score, which does not depend on the recurrence at all.  A staging launch forms the
// scores on every CU of the chip in the SAME arithmetic (the provider's own score()), and this sweep is left with the
// recurrence.
// Layout of one strip (64 rows): element t * 64 + lane = S(row lane, column t - lane), t = 0 .. m + 62: the line a wave
// needs at step t is one coalesced 512-byte read.  The loop runs in blocks of kStagedBlock = 16 steps, unrolled: the
// block's 16 lines sit in registers, requested TWO blocks ahead (the scores are in L2 / MALL, 200 .. 900 cycles away;
// 48 lines in flight per wave); shifts and word boundaries of a block are fixed at compile time.  A strip follows the
// one above by at least five blocks (63 + 16 steps), paced by progress words (below).  The strip region has
// staged_steps(m_max) lines: the requests two blocks past the last step stay inside it.
// LDS (doubles): NW + 1 hand-off rings of NB * kStagedRing | NW * 8 | NW dumps | progress words.  Decision words: as every other skewed sweep.
// ---------------------------------------------------------------------------------------------
constexpr int kStagedBlock = 16;         // steps per block with one or two rows per lane; 8 with three or four (registers)
constexpr int kStagedMaxWaves = 8;       // the blocks of score lines in registers need more than the 128 VGPRs of a 16-wave
                                         // workgroup: 512 rows per row of a lane
constexpr int kStagedMaxR = 4;
constexpr int kStagedMaxRows = kStagedMaxWaves * kWave * kStagedMaxR;
CR_HD int staged_steps(int m_max) { return (m_max + kWave - 1 + kStagedBlock - 1) / kStagedBlock * kStagedBlock + 2 * kStagedBlock; }

template <int R, bool RBF = true>
struct StagedScore {                               // what dp_column sees: the scores of the lane's cells of this step
    static constexpr bool kNonNegative = RBF;      // RBF scores (the staging kernels write what the RBF providers return);
                                                   // explicit score matrices (cr_dropins.h) may hold anything
    static constexpr bool kMaskRows = !RBF;        // RBF: rows past n were staged as the exact zeros the RBF gives them
    double v[R];
    CR_D double score(int q, const ExpEntry*) const { return v[q]; }
};

// doubles per wave that take the hand-off writes of lanes 0 .. 62: dump[lane + plane * 64 + step], up to three planes
constexpr int kStagedRing = 128;          // steps a plane of a hand-off ring of the staged sweeps holds
constexpr int kStagedDump = kWave + 2 * kStagedRing + kStagedBlock;

// progress words of the staged sweeps (as cr_duo.h paces its strips): blocks a strip has completed
CR_D void staged_publish(int* word, int blocks_done) {
    asm volatile("" ::: "memory");       // (compiler: the hand-off values of the block are written first)
    *reinterpret_cast<volatile int*>(word) = blocks_done;
}
CR_D void staged_wait(const int* word, int need) {
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(word)) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");       // (compiler: hand-off values are read behind the word)
}

template <int R, int MODE, bool RBF = true>
CR_D void sweep_staged(const double* __restrict__ strip, const int n, const int m, const SweepParams prm, double* lds,
                       uint32_t* __restrict__ sw_dirs, uint32_t* __restrict__ dtw_bits, SeedMax& seed_out,
                       AlignEnd& end_out, const StripGeom geom) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);
    constexpr int B = R <= 2 ? kStagedBlock : 8;          // steps per block (the block's R * B score lines sit in registers)
    constexpr int LAGB = R <= 2 ? 5 : 9;                  // blocks a strip lags the one above: 80 / 72 steps (>= 63 + B)
    constexpr bool FAR = R == 1;                          // score lines two blocks ahead (R >= 2: one, the registers are taken)
    constexpr int RING = kStagedRing;                     // slots of a plane of a hand-off ring
    constexpr int PH = 0, PM0 = (NB - 2) * RING, PM1 = (NB - 1) * RING;             // planes of a ring
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    // Hand-off rings, indexed by the WRITER's step (t & 63: a block's 16 slots are contiguous): ring 0 holds the DP border
    // above row 0 (constants), ring w + 1 the last row of strip w.  Strip w reads ring w: no special case for the first.
    const double* ring_in = lds + w * (NB * RING);
    double* ring_out = lds + (w + 1) * (NB * RING);
    double* red = lds + (NW + 1) * (NB * RING);
    double* dump = red + NW * 8 + w * kStagedDump;
    int* prog = reinterpret_cast<int*>(red + NW * 8 + NW * kStagedDump);     // prog[w]: blocks strip w has completed

    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;
    const bool mine = w < nstrips;
    const int rowbase = geom.rowbase0 + lane * R;
    const int rows_here = n - geom.rowbase0;
    const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
    const int my_blocks = mine ? (m + lanes_here - 1 + B - 1) / B : 0;
    // every ring starts as the DP border: ring 0 IS the border; in the others a lane 0 that is past its last column (the
    // ramps run unmasked) may read a slot its writer never reaches -- e.g. the one step of the writer's masked last block
    // in which its lane 63 is past the last column -- and must not find whatever the LDS held
    for (int x = threadIdx.x; x < (NW + 1) * NB * RING; x += blockDim.x) lds[x] = (DTW && (x / RING) % NB == NB - 2) ? col0_m2 : 0.0;
    if (threadIdx.x < kStagedMaxWaves) prog[threadIdx.x] = 0;

    DpState<R> st;
    st.sw_max = 0.0;
    st.reset_column0(col0_m2);
#pragma unroll
    for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;
    StagedScore<R, RBF> src;
    const double* __restrict__ line = strip + lane;      // line t: R sub-lines of 64 doubles (row slot q, lane)
    double cur[B][R], nxt[B][R], nx2[FAR ? B : 1][R];
    if (mine) {
#pragma unroll
        for (int k = 0; k < B; k++)
#pragma unroll
            for (int q = 0; q < R; q++) {
                nxt[k][q] = line[(k * R + q) * kWave];
                if constexpr (FAR) nx2[k][q] = line[((B + k) * R + q) * kWave];
            }
    }
    const int blocks_above = (m + kWave - 1 + B - 1) / B;    // blocks of the strip above (it has all 64 lanes)
    const bool hand_out = w + 1 < nstrips;
    // RAMPS WITHOUT MASKS.  A lane's column t - lane is outside [0, m) in the first 63 and the last 63 steps of its strip; a
    // block in which that happens for any lane runs the EXEC-masked step (269 against 206 cycles for the DTW, 250 against
    // 156 for the SW, tools/step_probe.hip) -- and with the strips 80 steps apart nearly every block of the WORKGROUP has
    // some strip in a ramp, so the whole fill ran at the masked step's pace.  The masks are not needed where the staging
    // kernels have written exact zeros for the columns outside [0, m) (cr_staged.h, stage_block) and the penalties are not
    // negative:
    //  * before its column 0 a lane then sits at a fixed point that its first real step cannot tell from the DP border:
    //    SW: h = max(0 + 0, 0, 0) = 0, decision code 0, no row maximum.  DTW: c1 = 0 + 0, the layer above gives
    //    m0 = max(m0' - extend, 0 - open) <= 0, m2 = max(0 - open, m2 - extend) = -open from the first such step on, so
    //    m1 = max(max(m0, 0), -open) = +0.0 = M[i][0][1]; the first real step reads m2 - extend, which is below 0 - open
    //    for -open as for the border's MIN - open (same maximum, same decision bit);
    //  * behind column m - 1 a lane's state is dead: its decision bits lie at positions no walk reads, the values it hands
    //    down belong to columns the strip below does not have; SW with gap 0 repeats the row's last value (no new row
    //    maximum), a global SW maximum only ever sees values of real cells again.  Two things do outlive the last column:
    //    the DTW layers of row n - 1 (the score) and, for an SW with a gap, each row's first maximum -- hence the LAST
    //    block of every strip stays masked (the owner of row n - 1 is the last lane of its strip to finish), and the SW
    //    trace with a gap keeps its masks altogether.
    // Explicit score matrices (RBF = false: scores of any sign, so a row of an SW is not monotone): the DTW and the SW
    // score run unmasked as well (neither argument above needs the sign of a REAL score), the SW trace keeps its masks.
    const bool unmasked = !kProbeMaskedRamps && !(TRACE && (!RBF || !(MODE & kZeroGap))) && prm.sw_gap >= 0.0 && prm.gap_open >= 0.0 &&
                          prm.gap_extend >= 0.0;
    // PACING.  The strips form a chain -- strip w needs, for its block tb, the last row of the strip above up to that
    // strip's step 16 tb + 15 + 63, i.e. its blocks up to tb + LAGB - 1 -- and used to advance together behind one
    // s_barrier per block: every block took what the slowest strip's block took.  Now every strip publishes the number of
    // blocks it has completed (an LDS word, written behind the block's hand-off values: the LDS serves a wave's requests in
    // order) and waits only for the strip above; a writer also waits until the strip below is past the values a block will
    // overwrite (the rings hold RING = 128 steps: five blocks of slack on top of the five of lag).
    lds_barrier();                                         // border ring and progress words
#pragma unroll 1
    for (int tb = 0; tb < my_blocks; tb++) {
        if (w > 0) staged_wait(prog + w - 1, tb + LAGB < blocks_above ? tb + LAGB : blocks_above);
        // (this block overwrites the values of block tb - RING / B, whose last one the strip below reads in its step
        // B (tb - RING / B) + B - 1 - 63)
        if (hand_out && tb >= RING / B + LAGB - 2) staged_wait(prog + w + 1, tb - (RING / B + LAGB - 3));
        const double* __restrict__ ahead = line + (int64_t)(tb + (FAR ? 2 : 1)) * (B * R * kWave);
#pragma unroll
        for (int k = 0; k < B; k++)
#pragma unroll
            for (int q = 0; q < R; q++) {
                cur[k][q] = nxt[k][q];
                if constexpr (FAR) {
                    nxt[k][q] = nx2[k][q];
                    nx2[k][q] = ahead[(k * R + q) * kWave];
                } else {
                    nxt[k][q] = ahead[(k * R + q) * kWave];
                }
            }
        // The row above the strip.  Lane 0's column at step t is t itself, written by the strip above at ITS step t + 63:
        // slot (t - 1) & 63.  Every lane reads it (one address: a broadcast) and hands it to the shift as lane 0's fill;
        // the read of step k + 1 is issued before the arithmetic of step k.
        // (the slots of a ring are the WRITER's steps mod RING; lane 0's column at step t is t, written above at step t + 63)
        const int q4 = (tb * B) & (RING - 1);              // this strip writes the slots q4 + k
        const int q4r = (tb * B + kWave) & (RING - 1);     // ... and reads slot0, then q4r + k - 1 for its step k >= 1
        const double* fills = ring_in + q4r - 1;           // step k >= 1: fills[k]
        const int slot0 = (tb * B + kWave - 1) & (RING - 1);
        // the strip's last row: lane 63 writes its values of step k to slot q4 + k of the ring, the other lanes write theirs
        // to a dump (one LDS instruction per step with no EXEC juggling)
        double* wr = (lane == kWave - 1 && w + 1 < nstrips) ? ring_out + q4 : dump + lane;
        double f_h = 0.0, f_m0 = 0.0, f_m1 = 0.0;
        if constexpr (SW) f_h = ring_in[PH + slot0];
        if constexpr (DTW) {
            f_m0 = ring_in[PM0 + slot0];
            f_m1 = ring_in[PM1 + slot0];
        }
        auto steps = [&](auto all_tag) {
            constexpr bool ALL = decltype(all_tag)::value;    // every lane's column of every step of the block is inside [0, m)
            static_for<0, B>([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                const int c = tb * B + k - lane;
                const bool active = ALL || (unsigned)c < (unsigned)m;
                double g_h = 0.0, g_m0 = 0.0, g_m1 = 0.0;
                if constexpr (k + 1 < B) {
                    if constexpr (SW) g_h = fills[PH + k + 1];
                    if constexpr (DTW) {
                        g_m0 = fills[PM0 + k + 1];
                        g_m1 = fills[PM1 + k + 1];
                    }
                }
                double h_top = 0.0, m0_top = 0.0, m1_top = 0.0;
                if constexpr (SW) h_top = wave_shr1(st.h_left[R - 1], f_h);
                if constexpr (DTW) {
                    m0_top = wave_shr1(st.m0_left[R - 1], f_m0);
                    m1_top = wave_shr1(st.m1_left[R - 1], f_m1);
                }
                if (active) {
#pragma unroll
                    for (int q = 0; q < R; q++) src.v[q] = cur[k][q];
                    dp_column<R, MODE>(src, st, prm, nullptr, c, rowbase, n, ((tb * B + k) & 15) * 2, (k & 7) * 4, h_top, m0_top, m1_top);
                    if constexpr (!kProbeNoDump) {
                        if constexpr (SW) wr[PH + k] = st.h_left[R - 1];
                        if constexpr (DTW) {
                            wr[PM0 + k] = st.m0_left[R - 1];
                            wr[PM1 + k] = st.m1_left[R - 1];
                        }
                    }
                }
                f_h = g_h;
                f_m0 = g_m0;
                f_m1 = g_m1;
                if constexpr (DTW) {
                    if ((k & 7) == 7 && tb * (B / 8) + (k >> 3) < TB_DTW) {
                        const int64_t base = ((int64_t)geom.slot0 * TB_DTW + (int64_t)(tb * (B / 8) + (k >> 3)) * R) * kWave + lane;
#pragma unroll
                        for (int q = 0; q < R; q++) {
                            dtw_bits[base + q * kWave] = st.dtbits[q];
                            st.dtbits[q] = 0;
                        }
                    }
                }
            });
        };
        if (unmasked ? tb != my_blocks - 1 : (tb * B >= kWave - 1 && tb * B + B - 1 < m)) steps(std::true_type{});
        else steps(std::false_type{});
        if (nstrips > 1 && lane == 0) staged_publish(prog + w, tb + 1);    // (the strip below follows it, the strip above must not lap it)
        if constexpr (TRACE) {
            if ((((tb + 1) * B) & 15) == 0 || tb == my_blocks - 1) {      // a decision word holds 16 steps
                const int64_t base = ((int64_t)geom.slot0 * TB_SW + (int64_t)((tb * B) >> 4) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    sw_dirs[base + q * kWave] = st.swbits[q];
                    st.swbits[q] = 0;
                }
            }
        }
    }
    wide_finish<R, MODE>(st, mine, w, lane, rowbase, geom, red, seed_out, end_out);
}

template <int MODE>
__host__ __device__ inline size_t sweep_staged_lds_doubles(int waves) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    return (size_t)(waves + 1) * (NB * kStagedRing) + (size_t)waves * (8 + kStagedDump) + kStagedMaxWaves / 2;
}
