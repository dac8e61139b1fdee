// The column sweeps: Smith-Waterman with gap 0 without the time skew (one wave per pair, one wave per strip, scores only).
// Part of cr_kernels.h (included there, inside namespace cr, in this order: cr_providers.h, cr_sweep.h, cr_sweep_cols.h,
// cr_sweep_wide.h, cr_trace.h, cr_pair_kernels.h); not a header of its own.

// ---------------------------------------------------------------------------------------------
// The column sweep: Smith-Waterman with gap 0 on non-negative scores (the reference's only use of smith_waterman in
// the pipeline, multiple_alignment.py:332-335) WITHOUT the time skew.
//
// With gap = 0 and S >= 0 the recurrence H = max(0, diag + S, left, up) (dynamic_time_warping.py:234-238) makes H
// non-decreasing along rows and columns, and max is exact and associative, so for one column j
//     H[i][j] = max over i' <= i of B[i'][j],      B[i][j] = max(H[i-1][j-1] + S[i][j], H[i][j-1]),
// i.e. the `up` dependency is a PREFIX MAXIMUM down the column.  All 64 lanes (R rows each) therefore work on the SAME
// column in every step: B from the previous column's values (registers), a sequential scan down the lane's R rows, a
// 6-step DPP max-scan across the lanes (row_shr 1/2/4/8, row_bcast 15/31), one more max per cell.  A strip takes m
// steps instead of m + 63, no lane ever idles in a ramp, and the column's features are wave-uniform: they are read
// with scalar loads into SGPRs (no LDS ring, no per-step ds_reads).  Every value is bit-identical to the
// cell-by-cell evaluation; the decisions (h == diag + S, then h == left, else up; 0 when h == 0) and the row-major
// first maximum are taken from the same values: a row's maximum is its last value and its first position is the column
// of the row's last strict increase (h != left).
// Decision words: ((strip * TB + (j >> 4)) * R + q) * 64 + lane, TB = ceil(m / 16), bits (j & 15) * 2: the layout of
// the skewed sweeps with time step = column (Walker<R, 2, 0>).
// Strips after the first take the row above them (the previous strip's last row, one value per column) from `hand_g`,
// 64 columns per coalesced load.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
CR_D double scan_step(double v) {
    // lanes without a source lane read +0.0 (bound_ctrl), rows outside ROW_MASK keep the +0.0 they are given:
    // max(v, 0) = v for v >= 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    return vmax(v, __hiloint2double(hi, lo));
}

// inclusive prefix maximum over the 64 lanes of non-negative doubles
CR_D double wave_scan_max(double v) {
    v = scan_step<0x111, 0xf>(v);      // row_shr:1
    v = scan_step<0x112, 0xf>(v);      // row_shr:2
    v = scan_step<0x114, 0xf>(v);      // row_shr:4
    v = scan_step<0x118, 0xf>(v);      // row_shr:8
    v = scan_step<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = scan_step<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}

// rows per lane up to which the column sweep keeps two sets of column features (ColSweep::step; 32 structures x 150:
// k_seed 0.114 -> 0.110 ms -- with one wave per SIMD and three rows per lane the step is bound by the latency of its
// dependent FP64 chains, about 6 cycles per instruction, more than by the scalar loads)
template <int R>
constexpr bool kTwoColumnSets = R <= 3;

// Per-lane state of the column sweep and one column step.
template <int R, int D>
struct ColSweep {
    double hprev[R];          // H of this lane's rows, previous column
    double eprev;             // H of the row above them, previous column
    int rowfirst[R];          // column of each row's last strict increase
    uint32_t bits[R];         // decisions of the current word

    CR_D void reset() {
#pragma unroll
        for (int q = 0; q < R; q++) {
            hprev[q] = 0.0;
            rowfirst[q] = 0;
            bits[q] = 0;
        }
        eprev = 0.0;
    }
    // The column's features are wave-uniform: scalar loads into SGPRs (src.col).  A step first forms the R squared
    // distances -- the only readers of the features -- and then requests the NEXT column into the same registers, so the
    // load's latency hides behind the exp / DP / scan arithmetic of this step even with a single wave on the SIMD, and one
    // set of SGPRs suffices.  (Always D loads: the tensor array is allocated with D doubles of slack and the padded
    // features are zeroed by scalar selects -- conditional loads would cost a branch each.)
    template <bool FULL>
    CR_D void prefetch(RbfTensor<R, D>& src, int j) {
        prefetch_into<FULL>(src, j, src.col);
    }
    template <bool FULL>
    CR_D void prefetch_into(RbfTensor<R, D>& src, int j, double (&set)[D]) {
        const double* __restrict__ cg = src.cols_g;
        const int d = FULL ? D : src.d;
#pragma unroll
        for (int k = 0; k < D; k++) {
            const double v = cg[(int64_t)j * d + k];
            set[k] = (FULL || k < d) ? v : 0.0;
        }
    }
    // Column j (prefetch<FULL>(src, j) has been called; `jn` = the column to request now, any valid column).  FULL: the
    // stored tensor width equals D (no padded features).  `top`: H of the row above the strip in this column
    // (wave-uniform; only read when TOP).
    // SET 0: one set of feature registers, as described above.  SET 1 / 2 (few rows per lane: the arithmetic behind the
    // squared distances is too short to cover a scalar load that misses): two sets in turn -- column j is in set SET,
    // column jn is requested into the other one BEFORE anything else, so the load has the whole step to arrive.
    template <bool FULL, bool TOP, int SET = 0>
    CR_D void step(RbfTensor<R, D>& src, const ExpEntry* tab, int j, int jn, double top) {
        double acc[R];
        if constexpr (SET == 1) prefetch_into<FULL>(src, jn, src.col2);
        if constexpr (SET == 2) prefetch_into<FULL>(src, jn, src.col);
#pragma unroll
        for (int q = 0; q < R; q++) acc[q] = SET == 2 ? src.dist2_of(q, src.col2) : src.dist2_of(q, src.col);
        if constexpr (SET == 0) prefetch<FULL>(src, jn);
        double dg[R], p[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double sc = exp_tab<true>(src.neg_gamma * acc[q], tab);
            dg[q] = (q == 0 ? eprev : hprev[q - 1]) + sc;
            const double b = vmax(dg[q], hprev[q]);
            p[q] = q == 0 ? b : vmax(p[q - 1], b);
        }
        double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
        if constexpr (TOP) e = vmax(e, top);
        const int sh2 = (j & 15) * 2;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double h = vmax(p[q], e);
            // decision replayed by the traceback's equality tests (:255-277): diag, then left, else up
            const bool same = h == hprev[q];
            uint32_t code = (h == dg[q]) ? 1u : same ? 2u : 3u;
            code = (h > 0.0) ? code : 0u;
            bits[q] |= code << sh2;
            rowfirst[q] = same ? rowfirst[q] : j;          // column of the row's last strict increase
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
    // fold the rows' maxima (= last values) into a running best, rows ascending
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

CR_D void wave_first_max(double& best_v, int& best_i, int& best_j) {
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(best_v, off);
        int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
        bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
        best_v = take ? ov : best_v;
        best_i = take ? oi : best_i;
        best_j = take ? oj : best_j;
    }
}

// One wave, strips one after the other.
template <int R, int D>
CR_D void sweep_cols(RbfTensor<R, D>& src, const int n, const int m, double* lds, uint32_t* __restrict__ sw_dirs,
                     double* __restrict__ hand_g, SeedMax& seed_out) {
    const int lane = threadIdx.x;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    load_exp_table(lds, lane);
    __syncthreads();

    const int nstrips = strips_of(n, R);
    const int TB = (m + 15) >> 4;
    const bool full = src.d == D;
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    ColSweep<R, D> st;

    for (int s = 0; s < nstrips; s++) {
        const int rowbase = (s * kWave + lane) * R;
        src.load_rows(rowbase, n);
        st.reset();
        const bool hand_out = s + 1 < nstrips;
        auto run = [&](auto full_tag, auto top_tag) {
            constexpr bool FULL = decltype(full_tag)::value, TOP = decltype(top_tag)::value;
            double top_vec = 0.0;                // row above the strip, 64 columns per load (lane x: column j0 + x)
            st.template prefetch<FULL>(src, 0);
            auto column = [&](auto set_tag, int j) {
                constexpr int SET = decltype(set_tag)::value;
                if (TOP && (j & (kWave - 1)) == 0) top_vec = (j + lane < m) ? hand_g[j + lane] : 0.0;
                st.template step<FULL, TOP, SET>(src, tab, j, j + 1 < m ? j + 1 : j, TOP ? lane_value(top_vec, j & (kWave - 1)) : 0.0);
                if (hand_out && lane == kWave - 1) hand_g[j] = st.hprev[R - 1];
                if ((j & 15) == 15 || j == m - 1) st.flush(sw_dirs, ((int64_t)(s * TB + (j >> 4)) * R) * kWave + lane);
            };
            if constexpr (kTwoColumnSets<R>) {
#pragma unroll 1
                for (int j = 0; j < m; j += 2) {
                    column(std::integral_constant<int, 1>{}, j);
                    if (j + 1 < m) column(std::integral_constant<int, 2>{}, j + 1);
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < m; j++) column(std::integral_constant<int, 0>{}, j);
            }
        };
        if (s == 0) {
            if (full) run(std::true_type{}, std::false_type{});
            else run(std::false_type{}, std::false_type{});
        } else {
            if (full) run(std::true_type{}, std::true_type{});
            else run(std::false_type{}, std::true_type{});
        }
        if (hand_out) {                        // the hand-off row: visible to this wave's loads in the next strip
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
        st.fold(rowbase, best_v, best_i, best_j);
    }
    wave_first_max(best_v, best_i, best_j);
    seed_out.score = best_v;
    seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
    seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    __syncthreads();                                   // the caller may reuse the LDS from here on
}

// One WORKGROUP per pair, one wave per strip, all strips in flight: strip s works on columns [c * B, (c + 1) * B) in
// phase c + s (B = kColChunk), i.e. only B columns behind the strip above it -- against 64 + in the skewed team sweeps.
// The row above a strip arrives through an LDS ring written by the previous strip's last lane (one double per column,
// two chunks deep); the waves meet at one barrier per phase.  LDS (doubles): exp table | NW rings of 2 * kColChunk |
// NW * 4 reduction slots.  Results in every lane of every wave.
constexpr int kColChunk = 8;

template <int R, int D>
CR_D void sweep_cols_team(RbfTensor<R, D>& src, const int n, const int m, double* lds,
                          uint32_t* __restrict__ sw_dirs, SeedMax& seed_out, const StripGeom geom) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* rings = lds + kExpDoubles;
    double* ring_out = rings + w * (2 * kColChunk);
    const double* ring_in = rings + (w > 0 ? w - 1 : 0) * (2 * kColChunk);
    double* red = rings + NW * (2 * kColChunk);
    load_exp_table(lds, threadIdx.x);

    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const int TB = (m + 15) >> 4;
    const bool mine = w < nstrips;
    const bool full = src.d == D;
    const int rowbase = geom.rowbase0 + lane * R;
    const bool hand_out = w + 1 < nstrips;
    ColSweep<R, D> st;
    st.reset();
    if (mine) src.load_rows(rowbase, n);
    const int chunks = (m + kColChunk - 1) / kColChunk;
    const int phases = chunks + nstrips - 1;

    auto run = [&](auto full_tag, auto top_tag) {
        constexpr bool FULL = decltype(full_tag)::value, TOP = decltype(top_tag)::value;
#pragma unroll 1
        for (int g = 0; g < phases; g++) {
            __syncthreads();                           // the chunk written in phase g - 1 is visible to the strip below
            const int c = g - w;
            if (!mine || c < 0 || c >= chunks) continue;
            const int j0 = c * kColChunk;
            const int jend = j0 + kColChunk < m ? j0 + kColChunk : m;
            // the row above the strip for this chunk: lane x holds column j0 + x
            double top_vec = 0.0;
            if (TOP && lane < kColChunk) top_vec = ring_in[(c & 1) * kColChunk + lane];
            if (c == 0) st.template prefetch<FULL>(src, 0);
#pragma unroll 1
            for (int j = j0; j < jend; j++) {
                st.template step<FULL, TOP>(src, tab, j, j + 1 < m ? j + 1 : j, TOP ? lane_value(top_vec, j - j0) : 0.0);
                if (hand_out && lane == kWave - 1) ring_out[(c & 1) * kColChunk + (j - j0)] = st.hprev[R - 1];
            }
            if (((jend - 1) & 15) == 15 || jend == m)                            // a decision word holds 16 columns
                st.flush(sw_dirs, ((int64_t)geom.slot0 * TB + (int64_t)((jend - 1) >> 4) * R) * kWave + lane);
        }
    };
    if (w == 0) {
        if (full) run(std::true_type{}, std::false_type{});
        else run(std::false_type{}, std::false_type{});
    } else {
        if (full) run(std::true_type{}, std::true_type{});
        else run(std::false_type{}, std::true_type{});
    }

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

__host__ __device__ inline size_t sweep_cols_team_lds_doubles(int waves) {
    return kExpDoubles + (size_t)waves * (2 * kColChunk + 4);
}

// ---------------------------------------------------------------------------------------------
// smith_waterman_score (gap 0) of a pair as a column sweep WITHOUT decisions: what MultipleAlignment.make_pairwise_matrix
// needs of a pair (multiple_alignment.py:164) -- the P x P matrix entry, no alignment.  Same recurrence and scan as
// sweep_cols; the provider's columns come through its LDS ring (RbfCoords transforms 64 columns per chunk with the seed
// superposition) and are read back with wave-uniform addresses (LDS broadcast).  np.max of the matrix is H[n][m]
// (monotone rows and columns).  One wave, strips one after the other; the row above a strip travels through `hand_g`.
// ---------------------------------------------------------------------------------------------
template <int R, class Src>
CR_D double sweep_cols_score(Src& src, const int n, const int m, double* lds, double* __restrict__ hand_g) {
    const int lane = threadIdx.x;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* ring = lds + kExpDoubles;
    load_exp_table(lds, lane);
    src.init_ring(ring, lane);
    __syncthreads();
    const int nstrips = strips_of(n, R);
    double hprev[R], eprev = 0.0;
    for (int s = 0; s < nstrips; s++) {
        const int rowbase = (s * kWave + lane) * R;
        src.load_rows(rowbase, n);
#pragma unroll
        for (int q = 0; q < R; q++) hprev[q] = 0.0;
        eprev = 0.0;
        const bool hand_out = s + 1 < nstrips;
        double top_vec = 0.0;
#pragma unroll 1
        for (int j = 0; j < m; j++) {
            if ((j & (kWave - 1)) == 0) {
                __syncthreads();
                src.load_chunk(ring, j >> 6, m, lane);
                if (s > 0) top_vec = (j + lane < m) ? hand_g[j + lane] : 0.0;
                __syncthreads();
            }
            src.fetch_col(ring, j & (kRing - 1));
            double p[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const double sc = src.score(q, tab);
                const double dg = (q == 0 ? eprev : hprev[q - 1]) + sc;
                const double b = vmax(dg, hprev[q]);
                p[q] = q == 0 ? b : vmax(p[q - 1], b);
            }
            double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
            if (s > 0) e = vmax(e, lane_value(top_vec, j & (kWave - 1)));
#pragma unroll
            for (int q = 0; q < R; q++) hprev[q] = vmax(p[q], e);
            eprev = e;
            if (hand_out && lane == kWave - 1) hand_g[j] = hprev[R - 1];
        }
        if (hand_out) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
    }
    // H[n][m]: row n - 1 lives in the last strip, lane ((n - 1) / R) % 64, slot (n - 1) % R
    const int qo = (n - 1) % R;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) v = (q == qo) ? hprev[q] : v;
    return lane_value(v, ((n - 1) / R) % kWave);
}

// The same with one wave per strip and all strips in flight (strip s works kColChunk columns behind strip s - 1, one
// barrier per phase, as sweep_cols_team): for pair lists too short to fill the chip with one wave per pair.  Every wave
// has its own column ring.  LDS (doubles): exp table | NW column rings | NW edge rings of 2 * kColChunk | NW slots.
template <int R, class Src>
CR_D double sweep_cols_score_team(Src& src, const int n, const int m, double* lds, const StripGeom geom) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* ring = lds + kExpDoubles + w * Src::kRingDoubles;
    double* edges = lds + kExpDoubles + NW * Src::kRingDoubles;
    double* edge_out = edges + w * (2 * kColChunk);
    const double* edge_in = edges + (w > 0 ? w - 1 : 0) * (2 * kColChunk);
    double* red = edges + NW * (2 * kColChunk);
    load_exp_table(lds, threadIdx.x);
    src.init_ring(ring, lane);
    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const bool mine = w < nstrips;
    const int rowbase = geom.rowbase0 + lane * R;
    const bool hand_out = w + 1 < nstrips;
    double hprev[R], eprev = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) hprev[q] = 0.0;
    if (mine) src.load_rows(rowbase, n);
    const int chunks = (m + kColChunk - 1) / kColChunk;
    const int phases = chunks + nstrips - 1;
#pragma unroll 1
    for (int g = 0; g < phases; g++) {
        __syncthreads();                               // the chunk written in phase g - 1 is visible to the strip below
        const int c = g - w;
        if (!mine || c < 0 || c >= chunks) continue;
        const int j0 = c * kColChunk;
        const int jend = j0 + kColChunk < m ? j0 + kColChunk : m;
        if ((j0 & (kWave - 1)) == 0) {                 // this wave's own ring: a wave-level fence is enough
            wave_sync();
            src.load_chunk(ring, j0 >> 6, m, lane);
            wave_sync();
        }
        double top_vec = 0.0;
        if (w > 0 && lane < kColChunk) top_vec = edge_in[(c & 1) * kColChunk + lane];
#pragma unroll 1
        for (int j = j0; j < jend; j++) {
            src.fetch_col(ring, j & (kRing - 1));
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
            if (hand_out && lane == kWave - 1) edge_out[(c & 1) * kColChunk + (j - j0)] = hprev[R - 1];
        }
    }
    const int qo = geom.owner_q;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) v = (q == qo) ? hprev[q] : v;
    if (w == geom.owner_wave && lane == geom.owner_lane) red[0] = v;
    __syncthreads();
    return red[0];
}

template <class Src>
__host__ __device__ inline size_t sweep_cols_score_team_lds_doubles(int waves) {
    return kExpDoubles + (size_t)waves * (Src::kRingDoubles + 2 * kColChunk) + 8;
}

// LDS doubles needed by a sweep of the given provider/mode for column count m and row count n
template <int R, int MODE, class Src>
__host__ __device__ inline size_t sweep_lds_doubles(int n_max, int m_max) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    size_t v = exp_doubles<Src>::value + Src::kRingDoubles;
    if (strips_of(n_max, R) > 1) v += (size_t)NB * (kWave + kRing);
    return v;
}
