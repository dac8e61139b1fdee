// The DP state of a lane, one column of its cells (dp_column) and the time-skewed sweep of one wave per pair (sweep).
// Part of cr_kernels.h (included there, inside namespace cr, in this order: cr_providers.h, cr_sweep.h, cr_sweep_cols.h,
// cr_sweep_wide.h, cr_trace.h, cr_pair_kernels.h); not a header of its own.

// Registers a lane carries from column to column of its R rows.
template <int R>
struct DpState {
    double h_left[R];                       // SW: H of this lane's rows, previous column
    double m0_left[R], m1_left[R], m2_left[R];   // DTW layers, previous column (m0: current column, kept for (n, m))
    double rowmax[R];                       // SW trace: running first maximum of each row ...
    int rowarg[R];                          // ... and its column
    uint32_t swbits[R], dtbits[R];          // decisions of the current word
    double h_diag, m1_diag;                 // row above the lane's block, previous column
    // (this lane's last row, current column -- h_left / m0_left / m1_left [R - 1] -- is handed down by DPP)
    double sw_max;                          // SW score: running maximum

    CR_D void reset_column0(double col0_m2) {   // DP border left of column 0
#pragma unroll
        for (int q = 0; q < R; q++) {
            h_left[q] = 0.0;
            m0_left[q] = 0.0;
            m1_left[q] = 0.0;          // M[i][0][1] = 0
            m2_left[q] = col0_m2;      // M[i][0][2] = MIN - open
            rowmax[q] = 0.0;
            rowarg[q] = 0;
        }
        h_diag = 0.0;
        m1_diag = 0.0;
    }
};

// v_max_f64 as is.  __builtin_fmax makes the compiler canonicalise operands it cannot prove quiet (values that came
// through DPP or LDS) with an extra v_max_f64 x, x; the data here is never NaN, and the instruction itself returns the
// larger operand unchanged.
CR_D double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

enum : int { kSwTrace = 1, kSwScore = 2, kDtw = 4, kZeroGap = 8 };   // kZeroGap: sw_gap == 0.0

// Diagnostic builds only (tools/step_probe.hip): what a step of the sweeps costs without its decision packing / without
// the hand-off writes of the lanes that hand nothing down.  Never defined in the library.
#ifdef CR_PROBE_NO_DECISIONS
constexpr bool kProbeNoDecisions = true;
#else
constexpr bool kProbeNoDecisions = false;
#endif
#ifdef CR_PROBE_NO_DUMP
constexpr bool kProbeNoDump = true;
#else
constexpr bool kProbeNoDump = false;
#endif
#ifdef CR_PROBE_MASKED_RAMPS
constexpr bool kProbeMaskedRamps = true;
#else
constexpr bool kProbeMaskedRamps = false;
#endif

struct SweepParams {
    double sw_gap, gap_open, gap_extend;
};

// The R cells of one column of one lane.  *_top: the row above the lane's block in this column.
// max(a, b) is v_max_f64: value-identical to the reference's compare-and-keep for non-NaN data.
template <int R, int MODE, class Src>
CR_D void dp_column(const Src& src, DpState<R>& st, const SweepParams& prm, const ExpEntry* tab, int c, int rowbase,
                    int n, int sh2, int sh4, double h_top, double m0_top, double m1_top, const double* ready = nullptr) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr bool ZG = (MODE & kZeroGap) != 0;        // x - 0.0 == x: the gap subtractions vanish
    constexpr bool NOFLOOR = ZG && Src::kNonNegative;  // all candidates >= +0: max(0, .) is the identity
    // Phase 1: everything that reads the PREVIOUS column's values of the row above (the diagonal terms) and of the row
    // itself (the horizontal gap layer), for all R rows, before any of them is overwritten: the old values die here, so
    // the new ones can take their registers (no copies of the loop-carried state).
    double dg[R], c1[R], m2n[R];
    bool b2[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
        // `ready`: the scores of this column, formed one step ahead (sweep_wide with few rows per lane)
        const double sc = ready ? ready[q] : src.score(q, tab);
        if constexpr (SW) dg[q] = (q == 0 ? st.h_diag : st.h_left[q - 1]) + sc;
        if constexpr (DTW) {
            c1[q] = (q == 0 ? st.m1_diag : st.m1_left[q - 1]) + sc;
            const double up0 = st.m1_left[q] - prm.gap_open;
            const double up1 = st.m2_left[q] - prm.gap_extend;
            b2[q] = up1 > up0;
            m2n[q] = vmax(up0, up1);
        }
    }
    // Phase 2: the chain down the lane's rows
    double h_up = h_top;
    double m0_up = m0_top, m1_up = m1_top;
#pragma unroll
    for (int q = 0; q < R; q++) {
        if constexpr (SW) {
            // H = max(0, diag + S, left - gap, up - gap)
            const double lf = ZG ? st.h_left[q] : st.h_left[q] - prm.sw_gap;
            const double up = ZG ? h_up : h_up - prm.sw_gap;
            const double h = NOFLOOR ? vmax(vmax(dg[q], lf), up)
                                     : vmax(vmax(vmax(0.0, dg[q]), lf), up);
            if constexpr (TRACE && !kProbeNoDecisions) {
                // decision replayed by the traceback's equality tests (:255-277)
                const bool same = h == lf;
                uint32_t code = (h == dg[q]) ? 1u : same ? 2u : 3u;
                code = (h > 0.0) ? code : 0u;
                // (gap 0 on non-negative scores: a row never decreases, its running maximum IS its last value -- a new
                // first maximum is a strict increase, which the decision's own comparison has already found)
                bool gt = NOFLOOR ? !same : h > st.rowmax[q];
                if constexpr (Src::kMaskRows) {
                    const bool rv = rowbase + q < n;
                    gt = gt & rv;
                    code = rv ? code : 0u;
                }
                st.swbits[q] |= code << sh2;
                if constexpr (NOFLOOR) st.rowmax[q] = h;          // (no instruction: the value h_left takes anyway)
                else if constexpr (Src::kMaskRows) st.rowmax[q] = gt ? h : st.rowmax[q];
                else st.rowmax[q] = vmax(st.rowmax[q], h);       // same value as the select, one instruction
                st.rowarg[q] = gt ? c : st.rowarg[q];
            } else {
                if constexpr (Src::kMaskRows) {
                    st.sw_max = (rowbase + q < n) ? vmax(st.sw_max, h) : st.sw_max;
                } else {
                    st.sw_max = vmax(st.sw_max, h);
                }
            }
            h_up = h;
            st.h_left[q] = h;
        }
        if constexpr (DTW) {
            const double lo0 = m0_up - prm.gap_extend;
            const double lo1 = m1_up - prm.gap_open;
            const bool b0 = lo1 > lo0;                  // np.argmax keeps the first maximum
            const double m0 = vmax(lo0, lo1);
            const bool g1 = c1[q] > m0;
            const double m01 = vmax(m0, c1[q]);
            const bool g2 = m2n[q] > m01;
            const double m1 = vmax(m01, m2n[q]);
            if constexpr (!kProbeNoDecisions) {
                const uint32_t nib = (b0 ? 1u : 0u) | (g2 ? 4u : (g1 ? 2u : 0u)) | (b2[q] ? 8u : 0u);
                st.dtbits[q] |= nib << sh4;
            }
            m0_up = m0;
            m1_up = m1;
            st.m0_left[q] = m0;
            st.m1_left[q] = m1;
            st.m2_left[q] = m2n[q];
        }
    }
    // (the values handed down to the next lane are the new h_left / m0_left / m1_left of the lane's last row)
    if constexpr (SW) st.h_diag = h_top;
    if constexpr (DTW) st.m1_diag = m1_top;
}

// ---------------------------------------------------------------------------------------------
// The sweep.  One wave, one pair.  MODE selects the recurrences evaluated per cell:
//   kSwTrace : SW fill + 2-bit decisions + first maximum   (dynamic_time_warping.py:226-247)
//   kSwScore : SW fill, maximum only                        (dynamic_time_warping.py:205-222)
//   kDtw     : 3-layer affine fill + 4-bit decisions        (dynamic_time_warping.py:8-86,181-182)
// LDS layout (doubles): [0,kExpDoubles) exp table | ring | hand-off in-ring NB*64 | hand-off out-ring NB*128
// (the last two only if the pair needs more than one strip).
//
// Lanes whose column c = t - lane lies outside [0, m) are switched off with the EXEC mask for the
// whole cell block, so their state registers keep the DP border values without any select.
// Rows past n (last strip only) are fed features of 1e150: their RBF score underflows to exactly 0,
// so they can only repeat values of valid cells and lose every first-maximum tie (larger row).
// Providers that cannot do that (explicit score matrix) set kMaskRows.
// max(a, b) is v_max_f64: value-identical to the reference's compare-and-keep for non-NaN data.
// ---------------------------------------------------------------------------------------------
template <int R, int MODE, class Src>
CR_D void sweep(Src& src, const int n, const int m, const SweepParams prm, double* lds,
                uint32_t* __restrict__ sw_dirs, uint32_t* __restrict__ dtw_bits, double* __restrict__ hand_g,
                SeedMax& seed_out, AlignEnd& end_out) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);   // values handed from strip to strip per column
    const int lane = threadIdx.x;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* ring = lds + exp_doubles<Src>::value;
    // A strip's last row is handed to the next strip through HBM (hand_g: NB planes of m doubles, L2
    // resident), staged on both sides through small LDS rings with coalesced transfers every 64 steps.
    double* hin = ring + Src::kRingDoubles;            // [NB][64]  row above lane 0, current 64 columns
    double* hout = hin + NB * kWave;                   // [NB][128] last row of lane 63, most recent columns

    if constexpr (exp_doubles<Src>::value != 0) load_exp_table(lds, lane);
    src.init_ring(ring, lane);
    __syncthreads();

    const int nstrips = strips_of(n, R);
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;      // M[i][0][2], M[0][j][0] (dynamic_time_warping.py:45,49)

    // first maximum of H in row-major order (smith_waterman, :241-247): lane-level running best
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    DpState<R> st;
    st.sw_max = 0.0;

    for (int s = 0; s < nstrips; s++) {
        const int rowbase = (s * kWave + lane) * R;
        const int rows_here = n - s * kWave * R;                        // rows left for this strip
        const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
        const int T = m + lanes_here - 1;
        src.load_rows(rowbase, n);
        st.reset_column0(col0_m2);
#pragma unroll
        for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;
    
        for (int t = 0; t < T; t++) {
            if ((t & (kWave - 1)) == 0) {
                __syncthreads();
                src.load_chunk(ring, t >> 6, m, lane);
                if (nstrips > 1) {
                    if (s + 1 < nstrips && t >= 2 * kWave) {    // columns [t-128, t-65] are complete
                        const int cc = t - 2 * kWave + lane;
                        if (cc < m)
                            for (int k = 0; k < NB; k++) hand_g[(int64_t)k * m + cc] = hout[k * kRing + (cc & (kRing - 1))];
                    }
                    if (s > 0 && t + lane < m)
                        for (int k = 0; k < NB; k++)
                            hin[k * kWave + lane] = __builtin_nontemporal_load(hand_g + (int64_t)k * m + t + lane);
                }
                __syncthreads();
            }
            if constexpr (is_streaming<Src>::value) src.step_begin(ring, t, m);
            const int c = t - lane;
            const bool active = (unsigned)c < (unsigned)m;

            // row above this lane's block: lane 0 reads the DP border (strip 0) or the hand-off row
            double h_top0 = 0.0, m0_top0 = col0_m2, m1_top0 = 0.0;   // M[0][j][0] = MIN - open, M[0][j][1] = 0
            if (s > 0 && lane == 0 && active) {
                if constexpr (SW) h_top0 = hin[c & (kWave - 1)];
                if constexpr (DTW) {
                    m0_top0 = hin[(NB - 2) * kWave + (c & (kWave - 1))];
                    m1_top0 = hin[(NB - 1) * kWave + (c & (kWave - 1))];
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
                if (s + 1 < nstrips && lane == kWave - 1) {
                    if constexpr (SW) hout[c & (kRing - 1)] = st.h_left[R - 1];
                    if constexpr (DTW) {
                        hout[(NB - 2) * kRing + (c & (kRing - 1))] = st.m0_left[R - 1];
                        hout[(NB - 1) * kRing + (c & (kRing - 1))] = st.m1_left[R - 1];
                    }
                }
            }
            // Decision words go out in the order the sweep forms them (one 256-byte row of words per store instruction), but
            // only the words a walk can ever read: a lane whose rows lie past n, or whose steps of this word all lie outside
            // the columns [0, m) -- the pipeline's ramps --, keeps out of the store (round 5: k_align wrote 503 MB where the
            // cells' decisions are 366 MB; the padding words of the ramps and of the last lanes were a fifth of it).
            if constexpr (TRACE) {
                if ((t & 15) == 15 || t == T - 1) {
                    const int64_t base = ((int64_t)(s * TB_SW + (t >> 4)) * R) * kWave + lane;
                    const bool used = rowbase < n && t >= lane && (t & ~15) - lane < m;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        if (used) sw_dirs[base + q * kWave] = st.swbits[q];
                        st.swbits[q] = 0;
                    }
                }
            }
            if constexpr (DTW) {
                if ((t & 7) == 7 || t == T - 1) {
                    const int64_t base = ((int64_t)(s * TB_DTW + (t >> 3)) * R) * kWave + lane;
                    const bool used = rowbase < n && t >= lane && (t & ~7) - lane < m;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        if (used) dtw_bits[base + q * kWave] = st.dtbits[q];
                        st.dtbits[q] = 0;
                    }
                }
            }
        }
        if (s + 1 < nstrips) {
            // flush the hand-off columns not yet written (at most 127) and make them visible to this
            // wave's own loads in the next strip
            __syncthreads();
            const int tl = (T - 1) & ~(kWave - 1);                        // last chunk boundary seen
            for (int cc = (tl >= 2 * kWave ? tl - kWave : 0) + lane; cc < m; cc += kWave)
                for (int k = 0; k < NB; k++) hand_g[(int64_t)k * m + cc] = hout[k * kRing + (cc & (kRing - 1))];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
        if constexpr (TRACE) {
            // fold this strip's per-row first maxima into the lane's running best (rows ascending)
#pragma unroll
            for (int q = 0; q < R; q++) {
                const bool gt = st.rowmax[q] > best_v;
                best_v = gt ? st.rowmax[q] : best_v;
                best_i = gt ? rowbase + q : best_i;
                best_j = gt ? st.rowarg[q] : best_j;
            }
        }
    }
    double sw_max = st.sw_max;

    // ---- wave reductions: results are returned in every lane ------------------------------------
    if constexpr (TRACE) {
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(best_v, off);
            int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
            bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
        seed_out.score = best_v;
        seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
        seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    }
    if constexpr ((MODE & kSwScore) != 0 || DTW) {
        if constexpr ((MODE & kSwScore) != 0) {
            for (int off = 32; off > 0; off >>= 1) sw_max = __builtin_fmax(sw_max, __shfl_xor(sw_max, off));
        }
        const int owner = ((n - 1) / R) % kWave;       // lane and register slot that own row n-1
        const int qo = (n - 1) % R;
        double fin0 = 0.0, fin1 = 0.0, fin2 = 0.0;     // M[n][m][0..2]
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
        int idx = 0;                                   // np.argmax of the three layers at (n, m), :181-182
        double best = fin0;
        if (fin1 > best) { best = fin1; idx = 1; }
        if (fin2 > best) { best = fin2; idx = 2; }
        end_out.dtw_score = DTW ? best : 0.0;
        end_out.start_layer = idx;
        end_out.pad = 0;
    }
    __syncthreads();                                   // the caller may reuse the LDS from here on
}
