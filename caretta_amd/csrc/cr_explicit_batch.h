// Many explicit score matrices per launch: smith_waterman_score and dtw_align over a LIST of (S, seq1, seq2)
// (dynamic_time_warping.py:205-222 and :148-184 as MultipleAlignment.make_pairwise_matrix / progressive_align call them
// for third-party SequenceBase plugins, multiple_alignment.py:158-170, :204-217).  The matrices stay resident in HBM
// (cr_explicit_batch), so the kernels below are the HBM-bound part of the path: 8 bytes of S per DP cell.
// Included at the end of cr_api.hip, after cr_dropins.h.
#pragma once

namespace cr {

struct ExplicitProblem {
    int64_t s_off, seq1_off, seq2_off;   // element offsets of the matrix in S and of the index sequences in seqs
    int64_t hand_off;                    // doubles: hand-off column (row sweep) / hand-off rows (skewed sweep)
    int64_t dirs_off, bits_off, aln_off; // decision words and alignment rows (dtw_align only)
    int64_t bits_off_s;                  // decision words of the streaming kernels (their own rows per lane)
    int32_t s_rows, s_cols, n, m;
    int32_t col0, ident;                 // ident: seq2[j] == col0 + j for every j (contiguous columns)
};

struct BatchTrace {      // a walk's result: entries, first entry of the back-to-front rows
    int32_t len, start;
};

// 16 bytes of a row that is only 8-byte aligned
struct __attribute__((packed, aligned(8))) Pair8 {
    double a, b;
};

// ---------------------------------------------------------------------------------------------
// smith_waterman_score with gap 0 (the reference's default and only use) as a ROW SWEEP.
//
// With gap = 0 the recurrence H = max(0, diag + S, left, up) (dynamic_time_warping.py:216-221) is non-decreasing
// along rows and columns for ANY scores (the floor at 0 included), max is exact and associative, hence
//     H[i][j] = max over j' <= j of A[i][j'],     A[i][j] = max(0, H[i-1][j-1] + S[i][j], H[i-1][j]):
// the `left` dependency is a prefix maximum along the row.  One wave per problem; lane l owns the CC consecutive
// columns l * CC ... of the current column strip (64 * CC columns); every step is ONE ROW of S: read with coalesced
// 16-byte loads (the whole row segment is contiguous in HBM), several rows in flight per wave, transposed through a
// small LDS row buffer into the lanes' column order; then A, a scan along the lane's columns, a 6-step DPP max-scan
// across lanes and one more max per cell.  np.max of the matrix (:222) is H[n][m] by monotonicity.  Every value is
// bit-identical to the cell-by-cell evaluation.  Algorithmic traffic: 8 B per cell, each byte of S read once.
// Problems whose columns are not contiguous (alphabet mode: seq2 arbitrary) gather their cells with per-lane loads.
// Wider matrices take column strips one after the other; the last column of a strip goes to the next one through
// `hand` (n doubles per problem).
// Round 6: the row loads are range-checked BUFFER loads (zeros past the row's last column) and the transpose writes all 128 NV
// loaded elements into a buffer that holds them -- no EXEC-masked branch is left in a row (a third of the scalar instructions and
// their hazard nops went with them): 8 128 x 300 x 300 1.00 -> 0.95 ms (6.15 TB/s = 0.77 of peak, 0.98 of what a copy reaches).
// ---------------------------------------------------------------------------------------------
constexpr int kRowsInFlight = 4;

template <int CC>
struct RowSweep {
    static constexpr int W = kWave * CC;                 // columns per strip
    static constexpr int NV = (CC + 1) / 2;              // 16-byte loads per lane and row
    static constexpr int kStride = (CC % 2 == 0) ? CC + 1 : CC;   // LDS doubles per lane: odd, conflict-free reads
    static constexpr int kBufDoubles = kWave * kStride + 2;
    static constexpr int kTraceBufDoubles = ((2 * NV * kWave + CC - 1) / CC) * kStride + 2;   // k_sw_trace_rows: room for all 128 NV loaded elements

    double hprev[CC];
    double left_prev;        // H[i-1][c0-1]: the strip's left neighbour column, previous row

    CR_D void reset() {
#pragma unroll
        for (int x = 0; x < CC; x++) hprev[x] = 0.0;
        left_prev = 0.0;
    }
    // one row: s[x] = S of this lane's columns; `left` = H[i][c0-1] (wave-uniform; 0 in the first strip)
    template <bool LEFT>
    CR_D void step(const double* s, double left) {
        // H[i-1][c-1] of the lane's first column: the previous lane's last column, previous row
        const double dleft = wave_shr1(hprev[CC - 1], LEFT ? left_prev : 0.0);
        double p[CC];
#pragma unroll
        for (int x = 0; x < CC; x++) {
            const double dg = (x == 0 ? dleft : hprev[x - 1]) + s[x];
            const double a = vmax(vmax(0.0, dg), hprev[x]);
            p[x] = x == 0 ? a : vmax(p[x - 1], a);
        }
        double e = wave_shr1(wave_scan_max(p[CC - 1]), 0.0);
        if constexpr (LEFT) e = vmax(e, left);
#pragma unroll
        for (int x = 0; x < CC; x++) hprev[x] = vmax(p[x], e);
        left_prev = left;
    }
    // The same row WITH the decisions smith_waterman's traceback replays (dynamic_time_warping.py:255-277: h == diag + S, then
    // h == left, else up; nothing where h == 0), 2 bits per cell appended to bits[x] at bit `sh2`, and the row's first maximum:
    // with gap 0 a row never decreases, so its maximum is its last value and that value's FIRST column is the column of the
    // row's last strict increase (h != left -- the compare the decision needs anyway).  The compare's lane mask sits in an
    // SGPR pair, so "the last lane with an increase in slot x" is scalar work (s_flbit); returns the strip-relative column
    // of the row's last strict increase in this strip, -1 when the row does not grow in this strip.  The value of the
    // cell to the left of a lane's first column is the lane's exclusive prefix maximum `e` itself.
    template <bool LEFT>
    CR_D int step_trace(const double* s, double left, int sh2, uint32_t (&bits)[CC]) {
        const double dleft = wave_shr1(hprev[CC - 1], LEFT ? left_prev : 0.0);
        double dg[CC], p[CC];
#pragma unroll
        for (int x = 0; x < CC; x++) {
            dg[x] = (x == 0 ? dleft : hprev[x - 1]) + s[x];
            const double a = vmax(vmax(0.0, dg[x]), hprev[x]);
            p[x] = x == 0 ? a : vmax(p[x - 1], a);
        }
        double e = wave_shr1(wave_scan_max(p[CC - 1]), 0.0);
        if constexpr (LEFT) e = vmax(e, left);
        int last = -1;
        double lf = e;
#pragma unroll
        for (int x = 0; x < CC; x++) {
            const double h = vmax(p[x], e);
            const bool same = h == lf;
            uint32_t code = (h == dg[x]) ? 1u : same ? 2u : 3u;
            code = (h > 0.0) ? code : 0u;
            bits[x] |= code << sh2;
            // (no branch: __clzll(0) = 64 makes the column negative, and `last` starts at -1)
            const int col = (63 - __clzll((long long)__ballot(!same))) * CC + x;
            last = col > last ? col : last;
            hprev[x] = h;
            lf = h;
        }
        left_prev = left;
        return last;
    }
};

template <int CC>
__global__ __launch_bounds__(kWave) void k_sw_score_rows(const ExplicitProblem* __restrict__ probs,
                                                        const double* __restrict__ S,
                                                        const int32_t* __restrict__ seqs, double* __restrict__ hand,
                                                        double* __restrict__ scores) {
    using RS = RowSweep<CC>;
    constexpr int kTraceBuf = RowSweep<CC>::kTraceBufDoubles;
    extern __shared__ double lds[];                      // two row buffers
    const ExplicitProblem pb = probs[blockIdx.x];
    const int lane = threadIdx.x;
    const double* __restrict__ Sp = S + pb.s_off;
    const int32_t* __restrict__ seq1 = seqs + pb.seq1_off;
    const int32_t* __restrict__ seq2 = seqs + pb.seq2_off;
    double* __restrict__ hcol = hand + pb.hand_off;
    const int n = pb.n, m = pb.m;
    const int nstrips = (m + RS::W - 1) / RS::W;
    RS st;
    double result = 0.0;

    for (int cs = 0; cs < nstrips; cs++) {
        const int c0 = cs * RS::W;
        const int cols = m - c0 < RS::W ? m - c0 : RS::W;
        const bool hand_out = cs + 1 < nstrips;
        st.reset();
        auto run = [&](auto left_tag) {
            constexpr bool LEFT = decltype(left_tag)::value;
            double left_vec = 0.0;                       // H[i][c0-1] of 64 rows (lane x: row i0 + x)
            if (pb.ident) {
                // ---- contiguous columns: stream whole row segments, kRowsInFlight rows ahead ------------------
                Pair8 buf[kRowsInFlight][RS::NV];
                auto issue = [&](int row, Pair8* dst) {
                    const double* rp = Sp + (int64_t)seq1[row] * pb.s_cols + pb.col0 + c0;
                    // (buffer loads: the row's range check returns zeros past its last column -- no EXEC-masked branches around the
                    // loads; the descriptor is built from wave-uniform values)
                    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)rp);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)rp >> 32));
                    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), 0, cols * 8, 0x00020000);
#pragma unroll
                    for (int y = 0; y < RS::NV; y++)
                        dst[y] = __builtin_bit_cast(Pair8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (y * kWave + lane) * 16, 0, 0));
                };
#pragma unroll
                for (int u = 0; u < kRowsInFlight; u++)
                    if (u < n) issue(u, buf[u]);
#pragma unroll 1
                for (int i0 = 0; i0 < n; i0 += kRowsInFlight) {
#pragma unroll
                    for (int u = 0; u < kRowsInFlight; u++) {
                        const int i = i0 + u;
                        if (i < n) {
                            if (LEFT && (i & (kWave - 1)) == 0) left_vec = (i + lane < n) ? hcol[i + lane] : 0.0;
                            double* rb = lds + (u & 1) * kTraceBuf;
                            // transpose: loaded element k (column c0 + k) belongs to lane k / CC, slot k % CC (the buffer holds all
                            // 128 NV loaded elements: no lane is left out, no branch)
#pragma unroll
                            for (int y = 0; y < RS::NV; y++) {
                                const int k = 2 * (y * kWave + lane);
                                rb[(k / CC) * RS::kStride + k % CC] = buf[u][y].a;
                                rb[((k + 1) / CC) * RS::kStride + (k + 1) % CC] = buf[u][y].b;
                            }
                            if (i + kRowsInFlight < n) issue(i + kRowsInFlight, buf[u]);
                            wave_sync();
                            double s[CC];
#pragma unroll
                            for (int x = 0; x < CC; x++) s[x] = rb[lane * RS::kStride + x];
                            st.template step<LEFT>(s, LEFT ? lane_value(left_vec, i & (kWave - 1)) : 0.0);
                            if (hand_out && lane == kWave - 1) hcol[i] = st.hprev[CC - 1];
                        }
                    }
                }
            } else {
                // ---- arbitrary columns (alphabet mode): per-lane gathers -------------------------------------
                int cidx[CC];
#pragma unroll
                for (int x = 0; x < CC; x++) {
                    const int c = c0 + lane * CC + x;
                    cidx[x] = c < m ? seq2[c] : -1;
                }
#pragma unroll 1
                for (int i = 0; i < n; i++) {
                    if (LEFT && (i & (kWave - 1)) == 0) left_vec = (i + lane < n) ? hcol[i + lane] : 0.0;
                    const double* rp = Sp + (int64_t)seq1[i] * pb.s_cols;
                    double s[CC];
#pragma unroll
                    for (int x = 0; x < CC; x++) s[x] = cidx[x] >= 0 ? rp[cidx[x]] : 0.0;
                    st.template step<LEFT>(s, LEFT ? lane_value(left_vec, i & (kWave - 1)) : 0.0);
                    if (hand_out && lane == kWave - 1) hcol[i] = st.hprev[CC - 1];
                }
            }
        };
        if (cs == 0) run(std::false_type{});
        else run(std::true_type{});
        if (hand_out) {                                    // the hand-off column: visible to this wave's later loads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            wave_sync();
        } else {
            // np.max(score_matrix) = H[n][m]: column m - 1 lives in lane (m - 1 - c0) / CC, slot (m - 1 - c0) % CC
            const int k = m - 1 - c0;
            double v = 0.0;
#pragma unroll
            for (int x = 0; x < CC; x++) v = (k % CC == x) ? st.hprev[x] : v;
            result = lane_value(v, k / CC);
        }
    }
    if (lane == 0) scores[blockIdx.x] = result;
}

// ---------------------------------------------------------------------------------------------
// smith_waterman WITH its traceback (dynamic_time_warping.py:226-278) at gap 0 -- the reference's default call
// (multiple_alignment.py:332-334) -- on the same row sweep: no skew, no ring, every byte of S read once with whole-row
// loads.  Per cell the 2-bit decision of RowSweep::step_trace; a lane owns COLUMNS here, so its decision words hold 16
// consecutive ROWS of one column: exactly the layout the column sweep of the pairwise kernels writes for the TRANSPOSED
// problem (word ((strip * TB + i / 16) * CC + x) * 64 + lane, TB = ceil(n / 16)), and the walk runs on Walker<CC, 2, 0>
// with rows and columns swapped (a horizontal run of the matrix is a vertical run of the block and the other way round).
// The first maximum in row-major order (:241-247): rows never decrease downwards either, so it is the first row whose
// last value exceeds every earlier row's, at the column of that row's last strict increase.  Column strips beyond the
// first hand the rows' first-maximum columns on through `hand` (ints behind the n hand-off doubles).
// The walk follows the fill in the same wave (as in k_seed): a launch, and the walks of the waves that finish first hide
// under the row streams of the others.
// ---------------------------------------------------------------------------------------------
template <int CC>
CR_D void sw_walk_transposed(const uint32_t* __restrict__ words, const int n, const int m, const SeedMax sm, double* lds,
                             int32_t* __restrict__ a1, BatchTrace* __restrict__ out);

template <int CC, int WAVES>
__global__ __launch_bounds__(kWave, WAVES) void k_sw_trace_rows(const ExplicitProblem* __restrict__ probs, const int64_t* __restrict__ dirs_off,
                                                        const double* __restrict__ S, const int32_t* __restrict__ seqs,
                                                        double* __restrict__ hand, uint32_t* __restrict__ dirs,
                                                        SeedMax* __restrict__ seeds, int32_t* __restrict__ aln,
                                                        BatchTrace* __restrict__ out, const int walk) {
    using RS = RowSweep<CC>;
    constexpr int kTraceBuf = RowSweep<CC>::kTraceBufDoubles;
    extern __shared__ double lds[];                      // two row buffers; then the walk's packed entries
    const ExplicitProblem pb = probs[blockIdx.x];
    const int lane = threadIdx.x;
    const double* __restrict__ Sp = S + pb.s_off;
    const int32_t* __restrict__ seq1 = seqs + pb.seq1_off;
    const int32_t* __restrict__ seq2 = seqs + pb.seq2_off;
    double* __restrict__ hcol = hand + pb.hand_off;
    const int n = pb.n, m = pb.m;
    int32_t* __restrict__ rowfirst = reinterpret_cast<int32_t*>(hcol + n);       // (hand holds 3 max(n, m) doubles per problem)
    uint32_t* __restrict__ words = dirs + dirs_off[blockIdx.x];
    const int nstrips = (m + RS::W - 1) / RS::W;
    const int TB = (n + 15) >> 4;
    RS st;
    double best_v = 0.0;
    int best_i = 0, best_j = 0;

    for (int cs = 0; cs < nstrips; cs++) {
        const int c0 = cs * RS::W;
        const int cols = m - c0 < RS::W ? m - c0 : RS::W;
        const bool hand_out = cs + 1 < nstrips;
        const bool used = c0 + lane * CC < m;              // this lane holds columns of the matrix: its words can be read
        st.reset();
        uint32_t bits[CC];
#pragma unroll
        for (int x = 0; x < CC; x++) bits[x] = 0;
        auto run = [&](auto left_tag) {
            constexpr bool LEFT = decltype(left_tag)::value;
            double left_vec = 0.0;                       // H[i][c0-1] of 64 rows (lane x: row i0 + x)
            int first_vec = 0;                           // first-maximum column of 64 rows from the strips before this one
            // behind every row: its decisions into the words, the strip hand-off, the running first maximum
            auto row_done = [&](const int i, const int last) {
                if ((i & 15) == 15 || i == n - 1) {
                    const int64_t base = ((int64_t)(cs * TB + (i >> 4)) * CC) * kWave + lane;
#pragma unroll
                    for (int x = 0; x < CC; x++) {
                        if (used) words[base + x * kWave] = bits[x];
                        bits[x] = 0;
                    }
                }
                int first = last >= 0 ? c0 + last : (LEFT ? __builtin_amdgcn_readlane(first_vec, i & (kWave - 1)) : 0);
                if (hand_out) {
                    if (lane == kWave - 1) {
                        hcol[i] = st.hprev[CC - 1];
                        rowfirst[i] = first;
                    }
                } else {
                    const double rm = lane_value(st.hprev[CC - 1], kWave - 1);       // H[i][m-1] (columns past m repeat it)
                    const bool gt = rm > best_v;
                    best_v = gt ? rm : best_v;
                    best_i = gt ? i : best_i;
                    best_j = gt ? first : best_j;
                }
            };
            if (pb.ident) {
                // ---- contiguous columns: stream whole row segments, kRowsInFlight rows ahead ------------------
                // (buffer loads: the row's range check returns zeros past its last column -- no EXEC-masked branches around the
                // loads, no selects; the descriptor is built from wave-uniform values)
                Pair8 buf[kRowsInFlight][RS::NV];
                auto issue = [&](int row, Pair8* dst) {
                    const double* rp = Sp + (int64_t)seq1[row] * pb.s_cols + pb.col0 + c0;
                    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)rp);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)rp >> 32));
                    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), 0, cols * 8, 0x00020000);
#pragma unroll
                    for (int y = 0; y < RS::NV; y++)
                        dst[y] = __builtin_bit_cast(Pair8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (y * kWave + lane) * 16, 0, 0));
                };
#pragma unroll
                for (int u = 0; u < kRowsInFlight; u++)
                    if (u < n) issue(u, buf[u]);
#pragma unroll 1
                for (int i0 = 0; i0 < n; i0 += kRowsInFlight) {
#pragma unroll
                    for (int u = 0; u < kRowsInFlight; u++) {
                        const int i = i0 + u;
                        if (i < n) {
                            if (LEFT && (i & (kWave - 1)) == 0) {
                                left_vec = (i + lane < n) ? hcol[i + lane] : 0.0;
                                first_vec = (i + lane < n) ? rowfirst[i + lane] : 0;
                            }
                            double* rb = lds + (u & 1) * kTraceBuf;
#pragma unroll
                            for (int y = 0; y < RS::NV; y++) {           // (the buffer holds all 128 NV loaded elements: no lane is left out)
                                const int k = 2 * (y * kWave + lane);
                                rb[(k / CC) * RS::kStride + k % CC] = buf[u][y].a;
                                rb[((k + 1) / CC) * RS::kStride + (k + 1) % CC] = buf[u][y].b;
                            }
                            if (i + kRowsInFlight < n) issue(i + kRowsInFlight, buf[u]);
                            wave_sync();
                            double s[CC];
#pragma unroll
                            for (int x = 0; x < CC; x++) s[x] = rb[lane * RS::kStride + x];
                            const int last = st.template step_trace<LEFT>(s, LEFT ? lane_value(left_vec, i & (kWave - 1)) : 0.0, (i & 15) * 2, bits);
                            row_done(i, last);
                        }
                    }
                }
            } else {
                // ---- arbitrary columns (alphabet mode): per-lane gathers -------------------------------------
                int cidx[CC];
#pragma unroll
                for (int x = 0; x < CC; x++) {
                    const int c = c0 + lane * CC + x;
                    cidx[x] = c < m ? seq2[c] : -1;
                }
#pragma unroll 1
                for (int i = 0; i < n; i++) {
                    if (LEFT && (i & (kWave - 1)) == 0) {
                        left_vec = (i + lane < n) ? hcol[i + lane] : 0.0;
                        first_vec = (i + lane < n) ? rowfirst[i + lane] : 0;
                    }
                    const double* rp = Sp + (int64_t)seq1[i] * pb.s_cols;
                    double s[CC];
#pragma unroll
                    for (int x = 0; x < CC; x++) s[x] = cidx[x] >= 0 ? rp[cidx[x]] : 0.0;
                    const int last = st.template step_trace<LEFT>(s, LEFT ? lane_value(left_vec, i & (kWave - 1)) : 0.0, (i & 15) * 2, bits);
                    row_done(i, last);
                }
            }
        };
        if (cs == 0) run(std::false_type{});
        else run(std::true_type{});
        // the hand-off column and this wave's own decision words: visible to its later loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        wave_sync();
    }
    SeedMax sm;
    sm.score = best_v;
    sm.i = best_v > 0.0 ? best_i + 1 : 0;
    sm.j = best_v > 0.0 ? best_j + 1 : 0;
    if (lane == 0) seeds[blockIdx.x] = sm;
    // (the walk is a chain of dependent scalar instructions on a SIMD it shares with three waves that wait for HBM most of the
    // time: asking the arbiter for priority shortens the chain and costs the streams nothing)
    __builtin_amdgcn_s_setprio(3);
    if (walk) sw_walk_transposed<CC>(words, n, m, sm, lds, aln + pb.aln_off, out + blockIdx.x);   // (walk == 0: calibration only)
}

// The walk of smith_waterman (:249-278) on the decision words of the row sweep: the transposed view -- block row = column of
// the matrix, block step = row of the matrix.  One wave; rows to HBM back to front, the record to *out.
template <int CC>
CR_D void sw_walk_transposed(const uint32_t* __restrict__ words, const int n, const int m, const SeedMax sm, double* lds,
                             int32_t* __restrict__ a1, BatchTrace* __restrict__ out) {
    const int lane = threadIdx.x;
    const int TB = (n + 15) >> 4;
    uint32_t* arow = reinterpret_cast<uint32_t*>(lds);
    const int cap = n + m;
    int idx = 0;
    if (sm.i > 0) {
        Walker<CC, 2, 0> wk;
        wk.init(words, TB, lane);
        int i = __builtin_amdgcn_readfirstlane(sm.i), j = __builtin_amdgcn_readfirstlane(sm.j);
        wk.set_row(j - 1);
#pragma unroll 1
        while (i > 0 && j > 0) {
            const uint32_t code = wk.get(j - 1, i - 1);
            if (code == 0) break;
            if (code == 1) {
                const int run = wk.diag_run(j - 1, i - 1, [](uint32_t f) { return f == 1u; });
                if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(i - 1 - lane, j - 1 - lane);
                idx += run;
                i -= run;
                j -= run;
                if (j > 0) wk.set_row(j - 1);
            } else if (code == 2) {                      // left in the matrix = up in the transposed block
                bool more;
                const int run = wk.template run_up<0>(j - 1, i - 1, [](uint32_t f) { return f == 2u; }, more);
                if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(-1, j - 1 - lane);
                idx += run;
                j -= run;
                if (j > 0) wk.set_row(j - 1);
            } else {                                     // up in the matrix = left in the transposed block
                bool more;
                const int run = wk.run_left(j - 1, i - 1, [](uint32_t f) { return f == 3u; }, more);
                if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(i - 1 - lane, -1);
                idx += run;
                i -= run;
            }
        }
    }
    wave_sync();
    const int first = cap - idx;
    int32_t* a2 = a1 + cap;
    for (int x = first + lane; x < cap; x += kWave) {
        const uint32_t u = arow[x];
        const uint32_t i = u & 0xffffu, j = u >> 16;
        a1[x] = i == kGap16 ? -1 : (int)i;
        a2[x] = j == kGap16 ? -1 : (int)j;
    }
    if (lane == 0) {
        out->len = idx;
        out->start = first;
    }
}

// ---------------------------------------------------------------------------------------------
// The general recurrences (smith_waterman_score with gap != 0, dtw_align) on many matrices: the time-skewed sweep of
// the single-call drop-ins (Explicit provider, the strip's tile of S staged in LDS), one block per problem.
// ---------------------------------------------------------------------------------------------
template <int R, int MODE>
__global__ __launch_bounds__(kWave) void k_explicit_batch(const ExplicitProblem* __restrict__ probs,
                                                         const double* __restrict__ S,
                                                         const int32_t* __restrict__ seqs, SweepParams prm,
                                                         uint32_t* __restrict__ bits, double* __restrict__ hand,
                                                         AlignEnd* __restrict__ ends) {
    extern __shared__ double lds[];
    const ExplicitProblem pb = probs[blockIdx.x];
    Explicit<R> src;
    src.S = S + pb.s_off;
    src.seq1 = seqs + pb.seq1_off;
    src.seq2 = seqs + pb.seq2_off;
    src.s_cols = pb.s_cols;
    SeedMax sm;
    AlignEnd ae;
    ae.sw = ae.dtw_score = 0.0;
    ae.start_layer = ae.pad = 0;
    if (pb.m > 0) sweep<R, MODE>(src, pb.n, pb.m, prm, lds, nullptr, bits + pb.bits_off, hand + pb.hand_off, sm, ae);
    if (threadIdx.x == 0) ends[blockIdx.x] = ae;
}

// ---------------------------------------------------------------------------------------------
// Streaming provider for the time-skewed sweep on an explicit matrix with CONTIGUOUS columns: no shared tile.  Every
// lane consumes its own R rows, one column per step, i.e. one 64-byte block of each row every 8 steps -- but lane o's
// position t - o and every row's start address put the block boundaries of different (lane, row) streams at different
// steps.  The provider therefore keeps, per lane and row, a ring of TWO cache-line-ALIGNED 64-byte blocks in LDS
// ([row slot][ring slot][position][lane]: the per-step read of 64 lanes is conflict-free) and at every step that is a
// multiple of 8 fetches for every stream the ONE aligned block its next 8 steps will enter: each block of S is
// requested exactly once (unaligned 64-byte pieces were measured at 1.93 x the bytes: every straddled line came
// twice).  The loads are full-wave instructions issued 8 steps ahead into registers and COOPERATIVE: in load y the four
// lanes 4g .. 4g+3 fetch the four 16-byte quarters of the block that lane o = 16y + g will consume -- one aligned
// 64-byte request per quad.  A wave keeps R * 4 KB in flight and needs R * 8.5 KB of LDS, so several waves share a CU and
// the sweep is fed from HBM -- against one 132 KB tile per CU of the Explicit provider.  Blocks reach up to 63 doubles
// before a row (lanes that have not started yet) and 7 after it: into the neighbouring rows, or into the slack the batch
// keeps on either side of S.
// ---------------------------------------------------------------------------------------------
template <int R>
struct ExplicitStream {
    static constexpr bool kNonNegative = false;
    static constexpr bool kMaskRows = true;
    static constexpr bool kStreams = true;
    static constexpr bool kNoExp = true;                  // explicit scores: no exp table in LDS
    static constexpr int kBlk = 8;                        // doubles per aligned block
    static constexpr int kLoads = 4;                      // load instructions per row slot and boundary (16 owners each)
    static constexpr int kRingDoubles = R * 2 * kBlk * kWave + R * kWave;     // block rings | stream offsets of every lane
    const double* __restrict__ S;                         // the batch's matrices (64-byte aligned)
    const int32_t* __restrict__ seq1;
    int64_t s_off, s_cols;
    int col0;
    int m_, lane_, t_;
    double* ring_;
    int low_[R];                 // this lane's stream offsets modulo 16 (which block / position a column falls in)
    Pair8 pend[R][kLoads];       // quarter blocks in flight: pend[q][y] belongs to lane 16y + lane / 4
    double val[R];

    CR_D int64_t* stream_offsets() const { return reinterpret_cast<int64_t*>(ring_ + R * 2 * kBlk * kWave); }
    // quarter `part` of aligned block `blk + extra` of every owner's streams, where blk = the block the owner's window
    // that starts at step t0 begins in (extra = 1: the block it ends in)
    template <bool LAST>
    CR_D void issue(int t0, Pair8 (&dst)[R][kLoads]) {
        const int64_t* so = stream_offsets();
        const int part = lane_ & 3;
#pragma unroll
        for (int y = 0; y < kLoads; y++) {
            const int o = 16 * y + (lane_ >> 2);
            const int cs = t0 - o;                           // first column of the window
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int64_t off = so[o * R + q];
                const int64_t blk = ((off + cs + (LAST ? kBlk - 1 : 0)) >> 3) << 3;     // stream offset of the aligned block
                const int64_t c0 = blk - off;                                              // its first column
                Pair8 v{0.0, 0.0};
                if (off >= 0 && c0 + kBlk > 0 && c0 < m_) v = *reinterpret_cast<const Pair8*>(S + blk + 2 * part);
                dst[q][y] = v;
            }
        }
    }
    // park quarter blocks in the ring slot their block number selects
    template <bool LAST>
    CR_D void park(int t0, const Pair8 (&src)[R][kLoads]) {
        const int64_t* so = stream_offsets();
        const int part = lane_ & 3;
#pragma unroll
        for (int y = 0; y < kLoads; y++) {
            const int o = 16 * y + (lane_ >> 2);
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int64_t a = so[o * R + q] + (t0 - o) + (LAST ? kBlk - 1 : 0);
                const int slot = (int)(a >> 3) & 1;
                ring_[((q * 2 + slot) * kBlk + 2 * part) * kWave + o] = src[q][y].a;
                ring_[((q * 2 + slot) * kBlk + 2 * part + 1) * kWave + o] = src[q][y].b;
            }
        }
    }
    CR_D void init_ring(double* ring, int) { ring_ = ring; }
    CR_D void load_rows(int rowbase, int n) {
        lane_ = threadIdx.x & (kWave - 1);
        wave_sync();                                  // the previous strip's readers are done with the offsets
        int64_t* so = stream_offsets();
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int64_t off = rowbase + q < n ? s_off + (int64_t)seq1[rowbase + q] * s_cols + col0 : (int64_t)-1;
            so[lane_ * R + q] = off;
            low_[q] = (int)(off & 15);
        }
        wave_sync();
        // the window [0, 8): the block it starts in (parked at once) and the block it ends in (parked at step 0)
        Pair8 first[R][kLoads];
        issue<false>(0, first);
        issue<true>(0, pend);
        park<false>(0, first);
    }
    // ---- the same streams with their per-window state kept in registers (sweep_stream below) ---------------------------
    // Window w of an owner = its steps [8 w, 8 w + 8); what a window newly needs is the aligned block it ENDS in.  Per load y
    // and row slot q this lane keeps: the address of its quarter of that block, the block's first column, and the ring
    // word it will be parked in.  A window later everything moves on by one block: + 64 bytes, + 8 columns, the other slot.
    // (Requesting BOTH 64-byte halves of a 128-byte line when a stream enters the first one, the second kept in registers
    // for the next window, was measured: 1.46 / 1.54 ms against 1.42 / 1.45 ms -- the kernel is not bound by the re-fetched
    // lines, and the sixteen more registers cost more than the traffic saved.)
    const double* wptr_[R][kLoads];
    int wcol_[R][kLoads];
    int wpark_[R][kLoads];
    CR_D void window_init(int m) {                       // behind load_rows(): state of window 0
        m_ = m;
        const int64_t* so = stream_offsets();
        const int part = lane_ & 3;
#pragma unroll
        for (int y = 0; y < kLoads; y++) {
            const int o = 16 * y + (lane_ >> 2);
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int64_t off = so[o * R + q];
                const int64_t blk = ((off - o + (kBlk - 1)) >> 3) << 3;
                wptr_[q][y] = S + blk + 2 * part;
                wcol_[q][y] = off >= 0 ? (int)(blk - off) : -(1 << 30);          // (no row: never inside [0, m))
                wpark_[q][y] = ((q * 2 + ((int)(blk >> 3) & 1)) * kBlk + 2 * part) * kWave + o;
            }
        }
    }
    // at step t0 = 8 w: park the block window w ends in (requested a window ago: `pend`), request window w + 1's
    CR_D void window_advance() {
#pragma unroll
        for (int y = 0; y < kLoads; y++) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                ring_[wpark_[q][y]] = pend[q][y].a;
                ring_[wpark_[q][y] + kWave] = pend[q][y].b;
                wpark_[q][y] ^= kBlk * kWave;                  // the other ring slot
                wptr_[q][y] += kBlk;
                wcol_[q][y] += kBlk;
                Pair8 v{0.0, 0.0};
                if (wcol_[q][y] + kBlk > 0 && wcol_[q][y] < m_) v = *reinterpret_cast<const Pair8*>(wptr_[q][y]);
                pend[q][y] = v;
            }
        }
    }
    CR_D void load_chunk(double*, int, int, int) {}
    CR_D void step_begin(double*, int t, int m) {
        m_ = m;
        t_ = t;
        if ((t & (kBlk - 1)) == 0) {
            park<true>(t, pend);                      // the block the window [t, t + 8) ends in
            wave_sync();
            issue<true>(t + kBlk, pend);              // ... and the one the next window will end in
        }
    }
    CR_D void fetch_col(const double* ring, int) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int a = low_[q] + t_ - lane_;          // the column's stream offset modulo 16 (two's complement is fine)
            val[q] = ring[((q * 2 + ((a >> 3) & 1)) * kBlk + (a & 7)) * kWave + lane_];
        }
    }
    CR_D double score(int q, const ExpEntry*) const { return val[q]; }
};

// ---------------------------------------------------------------------------------------------
// The time-skewed sweep over ExplicitStream, written for it (sweep() of cr_sweep.h serves every provider): the wave
// is bound by instruction issue -- one row per lane leaves ~50 instructions per cell of which 13 are the recurrence -- so
//   * eight steps per loop body: the positions inside a decision word and the window boundaries are compile-time;
//   * the streams' window state lives in registers (ExplicitStream::window_advance: 4 instructions per load instead of 17);
//   * the row above the strip is read by ALL lanes from one LDS address (a broadcast; only lane 0 uses it, as the fill of
//     the DPP shift), the border of strip 0 sits in the same buffer; lane 63 hands its last row down with the store all
//     lanes execute (the others into a dump word): no EXEC juggling in a step.  (Handing down with plain 8-byte stores
//     to HBM instead -- no ring, 9 KB of LDS, sixteen waves per CU -- was measured: 1.52 / 1.61 ms against 1.42 / 1.45.)
//   * windows whose 64 lanes are all inside the matrix run without the column mask;
//   * STORE = false (dtw_align's score alone, dynamic_time_warping.py:188-201): no decision words are formed or written.
// MODE: kSwScore (smith_waterman_score with a gap), kSwTrace (smith_waterman with a gap and its traceback: 2-bit decisions + first maximum) or
// kDtw.  Same arithmetic as sweep(): dp_column on the same values.
// LDS (doubles): ring R * 1088 | pad to 128 | hout NB x 128 | hin NB x 64 | dump NB.
// ---------------------------------------------------------------------------------------------
template <int R, int MODE>
__host__ __device__ inline size_t stream_lds_doubles() {
    constexpr int NB = ((MODE & (kSwScore | kSwTrace)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    return (size_t)((ExplicitStream<R>::kRingDoubles + 127) / 128 * 128) + (size_t)NB * (kRing + kWave) + 8;
}

template <int R, int MODE, bool STORE>
CR_D void sweep_stream(ExplicitStream<R>& src, const int n, const int m, const SweepParams prm, double* lds,
                       uint32_t* __restrict__ dtw_bits, double* __restrict__ hand_g, AlignEnd& end_out,
                       uint32_t* __restrict__ sw_dirs = nullptr, SeedMax* seed_out = nullptr) {
    constexpr bool SW = (MODE & (kSwScore | kSwTrace)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;       // smith_waterman with its traceback: 2-bit decisions + first maximum (sweep())
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);
    constexpr int kRingPad = (ExplicitStream<R>::kRingDoubles + 127) / 128 * 128;
    const int lane = threadIdx.x;
    double* ring = lds;
    double* hout = ring + kRingPad;                    // [NB][128], every plane 1 KB-aligned: last row of lane 63, recent columns
    double* hin = hout + NB * kRing;                   // [NB][64]: row above lane 0 (strip 0: the DP border), current 64 columns
    src.init_ring(ring, lane);
    const int nstrips = strips_of(n, R);
    const int TB_DTW = tblocks(m, 8), TB_SW = tblocks(m, 16);
    const double col0_m2 = kMinF64 - prm.gap_open;
    DpState<R> st;
    st.sw_max = 0.0;
    double best_v = 0.0;                                 // first maximum of H in row-major order (:241-247), as sweep()
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    int sh2hi = 0;                                       // bit position of the current 8-step window inside a 16-step word
    // byte offsets of this lane's hand-down stores: lane 63 the ring (plus the column's slot), the others the dump word
    uint32_t out_base[NB];
    const uint32_t out_mask = lane == kWave - 1 ? 0x3f8u : 0u;
#pragma unroll
    for (int k = 0; k < NB; k++)
        out_base[k] = lane == kWave - 1 ? (uint32_t)((kRingPad + k * kRing) * 8) : (uint32_t)((kRingPad + NB * kRing + NB * kWave + k) * 8);

    for (int s = 0; s < nstrips; s++) {
        const int rowbase = (s * kWave + lane) * R;
        const int rows_here = n - s * kWave * R;
        const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
        const int T = m + lanes_here - 1;
        const bool hand_out = s + 1 < nstrips;
        src.load_rows(rowbase, n);
        src.window_init(m);
        st.reset_column0(col0_m2);
#pragma unroll
        for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;
        if (s == 0) {                                   // the DP border above row 0 (dynamic_time_warping.py:45-49)
            if constexpr (SW) hin[lane] = 0.0;
            if constexpr (DTW) {
                hin[(NB - 2) * kWave + lane] = col0_m2;
                hin[(NB - 1) * kWave + lane] = 0.0;
            }
        }
        wave_sync();

        // one step; K = t & 7 is a compile-time constant, MASKED = some lane may be outside the matrix
        auto step = [&](auto ktag, auto masked_tag, const int t) {
            constexpr int K = decltype(ktag)::value;
            constexpr bool MASKED = decltype(masked_tag)::value;
            const int c = t - lane;
            src.t_ = t;
            double h_top0 = 0.0, m0_top0 = 0.0, m1_top0 = 0.0;       // wave-uniform addresses: LDS broadcasts
            if constexpr (SW) h_top0 = hin[t & (kWave - 1)];
            if constexpr (DTW) {
                m0_top0 = hin[(NB - 2) * kWave + (t & (kWave - 1))];
                m1_top0 = hin[(NB - 1) * kWave + (t & (kWave - 1))];
            }
            double h_top = 0.0, m0_top = 0.0, m1_top = 0.0;
            if constexpr (SW) h_top = wave_shr1(st.h_left[R - 1], h_top0);
            if constexpr (DTW) {
                m0_top = wave_shr1(st.m0_left[R - 1], m0_top0);
                m1_top = wave_shr1(st.m1_left[R - 1], m1_top0);
            }
            auto cell = [&]() {
                src.fetch_col(ring, 0);
                dp_column<R, MODE>(src, st, prm, nullptr, c, rowbase, n, sh2hi + K * 2, K * 4, h_top, m0_top, m1_top, src.val);
            };
            if constexpr (MASKED) {
                if ((unsigned)c < (unsigned)m) cell();
            } else {
                cell();
            }
            if (hand_out && t >= kWave - 1 && t - (kWave - 1) < m) {          // (uniform) lane 63 is inside the matrix
                const uint32_t slot = (uint32_t)((t - (kWave - 1)) & (kRing - 1)) * 8u;
                char* const base = reinterpret_cast<char*>(lds);
                if constexpr (SW) *reinterpret_cast<double*>(base + ((slot & out_mask) | out_base[0])) = st.h_left[R - 1];
                if constexpr (DTW) {
                    *reinterpret_cast<double*>(base + ((slot & out_mask) | out_base[NB - 2])) = st.m0_left[R - 1];
                    *reinterpret_cast<double*>(base + ((slot & out_mask) | out_base[NB - 1])) = st.m1_left[R - 1];
                }
            }
            // (decisions are packed where they are formed: left alone, the compiler sinks the compares of all eight steps to
            // the store behind the last one and keeps their operands alive -- 198 VGPRs with one row per lane instead of ~110)
            if constexpr (DTW && STORE) {
#pragma unroll
                for (int q = 0; q < R; q++) asm volatile("" : "+v"(st.dtbits[q]));
            }
            if constexpr (TRACE) {
#pragma unroll
                for (int q = 0; q < R; q++) asm volatile("" : "+v"(st.swbits[q]));
            }
        };
        for (int t0 = 0; t0 < T; t0 += 8) {
            if ((t0 & (kWave - 1)) == 0 && nstrips > 1) {
                wave_sync();
                if (hand_out && t0 >= 2 * kWave) {              // columns [t0 - 128, t0 - 65] are complete
                    const int cc = t0 - 2 * kWave + lane;
                    if (cc < m)
                        for (int k = 0; k < NB; k++) hand_g[(int64_t)k * m + cc] = hout[k * kRing + (cc & (kRing - 1))];
                }
                if (s > 0 && t0 + lane < m)
                    for (int k = 0; k < NB; k++) hin[k * kWave + lane] = __builtin_nontemporal_load(hand_g + (int64_t)k * m + t0 + lane);
                wave_sync();
            }
            src.window_advance();
            sh2hi = (t0 & 8) * 2;
            asm volatile("" ::: "memory");                    // (LDS executes the wave's own writes and reads in order)
            if (t0 >= kWave - 1 && t0 + 7 < m && t0 + 7 < T) {            // all 64 lanes inside the matrix for all eight steps
                static_for<0, 8>([&](auto k) { step(k, std::false_type{}, t0 + decltype(k)::value); });
            } else {
                static_for<0, 8>([&](auto k) {
                    if (t0 + decltype(k)::value < T) step(k, std::true_type{}, t0 + decltype(k)::value);
                });
            }
            if constexpr (DTW && STORE) {
                const int64_t base = ((int64_t)(s * TB_DTW + (t0 >> 3)) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    dtw_bits[base + q * kWave] = st.dtbits[q];
                    st.dtbits[q] = 0;
                }
            }
            if constexpr (TRACE) {
                if ((t0 & 8) != 0 || t0 + 8 >= T) {             // a decision word holds 16 steps
                    const int64_t base = ((int64_t)(s * TB_SW + (t0 >> 4)) * R) * kWave + lane;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        sw_dirs[base + q * kWave] = st.swbits[q];
                        st.swbits[q] = 0;
                    }
                }
            }
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
        if (hand_out) {
            // flush the hand-off columns not yet written (at most 127) and make them visible to this wave's own loads in the
            // next strip
            wave_sync();
            const int tl = (T - 1) & ~(kWave - 1);                        // last chunk boundary seen
            for (int cc = (tl >= 2 * kWave ? tl - kWave : 0) + lane; cc < m; cc += kWave)
                for (int k = 0; k < NB; k++) hand_g[(int64_t)k * m + cc] = hout[k * kRing + (cc & (kRing - 1))];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            wave_sync();
        }
    }
    if constexpr (TRACE) {
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best_v, off);
            const int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
            const bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
        seed_out->score = best_v;
        seed_out->i = best_v > 0.0 ? best_i + 1 : 0;
        seed_out->j = best_v > 0.0 ? best_j + 1 : 0;
    }
    double sw_max = st.sw_max;
    if constexpr ((MODE & kSwScore) != 0) {
        for (int off = 32; off > 0; off >>= 1) sw_max = __builtin_fmax(sw_max, __shfl_xor(sw_max, off));
    }
    const int owner = ((n - 1) / R) % kWave;           // lane and register slot that own row n - 1
    const int qo = (n - 1) % R;
    double fin0 = 0.0, fin1 = 0.0, fin2 = 0.0;         // M[n][m][0..2]
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
    int idx = 0;                                       // np.argmax of the three layers at (n, m), :181-182
    double best = fin0;
    if (fin1 > best) { best = fin1; idx = 1; }
    if (fin2 > best) { best = fin2; idx = 2; }
    end_out.dtw_score = DTW ? best : 0.0;
    end_out.start_layer = idx;
    end_out.pad = 0;
}

// STORE with `aln`: dtw_align's traceback (dynamic_time_warping.py:90-144) follows the fill IN THE SAME WAVE, as in k_align: one
// launch instead of two.  Measured neutral (8 128 x 300 x 300: 1.82-1.89 ms as two launches, 1.90 as one;
// profiles/r05/explicit_batch_rate.txt): this kernel is bound by how many row streams are open, and a wave that walks has
// none open -- the walk's 19 us per problem are not hidden under its neighbours' fills as they are in the issue-bound k_align.
template <int R, int MODE, bool STORE>
__global__ __launch_bounds__(kWave) void k_explicit_stream(const ExplicitProblem* __restrict__ probs,
                                                          const double* __restrict__ S,
                                                          const int32_t* __restrict__ seqs, SweepParams prm,
                                                          uint32_t* __restrict__ bits, double* __restrict__ hand,
                                                          AlignEnd* __restrict__ ends, int max_entries = 0,
                                                          int32_t* __restrict__ aln = nullptr, BatchTrace* __restrict__ trace = nullptr) {
    extern __shared__ double lds[];
    const ExplicitProblem pb = probs[blockIdx.x];
    ExplicitStream<R> src;
    src.S = S;
    src.s_off = pb.s_off;
    src.seq1 = seqs + pb.seq1_off;
    src.s_cols = pb.s_cols;
    src.col0 = pb.col0;
    src.m_ = pb.m;
    src.t_ = 0;
    src.lane_ = threadIdx.x & (kWave - 1);
    SeedMax sm;
    AlignEnd ae;
    ae.sw = ae.dtw_score = 0.0;
    ae.start_layer = ae.pad = 0;
    if (pb.m > 0) sweep_stream<R, MODE, STORE>(src, pb.n, pb.m, prm, lds, STORE ? bits + pb.bits_off_s : nullptr, hand + pb.hand_off, ae);
    if (threadIdx.x == 0) ends[blockIdx.x] = ae;
    if constexpr (STORE && (MODE & kDtw) != 0) {
        if (aln) {                                        // (uniform) the walk on this wave's own decision words
            drain_stores();
            __builtin_amdgcn_s_setprio(3);                // (a scalar chain beside waves that stream: see k_sw_trace_rows; medians 1.89-1.92 -> 1.84-1.85 ms)
            int len, pairs;
            dtw_walk<R>(pb.n, pb.m, max_entries, bits + pb.bits_off_s, ae.start_layer, lds, aln + pb.aln_off, len, pairs);
            if (threadIdx.x == 0) {
                trace[blockIdx.x].len = len;
                trace[blockIdx.x].start = pb.n + pb.m - len;
            }
        }
    }
}

// dtw_align's traceback (dynamic_time_warping.py:90-144) for every problem of the batch: one wave per problem on the
// register-resident decision blocks of the pairwise kernels (dtw_walk).  LDS: (n + m) packed entries.
template <int R, bool STREAM>
__global__ __launch_bounds__(kWave) void k_dtw_trace_batch(const ExplicitProblem* __restrict__ probs,
                                                          const uint32_t* __restrict__ bits,
                                                          const AlignEnd* __restrict__ ends, int max_entries,
                                                          int32_t* __restrict__ aln, BatchTrace* __restrict__ out) {
    extern __shared__ double lds[];
    const ExplicitProblem pb = probs[blockIdx.x];
    int len, pairs;
    dtw_walk<R>(pb.n, pb.m, max_entries, bits + (STREAM ? pb.bits_off_s : pb.bits_off), ends[blockIdx.x].start_layer, lds,
                aln + pb.aln_off, len, pairs);
    if (threadIdx.x == 0) {
        out[blockIdx.x].len = len;
        out[blockIdx.x].start = pb.n + pb.m - len;
    }
}

// smith_waterman WITH its traceback (dynamic_time_warping.py:226-278) over the list: the fill with 2-bit decisions and the
// first maximum in row-major order (kSwTrace), Explicit tile or the streaming provider; `dirs_off`: word offset of every
// problem's decisions (laid out for this launch's rows per lane).
template <int R>
CR_D void sw_walk_skewed(const ExplicitProblem& pb, const uint32_t* __restrict__ words, const SeedMax sm, double* lds, int32_t* __restrict__ aln,
                         BatchTrace* __restrict__ out);

template <int R, bool STREAM>
__global__ __launch_bounds__(kWave) void k_explicit_sw_batch(const ExplicitProblem* __restrict__ probs, const int64_t* __restrict__ dirs_off,
                                                            const double* __restrict__ S, const int32_t* __restrict__ seqs,
                                                            SweepParams prm, uint32_t* __restrict__ dirs, double* __restrict__ hand,
                                                            SeedMax* __restrict__ seeds, int32_t* __restrict__ aln, BatchTrace* __restrict__ out) {
    extern __shared__ double lds[];
    const ExplicitProblem pb = probs[blockIdx.x];
    SeedMax sm;
    sm.score = 0.0;
    sm.i = sm.j = 0;
    AlignEnd ae;
    if (pb.m > 0) {
        if constexpr (STREAM) {
            ExplicitStream<R> src;
            src.S = S;
            src.s_off = pb.s_off;
            src.seq1 = seqs + pb.seq1_off;
            src.s_cols = pb.s_cols;
            src.col0 = pb.col0;
            src.m_ = pb.m;
            src.t_ = 0;
            src.lane_ = threadIdx.x & (kWave - 1);
            sweep_stream<R, kSwTrace, false>(src, pb.n, pb.m, prm, lds, nullptr, hand + pb.hand_off, ae, dirs + dirs_off[blockIdx.x], &sm);
        } else {
            Explicit<R> src;
            src.S = S + pb.s_off;
            src.seq1 = seqs + pb.seq1_off;
            src.seq2 = seqs + pb.seq2_off;
            src.s_cols = pb.s_cols;
            sweep<R, kSwTrace>(src, pb.n, pb.m, prm, lds, dirs + dirs_off[blockIdx.x], nullptr, hand + pb.hand_off, sm, ae);
        }
    }
    if (threadIdx.x == 0) seeds[blockIdx.x] = sm;
    // the walk on this wave's own decision words, at priority (as k_sw_trace_rows and k_explicit_stream: until round 6 a second
    // launch, k_sw_trace_batch, 0.42 ms for the 8 128 x 300 x 300 list)
    drain_stores();
    __builtin_amdgcn_s_setprio(3);
    sw_walk_skewed<R>(pb, dirs + dirs_off[blockIdx.x], sm, lds, aln, out + blockIdx.x);
}

// The walk of smith_waterman (:249-278) WITH its gap entries, one wave per problem on the register-resident decision
// blocks (Walker<R, 2>): whole diagonal, horizontal and vertical runs per ballot, entries packed in LDS back to front, rows
// to HBM coalesced (as dtw_walk).  LDS: (n + m) packed entries.
template <int R>
CR_D void sw_walk_skewed(const ExplicitProblem& pb, const uint32_t* __restrict__ words, const SeedMax sm, double* lds, int32_t* __restrict__ aln,
                         BatchTrace* __restrict__ out) {
    const int lane = threadIdx.x;
    uint32_t* arow = reinterpret_cast<uint32_t*>(lds);
    const int cap = pb.n + pb.m;
    int idx = 0;
    if (sm.i > 0) {
        Walker<R, 2, 1> wk;
        wk.init(words, tblocks(pb.m, 16), lane);
        int i = __builtin_amdgcn_readfirstlane(sm.i), j = __builtin_amdgcn_readfirstlane(sm.j);
        wk.set_row(i - 1);
#pragma unroll 1
        while (i > 0 && j > 0) {
            const uint32_t code = wk.get(i - 1, j - 1);
            if (code == 0) break;
            if (code == 1) {
                const int run = wk.diag_run(i - 1, j - 1, [](uint32_t f) { return f == 1u; });
                if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(i - 1 - lane, j - 1 - lane);
                idx += run;
                i -= run;
                j -= run;
                if (i > 0) wk.set_row(i - 1);
            } else if (code == 2) {
                bool more;
                const int run = wk.run_left(i - 1, j - 1, [](uint32_t f) { return f == 2u; }, more);
                if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(-1, j - 1 - lane);
                idx += run;
                j -= run;
            } else {
                bool more;
                const int run = wk.template run_up<0>(i - 1, j - 1, [](uint32_t f) { return f == 3u; }, more);
                if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(i - 1 - lane, -1);
                idx += run;
                i -= run;
                if (i > 0) wk.set_row(i - 1);
            }
        }
    }
    wave_sync();
    const int first = cap - idx;
    int32_t* a1 = aln + pb.aln_off;
    int32_t* a2 = a1 + cap;
    for (int x = first + lane; x < cap; x += kWave) {
        const uint32_t u = arow[x];
        const uint32_t i = u & 0xffffu, j = u >> 16;
        a1[x] = i == kGap16 ? -1 : (int)i;
        a2[x] = j == kGap16 ? -1 : (int)j;
    }
    if (lane == 0) {
        out->len = idx;
        out->start = first;
    }
}

}  // namespace cr

struct cr_explicit_batch {
    cr_context* ctx = nullptr;
    int64_t count = 0;
    int m_max = 0, n_max = 0, cap_max = 0;   // longest row count, column count, n + m
    std::vector<cr::ExplicitProblem> h_probs;
    DevBuf<cr::ExplicitProblem> probs;
    // smith_waterman_score stops a row at the first -1 of seq2 (dynamic_time_warping.py:214-215): the same problems
    // with their column counts cut there (only kept when some sequence holds a -1)
    bool has_minus1 = false;
    int m_max_sw = 0;
    // every problem has contiguous columns (seq2[j] = col0 + j): the streaming kernels apply
    bool all_ident = false, all_ident_sw = false;
    int r_stream = 2;                   // rows per lane of the streaming kernels (rows_per_lane(n_max))
    DevBuf<cr::ExplicitProblem> probs_sw;
    DevBuf<double> S, hand, scores;
    DevBuf<int32_t> seqs, aln;
    DevBuf<uint32_t> bits;
    DevBuf<cr::AlignEnd> ends;
    DevBuf<cr::BatchTrace> trace;
    DevBuf<cr::SeedMax> seeds;          // cr_smith_waterman_batch: first maximum of every problem
    DevBuf<uint32_t> sw_dirs;           // ... its 2-bit decisions
    DevBuf<int64_t> sw_dirs_off;        // ... word offset of every problem (laid out for the rows per lane of the launch)
    int sw_dirs_r = 0;                  // ... rows per lane the offsets were laid out for (0: not yet)
    int64_t s_elems = 0;
    float last_ms = 0.f;             // device time of the last batch kernel (HIP events on the context's stream)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

constexpr size_t kSlackFront = 128, kSlackBack = 32;     // doubles; the front keeps S 64-byte aligned

template <int CC>
int launch_sw_rows(cr_explicit_batch* b) {
    const size_t lds = sizeof(double) * 2 * cr::RowSweep<CC>::kTraceBufDoubles;
    CR_LAUNCH(cr::k_sw_score_rows<CC>, dim3((unsigned)b->count), dim3(cr::kWave), lds, b->ctx->stream,
              b->has_minus1 ? b->probs_sw.p : b->probs.p, b->S.p + kSlackFront, b->seqs.p, b->hand.p, b->scores.p);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

// the streaming sweep (contiguous columns) with R rows per lane; bits == nullptr: no decision words (scores alone)
// (walk_entries > 0: dtw_align's traceback in the same launch -- rows into b->aln, lengths into b->trace)
template <int R, int MODE>
int launch_stream(cr_explicit_batch* b, const cr::ExplicitProblem* probs, const cr::SweepParams& prm, uint32_t* bits, int walk_entries = 0) {
    size_t lds = cr::stream_lds_doubles<R, MODE>() * sizeof(double);
    if (walk_entries > 0) lds = std::max(lds, sizeof(double) * cr::trace_lds_doubles(R, walk_entries));
    if (g_cfg.stream_lds_kb > 0) lds = std::max(lds, (size_t)g_cfg.stream_lds_kb * 1024);   // calibration: waves per CU
    auto go = [&](auto kernel) -> int {
        int rc = allow_lds(kernel, lds);
        if (rc) return rc;
        CR_LAUNCH(kernel, dim3((unsigned)b->count), dim3(cr::kWave), lds, b->ctx->stream, probs, b->S.p + kSlackFront, b->seqs.p, prm, bits,
                  b->hand.p, b->ends.p, walk_entries, walk_entries > 0 ? b->aln.p : (int32_t*)nullptr,
                  walk_entries > 0 ? b->trace.p : (cr::BatchTrace*)nullptr);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    if constexpr ((MODE & cr::kDtw) != 0) {
        if (bits) return go(cr::k_explicit_stream<R, MODE, true>);
    }
    return go(cr::k_explicit_stream<R, MODE, false>);
}

template <int MODE>
int launch_stream_r(int R, cr_explicit_batch* b, const cr::ExplicitProblem* probs, const cr::SweepParams& prm, uint32_t* bits, int walk_entries = 0) {
    return R == 1 ? launch_stream<1, MODE>(b, probs, prm, bits, walk_entries) : R == 2 ? launch_stream<2, MODE>(b, probs, prm, bits, walk_entries)
         : R == 3 ? launch_stream<3, MODE>(b, probs, prm, bits, walk_entries) : R == 4 ? launch_stream<4, MODE>(b, probs, prm, bits, walk_entries)
         : launch_stream<5, MODE>(b, probs, prm, bits, walk_entries);
}

template <int R, bool STREAM>
int launch_trace(cr_explicit_batch* b, int entries) {
    const size_t tl = sizeof(double) * cr::trace_lds_doubles(R, entries);
    int rc = allow_lds(cr::k_dtw_trace_batch<R, STREAM>, tl);
    if (rc) return rc;
    CR_LAUNCH((cr::k_dtw_trace_batch<R, STREAM>), dim3((unsigned)b->count), dim3(cr::kWave), tl, b->ctx->stream, b->probs.p, b->bits.p,
              b->ends.p, entries, b->aln.p, b->trace.p);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

// seeds, trace records and alignment rows of a smith_waterman launch sequence -> the caller's arrays (rows left-aligned, -2 behind them)
int sw_results_to_caller(cr_explicit_batch* b, int64_t aln_total, int64_t* aln, int64_t aln_stride, int64_t* aln_len, double* scores,
                         int32_t* all_zero) {
    hipStream_t st = b->ctx->stream;
    std::vector<cr::SeedMax> seeds((size_t)b->count);
    std::vector<cr::BatchTrace> tr((size_t)b->count);
    std::vector<int32_t> h_aln((size_t)aln_total);
    CR_DOWNLOAD(b->ctx, seeds.data(), b->seeds.p, sizeof(cr::SeedMax) * seeds.size());
    CR_DOWNLOAD(b->ctx, tr.data(), b->trace.p, sizeof(cr::BatchTrace) * tr.size());
    CR_DOWNLOAD(b->ctx, h_aln.data(), b->aln.p, sizeof(int32_t) * h_aln.size());
    CR_HIP(hipStreamSynchronize(st));
    CR_HIP(hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1));
    for (int64_t p = 0; p < b->count; p++) {
        if (scores) scores[p] = seeds[(size_t)p].score;
        if (all_zero) all_zero[p] = seeds[(size_t)p].i == 0 ? 1 : 0;
        const cr::ExplicitProblem& e = b->h_probs[(size_t)p];
        const int cap = e.n + e.m;
        const int32_t* a1 = h_aln.data() + e.aln_off + tr[(size_t)p].start;
        const int32_t* a2 = a1 + cap;
        int64_t* o1 = aln + (size_t)p * 2 * (size_t)aln_stride;
        int64_t* o2 = o1 + aln_stride;
        const int len = tr[(size_t)p].len;
        for (int x = 0; x < len; x++) {
            o1[x] = a1[x];
            o2[x] = a2[x];
        }
        for (int64_t x = len; x < aln_stride; x++) o1[x] = o2[x] = -2;
        aln_len[p] = len;
    }
    return CR_OK;
}

int columns_per_lane(int m_max) {
    const int cc = (m_max + cr::kWave - 1) / cr::kWave;
    return cc <= 5 ? std::max(cc, 1) : 8;                // 1..5, then 8 (wider matrices: column strips of 512)
}

// smith_waterman with gap 0 over the list: fill with decisions, first maximum and the walk in ONE launch of the row sweep
template <int CC, int WAVES = (CC <= 4 ? 5 : CC == 5 ? 4 : 3)>
int launch_sw_trace_rows(cr_explicit_batch* b) {
    const size_t lds = std::max(sizeof(double) * 2 * cr::RowSweep<CC>::kTraceBufDoubles, sizeof(uint32_t) * ((size_t)b->cap_max + 2));
    int rc = allow_lds(cr::k_sw_trace_rows<CC, WAVES>, lds);
    if (rc) return rc;
    CR_LAUNCH((cr::k_sw_trace_rows<CC, WAVES>), dim3((unsigned)b->count), dim3(cr::kWave), lds, b->ctx->stream, b->probs.p, b->sw_dirs_off.p,
              b->S.p + kSlackFront, b->seqs.p, b->hand.p, b->sw_dirs.p, b->seeds.p, b->aln.p, b->trace.p, g_cfg.sw_rows_nowalk ? 0 : 1);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

int smith_waterman_rows(cr_explicit_batch* b, int64_t* aln, int64_t aln_stride, int64_t* aln_len, double* scores, int32_t* all_zero) {
    hipStream_t st = b->ctx->stream;
    const int CC = columns_per_lane(std::max(b->m_max, 1));
    const int W = CC * cr::kWave;
    int rc;
    if (b->sw_dirs_r != 100 + CC) {                      // decision layout of the row sweep: column strips x blocks of 16 rows x CC x 64 words
        std::vector<int64_t> off((size_t)b->count);
        int64_t total = 0;
        for (int64_t p = 0; p < b->count; p++) {
            const auto& e = b->h_probs[(size_t)p];
            off[(size_t)p] = total;
            total += (int64_t)((e.m + W - 1) / W) * ((e.n + 15) / 16) * CC * cr::kWave;
        }
        if ((rc = upload(b->sw_dirs_off, off.data(), off.size(), b->ctx))) return rc;
        CR_HIP(b->sw_dirs.ensure((size_t)std::max<int64_t>(total, 1)));
        b->sw_dirs_r = 100 + CC;
    }
    int64_t aln_total = 0;
    for (const auto& e : b->h_probs) aln_total = std::max(aln_total, e.aln_off + 2 * (int64_t)(e.n + e.m));
    CR_HIP(b->seeds.ensure((size_t)b->count));
    CR_HIP(b->aln.ensure((size_t)aln_total));
    CR_HIP(b->trace.ensure((size_t)b->count));
    CR_HIP(hipEventRecord(b->ev0, st));
    switch (CC) {
        case 1: rc = launch_sw_trace_rows<1>(b); break;
        case 2: rc = launch_sw_trace_rows<2>(b); break;
        case 3: rc = launch_sw_trace_rows<3>(b); break;
        case 4: rc = launch_sw_trace_rows<4>(b); break;
        case 5: rc = g_cfg.sw_rows_waves == 5 ? launch_sw_trace_rows<5, 5>(b) : g_cfg.sw_rows_waves == 3 ? launch_sw_trace_rows<5, 3>(b) : launch_sw_trace_rows<5>(b); break;
        default: rc = launch_sw_trace_rows<8>(b); break;
    }
    if (rc) return rc;
    CR_HIP(hipEventRecord(b->ev1, st));
    return sw_results_to_caller(b, aln_total, aln, aln_stride, aln_len, scores, all_zero);
}

}  // namespace

extern "C" {

int cr_explicit_batch_create(cr_context* ctx, const double* S, int64_t s_elems, const int64_t* seqs, int64_t seq_elems,
                             const cr_explicit_problem* problems, int64_t count, cr_explicit_batch** out) {
    CR_REQUIRE(out != nullptr, "null out");
    *out = nullptr;
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(S && seqs && problems && count >= 1 && s_elems >= 1 && seq_elems >= 1, "bad argument");
    CR_REQUIRE(count < (int64_t)1 << 30, "too many problems for one batch");
    std::vector<int32_t> h_seq((size_t)seq_elems);
    std::vector<cr::ExplicitProblem> hp((size_t)count);
    std::vector<cr::ExplicitProblem> hp_sw((size_t)count);
    bool has_minus1 = false, all_ident = true, all_ident_sw = true;
    int n_all = 0;
    for (int64_t p = 0; p < count; p++) n_all = std::max<int>(n_all, problems[p].n);
    // rows per lane of the streaming kernels: 1.  A stream costs 1 KB of LDS per lane-row pair of blocks, so the rows per
    // lane set the waves per CU: 1 row -> 12 waves (three per SIMD: FP64 at the full issue rate), 2 -> 7, 3 -> 4; and 300 rows
    // are 4.7 strips of 64 but 2.3 of 128.  Measured on 8128 x 300 x 300 (without the exp table these providers never read):
    // R = 1 / 2 / 3 -> 1.52 / 1.76 / 2.13 ms (SW, gap 0.1: 3.84 TB/s), 2.22 / 2.31 / 2.69 ms (DTW); CARETTA_STREAM_R overrides
    // (a list too short to give every CU its twelve waves is bound by the strips one wave walks one after the other: 2 rows
    // per lane halve them -- 600 x 1200 x 1200: 4.2 vs 5.8 ms)
    int r_stream = g_cfg.force_r ? rows_per_lane(std::max(n_all, 1)) : (count >= 3072 ? 1 : 2);
    if (g_cfg.stream_r >= 1 && g_cfg.stream_r <= 5) r_stream = g_cfg.stream_r;        // calibration
    int64_t bits_off_s = 0;
    int64_t hand_off = 0, bits_off = 0, aln_off = 0;
    int m_max = 0, n_max = 0, m_max_sw = 0;
    for (int64_t p = 0; p < count; p++) {
        const cr_explicit_problem& q = problems[p];
        CR_REQUIRE(q.n >= 1 && q.m >= 1 && q.s_rows >= 1 && q.s_cols >= 1, "empty sequence or score matrix");
        CR_REQUIRE(q.n <= cr::kMaxLength && q.m <= cr::kMaxLength, "sequence longer than 65534");
        CR_REQUIRE(q.s_off >= 0 && q.s_off + (int64_t)q.s_rows * q.s_cols <= s_elems, "score matrix outside S");
        CR_REQUIRE(q.seq1_off >= 0 && q.seq1_off + q.n <= seq_elems && q.seq2_off >= 0 && q.seq2_off + q.m <= seq_elems,
                   "index sequence outside seqs");
        cr::ExplicitProblem& e = hp[(size_t)p];
        e.s_off = q.s_off;
        e.seq1_off = q.seq1_off;
        e.seq2_off = q.seq2_off;
        e.s_rows = q.s_rows;
        e.s_cols = q.s_cols;
        e.n = q.n;
        e.m = q.m;
        for (int64_t x = 0; x < q.n; x++) {
            const int64_t v = seqs[q.seq1_off + x];
            CR_REQUIRE(v >= 0 && v < q.s_rows, "seq1: index outside the score matrix");
            h_seq[(size_t)(q.seq1_off + x)] = (int32_t)v;
        }
        bool ident = true, ident_sw = true;
        int m_sw = q.m;
        for (int64_t x = 0; x < q.m; x++) {
            const int64_t v = seqs[q.seq2_off + x];
            // -1 ends a row of smith_waterman_score (dynamic_time_warping.py:214-215); dtw_align has no such rule and
            // rejects it when it runs
            CR_REQUIRE(v >= -1 && v < q.s_cols, "seq2: index outside the score matrix");
            h_seq[(size_t)(q.seq2_off + x)] = (int32_t)v;
            if (v == -1 && x < m_sw) m_sw = (int)x;
            ident = ident && v == seqs[q.seq2_off] + x;
            if (x < m_sw) ident_sw = ident;
        }
        has_minus1 = has_minus1 || m_sw != q.m;
        m_max_sw = std::max(m_max_sw, m_sw);
        e.ident = ident ? 1 : 0;
        e.col0 = (int32_t)std::max<int64_t>(seqs[q.seq2_off], 0);
        e.hand_off = hand_off;
        e.bits_off = bits_off;
        e.bits_off_s = bits_off_s;
        bits_off_s += (int64_t)cr::strips_of(q.n, r_stream) * cr::tblocks(q.m, 8) * r_stream * cr::kWave;
        all_ident = all_ident && ident;
        all_ident_sw = all_ident_sw && ident_sw;
        e.dirs_off = 0;
        e.aln_off = aln_off;
        hand_off += 3 * (int64_t)std::max(q.n, q.m);
        bits_off += (int64_t)cr::strips_of(q.n, kExplicitR) * cr::tblocks(q.m, 8) * kExplicitR * cr::kWave;
        aln_off += 2 * (int64_t)(q.n + q.m);
        m_max = std::max(m_max, (int)q.m);
        n_max = std::max(n_max, (int)q.n);
        hp_sw[(size_t)p] = e;
        hp_sw[(size_t)p].m = m_sw;
        hp_sw[(size_t)p].ident = ident_sw ? 1 : 0;
    }
    CR_REQUIRE(all_finite(S, (size_t)s_elems), "score matrices contain NaN or infinity");
    cr_explicit_batch* b = new (std::nothrow) cr_explicit_batch();
    if (!b) return fail(CR_ERR_MEMORY, "out of host memory");
    struct Guard {
        cr_explicit_batch* b;
        ~Guard() { delete b; }
    } guard{b};
    b->ctx = ctx;
    b->count = count;
    b->m_max = m_max;
    b->n_max = n_max;
    for (const auto& e : hp) b->cap_max = std::max(b->cap_max, e.n + e.m);
    b->s_elems = s_elems;
    b->h_probs = hp;
    b->has_minus1 = has_minus1;
    b->m_max_sw = m_max_sw;
    b->all_ident = all_ident;
    b->all_ident_sw = all_ident_sw;
    b->r_stream = r_stream;
    if (has_minus1 && (rc = upload(b->probs_sw, hp_sw.data(), hp_sw.size(), ctx))) return rc;
    // the gather path reads S[row, seq2[c]] only for c < m; the streaming path reads whole row segments inside the row
    // S sits kSlackFront doubles into its buffer and has kSlackBack behind it: the streaming sweep reads whole aligned
    // 64-byte blocks that may start before the first row (lanes that have not reached column 0) and end after the last
    CR_HIP(b->S.ensure((size_t)s_elems + kSlackFront + kSlackBack));
    CR_HIP(hipMemsetAsync(b->S.p, 0, sizeof(double) * kSlackFront, ctx->stream));
    CR_HIP(hipMemsetAsync(b->S.p + kSlackFront + s_elems, 0, sizeof(double) * kSlackBack, ctx->stream));
    if ((rc = upload_async(ctx, b->S.p + kSlackFront, S, sizeof(double) * (size_t)s_elems))) return rc;
    if ((rc = upload(b->seqs, h_seq.data(), h_seq.size(), ctx))) return rc;
    if ((rc = upload(b->probs, hp.data(), hp.size(), ctx))) return rc;
    CR_HIP(b->hand.ensure((size_t)hand_off));
    CR_HIP(b->scores.ensure((size_t)count));
    CR_HIP(hipEventCreate(&b->ev0));
    CR_HIP(hipEventCreate(&b->ev1));
    CR_HIP(hipStreamSynchronize(ctx->stream));
    guard.b = nullptr;
    *out = b;
    return CR_OK;
}

int cr_explicit_batch_destroy(cr_explicit_batch* b) {
    if (!b) return CR_OK;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    delete b;
    return CR_OK;
}

int cr_explicit_batch_last_ms(cr_explicit_batch* b, float* ms) {
    CR_REQUIRE(b && ms, "null argument");
    *ms = b->last_ms;
    return CR_OK;
}

int cr_smith_waterman_score_batch(cr_explicit_batch* b, double gap, double* scores) {
    CR_REQUIRE(b != nullptr && scores != nullptr, "null argument");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    CR_REQUIRE(std::isfinite(gap), "gap must be finite");
    hipStream_t st = b->ctx->stream;
    CR_HIP(hipEventRecord(b->ev0, st));
    if (gap == 0.0) {
        switch (columns_per_lane(std::max(b->m_max_sw, 1))) {
            case 1: rc = launch_sw_rows<1>(b); break;
            case 2: rc = launch_sw_rows<2>(b); break;
            case 3: rc = launch_sw_rows<3>(b); break;
            case 4: rc = launch_sw_rows<4>(b); break;
            case 5: rc = launch_sw_rows<5>(b); break;
            default: rc = launch_sw_rows<8>(b); break;
        }
        if (rc) return rc;
    } else {
        constexpr int R = kExplicitR;
        constexpr int MODE = cr::kSwScore;
        CR_HIP(b->ends.ensure((size_t)b->count));
        cr::SweepParams prm{gap, 0.0, 0.0};
        const cr::ExplicitProblem* probs = b->has_minus1 ? b->probs_sw.p : b->probs.p;
        if (b->all_ident_sw) {                           // contiguous columns everywhere: the streaming sweep
            if ((rc = launch_stream_r<MODE>(b->r_stream, b, probs, prm, nullptr))) return rc;
        } else {
            const size_t lds = cr::sweep_lds_doubles<R, MODE, cr::Explicit<R>>(b->n_max, b->m_max) * sizeof(double);
            if ((rc = allow_lds(cr::k_explicit_batch<R, MODE>, lds))) return rc;
            CR_LAUNCH((cr::k_explicit_batch<R, MODE>), dim3((unsigned)b->count), dim3(cr::kWave), lds, st, probs, b->S.p + kSlackFront,
                      b->seqs.p, prm, (uint32_t*)nullptr, b->hand.p, b->ends.p);
            CR_HIP(hipGetLastError());
        }
        CR_HIP(hipMemcpy2DAsync(b->scores.p, sizeof(double), b->ends.p, sizeof(cr::AlignEnd), sizeof(double), (size_t)b->count,
                                hipMemcpyDeviceToDevice, st));
    }
    CR_HIP(hipEventRecord(b->ev1, st));
    CR_DOWNLOAD(b->ctx, scores, b->scores.p, sizeof(double) * (size_t)b->count);
    CR_HIP(hipStreamSynchronize(st));
    CR_HIP(hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1));
    return CR_OK;
}

int cr_smith_waterman_batch(cr_explicit_batch* b, double gap, int64_t* aln, int64_t aln_stride, int64_t* aln_len, double* scores,
                            int32_t* all_zero) {
    CR_REQUIRE(b != nullptr && aln != nullptr && aln_len != nullptr, "null argument");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    CR_REQUIRE(std::isfinite(gap), "gap must be finite");
    CR_REQUIRE(aln_stride >= b->cap_max, "aln needs a stride of at least the longest n + m");
    CR_REQUIRE(!b->has_minus1, "seq2: index outside the score matrix (-1 only ends a row of smith_waterman_score)");
    hipStream_t st = b->ctx->stream;
    if (gap == 0.0 && !g_cfg.no_sw_rows) return smith_waterman_rows(b, aln, aln_stride, aln_len, scores, all_zero);
    const bool stream = b->all_ident;                    // contiguous columns everywhere: the streaming sweep
    const int R = stream ? b->r_stream : kExplicitR;
    int64_t aln_total = 0;
    if (b->sw_dirs_r != R) {                             // decision layout of this launch's rows per lane
        std::vector<int64_t> off((size_t)b->count);
        int64_t total = 0;
        for (int64_t p = 0; p < b->count; p++) {
            const auto& e = b->h_probs[(size_t)p];
            off[(size_t)p] = total;
            total += (int64_t)cr::strips_of(e.n, R) * cr::tblocks(e.m, 16) * R * cr::kWave;
        }
        if ((rc = upload(b->sw_dirs_off, off.data(), off.size(), b->ctx))) return rc;
        CR_HIP(b->sw_dirs.ensure((size_t)std::max<int64_t>(total, 1)));
        b->sw_dirs_r = R;
    }
    for (const auto& e : b->h_probs) aln_total = std::max(aln_total, e.aln_off + 2 * (int64_t)(e.n + e.m));
    CR_HIP(b->seeds.ensure((size_t)b->count));
    CR_HIP(b->aln.ensure((size_t)aln_total));
    CR_HIP(b->trace.ensure((size_t)b->count));
    cr::SweepParams prm{gap, 0.0, 0.0};
    CR_HIP(hipEventRecord(b->ev0, st));
    auto fill = [&](auto rt, auto stream_tag) -> int {
        constexpr int RR = decltype(rt)::value;
        constexpr bool STREAM = decltype(stream_tag)::value;
        using Src = std::conditional_t<STREAM, cr::ExplicitStream<RR>, cr::Explicit<RR>>;
        // (fill and walk in one launch: the LDS of the larger of the two)
        const size_t fill_lds = STREAM ? cr::stream_lds_doubles<RR, cr::kSwTrace>() : cr::sweep_lds_doubles<RR, cr::kSwTrace, Src>(b->n_max, b->m_max);
        const size_t lds = std::max(fill_lds, cr::trace_lds_doubles(RR, b->cap_max)) * sizeof(double);
        int rc2 = allow_lds(cr::k_explicit_sw_batch<RR, STREAM>, lds);
        if (rc2) return rc2;
        CR_LAUNCH((cr::k_explicit_sw_batch<RR, STREAM>), dim3((unsigned)b->count), dim3(cr::kWave), lds, st, b->probs.p, b->sw_dirs_off.p,
                  b->S.p + kSlackFront, b->seqs.p, prm, b->sw_dirs.p, b->hand.p, b->seeds.p, b->aln.p, b->trace.p);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    if (!stream) rc = fill(std::integral_constant<int, kExplicitR>{}, std::false_type{});
    else rc = R == 1 ? fill(std::integral_constant<int, 1>{}, std::true_type{}) : R == 2 ? fill(std::integral_constant<int, 2>{}, std::true_type{})
            : R == 3 ? fill(std::integral_constant<int, 3>{}, std::true_type{}) : R == 4 ? fill(std::integral_constant<int, 4>{}, std::true_type{})
            : fill(std::integral_constant<int, 5>{}, std::true_type{});
    if (rc) return rc;
    CR_HIP(hipEventRecord(b->ev1, st));
    return sw_results_to_caller(b, aln_total, aln, aln_stride, aln_len, scores, all_zero);
}

int cr_dtw_align_batch(cr_explicit_batch* b, double gap_open, double gap_extend, int64_t* aln, int64_t aln_stride,
                       int64_t* aln_len, double* scores) {
    CR_REQUIRE(b != nullptr, "null batch");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    CR_REQUIRE(std::isfinite(gap_open) && std::isfinite(gap_extend), "gap penalties must be finite");
    CR_REQUIRE(!aln || (aln_len && aln_stride >= b->cap_max), "aln needs aln_len and a stride of at least the longest n + m");
    CR_REQUIRE(!b->has_minus1, "seq2: index outside the score matrix (-1 only ends a row of smith_waterman_score)");
    constexpr int MODE = cr::kDtw;
    hipStream_t st = b->ctx->stream;
    const bool stream = b->all_ident;                    // contiguous columns everywhere: the streaming sweep
    const int R = stream ? b->r_stream : kExplicitR;
    int64_t bits_total = 0, aln_total = 0;
    for (const auto& e : b->h_probs) {
        bits_total = std::max(bits_total, (stream ? e.bits_off_s : e.bits_off) + (int64_t)cr::strips_of(e.n, R) * cr::tblocks(e.m, 8) * R * cr::kWave);
        aln_total = std::max(aln_total, e.aln_off + 2 * (int64_t)(e.n + e.m));
    }
    CR_HIP(b->bits.ensure((size_t)bits_total));
    CR_HIP(b->ends.ensure((size_t)b->count));
    cr::SweepParams prm{0.0, gap_open, gap_extend};
    if (aln) {
        CR_HIP(b->aln.ensure((size_t)aln_total));
        CR_HIP(b->trace.ensure((size_t)b->count));
    }
    CR_HIP(hipEventRecord(b->ev0, st));
    if (stream) {
        // (with alignments: fill, decisions and the walk of every problem in ONE launch)
        if ((rc = launch_stream_r<MODE>(R, b, b->probs.p, prm, aln ? b->bits.p : nullptr, aln ? b->cap_max : 0))) return rc;
    } else {
        const size_t lds = cr::sweep_lds_doubles<kExplicitR, MODE, cr::Explicit<kExplicitR>>(b->n_max, b->m_max) * sizeof(double);
        if ((rc = allow_lds(cr::k_explicit_batch<kExplicitR, MODE>, lds))) return rc;
        CR_LAUNCH((cr::k_explicit_batch<kExplicitR, MODE>), dim3((unsigned)b->count), dim3(cr::kWave), lds, st, b->probs.p,
                  b->S.p + kSlackFront, b->seqs.p, prm, b->bits.p, b->hand.p, b->ends.p);
        CR_HIP(hipGetLastError());
    }
    std::vector<cr::BatchTrace> tr;
    if (aln && !stream) {                                // (the tile kernels: the walks as a launch of their own)
        if ((rc = launch_trace<kExplicitR, false>(b, b->cap_max))) return rc;
    }
    CR_HIP(hipEventRecord(b->ev1, st));
    std::vector<cr::AlignEnd> ends((size_t)b->count);
    CR_DOWNLOAD(b->ctx, ends.data(), b->ends.p, sizeof(cr::AlignEnd) * ends.size());
    std::vector<int32_t> h_aln;
    if (aln) {
        tr.resize((size_t)b->count);
        h_aln.resize((size_t)aln_total);
        CR_DOWNLOAD(b->ctx, tr.data(), b->trace.p, sizeof(cr::BatchTrace) * tr.size());
        CR_DOWNLOAD(b->ctx, h_aln.data(), b->aln.p, sizeof(int32_t) * h_aln.size());
    }
    CR_HIP(hipStreamSynchronize(st));
    CR_HIP(hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1));
    for (int64_t p = 0; p < b->count; p++) {
        if (scores) scores[p] = ends[(size_t)p].dtw_score;
        if (!aln) continue;
        const cr::ExplicitProblem& e = b->h_probs[(size_t)p];
        const int cap = e.n + e.m;
        const int32_t* a1 = h_aln.data() + e.aln_off + tr[(size_t)p].start;
        const int32_t* a2 = a1 + cap;
        int64_t* o1 = aln + (size_t)p * 2 * (size_t)aln_stride;
        int64_t* o2 = o1 + aln_stride;
        const int len = tr[(size_t)p].len;
        for (int x = 0; x < len; x++) {
            o1[x] = a1[x];
            o2[x] = a2[x];
        }
        for (int64_t x = len; x < aln_stride; x++) o1[x] = o2[x] = -2;
        aln_len[p] = len;
    }
    return CR_OK;
}

}  // extern "C"
