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
    int32_t s_rows, s_cols, n, m;
    int32_t col0, ident;                 // ident: seq2[j] == col0 + j for every j (contiguous columns)
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
// ---------------------------------------------------------------------------------------------
constexpr int kRowsInFlight = 4;

template <int CC>
struct RowSweep {
    static constexpr int W = kWave * CC;                 // columns per strip
    static constexpr int NV = (CC + 1) / 2;              // 16-byte loads per lane and row
    static constexpr int kStride = (CC % 2 == 0) ? CC + 1 : CC;   // LDS doubles per lane: odd, conflict-free reads
    static constexpr int kBufDoubles = kWave * kStride + 2;

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
};

template <int CC>
__global__ __launch_bounds__(kWave) void k_sw_score_rows(const ExplicitProblem* __restrict__ probs,
                                                        const double* __restrict__ S,
                                                        const int32_t* __restrict__ seqs, double* __restrict__ hand,
                                                        double* __restrict__ scores) {
    using RS = RowSweep<CC>;
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
#pragma unroll
                    for (int y = 0; y < RS::NV; y++) {
                        const int k = 2 * (y * kWave + lane);
                        Pair8 v{0.0, 0.0};
                        if (k + 1 < cols) v = *reinterpret_cast<const Pair8*>(rp + k);
                        else if (k < cols) v.a = rp[k];
                        dst[y] = v;
                    }
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
                            double* rb = lds + (u & 1) * RS::kBufDoubles;
                            // transpose: loaded element k (column c0 + k) belongs to lane k / CC, slot k % CC
#pragma unroll
                            for (int y = 0; y < RS::NV; y++) {
                                const int k = 2 * (y * kWave + lane);
                                if (k < RS::W) {
                                    rb[(k / CC) * RS::kStride + k % CC] = buf[u][y].a;
                                    rb[((k + 1) / CC) * RS::kStride + (k + 1) % CC] = buf[u][y].b;
                                }
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

struct BatchTrace {
    int32_t len, start;
};

// dtw_align's traceback (dynamic_time_warping.py:90-144) for every problem of the batch: one wave per problem on the
// register-resident decision blocks of the pairwise kernels (dtw_walk).  LDS: (n + m) packed entries.
template <int R>
__global__ __launch_bounds__(kWave) void k_dtw_trace_batch(const ExplicitProblem* __restrict__ probs,
                                                          const uint32_t* __restrict__ bits,
                                                          const AlignEnd* __restrict__ ends, int max_entries,
                                                          int32_t* __restrict__ aln, BatchTrace* __restrict__ out) {
    extern __shared__ double lds[];
    const ExplicitProblem pb = probs[blockIdx.x];
    int len, pairs;
    dtw_walk<R>(pb.n, pb.m, max_entries, bits + pb.bits_off, ends[blockIdx.x].start_layer, lds, aln + pb.aln_off, len, pairs);
    if (threadIdx.x == 0) {
        out[blockIdx.x].len = len;
        out[blockIdx.x].start = pb.n + pb.m - len;
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
    DevBuf<cr::ExplicitProblem> probs_sw;
    DevBuf<double> S, hand, scores;
    DevBuf<int32_t> seqs, aln;
    DevBuf<uint32_t> bits;
    DevBuf<cr::AlignEnd> ends;
    DevBuf<cr::BatchTrace> trace;
    int64_t s_elems = 0;
    float last_ms = 0.f;             // device time of the last batch kernel (HIP events on the context's stream)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

template <int CC>
int launch_sw_rows(cr_explicit_batch* b) {
    const size_t lds = sizeof(double) * 2 * cr::RowSweep<CC>::kBufDoubles;
    CR_LAUNCH(cr::k_sw_score_rows<CC>, dim3((unsigned)b->count), dim3(cr::kWave), lds, b->ctx->stream,
              b->has_minus1 ? b->probs_sw.p : b->probs.p, b->S.p, b->seqs.p, b->hand.p, b->scores.p);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

int columns_per_lane(int m_max) {
    const int cc = (m_max + cr::kWave - 1) / cr::kWave;
    return cc <= 5 ? std::max(cc, 1) : 8;                // 1..5, then 8 (wider matrices: column strips of 512)
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
    bool has_minus1 = false;
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
    if (has_minus1 && (rc = upload(b->probs_sw, hp_sw.data(), hp_sw.size(), ctx->stream))) return rc;
    // the gather path reads S[row, seq2[c]] only for c < m; the streaming path reads whole row segments inside the row
    if ((rc = upload(b->S, S, (size_t)s_elems, ctx->stream))) return rc;
    if ((rc = upload(b->seqs, h_seq.data(), h_seq.size(), ctx->stream))) return rc;
    if ((rc = upload(b->probs, hp.data(), hp.size(), ctx->stream))) return rc;
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
        const size_t lds = cr::sweep_lds_doubles<R, MODE, cr::Explicit<R>>(b->n_max, b->m_max) * sizeof(double);
        if ((rc = allow_lds(cr::k_explicit_batch<R, MODE>, lds))) return rc;
        cr::SweepParams prm{gap, 0.0, 0.0};
        CR_LAUNCH((cr::k_explicit_batch<R, MODE>), dim3((unsigned)b->count), dim3(cr::kWave), lds, st,
                  b->has_minus1 ? b->probs_sw.p : b->probs.p, b->S.p, b->seqs.p, prm, (uint32_t*)nullptr, b->hand.p, b->ends.p);
        CR_HIP(hipGetLastError());
        CR_HIP(hipMemcpy2DAsync(b->scores.p, sizeof(double), b->ends.p, sizeof(cr::AlignEnd), sizeof(double), (size_t)b->count,
                                hipMemcpyDeviceToDevice, st));
    }
    CR_HIP(hipEventRecord(b->ev1, st));
    CR_HIP(hipMemcpyAsync(scores, b->scores.p, sizeof(double) * (size_t)b->count, hipMemcpyDeviceToHost, st));
    CR_HIP(hipStreamSynchronize(st));
    CR_HIP(hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1));
    return CR_OK;
}

int cr_dtw_align_batch(cr_explicit_batch* b, double gap_open, double gap_extend, int64_t* aln, int64_t aln_stride,
                       int64_t* aln_len, double* scores) {
    CR_REQUIRE(b != nullptr, "null batch");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    CR_REQUIRE(std::isfinite(gap_open) && std::isfinite(gap_extend), "gap penalties must be finite");
    CR_REQUIRE(!aln || (aln_len && aln_stride >= b->cap_max), "aln needs aln_len and a stride of at least the longest n + m");
    CR_REQUIRE(!b->has_minus1, "seq2: index outside the score matrix (-1 only ends a row of smith_waterman_score)");
    constexpr int R = kExplicitR;
    constexpr int MODE = cr::kDtw;
    hipStream_t st = b->ctx->stream;
    int64_t bits_total = 0, aln_total = 0;
    for (const auto& e : b->h_probs) {
        bits_total = std::max(bits_total, e.bits_off + (int64_t)cr::strips_of(e.n, R) * cr::tblocks(e.m, 8) * R * cr::kWave);
        aln_total = std::max(aln_total, e.aln_off + 2 * (int64_t)(e.n + e.m));
    }
    CR_HIP(b->bits.ensure((size_t)bits_total));
    CR_HIP(b->ends.ensure((size_t)b->count));
    const size_t lds = cr::sweep_lds_doubles<R, MODE, cr::Explicit<R>>(b->n_max, b->m_max) * sizeof(double);
    if ((rc = allow_lds(cr::k_explicit_batch<R, MODE>, lds))) return rc;
    cr::SweepParams prm{0.0, gap_open, gap_extend};
    CR_HIP(hipEventRecord(b->ev0, st));
    CR_LAUNCH((cr::k_explicit_batch<R, MODE>), dim3((unsigned)b->count), dim3(cr::kWave), lds, st, b->probs.p, b->S.p,
              b->seqs.p, prm, b->bits.p, b->hand.p, b->ends.p);
    CR_HIP(hipGetLastError());
    std::vector<cr::BatchTrace> tr;
    if (aln) {
        const int entries = b->cap_max;
        CR_HIP(b->aln.ensure((size_t)aln_total));
        CR_HIP(b->trace.ensure((size_t)b->count));
        const size_t tl = sizeof(double) * cr::trace_lds_doubles(R, entries);
        if ((rc = allow_lds(cr::k_dtw_trace_batch<R>, tl))) return rc;
        CR_LAUNCH(cr::k_dtw_trace_batch<R>, dim3((unsigned)b->count), dim3(cr::kWave), tl, st, b->probs.p, b->bits.p, b->ends.p,
                  entries, b->aln.p, b->trace.p);
        CR_HIP(hipGetLastError());
    }
    CR_HIP(hipEventRecord(b->ev1, st));
    std::vector<cr::AlignEnd> ends((size_t)b->count);
    CR_HIP(hipMemcpyAsync(ends.data(), b->ends.p, sizeof(cr::AlignEnd) * ends.size(), hipMemcpyDeviceToHost, st));
    std::vector<int32_t> h_aln;
    if (aln) {
        tr.resize((size_t)b->count);
        h_aln.resize((size_t)aln_total);
        CR_HIP(hipMemcpyAsync(tr.data(), b->trace.p, sizeof(cr::BatchTrace) * tr.size(), hipMemcpyDeviceToHost, st));
        CR_HIP(hipMemcpyAsync(h_aln.data(), b->aln.p, sizeof(int32_t) * h_aln.size(), hipMemcpyDeviceToHost, st));
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
