// Single-call drop-ins of the reference's njit functions (host buffers in/out, device compute) and
// the host-side integer / tree work.  Included at the end of cr_api.hip.
#pragma once

namespace cr {

// make_score_matrix (score_functions.py:23-51): one thread per cell, coalesced along j.
__global__ void k_score_matrix(const double* __restrict__ a, int n, const double* __restrict__ b, int m, int k,
                               double neg_gamma, double* __restrict__ S) {
    __shared__ ExpEntry tab[kExpEntries];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    for (int x = tid; x < kExpEntries; x += blockDim.x * blockDim.y) tab[x] = kExpTable[x];
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= n || j >= m) return;
    const double* ai = a + (int64_t)i * k;
    const double* bj = b + (int64_t)j * k;
    double df = ai[0] - bj[0];
    double acc = df * df;
    for (int x = 1; x < k; x++) {
        df = ai[x] - bj[x];
        acc = acc + df * df;
    }
    S[(int64_t)i * m + j] = exp_tab<false>(neg_gamma * acc, tab);
}

// coordinate score matrix on the seed-superposed frames (multiple_alignment.py:344-349)
__global__ void k_score_matrix_xf(const double* __restrict__ xi, int n, const double* __restrict__ xj, int m,
                                  const Transform* __restrict__ xf, double neg_gamma, double* __restrict__ S) {
    __shared__ ExpEntry tab[kExpEntries];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    for (int x = tid; x < kExpEntries; x += blockDim.x * blockDim.y) tab[x] = kExpTable[x];
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= n || j >= m) return;
    double a[3], o[3];
    const bool raw = xf->flags & kFlagSeedSkipped;
    for (int x = 0; x < 3; x++) {
        const double v = xi[(int64_t)i * 3 + x];
        a[x] = raw ? v : v - xf->c1[x];
    }
    if (raw) {
        for (int x = 0; x < 3; x++) o[x] = xj[(int64_t)j * 3 + x];
    } else {
        double w[3];
        for (int x = 0; x < 3; x++) w[x] = xj[(int64_t)j * 3 + x] - xf->c2[x];
        rot3(w, xf->R, o);
    }
    const double dx = a[0] - o[0], dy = a[1] - o[1], dz = a[2] - o[2];
    const double acc = (dx * dx + dy * dy) + dz * dz;
    S[(int64_t)i * m + j] = exp_tab<false>(neg_gamma * acc, tab);
}

// DP on an explicit score matrix (dtw_align / smith_waterman(_score) drop-ins).  One wave.
template <int R, int MODE>
__global__ __launch_bounds__(kWave) void k_explicit(const int32_t* __restrict__ seq1, int n,
                                                   const int32_t* __restrict__ seq2, int m,
                                                   const double* __restrict__ S, int64_t s_cols, SweepParams prm,
                                                   uint32_t* __restrict__ dirs, uint32_t* __restrict__ bits,
                                                   double* __restrict__ hand, SeedMax* __restrict__ seed,
                                                   AlignEnd* __restrict__ end) {
    extern __shared__ double lds[];
    Explicit<R> src;
    src.S = S;
    src.seq1 = seq1;
    src.seq2 = seq2;
    src.s_cols = s_cols;
    SeedMax sm;
    AlignEnd ae;
    sweep<R, MODE>(src, n, m, prm, lds, dirs, bits, hand, sm, ae);
    if (threadIdx.x == 0) {
        if constexpr ((MODE & kSwTrace) != 0) *seed = sm;
        if constexpr ((MODE & (kSwScore | kDtw)) != 0) *end = ae;
    }
}

struct TraceOut {
    int32_t len, start;
};

__global__ void k_dtw_trace_full(int n, int m, int R, const uint32_t* __restrict__ bits,
                                 const AlignEnd* __restrict__ end, int32_t* __restrict__ aln, TraceOut* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int cap = n + m;
    const int len = dtw_traceback(bits, R, tblocks(m, 8), n, m, end->start_layer, aln, aln + cap, cap);
    out->len = len;
    out->start = cap - len;
}

__global__ void k_sw_trace_full(int n, int m, int R, const uint32_t* __restrict__ dirs,
                                const SeedMax* __restrict__ seed, int32_t* __restrict__ aln, TraceOut* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int cap = n + m;
    int len = 0;
    if (seed->i > 0) len = sw_traceback(dirs, R, tblocks(m, 16), seed->i, seed->j, aln, aln + cap, cap);
    out->len = len;
    out->start = cap - len;
}

// paired_svd_superpose (superposition_functions.py:7-35), sequential sums as under numba.  Host and device: the
// single-call drop-ins run it on the host for small inputs (small_on_host()), the kernels for the rest.
CR_HD void kabsch_seq(const double* __restrict__ x1, const double* __restrict__ x2, int k, double* R, double* t,
                     double* c1, double* c2) {
    for (int a = 0; a < 3; a++) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < k; i++) {
            s1 += x1[(int64_t)i * 3 + a];
            s2 += x2[(int64_t)i * 3 + a];
        }
        c1[a] = s1 / (double)k;
        c2[a] = s2 / (double)k;
    }
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < k; i++) {
        const double a[3] = {x2[(int64_t)i * 3] - c2[0], x2[(int64_t)i * 3 + 1] - c2[1], x2[(int64_t)i * 3 + 2] - c2[2]};
        const double b[3] = {x1[(int64_t)i * 3] - c1[0], x1[(int64_t)i * 3 + 1] - c1[1], x1[(int64_t)i * 3 + 2] - c1[2]};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) C[3 * r + c] += a[r] * b[c];
    }
    kabsch_from_correlation(C, c1, c2, R, t);
}

// out layout: R[9], t[3], c1[3], c2[3]
__global__ void k_kabsch(const double* __restrict__ x1, const double* __restrict__ x2, int k, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    kabsch_seq(x1, x2, k, out, out + 9, out + 12, out + 15);
}

// apply_rotran (superposition_functions.py:64-80); with center != nullptr computes (x - center) @ R
// and with R == nullptr just x - center (the two lines superposition_functions.py:57-58).
__global__ void k_transform(const double* __restrict__ x, int k, const double* __restrict__ R,
                            const double* __restrict__ t, const double* __restrict__ center, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    double v[3] = {x[(int64_t)i * 3], x[(int64_t)i * 3 + 1], x[(int64_t)i * 3 + 2]};
    if (center)
        for (int a = 0; a < 3; a++) v[a] = v[a] - center[a];
    double o[3] = {v[0], v[1], v[2]};
    if (R) rot3(v, R, o);
    if (t)
        for (int a = 0; a < 3; a++) o[a] = o[a] + t[a];
    for (int a = 0; a < 3; a++) out[(int64_t)i * 3 + a] = o[a];
}

// get_rmsd (score_functions.py:15-19) and tm_score (multiple_alignment.py:59-70); out[0]=rmsd, out[1]=tm
CR_HD void rmsd_tm_seq(const double* __restrict__ x1, const double* __restrict__ x2, int k, int64_t l1, int64_t l2,
                       double* __restrict__ out) {
    const double d1 = 1.24 * (double)(l1 - 15) / 3.0 - 1.8;
    const double d2 = 1.24 * (double)(l2 - 15) / 3.0 - 1.8;
    double ss = 0.0, sum1 = 0.0, sum2 = 0.0;
    for (int i = 0; i < k; i++) {
        const double e0 = x1[(int64_t)i * 3] - x2[(int64_t)i * 3];
        const double e1 = x1[(int64_t)i * 3 + 1] - x2[(int64_t)i * 3 + 1];
        const double e2 = x1[(int64_t)i * 3 + 2] - x2[(int64_t)i * 3 + 2];
        ss += e0 * e0;
        ss += e1 * e1;
        ss += e2 * e2;
        const double sg = (e0 + e1) + e2;
        const double q1 = sg / d1, q2 = sg / d2;
        sum1 += 1.0 / (1.0 + q1 * q1);
        sum2 += 1.0 / (1.0 + q2 * q2);
    }
    out[0] = sqrt(ss / (double)k);
    const double t1 = (1.0 / (double)l1) * sum1;
    const double t2 = (1.0 / (double)l2) * sum2;
    out[1] = t1 > t2 ? t1 : t2;
}

__global__ void k_rmsd_tm(const double* __restrict__ x1, const double* __restrict__ x2, int k, int64_t l1, int64_t l2,
                          double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    rmsd_tm_seq(x1, x2, k, l1, l2, out);
}

// superpose_core (multiple_alignment.py:914-950): every structure fitted onto the reference over the gap-free
// columns of the alignment.  k_core_reference (one wave): centroid of the reference's core coordinates (sequential
// mean, helper.py:46-53) and the centred core coordinates x1[e].  k_core_superpose (one wave per structure):
// paired_svd_superpose(x1, own core coordinates) with every sum in column order, then apply_rotran to the whole
// structure; the reference itself is only shifted by the centroid.
__global__ __launch_bounds__(kWave) void k_core_reference(const double* __restrict__ ref_coords,
                                                         const int32_t* __restrict__ ref_row,
                                                         const int32_t* __restrict__ core, int ncore,
                                                         double* __restrict__ x1, double* __restrict__ centroid) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const double s = ordered_sums<3>(ncore, lane, lds, [&](int e, double* out) {
        const double* v = ref_coords + (int64_t)ref_row[core[e]] * 3;
        out[0] = v[0];
        out[1] = v[1];
        out[2] = v[2];
    });
    const double mean = s / (double)ncore;
    const double c[3] = {lane_value(mean, 0), lane_value(mean, 1), lane_value(mean, 2)};
    for (int e = lane; e < ncore; e += kWave) {
        const double* v = ref_coords + (int64_t)ref_row[core[e]] * 3;
        for (int a = 0; a < 3; a++) x1[(int64_t)e * 3 + a] = v[a] - c[a];
    }
    if (lane < 3) centroid[lane] = c[lane];
}

__global__ __launch_bounds__(kWave) void k_core_superpose(const double* __restrict__ coords,
                                                         const int64_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ msa, int W,
                                                         const int32_t* __restrict__ core, int ncore, int ref,
                                                         const double* __restrict__ x1,
                                                         const double* __restrict__ centroid, double* __restrict__ out) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, s = blockIdx.x;
    const int64_t off = offsets[s], len = offsets[s + 1] - off;
    const double* X = coords + off * 3;
    double* O = out + off * 3;
    if (s == ref) {
        for (int64_t r = lane; r < len; r += kWave)
            for (int a = 0; a < 3; a++) O[r * 3 + a] = X[r * 3 + a] - centroid[a];
        return;
    }
    uint32_t* ent = reinterpret_cast<uint32_t*>(lds);
    double* scratch = lds + ((size_t)ncore + 3) / 4 * 2;
    for (int e = lane; e < ncore; e += kWave) ent[e] = pack_entry(e, msa[(int64_t)s * W + core[e]]);
    wave_sync();
    double c1[3], c2[3], Rm[9], t[3];
    kabsch_ordered(x1, X, ent, ncore, ncore, lane, scratch, c1, c2, Rm, t);
    for (int64_t r = lane; r < len; r += kWave) {              // apply_rotran (superposition_functions.py:64-80)
        const double v[3] = {X[r * 3], X[r * 3 + 1], X[r * 3 + 2]};
        double o[3];
        rot3(v, Rm, o);
        for (int a = 0; a < 3; a++) O[r * 3 + a] = o[a] + t[a];
    }
}

// superpose_reference (multiple_alignment.py:953-972): the structures listed in `which` fitted onto the reference
// over the alignment columns the two share.  `ref_coords` are the reference's coordinates to fit on (the loop of the
// reference refits the reference structure itself when it reaches it, so later structures see the refitted copy).
// counts[b] = number of shared columns (the caller checks > 3, :965).
__global__ __launch_bounds__(kWave) void k_reference_superpose(const double* __restrict__ coords,
                                                              const int64_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ msa, int W, int ref,
                                                              const double* __restrict__ ref_coords,
                                                              const int32_t* __restrict__ which, double* __restrict__ out,
                                                              int32_t* __restrict__ counts) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, s = which[blockIdx.x];
    const int64_t off = offsets[s], len = offsets[s + 1] - off;
    const double* X = coords + off * 3;
    double* O = out + off * 3;
    uint32_t* ent = reinterpret_cast<uint32_t*>(lds);
    double* scratch = lds + ((size_t)W + 3) / 4 * 2;
    int kloc = 0;
    for (int x = lane; x < W; x += kWave) {
        const int a = msa[(int64_t)ref * W + x], b = msa[(int64_t)s * W + x];
        const bool pair = a != -1 && b != -1;
        ent[x] = pair ? pack_entry(a, b) : pack_entry(-1, -1);
        kloc += pair ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) kloc += __shfl_xor(kloc, o);
    wave_sync();
    if (lane == 0) counts[blockIdx.x] = kloc;
    if (kloc <= 3) return;
    double c1[3], c2[3], Rm[9], t[3];
    kabsch_ordered(ref_coords, X, ent, W, kloc, lane, scratch, c1, c2, Rm, t);
    for (int64_t r = lane; r < len; r += kWave) {
        const double v[3] = {X[r * 3], X[r * 3 + 1], X[r * 3 + 2]};
        double o[3];
        rot3(v, Rm, o);
        for (int a = 0; a < 3; a++) O[r * 3 + a] = o[a] + t[a];
    }
}

}  // namespace cr

namespace {

template <class T>
int upload(DevBuf<T>& buf, const T* host, size_t count, cr_context* ctx) {
    hipError_t e = buf.ensure(count);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? CR_ERR_MEMORY : CR_ERR_HIP, std::string("upload: ") + hipGetErrorString(e));
    return count ? upload_async(ctx, buf.p, host, sizeof(T) * count) : CR_OK;
}

int to_i32(const int64_t* seq, int64_t len, int64_t bound, std::vector<int32_t>& out, const char* what) {
    out.resize((size_t)len);
    for (int64_t x = 0; x < len; x++) {
        if (seq[x] < 0 || seq[x] >= bound) return fail(CR_ERR_ARGUMENT, std::string(what) + ": index outside the score matrix");
        out[(size_t)x] = (int32_t)seq[x];
    }
    return CR_OK;
}

struct ExplicitRun {
    DevBuf<double> S;
    DevBuf<int32_t> s1, s2, aln;
    DevBuf<uint32_t> dirs, bits;
    DevBuf<double> hand;
    DevBuf<cr::SeedMax> seed;
    DevBuf<cr::AlignEnd> end;
    DevBuf<cr::TraceOut> tout;
    DevBuf<double> staged;              // the matrix in the staged sweep's step order (cr_staged.h)
    int r = 0;                          // rows per lane of the decision words
    bool walked = false;                // dtw_align's traceback already ran (inside the staged kernel)
};
static_assert(sizeof(cr::TraceOut) == sizeof(cr::StagedTrace), "trace records differ");

constexpr int kExplicitR = 2;     // 128 rows per strip: the strip's tile (128 x 129 doubles) fits the LDS

// shared body of the three explicit-matrix drop-ins
// One matrix of up to 1024 rows: gathered into the staged sweep's step order by its own launch (every CU), then ONE
// workgroup with a wave per 64 (128) rows instead of one wave for everything; dtw_align's traceback by the walker of the
// pairwise kernels in the same launch.  300 x 300: dtw_align 0.63 -> see profiles/r03/dropin_latency.txt.
template <int R, int MODE>
int run_explicit_staged(cr_context* ctx, int64_t n, int64_t m, int64_t s_cols, cr::SweepParams prm, ExplicitRun& r, bool walk,
                        const cr::StagedShape shape) {
    CR_HIP(r.staged.ensure((size_t)shape.pair_doubles()));
    const int steps = (int)m + cr::kWave - 1, tc = kStageSteps;
    CR_LAUNCH(cr::k_stage_explicit<R>, dim3((unsigned)((steps + tc - 1) / tc)), dim3(shape.waves * cr::kWave), 0, ctx->stream, r.s1.p,
              (int)n, r.s2.p, (int)m, r.S.p, s_cols, tc, r.staged.p, shape);
    CR_HIP(hipGetLastError());
    const int entries = (int)(n + m);
    const size_t lds = sizeof(double) * std::max(cr::sweep_staged_lds_doubles<MODE>(shape.waves), walk ? cr::trace_lds_doubles(R, entries) : (size_t)0);
    auto go = [&](auto kernel) -> int {
        int rc = allow_lds(kernel, lds);
        if (rc) return rc;
        CR_LAUNCH(kernel, dim3(1), dim3(shape.waves * cr::kWave), lds, ctx->stream, (int)n, (int)m, prm, r.staged.p, shape, r.dirs.p,
                  r.bits.p, r.seed.p, r.end.p, entries, r.aln.p, reinterpret_cast<cr::StagedTrace*>(r.tout.p));
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    r.r = R;
    r.walked = walk;
    if constexpr ((MODE & cr::kDtw) != 0) {
        if (walk) return go(cr::k_explicit_staged<R, MODE, true>);
    }
    return go(cr::k_explicit_staged<R, MODE, false>);
}

template <int MODE>
int run_explicit(cr_context* ctx, const int64_t* seq1, int64_t n, const int64_t* seq2, int64_t m, const double* S,
                 int64_t s_rows, int64_t s_cols, cr::SweepParams prm, ExplicitRun& r, bool walk = false) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(seq1 && seq2 && S, "null input array");
    CR_REQUIRE(n >= 1 && m >= 1 && s_rows >= 1 && s_cols >= 1, "empty sequence or score matrix");
    CR_REQUIRE(n < (1 << 24) && m < (1 << 24), "sequence too long");
    std::vector<int32_t> h1, h2;
    if ((rc = to_i32(seq1, n, s_rows, h1, "seq1"))) return rc;
    if ((rc = to_i32(seq2, m, s_cols, h2, "seq2"))) return rc;
    if ((rc = upload(r.S, S, (size_t)s_rows * s_cols, ctx))) return rc;
    if ((rc = upload(r.s1, h1.data(), (size_t)n, ctx))) return rc;
    if ((rc = upload(r.s2, h2.data(), (size_t)m, ctx))) return rc;
    constexpr int R = kExplicitR;
    // (sized for either layout: strips of 64 R' rows hold ceil(n / 64 R') R' row slots, at most ceil(n / 64) + 3)
    const size_t slots = (size_t)((n + cr::kWave - 1) / cr::kWave) + 4;
    const size_t nd = slots * cr::tblocks((int)m, 16) * cr::kWave;
    const size_t nb = slots * cr::tblocks((int)m, 8) * cr::kWave;
    CR_HIP(r.dirs.ensure((MODE & cr::kSwTrace) ? nd : 1));
    CR_HIP(r.bits.ensure((MODE & cr::kDtw) ? nb : 1));
    CR_HIP(r.hand.ensure(3 * (size_t)m));
    CR_HIP(r.seed.ensure(1));
    CR_HIP(r.end.ensure(1));
    CR_HIP(r.tout.ensure(1));
    CR_HIP(r.aln.ensure(2 * (size_t)(n + m)));
    r.r = R;
    r.walked = false;
    {
        const cr::StagedShape shape = staged_shape((int)std::min<int64_t>(n, cr::kStagedMaxRows), (int)m);
        if (n <= cr::kStagedMaxRows && g_cfg.staged &&
            (double)shape.pair_doubles() * sizeof(double) <= 2.0 * 1024 * 1024 * 1024 &&
            (!walk || sizeof(double) * cr::trace_lds_doubles(shape.r, (int)(n + m)) <= 159 * 1024))
            return by_rows(shape.r, [&](auto rt) { return run_explicit_staged<decltype(rt)::value, MODE>(ctx, n, m, s_cols, prm, r, walk, shape); });
    }
    const size_t lds = cr::sweep_lds_doubles<R, MODE, cr::Explicit<R>>((int)n, (int)m) * sizeof(double);
    if ((rc = allow_lds(cr::k_explicit<R, MODE>, lds))) return rc;
    CR_LAUNCH((cr::k_explicit<R, MODE>), dim3(1), dim3(cr::kWave), lds, ctx->stream, r.s1.p, (int)n, r.s2.p,
                       (int)m, r.S.p, s_cols, prm, r.dirs.p, r.bits.p, r.hand.p, r.seed.p, r.end.p);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

int fetch_alignment(cr_context* ctx, ExplicitRun& r, int64_t n, int64_t m, int64_t* aln1, int64_t* aln2,
                    int64_t* aln_len) {
    cr::TraceOut to;
    CR_DOWNLOAD(ctx, &to, r.tout.p, sizeof(to));
    CR_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t cap = n + m;
    std::vector<int32_t> h((size_t)(2 * cap));
    CR_DOWNLOAD_WAIT(ctx, h.data(), r.aln.p, sizeof(int32_t) * (size_t)(2 * cap));
    for (int x = 0; x < to.len; x++) {
        aln1[x] = h[(size_t)(to.start + x)];
        aln2[x] = h[(size_t)(cap + to.start + x)];
    }
    if (aln_len) *aln_len = to.len;
    return CR_OK;
}

// k_node<R> over `count` tree nodes (one wave each); n_max / m_max / entries bound the LDS of the launch
template <int R>
int launch_node_r(hipStream_t stream, int count, int n_max, int m_max, int entries, const cr::PairDesc* pairs,
                  const double* coords, const double* tensors, int d, const double* weights, const cr::NodeDesc* nodes,
                  const cr::Transform* xf, const cr_params& prm, double gamma_weight, uint32_t* bits, double* hand,
                  int32_t* aln, double* xn, double* tn, double* wn, cr::NodeOut* out) {
    const size_t lds = sizeof(double) * std::max(cr::sweep_lds_doubles<R, cr::kDtw, cr::RbfNode<R>>(n_max, m_max),
                                                 (size_t)cr::kExpDoubles + cr::trace_lds_doubles(R, entries));
    int rc = allow_lds(cr::k_node<R>, lds);
    if (rc) return rc;
    CR_LAUNCH(cr::k_node<R>, dim3((unsigned)count), dim3(cr::kWave), lds, stream, pairs, coords, tensors, d, weights,
                       nodes, xf, prm.gamma_coords, gamma_weight, prm.gap_open, prm.gap_extend, entries, bits, hand, aln, xn,
                       tn, wn, out);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

// team kernels (kTeamWaves waves per node): R = 1 .. 5 rows per lane, strips_of(n_max, R) <= kTeamWaves (n_max <= 1280)
template <int R>
int launch_node_team_r(hipStream_t stream, int count, int n_max, int m_max, int entries, const cr::PairDesc* pairs,
                       const double* coords, const double* tensors, int d, const double* weights,
                       const cr::NodeDesc* nodes, const cr::Transform* xf, const cr_params& prm, double gamma_weight,
                       uint32_t* bits, double* hand, int32_t* aln, double* xn, double* tn, double* wn, cr::NodeOut* out) {
    const size_t lds = sizeof(double) * std::max(cr::sweep_wide_lds_doubles<cr::kDtw, cr::RbfNode<R>>(cr::kTeamWaves, m_max),
                                                 (size_t)cr::kExpDoubles + cr::trace_lds_doubles(R, entries));
    int rc = allow_lds(cr::k_node_team<R>, lds);
    if (rc) return rc;
    CR_LAUNCH(cr::k_node_team<R>, dim3((unsigned)count), dim3(cr::kTeamWaves * cr::kWave), lds, stream, pairs, coords,
                       tensors, d, weights, nodes, xf, prm.gamma_coords, gamma_weight, prm.gap_open, prm.gap_extend, entries,
                       bits, hand, aln, xn, tn, wn, out);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <class... A>
int launch_node_team(int R, A... a) {
    return R == 1 ? launch_node_team_r<1>(a...) : R == 2 ? launch_node_team_r<2>(a...) : R == 3 ? launch_node_team_r<3>(a...)
         : R == 4 ? launch_node_team_r<4>(a...) : launch_node_team_r<5>(a...);
}

template <int R, int D, bool ZG>
int launch_seed_team_zg(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    using Src = cr::RbfTensor<R, D>;
    const int entries = std::min(ck.n_max, ck.m_max);
    const size_t fill = ZG ? cr::sweep_cols_team_lds_doubles(cr::kTeamWaves) : cr::sweep_team_lds_doubles<R, cr::kSwTrace, Src>(cr::kTeamWaves);
    const size_t lds = sizeof(double) * std::max(fill, (size_t)cr::kExpDoubles + cr::trace_lds_doubles(R, entries));
    int rc = allow_lds(cr::k_seed_team<R, D, ZG>, lds);
    if (rc) return rc;
    CR_LAUNCH((cr::k_seed_team<R, D, ZG>), dim3((unsigned)ck.count), dim3(cr::kTeamWaves * cr::kWave), lds,
                       b->launch_stream ? b->launch_stream : b->ctx->stream, b->pairs.p + ck.first, b->tensors.p, (int)b->d, b->coords.p, prm.gamma_tensor,
                       prm.sw_gap, entries, b->dirs.p, b->xf.p + ck.first, b->seed_score.p + ck.first);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int R>
int launch_seed_team_r(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const bool zg = prm.sw_gap == 0.0;
    switch (b->d_pad) {
        case 4: return zg ? launch_seed_team_zg<R, 4, true>(b, ck, prm) : launch_seed_team_zg<R, 4, false>(b, ck, prm);
        case 8: return zg ? launch_seed_team_zg<R, 8, true>(b, ck, prm) : launch_seed_team_zg<R, 8, false>(b, ck, prm);
        case 10: return zg ? launch_seed_team_zg<R, 10, true>(b, ck, prm) : launch_seed_team_zg<R, 10, false>(b, ck, prm);
        case 16: return zg ? launch_seed_team_zg<R, 16, true>(b, ck, prm) : launch_seed_team_zg<R, 16, false>(b, ck, prm);
        case 24: return zg ? launch_seed_team_zg<R, 24, true>(b, ck, prm) : launch_seed_team_zg<R, 24, false>(b, ck, prm);
        case 32: return zg ? launch_seed_team_zg<R, 32, true>(b, ck, prm) : launch_seed_team_zg<R, 32, false>(b, ck, prm);
        default: return fail(CR_ERR_ARGUMENT, "unsupported tensor width");
    }
}

int launch_seed_team(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    return R == 1 ? launch_seed_team_r<1>(b, ck, prm) : R == 2 ? launch_seed_team_r<2>(b, ck, prm)
         : R == 3 ? launch_seed_team_r<3>(b, ck, prm) : R == 4 ? launch_seed_team_r<4>(b, ck, prm) : launch_seed_team_r<5>(b, ck, prm);
}

// the seed kernel that matches the batch's layout (cr_batch_set_pairs chose team / rows per lane)
int launch_seed_auto(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    if (b->staged) {
        const cr::StagedShape shape = staged_shape(b->n_max, b->m_max);
        const int rc = launch_stage_tensor(b, ck, prm, b->staged_scores.p, shape);
        return rc ? rc : launch_seed_staged(b, ck, prm, b->staged_scores.p, shape);
    }
    if (b->wide_sync) return launch_seed_wide(b->r_seed, b, ck, prm);
    if (b->team) return launch_seed_team(b->r_seed, b, ck, prm);
    return launch_seed_r(b->r_seed, b, ck, prm);
}

template <class... A>
int launch_node(int R, A... a) {
    return R == 2 ? launch_node_r<2>(a...) : R == 3 ? launch_node_r<3>(a...) : R == 4 ? launch_node_r<4>(a...) : launch_node_r<5>(a...);
}

}  // namespace

extern "C" {

int cr_make_score_matrix(cr_context* ctx, const double* a, int64_t n, const double* b, int64_t m, int64_t k,
                         double gamma, double* S) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(n >= 0 && m >= 0 && k >= 1, "bad shape");
    if (n == 0 || m == 0) return CR_OK;
    CR_REQUIRE(a && b && S, "null array");
    CR_REQUIRE(n < (1 << 24) && m < (1 << 24), "matrix too large");
    CR_REQUIRE(all_finite(a, (size_t)n * k) && all_finite(b, (size_t)m * k) && std::isfinite(gamma),
               "inputs contain NaN or infinity");
    DevBuf<double> da, db, ds;
    if ((rc = upload(da, a, (size_t)n * k, ctx))) return rc;
    if ((rc = upload(db, b, (size_t)m * k, ctx))) return rc;
    CR_HIP(ds.ensure((size_t)n * m));
    dim3 block(64, 4), grid((unsigned)((m + 63) / 64), (unsigned)((n + 3) / 4));
    CR_LAUNCH(cr::k_score_matrix, grid, block, 0, ctx->stream, da.p, (int)n, db.p, (int)m, (int)k, -gamma, ds.p);
    CR_HIP(hipGetLastError());
    return download(ctx, S, ds.p, sizeof(double) * (size_t)n * m);
}

int cr_protein_score_function(cr_context* ctx, const double* coords_i, const double* tensors_i, int64_t n,
                              const double* coords_j, const double* tensors_j, int64_t m, int64_t d,
                              double gamma_tensor, double gamma_coords, double* S, uint32_t* flags) {
    CR_REQUIRE(coords_i && tensors_i && coords_j && tensors_j && S, "null array");
    CR_REQUIRE(n >= 1 && m >= 1, "empty structure");
    CR_REQUIRE(gamma_ok(gamma_tensor) && gamma_ok(gamma_coords),
               "gamma_tensor and gamma_coords must be finite and >= 1e-290 (below that every score is exactly 1.0)");
    // a two-structure batch driven through stages 1-2, then the explicit matrix
    std::vector<double> coords((size_t)(n + m) * 3), tensors((size_t)(n + m) * d);
    std::memcpy(coords.data(), coords_i, sizeof(double) * (size_t)n * 3);
    std::memcpy(coords.data() + n * 3, coords_j, sizeof(double) * (size_t)m * 3);
    std::memcpy(tensors.data(), tensors_i, sizeof(double) * (size_t)n * d);
    std::memcpy(tensors.data() + n * d, tensors_j, sizeof(double) * (size_t)m * d);
    const int64_t offsets[3] = {0, n, n + m};
    const int32_t pair[2] = {0, 1};
    cr_batch* b = nullptr;
    int rc = cr_batch_create(ctx, coords.data(), tensors.data(), offsets, 2, d, &b);
    if (rc) return rc;
    rc = cr_batch_set_pairs(b, pair, 1);
    if (rc == CR_OK) {
        cr_params prm{gamma_tensor, gamma_coords, 1.0, 0.01, 0.0};
        rc = launch_seed_auto(b, b->chunks[0], prm);
    }
    if (rc == CR_OK) {
        DevBuf<double> ds;
        hipError_t e = ds.ensure((size_t)n * m);
        if (e == hipSuccess) {
            dim3 block(64, 4), grid((unsigned)((m + 63) / 64), (unsigned)((n + 3) / 4));
            CR_LAUNCH(cr::k_score_matrix_xf, grid, block, 0, ctx->stream, b->coords.p, (int)n,
                               b->coords.p + n * 3, (int)m, b->xf.p, -gamma_coords, ds.p);
            e = hipGetLastError();
        }
        cr::Transform tr;
        if (e == hipSuccess) e = hipMemcpyAsync(&tr, b->xf.p, sizeof(tr), hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) rc = fail(CR_ERR_HIP, std::string("score_function: ") + hipGetErrorString(e));
        else rc = download(ctx, S, ds.p, sizeof(double) * (size_t)n * m);
        if (!rc && flags) *flags = tr.flags;
    }
    cr_batch_destroy(b);
    return rc;
}

int cr_progressive_node(cr_context* ctx, const double* coords_1, const double* tensors_1, const double* weights_1,
                        int64_t n, const double* coords_2, const double* tensors_2, const double* weights_2, int64_t m,
                        int64_t d, double mult1, double mult2, const cr_params* params, double gamma_weight,
                        int64_t* aln1, int64_t* aln2, int64_t* aln_len, double* coords_out, double* tensors_out,
                        double* weights_out, uint32_t* flags) {
    CR_REQUIRE(coords_1 && tensors_1 && weights_1 && coords_2 && tensors_2 && weights_2 && params, "null input");
    CR_REQUIRE(aln1 && aln2 && aln_len && coords_out && tensors_out && weights_out, "null output");
    CR_REQUIRE(n >= 1 && m >= 1, "empty node");
    CR_REQUIRE(std::isfinite(gamma_weight) && gamma_weight >= 0.0 && std::isfinite(mult1) && std::isfinite(mult2),
               "gamma_weight must be finite and >= 0, multipliers finite");
    CR_REQUIRE(all_finite(weights_1, (size_t)n) && all_finite(weights_2, (size_t)m), "weights contain NaN or infinity");
    // the two children as a two-structure batch: k_seed gives the seed superposition, k_node the rest
    std::vector<double> coords((size_t)(n + m) * 3), tensors((size_t)(n + m) * d), weights((size_t)(n + m));
    std::memcpy(coords.data(), coords_1, sizeof(double) * (size_t)n * 3);
    std::memcpy(coords.data() + n * 3, coords_2, sizeof(double) * (size_t)m * 3);
    std::memcpy(tensors.data(), tensors_1, sizeof(double) * (size_t)n * d);
    std::memcpy(tensors.data() + n * d, tensors_2, sizeof(double) * (size_t)m * d);
    std::memcpy(weights.data(), weights_1, sizeof(double) * (size_t)n);
    std::memcpy(weights.data() + n, weights_2, sizeof(double) * (size_t)m);
    const int64_t offsets[3] = {0, n, n + m};
    const int32_t pair[2] = {0, 1};
    cr_batch* b = nullptr;
    int rc = cr_batch_create(ctx, coords.data(), tensors.data(), offsets, 2, d, &b);
    if (rc) return rc;
    struct Guard {
        cr_batch* b;
        ~Guard() { cr_batch_destroy(b); }
    } guard{b};
    rc = cr_batch_set_pairs(b, pair, 1);     // on staged scores up to 1024 rows (cr_staged.h) ...
    if (!rc && !b->staged) {                 // ... else single-wave or four-wave team kernels: the node kernel has no wide version
        g_no_wide = true;
        rc = cr_batch_set_pairs(b, pair, 1);
        g_no_wide = false;
    }
    if (rc) return rc;
    const cr_params prm = *params;
    CR_REQUIRE(gamma_ok(prm.gamma_tensor) && gamma_ok(prm.gamma_coords) && std::isfinite(prm.gap_open) && std::isfinite(prm.gap_extend) &&
                   std::isfinite(prm.sw_gap),
               "parameters must be finite, gamma_tensor and gamma_coords >= 1e-290 (below that every score is exactly 1.0)");
    const cr_batch::Chunk& ck = b->chunks[0];
    rc = launch_seed_auto(b, ck, prm);
    if (rc) return rc;
    const int64_t cap = n + m;
    DevBuf<double> dw, dxn, dtn, dwn;
    DevBuf<cr::NodeOut> dout;
    if ((rc = upload(dw, weights.data(), (size_t)cap, ctx))) return rc;
    CR_HIP(dxn.ensure((size_t)cap * 3));
    CR_HIP(dtn.ensure((size_t)cap * d));
    CR_HIP(dwn.ensure((size_t)cap));
    CR_HIP(dout.ensure(1));
    const int R = b->r_align;
    const int entries = (int)cap;
    const cr::NodeDesc hnd{mult1, mult2, 0};
    DevBuf<cr::NodeDesc> dnd;
    if ((rc = upload(dnd, &hnd, 1, ctx))) return rc;
    if (b->staged) {
        const cr::StagedShape shape = staged_shape(b->n_max, b->m_max);
        rc = launch_stage_node(ctx->stream, 1, (int)m, b->pairs.p, b->coords.p, dw.p, dnd.p, b->xf.p, prm, gamma_weight,
                               b->staged_scores.p, shape);
        if (!rc)
            rc = launch_node_staged(ctx->stream, 1, entries, b->pairs.p, b->coords.p, b->tensors.p, (int)d, dw.p, dnd.p, b->xf.p, prm,
                                    b->staged_scores.p, shape, b->bits.p, b->aln.p, dxn.p, dtn.p, dwn.p, dout.p);
    } else {
        rc = b->team ? launch_node_team(R, ctx->stream, 1, (int)n, (int)m, entries, b->pairs.p, b->coords.p, b->tensors.p, (int)d,
                                        dw.p, dnd.p, b->xf.p, prm, gamma_weight, b->bits.p, b->hand.p, b->aln.p, dxn.p, dtn.p,
                                        dwn.p, dout.p)
                     : launch_node(R, ctx->stream, 1, (int)n, (int)m, entries, b->pairs.p, b->coords.p, b->tensors.p, (int)d, dw.p,
                                   dnd.p, b->xf.p, prm, gamma_weight, b->bits.p, b->hand.p, b->aln.p, dxn.p, dtn.p, dwn.p, dout.p);
    }
    if (rc) return rc;
    CR_HIP(hipGetLastError());
    cr::NodeOut no;
    CR_DOWNLOAD(ctx, &no, dout.p, sizeof(no));
    CR_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int32_t> ha((size_t)(2 * cap));
    CR_DOWNLOAD_WAIT(ctx, ha.data(), b->aln.p, sizeof(int32_t) * (size_t)(2 * cap));
    for (int x = 0; x < no.len; x++) {
        aln1[x] = ha[(size_t)(no.first + x)];
        aln2[x] = ha[(size_t)(cap + no.first + x)];
    }
    CR_DOWNLOAD_WAIT(ctx, coords_out, dxn.p + (size_t)no.first * 3, sizeof(double) * (size_t)no.len * 3);
    CR_DOWNLOAD_WAIT(ctx, tensors_out, dtn.p + (size_t)no.first * d, sizeof(double) * (size_t)no.len * d);
    CR_DOWNLOAD_WAIT(ctx, weights_out, dwn.p + no.first, sizeof(double) * (size_t)no.len);
    *aln_len = no.len;
    if (flags) *flags = no.flags;
    return CR_OK;
}

int cr_dtw_align(cr_context* ctx, const int64_t* seq1, int64_t n, const int64_t* seq2, int64_t m, const double* S,
                 int64_t s_rows, int64_t s_cols, double gap_open, double gap_extend, int64_t* aln1, int64_t* aln2,
                 int64_t* aln_len, double* score) {
    ExplicitRun r;
    cr::SweepParams prm{0.0, gap_open, gap_extend};
    int rc = run_explicit<cr::kDtw>(ctx, seq1, n, seq2, m, S, s_rows, s_cols, prm, r, aln1 && aln2);
    if (rc) return rc;
    cr::AlignEnd e;
    if (aln1 && aln2 && !r.walked) {
        CR_LAUNCH(cr::k_dtw_trace_full, dim3(1), dim3(1), 0, ctx->stream, (int)n, (int)m, r.r, r.bits.p,
                           r.end.p, r.aln.p, r.tout.p);
        CR_HIP(hipGetLastError());
    }
    CR_DOWNLOAD(ctx, &e, r.end.p, sizeof(e));
    CR_HIP(hipStreamSynchronize(ctx->stream));
    if (score) *score = e.dtw_score;
    if (aln1 && aln2) return fetch_alignment(ctx, r, n, m, aln1, aln2, aln_len);
    return CR_OK;
}

int cr_smith_waterman_score(cr_context* ctx, const int64_t* seq1, int64_t n, const int64_t* seq2, int64_t m,
                            const double* S, int64_t s_rows, int64_t s_cols, double gap, double* score) {
    CR_REQUIRE(score != nullptr && seq2 != nullptr, "null argument");
    // dynamic_time_warping.py:214-215: a row stops at the first -1 of seq2; later columns stay 0
    int64_t m_eff = m;
    for (int64_t x = 0; x < m; x++)
        if (seq2[x] == -1) { m_eff = x; break; }
    if (m_eff == 0 || n == 0) { *score = 0.0; return CR_OK; }
    ExplicitRun r;
    cr::SweepParams prm{gap, 0.0, 0.0};
    int rc = run_explicit<cr::kSwScore>(ctx, seq1, n, seq2, m_eff, S, s_rows, s_cols, prm, r);
    if (rc) return rc;
    cr::AlignEnd e;
    CR_DOWNLOAD(ctx, &e, r.end.p, sizeof(e));
    CR_HIP(hipStreamSynchronize(ctx->stream));
    *score = e.sw;
    return CR_OK;
}

int cr_smith_waterman(cr_context* ctx, const int64_t* seq1, int64_t n, const int64_t* seq2, int64_t m, const double* S,
                      int64_t s_rows, int64_t s_cols, double gap, int64_t* aln1, int64_t* aln2, int64_t* aln_len,
                      double* score, int* all_zero) {
    CR_REQUIRE(aln1 && aln2 && aln_len && score, "null output");
    ExplicitRun r;
    cr::SweepParams prm{gap, 0.0, 0.0};
    int rc = run_explicit<cr::kSwTrace>(ctx, seq1, n, seq2, m, S, s_rows, s_cols, prm, r);
    if (rc) return rc;
    CR_LAUNCH(cr::k_sw_trace_full, dim3(1), dim3(1), 0, ctx->stream, (int)n, (int)m, r.r, r.dirs.p,
                       r.seed.p, r.aln.p, r.tout.p);
    CR_HIP(hipGetLastError());
    cr::SeedMax sm;
    CR_DOWNLOAD(ctx, &sm, r.seed.p, sizeof(sm));
    CR_HIP(hipStreamSynchronize(ctx->stream));
    *score = sm.score;
    if (all_zero) *all_zero = (sm.i == 0) ? 1 : 0;
    return fetch_alignment(ctx, r, n, m, aln1, aln2, aln_len);
}

// The single-call Kabsch / RMSD / TM drop-ins work on a 3 x 3 problem fed by sequential sums: one launch and two
// copies (0.1 ms) for what the reference's numba does in microseconds.  Up to this many positions they run the SAME
// CR_HD code on the host (bit-identical: FP64 +, -, *, /, sqrt are correctly rounded on both sides, the library is
// built without FMA contraction); CARETTA_HOST_SMALL_K=0 sends everything to the kernels (the parity test does).
static bool small_on_host(int64_t k) {
    return k <= (int64_t)g_cfg.host_small_k;
}

int cr_paired_svd_superpose(cr_context* ctx, const double* x1, const double* x2, int64_t k, double* R, double* t) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(x1 && x2 && R && t && k >= 1 && k < (1 << 28), "bad argument");
    if (small_on_host(k)) {
        double c1[3], c2[3];
        cr::kabsch_seq(x1, x2, (int)k, R, t, c1, c2);
        return CR_OK;
    }
    DevBuf<double> d1, d2, out;
    if ((rc = upload(d1, x1, (size_t)k * 3, ctx))) return rc;
    if ((rc = upload(d2, x2, (size_t)k * 3, ctx))) return rc;
    CR_HIP(out.ensure(18));
    CR_LAUNCH(cr::k_kabsch, dim3(1), dim3(1), 0, ctx->stream, d1.p, d2.p, (int)k, out.p);
    CR_HIP(hipGetLastError());
    double h[18];
    CR_DOWNLOAD(ctx, h, out.p, sizeof(h));
    CR_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(R, h, sizeof(double) * 9);
    std::memcpy(t, h + 9, sizeof(double) * 3);
    return CR_OK;
}

int cr_paired_svd_superpose_with_subset(cr_context* ctx, const double* c1, int64_t n, const double* c2, int64_t m,
                                        const double* s1, const double* s2, int64_t k, double* o1, double* o2,
                                        double* o3) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(c1 && c2 && s1 && s2 && o1 && o2 && k >= 1 && n >= 1 && m >= 1, "bad argument");
    DevBuf<double> dc1, dc2, ds1, ds2, kab, r1, r2, r3;
    if ((rc = upload(dc1, c1, (size_t)n * 3, ctx))) return rc;
    if ((rc = upload(dc2, c2, (size_t)m * 3, ctx))) return rc;
    if ((rc = upload(ds1, s1, (size_t)k * 3, ctx))) return rc;
    if ((rc = upload(ds2, s2, (size_t)k * 3, ctx))) return rc;
    CR_HIP(kab.ensure(18));
    CR_HIP(r1.ensure((size_t)n * 3));
    CR_HIP(r2.ensure((size_t)m * 3));
    CR_HIP(r3.ensure((size_t)k * 3));
    CR_LAUNCH(cr::k_kabsch, dim3(1), dim3(1), 0, ctx->stream, ds1.p, ds2.p, (int)k, kab.p);
    const int th = 256;
    CR_LAUNCH(cr::k_transform, dim3((unsigned)((n + th - 1) / th)), dim3(th), 0, ctx->stream, dc1.p, (int)n,
                       (const double*)nullptr, (const double*)nullptr, kab.p + 12, r1.p);
    CR_LAUNCH(cr::k_transform, dim3((unsigned)((m + th - 1) / th)), dim3(th), 0, ctx->stream, dc2.p, (int)m,
                       kab.p, (const double*)nullptr, kab.p + 15, r2.p);
    CR_LAUNCH(cr::k_transform, dim3((unsigned)((k + th - 1) / th)), dim3(th), 0, ctx->stream, ds2.p, (int)k,
                       kab.p, kab.p + 9, (const double*)nullptr, r3.p);
    CR_HIP(hipGetLastError());
    CR_DOWNLOAD(ctx, o1, r1.p, sizeof(double) * (size_t)n * 3);
    CR_DOWNLOAD(ctx, o2, r2.p, sizeof(double) * (size_t)m * 3);
    if (o3) CR_DOWNLOAD(ctx, o3, r3.p, sizeof(double) * (size_t)k * 3);
    CR_HIP(hipStreamSynchronize(ctx->stream));
    return CR_OK;
}

int cr_apply_rotran(cr_context* ctx, const double* x, int64_t k, const double* R, const double* t, double* out) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(x && R && t && out && k >= 0, "bad argument");
    if (k == 0) return CR_OK;
    DevBuf<double> dx, dr, dout;
    double rt[12];
    std::memcpy(rt, R, sizeof(double) * 9);
    std::memcpy(rt + 9, t, sizeof(double) * 3);
    if ((rc = upload(dx, x, (size_t)k * 3, ctx))) return rc;
    if ((rc = upload(dr, rt, 12, ctx))) return rc;
    CR_HIP(dout.ensure((size_t)k * 3));
    const int th = 256;
    CR_LAUNCH(cr::k_transform, dim3((unsigned)((k + th - 1) / th)), dim3(th), 0, ctx->stream, dx.p, (int)k, dr.p,
                       dr.p + 9, (const double*)nullptr, dout.p);
    CR_HIP(hipGetLastError());
    CR_DOWNLOAD(ctx, out, dout.p, sizeof(double) * (size_t)k * 3);
    CR_HIP(hipStreamSynchronize(ctx->stream));
    return CR_OK;
}

static int rmsd_tm(cr_context* ctx, const double* x1, const double* x2, int64_t k, int64_t l1, int64_t l2, double* h) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(x1 && x2 && k >= 1, "bad argument");
    if (small_on_host(k)) {
        cr::rmsd_tm_seq(x1, x2, (int)k, l1, l2, h);
        return CR_OK;
    }
    DevBuf<double> d1, d2, out;
    if ((rc = upload(d1, x1, (size_t)k * 3, ctx))) return rc;
    if ((rc = upload(d2, x2, (size_t)k * 3, ctx))) return rc;
    CR_HIP(out.ensure(2));
    CR_LAUNCH(cr::k_rmsd_tm, dim3(1), dim3(1), 0, ctx->stream, d1.p, d2.p, (int)k, l1, l2, out.p);
    CR_HIP(hipGetLastError());
    CR_DOWNLOAD(ctx, h, out.p, sizeof(double) * 2);
    CR_HIP(hipStreamSynchronize(ctx->stream));
    return CR_OK;
}

int cr_get_rmsd(cr_context* ctx, const double* x1, const double* x2, int64_t k, double* out) {
    CR_REQUIRE(out != nullptr, "null output");
    double h[2];
    int rc = rmsd_tm(ctx, x1, x2, k, 100, 100, h);
    if (rc == CR_OK) *out = h[0];
    return rc;
}

int cr_tm_score(cr_context* ctx, const double* x1, const double* x2, int64_t k, int64_t l1, int64_t l2, double* out) {
    CR_REQUIRE(out != nullptr, "null output");
    double h[2];
    int rc = rmsd_tm(ctx, x1, x2, k, l1, l2, h);
    if (rc == CR_OK) *out = h[1];
    return rc;
}

int cr_msa_metrics(cr_context* ctx, const double* coords, const int64_t* offsets, int64_t P, const int32_t* msa,
                   int64_t W, int superpose, double* rmsd, double* coverage, double* tm) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(coords && offsets && msa && rmsd && coverage && tm, "null argument");
    CR_REQUIRE(P >= 1 && W >= 1 && W <= 65534 && P < 46000, "bad alignment shape");
    const int64_t total = offsets[P];
    CR_REQUIRE(all_finite(coords, (size_t)total * 3), "coordinates contain NaN or infinity");
    for (int64_t s = 0; s < P; s++) {
        const int64_t len = offsets[s + 1] - offsets[s];
        CR_REQUIRE(len >= 1 && len <= cr::kMaxLength, "bad structure length");
        for (int64_t x = 0; x < W; x++)
            CR_REQUIRE(msa[s * W + x] >= -1 && msa[s * W + x] < len, "alignment index outside its structure");
    }
    for (int64_t x = 0; x < P * P; x++) {
        rmsd[x] = 0.0;
        coverage[x] = 1.0;
        tm[x] = 1.0;
    }
    const int64_t npairs = P * (P - 1) / 2;
    if (npairs == 0) return CR_OK;
    std::vector<int32_t> pairs((size_t)npairs * 2);
    size_t q = 0;
    for (int64_t i = 0; i < P - 1; i++)
        for (int64_t j = i + 1; j < P; j++) {
            pairs[q++] = (int32_t)i;
            pairs[q++] = (int32_t)j;
        }
    DevBuf<double> dc, dout;
    DevBuf<int64_t> doff;
    DevBuf<int32_t> dmsa, dpairs;
    if ((rc = upload(dc, coords, (size_t)total * 3, ctx))) return rc;
    if ((rc = upload(doff, offsets, (size_t)P + 1, ctx))) return rc;
    if ((rc = upload(dmsa, msa, (size_t)P * W, ctx))) return rc;
    if ((rc = upload(dpairs, pairs.data(), pairs.size(), ctx))) return rc;
    CR_HIP(dout.ensure((size_t)npairs * 4));
    const size_t lds = sizeof(double) * (((size_t)W + 3) / 4 * 2 + (size_t)cr::kWave * cr::kMaxAcc);
    if ((rc = allow_lds(cr::k_msa_metrics, lds))) return rc;
    CR_LAUNCH(cr::k_msa_metrics, dim3((unsigned)npairs), dim3(cr::kWave), lds, ctx->stream, dc.p, doff.p, dmsa.p,
                       (int)P, (int)W, superpose, dpairs.p, dout.p);
    CR_HIP(hipGetLastError());
    std::vector<double> h((size_t)npairs * 4);
    CR_DOWNLOAD(ctx, h.data(), dout.p, sizeof(double) * h.size());
    CR_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t p = 0; p < npairs; p++) {
        const int64_t i = pairs[(size_t)2 * p], j = pairs[(size_t)2 * p + 1];
        const bool ok = h[(size_t)4 * p + 3] >= 3.0;
        const double nan = std::numeric_limits<double>::quiet_NaN();
        rmsd[i * P + j] = rmsd[j * P + i] = ok ? h[(size_t)4 * p] : nan;
        coverage[i * P + j] = coverage[j * P + i] = h[(size_t)4 * p + 1];
        tm[i * P + j] = tm[j * P + i] = ok ? h[(size_t)4 * p + 2] : nan;
    }
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------
// host-side integer / tree work
// ---------------------------------------------------------------------------------------------
int cr_mean_axis0(const double* x, int64_t rows, int64_t cols, double* out) {
    CR_REQUIRE(x && out && rows >= 1 && cols >= 1, "bad argument");
    for (int64_t c = 0; c < cols; c++) {
        double s = 0.0;
        for (int64_t r = 0; r < rows; r++) s += x[r * cols + c];
        out[c] = s / (double)rows;
    }
    return CR_OK;
}

int cr_superpose_core(cr_context* ctx, const double* coords, const int64_t* offsets, int64_t P, const int32_t* msa,
                      int64_t W, const int32_t* core, int64_t ncore, int64_t ref, double* coords_out) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(coords && offsets && msa && core && coords_out, "null argument");
    CR_REQUIRE(P >= 1 && W >= 1 && W <= 65534 && ref >= 0 && ref < P, "bad alignment shape");
    CR_REQUIRE(ncore >= 1 && ncore <= 65534, "superpose_core needs at least one gap-free column");
    const int64_t total = offsets[P];
    CR_REQUIRE(all_finite(coords, (size_t)total * 3), "coordinates contain NaN or infinity");
    for (int64_t e = 0; e < ncore; e++) {
        CR_REQUIRE(core[e] >= 0 && core[e] < W, "core column outside the alignment");
        for (int64_t s = 0; s < P; s++) {
            const int32_t r = msa[s * W + core[e]];
            CR_REQUIRE(r >= 0 && r < offsets[s + 1] - offsets[s], "a core column holds a gap or an index outside its structure");
        }
    }
    DevBuf<double> dc, dout, dx1, dcen;
    DevBuf<int64_t> doff;
    DevBuf<int32_t> dmsa, dcore;
    if ((rc = upload(dc, coords, (size_t)total * 3, ctx))) return rc;
    if ((rc = upload(doff, offsets, (size_t)P + 1, ctx))) return rc;
    if ((rc = upload(dmsa, msa, (size_t)P * W, ctx))) return rc;
    if ((rc = upload(dcore, core, (size_t)ncore, ctx))) return rc;
    CR_HIP(dout.ensure((size_t)total * 3));
    CR_HIP(dx1.ensure((size_t)ncore * 3));
    CR_HIP(dcen.ensure(3));
    const size_t lds1 = sizeof(double) * (size_t)cr::kWave * 3;
    CR_LAUNCH(cr::k_core_reference, dim3(1), dim3(cr::kWave), lds1, ctx->stream, dc.p + offsets[ref] * 3,
                       dmsa.p + ref * W, dcore.p, (int)ncore, dx1.p, dcen.p);
    CR_HIP(hipGetLastError());
    const size_t lds2 = sizeof(double) * (((size_t)ncore + 3) / 4 * 2 + (size_t)cr::kWave * cr::kMaxAcc);
    if ((rc = allow_lds(cr::k_core_superpose, lds2))) return rc;
    CR_LAUNCH(cr::k_core_superpose, dim3((unsigned)P), dim3(cr::kWave), lds2, ctx->stream, dc.p, doff.p, dmsa.p, (int)W,
                       dcore.p, (int)ncore, (int)ref, dx1.p, dcen.p, dout.p);
    CR_HIP(hipGetLastError());
    CR_DOWNLOAD(ctx, coords_out, dout.p, sizeof(double) * (size_t)total * 3);
    CR_HIP(hipStreamSynchronize(ctx->stream));
    return CR_OK;
}

int cr_superpose_reference(cr_context* ctx, const double* coords, const int64_t* offsets, int64_t P, const int32_t* msa,
                           int64_t W, int64_t ref, double* coords_out) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(coords && offsets && msa && coords_out, "null argument");
    CR_REQUIRE(P >= 1 && W >= 1 && W <= 65534 && ref >= 0 && ref < P, "bad alignment shape");
    const int64_t total = offsets[P];
    CR_REQUIRE(all_finite(coords, (size_t)total * 3), "coordinates contain NaN or infinity");
    for (int64_t s = 0; s < P; s++)
        for (int64_t x = 0; x < W; x++)
            CR_REQUIRE(msa[s * W + x] >= -1 && msa[s * W + x] < offsets[s + 1] - offsets[s], "alignment index outside its structure");
    DevBuf<double> dc, dout;
    DevBuf<int64_t> doff;
    DevBuf<int32_t> dmsa, dwhich, dcounts;
    // list order of the reference's loop: structures before the reference, the reference itself, the rest
    std::vector<int32_t> which((size_t)P);
    for (int64_t s = 0; s < P; s++) which[(size_t)s] = (int32_t)s;
    if ((rc = upload(dc, coords, (size_t)total * 3, ctx))) return rc;
    if ((rc = upload(doff, offsets, (size_t)P + 1, ctx))) return rc;
    if ((rc = upload(dmsa, msa, (size_t)P * W, ctx))) return rc;
    if ((rc = upload(dwhich, which.data(), (size_t)P, ctx))) return rc;
    CR_HIP(dout.ensure((size_t)total * 3));
    CR_HIP(dcounts.ensure((size_t)P));
    const size_t lds = sizeof(double) * (((size_t)W + 3) / 4 * 2 + (size_t)cr::kWave * cr::kMaxAcc);
    if ((rc = allow_lds(cr::k_reference_superpose, lds))) return rc;
    auto launch = [&](int64_t first, int64_t count, const double* ref_coords) {
        if (count <= 0) return;
        CR_LAUNCH(cr::k_reference_superpose, dim3((unsigned)count), dim3(cr::kWave), lds, ctx->stream, dc.p, doff.p, dmsa.p,
                           (int)W, (int)ref, ref_coords, dwhich.p + first, dout.p, dcounts.p + first);
    };
    launch(0, ref, dc.p + offsets[ref] * 3);                       // before the reference: its original coordinates
    launch(ref, 1, dc.p + offsets[ref] * 3);                       // the reference onto itself (:966-968)
    launch(ref + 1, P - ref - 1, dout.p + offsets[ref] * 3);       // after it: the refitted reference
    CR_HIP(hipGetLastError());
    std::vector<int32_t> counts((size_t)P);
    CR_DOWNLOAD(ctx, counts.data(), dcounts.p, sizeof(int32_t) * (size_t)P);
    CR_DOWNLOAD(ctx, coords_out, dout.p, sizeof(double) * (size_t)total * 3);
    CR_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t s = 0; s < P; s++)
        CR_REQUIRE(counts[(size_t)s] > 3, "a structure shares 3 or fewer alignment columns with the reference (reference: assert len(pos_1) > 3)");
    return CR_OK;
}

int cr_superpose_members(cr_context* ctx, double* coords, const int64_t* offsets, int64_t P, const int32_t* msa, int64_t W,
                         int64_t ref, const int32_t* which, int64_t nwhich) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(coords && offsets && msa && (which || nwhich == 0), "null argument");
    CR_REQUIRE(P >= 1 && W >= 1 && W <= 65534 && ref >= 0 && ref < P && nwhich >= 0, "bad alignment shape");
    if (nwhich == 0) return CR_OK;
    const int64_t total = offsets[P];
    CR_REQUIRE(all_finite(coords, (size_t)total * 3), "coordinates contain NaN or infinity");
    for (int64_t b = 0; b < nwhich; b++) CR_REQUIRE(which[b] >= 0 && which[b] < P, "structure index out of range");
    for (int64_t s = 0; s < P; s++)
        for (int64_t x = 0; x < W; x++)
            CR_REQUIRE(msa[s * W + x] >= -1 && msa[s * W + x] < offsets[s + 1] - offsets[s], "alignment index outside its structure");
    DevBuf<double> dc, dout;
    DevBuf<int64_t> doff;
    DevBuf<int32_t> dmsa, dwhich, dcounts;
    if ((rc = upload(dc, coords, (size_t)total * 3, ctx))) return rc;
    if ((rc = upload(doff, offsets, (size_t)P + 1, ctx))) return rc;
    if ((rc = upload(dmsa, msa, (size_t)P * W, ctx))) return rc;
    if ((rc = upload(dwhich, which, (size_t)nwhich, ctx))) return rc;
    CR_HIP(dout.ensure((size_t)total * 3));
    CR_HIP(dcounts.ensure((size_t)nwhich));
    CR_HIP(hipMemcpyAsync(dout.p, dc.p, sizeof(double) * (size_t)total * 3, hipMemcpyDeviceToDevice, ctx->stream));
    const size_t lds = sizeof(double) * (((size_t)W + 3) / 4 * 2 + (size_t)cr::kWave * cr::kMaxAcc);
    if ((rc = allow_lds(cr::k_reference_superpose, lds))) return rc;
    CR_LAUNCH(cr::k_reference_superpose, dim3((unsigned)nwhich), dim3(cr::kWave), lds, ctx->stream, dc.p, doff.p, dmsa.p,
                       (int)W, (int)ref, dc.p + offsets[ref] * 3, dwhich.p, dout.p, dcounts.p);
    CR_HIP(hipGetLastError());
    std::vector<int32_t> counts((size_t)nwhich);
    std::vector<double> moved((size_t)total * 3);
    CR_DOWNLOAD(ctx, counts.data(), dcounts.p, sizeof(int32_t) * (size_t)nwhich);
    CR_DOWNLOAD(ctx, moved.data(), dout.p, sizeof(double) * (size_t)total * 3);
    CR_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t b = 0; b < nwhich; b++)
        CR_REQUIRE(counts[(size_t)b] > 3, "a structure shares 3 or fewer alignment columns with the reference (reference: assert len(pos_1) > 3)");
    std::memcpy(coords, moved.data(), sizeof(double) * (size_t)total * 3);
    return CR_OK;
}

int cr_get_common_positions(const int64_t* a1, const int64_t* a2, int64_t len, int64_t* p1, int64_t* p2, int64_t* k) {
    CR_REQUIRE(len >= 0 && (len == 0 || (a1 && a2 && p1 && p2)) && k, "bad argument");
    int64_t c = 0;
    for (int64_t x = 0; x < len; x++)
        if (a1[x] != -1 && a2[x] != -1) {
            p1[c] = a1[x];
            p2[c] = a2[x];
            c++;
        }
    *k = c;
    return CR_OK;
}

int cr_assemble_matrix(const int32_t* pairs, const double* scores, int64_t npairs, int64_t P, double* M) {
    CR_REQUIRE(pairs && scores && M && P >= 1 && npairs >= 0, "bad argument");
    std::fill(M, M + P * P, 0.0);
    for (int64_t p = 0; p < npairs; p++) {
        int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
        CR_REQUIRE(i >= 0 && i < P && j >= 0 && j < P, "pair index out of range");
        M[i * P + j] = M[j * P + i] = scores[p];
    }
    return CR_OK;
}

}  // extern "C"

namespace {

// Helper threads for the two row-parallel passes of neighbor joining (row sums, Q search) on large matrices.
// Every row is still summed / searched by ONE thread in the reference's order, so results do not depend on the
// thread count.  run(f) executes f(t) for t = 0 .. threads-1 (t = 0 on the caller); helpers poll for the next
// pass (passes of one call are microseconds apart) and yield between polls.
struct NjTeam {
    int threads = 1;
    std::vector<std::thread> helpers;
    std::atomic<int64_t> generation{0};
    std::atomic<int> done{0};
    std::atomic<bool> quit{false};
    const std::function<void(int)>* job = nullptr;
    explicit NjTeam(int t) : threads(t) {
        for (int x = 1; x < threads; x++) helpers.emplace_back([this, x] { work(x); });
    }
    ~NjTeam() { stop(); }
    void work(int t) {
        int64_t seen = 0;
        for (;;) {
            int spins = 0;
            while (generation.load(std::memory_order_acquire) == seen) {
                if (quit.load(std::memory_order_acquire)) return;
                if (++spins > 256) {
                    std::this_thread::yield();
                    spins = 0;
                } else {
                    __builtin_ia32_pause();
                }
            }
            seen++;
            (*job)(t);
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void run(const std::function<void(int)>& f) {
        job = &f;
        done.store(0, std::memory_order_relaxed);
        generation.fetch_add(1, std::memory_order_release);
        f(0);
        const int need = (int)helpers.size();
        int spins = 0;
        while (done.load(std::memory_order_acquire) < need) {
            if (++spins > 256) {
                std::this_thread::yield();
                spins = 0;
            } else {
                __builtin_ia32_pause();
            }
        }
    }
    void stop() {
        if (helpers.empty()) return;
        quit.store(true, std::memory_order_release);
        for (std::thread& h : helpers) h.join();
        helpers.clear();
        threads = 1;
    }
};

// CPUs this process may really use: the cgroup quota (cpu.max) when there is one -- spinning helpers beyond the
// quota get the whole process throttled --, else the affinity mask.  CARETTA_NJ_THREADS overrides.
int nj_threads() {
    if (g_cfg.nj_threads >= 1) return std::min(g_cfg.nj_threads, 64);
    int cpus = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = std::min(cpus, CPU_COUNT(&set));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0)
            cpus = std::min<long long>(cpus, std::max<long long>(1, std::atoll(quota) / period));
        std::fclose(f);
    }
    // half of what is there, at most 16: the passes are memory bound well before that
    return std::max(1, std::min(16, cpus / 2));
}

constexpr int64_t kNjParallelNodes = 640;      // below this one thread is as fast (the matrix sits in its L2)

}  // namespace

extern "C" {

// neighbor_joining.py:19-157.  The reference recomputes both row sums for every (i, j) (O(P^4));
// here each row sum is formed once per iteration with the same sequential left-to-right order
// (numba's np.sum), so every Q value, hence every decision, is bit-identical to the reference's.
//
// Layout.  The reference rebuilds the matrix every iteration with the new node FIRST and the survivors behind
// it in their old order (:60-83); that order decides ties.  Here the matrix keeps a fixed row stride between
// compactions: slots are never moved, a new node takes the next free slot of a LEFT margin (so slot order is
// the reference's order), and the two joined slots are zeroed.  A zero adds nothing to a left-to-right sum
// (the sums start at +0.0, so they are never -0.0) and a dead column gets the row-sum -inf, i.e. Q = +inf.
// When the margin is used up the live slots are packed again (once per ~n/16 iterations).
int cr_neighbor_joining(const double* D0, int64_t P, uint64_t* tree, double* bl) {
    CR_REQUIRE(D0 && tree && bl, "null argument");
    CR_REQUIRE(P >= 3, "neighbor joining needs at least 3 taxa");
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> A, B, rs, rsx;
    std::vector<int64_t> ident, ident2;          // node id of every slot
    std::vector<char> alive, alive2;
    int64_t W = 0, left = -1, n = P;            // row stride, next free margin slot, live slots
    // pack the live slots of (src, sw) in order behind a fresh margin
    auto compact = [&](const double* src, int64_t sw, const std::vector<int64_t>& slots, const std::vector<int64_t>& ids) {
        const int64_t cnt = (int64_t)slots.size(), g = std::max<int64_t>(8, cnt / 16), w = cnt + g;
        B.assign((size_t)(w * w), 0.0);
        for (int64_t a = 0; a < cnt; a++) {
            const double* r = src + slots[(size_t)a] * sw;
            double* d = &B[(size_t)((g + a) * w + g)];
            for (int64_t b = 0; b < cnt; b++) d[b] = r[slots[(size_t)b]];
        }
        A.swap(B);
        ident2.assign((size_t)w, -1);
        alive2.assign((size_t)w, 0);
        for (int64_t a = 0; a < cnt; a++) {
            ident2[(size_t)(g + a)] = ids[(size_t)a];
            alive2[(size_t)(g + a)] = 1;
        }
        ident.swap(ident2);
        alive.swap(alive2);
        W = w;
        left = g - 1;
        rs.assign((size_t)w, 0.0);
        rsx.assign((size_t)w, 0.0);
    };
    {
        std::vector<int64_t> slots((size_t)P), ids((size_t)P);
        for (int64_t i = 0; i < P; i++) slots[(size_t)i] = ids[(size_t)i] = i;
        compact(D0, P, slots, ids);
    }
    // Row sums of 8 rows at a time: each row is still summed left to right, but the 8 add chains are
    // independent, so the loop runs at the adder's throughput, not its latency.
    auto rowsums = [&](int64_t lo, int64_t row_begin, int64_t row_end) {
        const int64_t cnt = W - lo;
        int64_t i = row_begin;
        for (; i + 8 <= row_end; i += 8) {
            const double* r = &A[(size_t)(i * W + lo)];
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
            for (int64_t k = 0; k < cnt; k++) {
                s0 += r[k];
                s1 += r[W + k];
                s2 += r[2 * W + k];
                s3 += r[3 * W + k];
                s4 += r[4 * W + k];
                s5 += r[5 * W + k];
                s6 += r[6 * W + k];
                s7 += r[7 * W + k];
            }
            rs[(size_t)i] = s0; rs[(size_t)i + 1] = s1; rs[(size_t)i + 2] = s2; rs[(size_t)i + 3] = s3;
            rs[(size_t)i + 4] = s4; rs[(size_t)i + 5] = s5; rs[(size_t)i + 6] = s6; rs[(size_t)i + 7] = s7;
        }
        for (; i < row_end; i++) {
            const double* r = &A[(size_t)(i * W + lo)];
            double s = 0.0;
            for (int64_t k = 0; k < cnt; k++) s += r[k];
            rs[(size_t)i] = s;
        }
    };
    typedef double v4 __attribute__((ext_vector_type(4)));
    NjTeam team(P >= kNjParallelNodes ? nj_threads() : 1);
    int64_t index = 0, nint = 0;
    while (n > 3) {
        if (left < 0) {
            std::vector<int64_t> slots, ids;
            for (int64_t x = 0; x < W; x++)
                if (alive[(size_t)x]) {
                    slots.push_back(x);
                    ids.push_back(ident[(size_t)x]);
                }
            std::vector<double> old;
            old.swap(A);
            const int64_t sw = W;
            compact(old.data(), sw, slots, ids);
        }
        const int64_t lo = left + 1;
        const double nm2 = (double)(n - 2);
        struct Best {
            double q;
            int64_t i, j;
        };
        // first minimum of Q in row-major order (neighbor_joining.py:98-121, strict <) over the rows [row_begin, row_end)
        auto search = [&](int64_t row_begin, int64_t row_end) {
            double min_q = inf;
            int64_t mi = 0, mj = 0;
            for (int64_t i = row_begin; i < row_end; i++) {
                if (!alive[(size_t)i]) continue;
                const double* r = &A[(size_t)(i * W)];
                const double ri = rs[(size_t)i];
                auto qv = [&](int64_t j) { return (nm2 * r[j] - ri) - rsx[(size_t)j]; };
                double m = inf;
                v4 mv = {inf, inf, inf, inf};
                int64_t j = lo;
                for (; j + 4 <= W; j += 4) {
                    if (i >= j && i < j + 4) {                       // the block holding the diagonal: i != j
                        for (int64_t c = j; c < j + 4; c++)
                            if (c != i) {
                                const double q = qv(c);
                                m = q < m ? q : m;
                            }
                        continue;
                    }
                    v4 rv, sv;
                    std::memcpy(&rv, r + j, sizeof(rv));
                    std::memcpy(&sv, &rsx[(size_t)j], sizeof(sv));
                    const v4 q = (rv * nm2 - ri) - sv;
                    mv = q < mv ? q : mv;
                }
                for (; j < W; j++)
                    if (j != i) {
                        const double q = qv(j);
                        m = q < m ? q : m;
                    }
                for (int c = 0; c < 4; c++) m = mv[c] < m ? mv[c] : m;
                if (m < min_q) {
                    int64_t c = lo;
                    while (c == i || qv(c) != m) c++;
                    mi = i;
                    mj = c;
                    min_q = m;
                }
            }
            return Best{min_q, mi, mj};
        };
        double min_q = inf;
        int64_t mi = 0, mj = 0;
        if (team.threads > 1 && n >= kNjParallelNodes) {
            // contiguous row ranges, one per thread; the per-range minima combined in range order with a strict <
            // give the first minimum in row-major order again
            const int T = team.threads;
            const int64_t chunk = (((W - lo) + T - 1) / T + 7) / 8 * 8;
            auto range = [&](int t, int64_t& b0, int64_t& b1) {
                b0 = std::min(W, lo + t * chunk);
                b1 = std::min(W, b0 + chunk);
            };
            const std::function<void(int)> sums = [&](int t) {
                int64_t b0, b1;
                range(t, b0, b1);
                rowsums(lo, b0, b1);
            };
            team.run(sums);
            for (int64_t x = lo; x < W; x++) rsx[(size_t)x] = alive[(size_t)x] ? rs[(size_t)x] : -inf;
            std::vector<Best> best((size_t)T);
            const std::function<void(int)> find = [&](int t) {
                int64_t b0, b1;
                range(t, b0, b1);
                best[(size_t)t] = search(b0, b1);
            };
            team.run(find);
            for (int t = 0; t < T; t++)
                if (best[(size_t)t].q < min_q) {
                    min_q = best[(size_t)t].q;
                    mi = best[(size_t)t].i;
                    mj = best[(size_t)t].j;
                }
        } else {
            team.stop();
            rowsums(lo, lo, W);
            for (int64_t x = lo; x < W; x++) rsx[(size_t)x] = alive[(size_t)x] ? rs[(size_t)x] : -inf;
            const Best bst = search(lo, W);
            min_q = bst.q;
            mi = bst.i;
            mj = bst.j;
        }
        const double dij = A[(size_t)(mi * W + mj)];
        const double di = 0.5 * dij + (0.5 / (double)(n - 2)) * (rs[(size_t)mi] - rs[(size_t)mj]);
        const double dj = dij - di;
        const int64_t node = nint + P;
        nint++;
        tree[2 * index] = (uint64_t)ident[(size_t)mi]; tree[2 * index + 1] = (uint64_t)node; bl[index++] = di;
        tree[2 * index] = (uint64_t)ident[(size_t)mj]; tree[2 * index + 1] = (uint64_t)node; bl[index++] = dj;
        // the new node takes the next margin slot (:60-83)
        const int64_t v = left--;
        double* rv = &A[(size_t)(v * W)];
        const double* rmi = &A[(size_t)(mi * W)];
        const double* rmj = &A[(size_t)(mj * W)];
        for (int64_t k = lo; k < W; k++) {
            if (!alive[(size_t)k] || k == mi || k == mj) continue;
            const double val = 0.5 * ((rmi[k] + rmj[k]) - dij);
            rv[k] = val;
            A[(size_t)(k * W + v)] = val;
        }
        rv[v] = 0.0;
        for (int64_t dead : {mi, mj}) {
            std::fill(&A[(size_t)(dead * W + v)], &A[(size_t)(dead * W + W)], 0.0);
            for (int64_t k = v; k < W; k++) A[(size_t)(k * W + dead)] = 0.0;
            alive[(size_t)dead] = 0;
        }
        alive[(size_t)v] = 1;
        ident[(size_t)v] = node;
        n--;
    }
    // the last three nodes in order (:85-94)
    int64_t s3[3], c3 = 0;
    for (int64_t x = 0; x < W && c3 < 3; x++)
        if (alive[(size_t)x]) s3[c3++] = x;
    auto at = [&](int a, int b) { return A[(size_t)(s3[a] * W + s3[b])]; };
    const double s1 = ((0.0 + at(1, 0)) + at(1, 1)) + at(1, 2), s2 = ((0.0 + at(2, 0)) + at(2, 1)) + at(2, 2);
    const double d12 = at(1, 2);
    const double di = 0.5 * d12 + (0.5 / (double)(n - 2)) * (s1 - s2);
    const int64_t node = nint + P;
    tree[2 * index] = (uint64_t)ident[(size_t)s3[1]]; tree[2 * index + 1] = (uint64_t)node; bl[index++] = di;
    tree[2 * index] = (uint64_t)ident[(size_t)s3[2]]; tree[2 * index + 1] = (uint64_t)node; bl[index++] = d12 - di;
    tree[2 * index] = (uint64_t)ident[(size_t)s3[0]]; tree[2 * index + 1] = (uint64_t)node;
    bl[index++] = 0.5 * ((at(1, 0) + at(2, 0)) - d12);
    return CR_OK;
}

}  // extern "C"
