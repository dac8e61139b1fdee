// Score providers of the sweeps: the RBFs of the pipeline (tensor, coordinates in the frame of a superposition, tree nodes), an explicit score matrix.
// Part of cr_kernels.h (included there, inside namespace cr, in this order: cr_providers.h, cr_sweep.h, cr_sweep_cols.h,
// cr_sweep_wide.h, cr_trace.h, cr_pair_kernels.h); not a header of its own.

// ---------------------------------------------------------------------------------------------
// Score providers.  load_rows(): once per strip, lane-private row data into registers.
// load_chunk(): once per 64 steps, the next 64 columns into the LDS ring.  fetch_col(): once per
// step, this lane's column.  score(q): S(row q of this lane, current column).
// ---------------------------------------------------------------------------------------------

// exp(-gamma * sum_k (a_ik - b_jk)^2), k ascending (score_functions.py:7-11).  D is the padded
// width (zero padding adds exact zeros to the sum); `d` is the stored width.
template <int R, int D>
struct RbfTensor {
    static constexpr bool kNonNegative = true;   // scores are exp(.) >= 0
    const double* __restrict__ rows_g;   // (n, d)
    const double* __restrict__ cols_g;   // (m, d)
    int d;
    double neg_gamma;
    double row[R][D];
    double col[D];
    double col2[D];                       // second set of column features (column sweep with few rows per lane)
    static constexpr int kRingDoubles = D * kRing;
    static constexpr bool kMaskRows = false;

    CR_D void load_rows(int rowbase, int n) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int r = rowbase + q;
            const bool rv = r < n;
#pragma unroll
            for (int k = 0; k < D; k++)
                row[q][k] = (k < d) ? (rv ? rows_g[(int64_t)r * d + k] : kFarAway) : 0.0;
        }
    }
    CR_D void init_ring(double* ring, int lane) {
        for (int x = lane; x < D * kRing; x += kWave) ring[x] = 0.0;
    }
    CR_D void load_chunk(double* ring, int chunk, int m, int lane) {
        const int c0 = chunk * kWave;
        const int total = kWave * d;
        for (int e = lane; e < total; e += kWave) {
            int cc = e / d, k = e - cc * d;
            int c = c0 + cc;
            if (c < m) ring[k * kRing + (c & (kRing - 1))] = cols_g[(int64_t)c * d + k];
        }
    }
    CR_D void fetch_col(const double* ring, int slot) {
#pragma unroll
        for (int k = 0; k < D; k++) col[k] = ring[k * kRing + slot];
    }
    // wide sweep: all m columns resident in LDS, feature-major planes of `stride` doubles (consecutive lanes read
    // consecutive doubles of a plane: conflict-free ds_read_b64)
    static constexpr int kColDoubles = D;
    CR_D void load_resident(double* res, int stride, int m, int tid, int nth) { load_resident_range(res, stride, 0, m, tid, nth); }
    // columns [c0, c1) only, column c at index c - c0 of every plane (the score staging kernels, cr_staged.h)
    CR_D void load_resident_range(double* res, int stride, int c0, int c1, int tid, int nth) {
        const int total = (c1 - c0) * d;
        const double* __restrict__ from = cols_g + (int64_t)c0 * d;
        for (int e = tid; e < total; e += nth) {
            const int c = e / d, k = e - c * d;
            res[k * stride + c] = from[e];
        }
        for (int e = tid; e < (D - d) * stride; e += nth) res[d * stride + e] = 0.0;   // padded features
    }
    CR_D void fetch_resident(const double* res, int stride, int c) {
#pragma unroll
        for (int k = 0; k < D; k++) col[k] = res[k * stride + c];
    }
    // sum_k (a_ik - b_jk)^2, k ascending
    CR_D double dist2(int q) const { return dist2_of(q, col); }
    CR_D double dist2_of(int q, const double (&c)[D]) const {
        double df = row[q][0] - c[0];
        double acc = df * df;
#pragma unroll
        for (int k = 1; k < D; k++) {
            df = row[q][k] - c[k];
            acc = acc + df * df;
        }
        return acc;
    }
    CR_D double score(int q, const ExpEntry* tab) const { return exp_tab<true>(neg_gamma * dist2(q), tab); }
};

// The same score for ANY stored width (the reference takes any (L, d) tensor array, multiple_alignment.py:312-331; the kernels
// above keep a lane's row features in registers and are instantiated for widths padded to at most 32): the staging kernel of
// cr_staged.h only -- columns resident in LDS as d feature planes, a lane's row features read from L1 / L2 per cell, the sum in
// the same order k = 0, 1, ... (no padding: nothing is added).  Rows past n score exactly 0, as the far-away features make them.
template <int R>
struct RbfTensorAny {
    static constexpr bool kNonNegative = true;
    static constexpr bool kMaskRows = false;
    const double* __restrict__ rows_g;   // (n, d)
    const double* __restrict__ cols_g;   // (m, d)
    int d;
    double neg_gamma;
    const double* rowp[R];               // this lane's rows (nullptr: past n)
    const double* colp;                  // this lane's column inside the resident planes
    int stride_;

    CR_D void load_rows(int rowbase, int n) {
#pragma unroll
        for (int q = 0; q < R; q++) rowp[q] = rowbase + q < n ? rows_g + (int64_t)(rowbase + q) * d : nullptr;
    }
    CR_D void load_resident_range(double* res, int stride, int c0, int c1, int tid, int nth) {
        const int total = (c1 - c0) * d;
        const double* __restrict__ from = cols_g + (int64_t)c0 * d;
        for (int e = tid; e < total; e += nth) {
            const int c = e / d, k = e - c * d;
            res[k * stride + c] = from[e];
        }
    }
    CR_D void fetch_resident(const double* res, int stride, int c) {
        colp = res + c;
        stride_ = stride;
    }
    CR_D double score(int q, const ExpEntry* tab) const {
        const double* __restrict__ r = rowp[q];
        if (!r) return 0.0;
        double df = r[0] - colp[0];
        double acc = df * df;
        for (int k = 1; k < d; k++) {
            df = r[k] - colp[(int64_t)k * stride_];
            acc = acc + df * df;
        }
        return exp_tab<true>(neg_gamma * acc, tab);
    }
};

// Coordinate RBF on the seed-superposed frames: rows X_i - c1, columns (X_j - c2) @ R
// (superposition_functions.py:57-58), or the raw coordinates when the seed was skipped.
template <int R>
struct RbfCoords {
    static constexpr bool kNonNegative = true;
    const double* __restrict__ rows_g;   // (n, 3)
    const double* __restrict__ cols_g;   // (m, 3)
    const Transform* __restrict__ xf;
    double neg_gamma;
    double row[R][3];
    double col[3];
    static constexpr int kRingDoubles = 3 * kRing;
    static constexpr bool kMaskRows = false;

    CR_D void load_rows(int rowbase, int n) {
        const bool raw = xf->flags & kFlagSeedSkipped;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int r = rowbase + q;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (r < n) {
                    double v = rows_g[(int64_t)r * 3 + k];
                    row[q][k] = raw ? v : v - xf->c1[k];
                } else {
                    row[q][k] = kFarAway;
                }
            }
        }
    }
    CR_D void init_ring(double*, int) {}
    CR_D void load_chunk(double* ring, int chunk, int m, int lane) {
        int c = chunk * kWave + lane;
        if (c < m) {
            double v[3] = {cols_g[(int64_t)c * 3], cols_g[(int64_t)c * 3 + 1], cols_g[(int64_t)c * 3 + 2]};
            double o[3];
            if (xf->flags & kFlagSeedSkipped) {
                o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
            } else {
                double w[3] = {v[0] - xf->c2[0], v[1] - xf->c2[1], v[2] - xf->c2[2]};
                rot3(w, xf->R, o);
            }
            const int slot = c & (kRing - 1);
            ring[slot] = o[0];
            ring[kRing + slot] = o[1];
            ring[2 * kRing + slot] = o[2];
        }
    }
    CR_D void fetch_col(const double* ring, int slot) {
        col[0] = ring[slot];
        col[1] = ring[kRing + slot];
        col[2] = ring[2 * kRing + slot];
    }
    static constexpr int kColDoubles = 3;
    CR_D void load_resident(double* res, int stride, int m, int tid, int nth) { load_resident_range(res, stride, 0, m, tid, nth); }
    CR_D void load_resident_range(double* res, int stride, int c0, int c1, int tid, int nth) {
        const bool raw = xf->flags & kFlagSeedSkipped;
        for (int c = c0 + tid; c < c1; c += nth) {
            const double v[3] = {cols_g[(int64_t)c * 3], cols_g[(int64_t)c * 3 + 1], cols_g[(int64_t)c * 3 + 2]};
            double o[3];
            if (raw) {
                o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
            } else {
                const double w[3] = {v[0] - xf->c2[0], v[1] - xf->c2[1], v[2] - xf->c2[2]};
                rot3(w, xf->R, o);
            }
            res[c - c0] = o[0];
            res[stride + c - c0] = o[1];
            res[2 * stride + c - c0] = o[2];
        }
    }
    CR_D void fetch_resident(const double* res, int stride, int c) {
        col[0] = res[c];
        col[1] = res[stride + c];
        col[2] = res[2 * stride + c];
    }
    CR_D double score(int q, const ExpEntry* tab) const {
        double dx = row[q][0] - col[0], dy = row[q][1] - col[1], dz = row[q][2] - col[2];
        double acc = (dx * dx + dy * dy) + dz * dz;
        return exp_tab<true>(neg_gamma * acc, tab);
    }
};

// Progressive-alignment node score (multiple_alignment.py:204-210): the coordinate RBF of RbfCoords
// PLUS the RBF of the scaled consensus weights, exp(-gw * (w1[i]*mult1 - w2[j]*mult2)^2).
template <int R>
struct RbfNode {
    static constexpr bool kNonNegative = true;
    RbfCoords<R> xyz;
    const double* __restrict__ w_rows;   // (n) consensus weights of node 1
    const double* __restrict__ w_cols;   // (m) consensus weights of node 2
    double mult1, mult2, neg_gamma_w;
    double wrow[R], wcol;
    static constexpr int kRingDoubles = 4 * kRing;
    static constexpr bool kMaskRows = false;

    CR_D void load_rows(int rowbase, int n) {
        xyz.load_rows(rowbase, n);
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int r = rowbase + q;
            wrow[q] = r < n ? w_rows[r] * mult1 : kFarAway;
        }
    }
    CR_D void init_ring(double*, int) {}
    CR_D void load_chunk(double* ring, int chunk, int m, int lane) {
        xyz.load_chunk(ring, chunk, m, lane);
        const int c = chunk * kWave + lane;
        if (c < m) ring[3 * kRing + (c & (kRing - 1))] = w_cols[c] * mult2;
    }
    CR_D void fetch_col(const double* ring, int slot) {
        xyz.fetch_col(ring, slot);
        wcol = ring[3 * kRing + slot];
    }
    static constexpr int kColDoubles = 4;
    CR_D void load_resident(double* res, int stride, int m, int tid, int nth) { load_resident_range(res, stride, 0, m, tid, nth); }
    CR_D void load_resident_range(double* res, int stride, int c0, int c1, int tid, int nth) {
        xyz.load_resident_range(res, stride, c0, c1, tid, nth);
        for (int c = c0 + tid; c < c1; c += nth) res[3 * stride + c - c0] = w_cols[c] * mult2;
    }
    CR_D void fetch_resident(const double* res, int stride, int c) {
        xyz.fetch_resident(res, stride, c);
        wcol = res[3 * stride + c];
    }
    CR_D double score(int q, const ExpEntry* tab) const {
        const double dw = wrow[q] - wcol;
        return xyz.score(q, tab) + exp_tab<true>(neg_gamma_w * (dw * dw), tab);
    }
};

// The node score of the progressive alignment with flexible=True: Protein.score_function(flexible=True) is the TENSOR score
// matrix alone (multiple_alignment.py:323-326), make_intermediate_node adds the consensus-weight term (:207-210).  Used by
// the score staging kernel of cr_staged.h (columns resident, one plane per feature + one for the weights).
template <int R, int D>
struct RbfFlexNode {
    static constexpr bool kNonNegative = true;
    static constexpr bool kMaskRows = false;
    RbfTensor<R, D> ten;
    const double* __restrict__ w_rows;   // (n) consensus weights of node 1
    const double* __restrict__ w_cols;   // (m) consensus weights of node 2
    double mult1, mult2, neg_gamma_w;
    double wrow[R], wcol;
    static constexpr int kColDoubles = D + 1;

    CR_D void load_rows(int rowbase, int n) {
        ten.load_rows(rowbase, n);
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int r = rowbase + q;
            wrow[q] = r < n ? w_rows[r] * mult1 : kFarAway;
        }
    }
    CR_D void load_resident_range(double* res, int stride, int c0, int c1, int tid, int nth) {
        ten.load_resident_range(res, stride, c0, c1, tid, nth);
        for (int c = c0 + tid; c < c1; c += nth) res[D * stride + c - c0] = w_cols[c] * mult2;
    }
    CR_D void fetch_resident(const double* res, int stride, int c) {
        ten.fetch_resident(res, stride, c);
        wcol = res[D * stride + c];
    }
    CR_D double score(int q, const ExpEntry* tab) const {
        const double dw = wrow[q] - wcol;
        return ten.score(q, tab) + exp_tab<true>(neg_gamma_w * (dw * dw), tab);
    }
};

// Explicit score matrix with index sequences: S[seq1[i], seq2[j]] (dynamic_time_warping.py:24-26,79).
// The strip's 64*R rows x the 128 most recent columns are staged in LDS: every 64 steps all lanes copy the next
// 64 columns of every row of the strip with row-contiguous (coalesced when seq2 is a range) loads, so the sweep
// itself never waits on HBM.  A lane reads tile[(lane*R + q) * kStride + (t - lane) mod 128]; kStride makes
// R * kStride - 1 odd, so the 64 lanes of a step fall into distinct banks.
template <int R>
struct Explicit {
    static constexpr bool kNonNegative = false;
    static constexpr int kStride = kRing + 1 + (R & 1);
    const double* __restrict__ S;
    const int32_t* __restrict__ seq1;
    const int32_t* __restrict__ seq2;
    int64_t s_cols;
    int row0, rows;          // first row of the current strip, number of rows of the matrix
    int lane_;
    int myrow[R];
    double val[R];
    static constexpr int kRingDoubles = kWave * R * kStride + kWave * R / 2 + 1;   // tile + the strip's row indices
    static constexpr bool kMaskRows = true;

    CR_D void load_rows(int rowbase, int n) {
        lane_ = threadIdx.x & (kWave - 1);
        row0 = __builtin_amdgcn_readfirstlane(rowbase - lane_ * R);
        rows = n;
#pragma unroll
        for (int q = 0; q < R; q++) myrow[q] = rowbase + q < n ? seq1[rowbase + q] : 0;   // row indices of this lane
    }
    CR_D void init_ring(double*, int) {}
    CR_D void load_chunk(double* ring, int chunk, int m, int lane) {
        // the strip's row indices go through LDS once (LDS operations of one wave execute in order), so that the
        // copy loop's addresses come from a broadcast ds_read instead of a chain of scalar loads
        int* rowidx = reinterpret_cast<int*>(ring + kWave * R * kStride);
        if (chunk == 0) {
#pragma unroll
            for (int q = 0; q < R; q++) rowidx[lane * R + q] = myrow[q];
        }
        const int c = chunk * kWave + lane;
        const bool cv = c < m;
        const int64_t col = cv ? seq2[c] : 0;
        const int slot = c & (kRing - 1);
        const int left = rows - row0 < kWave * R ? rows - row0 : kWave * R;
        // 16 rows at a time: indices, then 16 loads in flight, then the stores (the tile and the index list are
        // both LDS, so interleaving them would serialise the loads behind the stores)
        for (int base = 0; base < kWave * R; base += 16) {
            int idx[16];
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) idx[k] = rowidx[base + k];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = (cv && base + k < left) ? S[(int64_t)idx[k] * s_cols + col] : 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (cv) ring[(base + k) * kStride + slot] = v[k];   // rows past n: masked in the DP, kept finite
        }
    }
    CR_D void fetch_col(const double* ring, int slot) {
#pragma unroll
        for (int q = 0; q < R; q++) val[q] = ring[(lane_ * R + q) * kStride + slot];
    }
    CR_D double score(int q, const ExpEntry*) const { return val[q]; }
};

// Providers that stream their scores lane by lane (ExplicitStream, cr_explicit_batch.h) get a call at the top of EVERY
// step from every lane, active or not: `static constexpr bool kStreams = true` + `step_begin(ring, t, m)`.
template <class S, class = void>
struct is_streaming : std::false_type {};
template <class S>
struct is_streaming<S, std::void_t<decltype(S::kStreams)>> : std::bool_constant<S::kStreams> {};

// doubles of LDS in front of a sweep's rings: the exp table, for providers that evaluate an RBF (explicit score matrices
// declare `static constexpr bool kNoExp = true` and get the 2 KB back: one more wave per CU for the streaming sweep)
template <class S, class = void>
struct exp_doubles : std::integral_constant<int, kExpDoubles> {};
template <class S>
struct exp_doubles<S, std::void_t<decltype(S::kNoExp)>> : std::integral_constant<int, S::kNoExp ? 0 : kExpDoubles> {};
