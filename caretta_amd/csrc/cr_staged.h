// Launches with few workgroups -- a level of the progressive alignment, a short pair list, a single explicit score matrix:
// the scores are formed (or gathered) by their own launch on every CU of the chip and the sweeps that hold the recurrence
// read them back (cr::sweep_staged in cr_sweep_wide.h).  DESIGN.md section 4.1d.
//
// Why: one tree level is 1 .. P/2 nodes, every node one workgroup, and its levels come one after the other
// (multiple_alignment.py:193-234 needs both children).  In the fused kernels the 4 .. 8 waves of a node form the scores AND
// run the recurrence; a wave that has its SIMD to itself issues an FP64-rate instruction every 5.5 .. 8.75 cycles
// (profiles/r03/valu_latency.txt), so a level took 0.49 ms whether it held 64 nodes or one, and 250 of the chip's 256
// CUs idled.  The score of a cell does not depend on the recurrence: 50 of the 59 instructions of a seed cell (tensor RBF,
// d = 10) and 55 of the ~80 of a node cell (two RBFs) go to a launch that has as many workgroups as there are (pair, 16 .. 64
// step) pieces, and the sweep keeps ~15 / ~25 instructions per cell, with as few rows per lane as its eight waves allow
// (one up to 512 rows: the shortest pipeline; up to four, 2048 rows).
// The arithmetic is the providers' own score() -- the values are the same doubles, so every result is bit-identical to the
// fused kernels (tests/test_gpu_parity.py: test_progressive_staged_scores_equal_fused,
// test_staged_pair_batches_vs_oracle_and_fused, test_staged_pair_batches_two_rows_per_lane, test_dp_multistrip_vs_oracle).
//
// Layout: one pair = `waves` strips of `steps` lines of r * 64 doubles in the consumer's step order; 8 bytes per cell are
// written once and read once (L2 / MALL resident: one level of 64 nodes of 450 columns is 140 MB).
// Included by cr_api.hip in front of its launch code (kernels, then the launchers).
#pragma once

namespace cr {

struct StagedShape {       // the same for every pair of a launch (sized for the launch's length bound)
    int waves;             // strips per pair
    int steps;             // score lines per strip: staged_steps(m bound)
    int r;                 // rows per lane: 1 up to 512 rows, 2 up to 1024, 3, 4 up to 2048 (a line is r sub-lines of 64 doubles)
    CR_HD int64_t strip_doubles() const { return (int64_t)steps * r * kWave; }
    CR_HD int64_t pair_doubles() const { return (int64_t)waves * strip_doubles(); }
};

// One workgroup = the steps [t0, t0 + tc) of every strip of one pair (wave w = strip w = rows 64 R w ..; lane l: R rows) in the layout of
// sweep_staged -- line t holds column t - lane, the workgroup needs the columns [t0 - 63, t0 + tc), which go through LDS
// once for all strips.
template <int R, class Src>
CR_D void stage_block(Src& src, const int n, const int m, const int tc, double* __restrict__ pair_base,
                      const StagedShape shape, double* lds) {
    constexpr int kBack = kWave - 1;
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int t0 = (int)blockIdx.x * tc;
    // the lines of every block of 16 steps a strip runs through, with EXACT ZEROS where a lane's column is outside
    // [0, m) -- sweep_staged runs its ramps without masks on them (cr_sweep_wide.h)
    const int t_end = (m + kBack + kStagedBlock - 1) / kStagedBlock * kStagedBlock;
    if (t0 >= t_end) return;                               // (whole workgroup) past the last step of this pair
    const int c_lo = t0 - kBack > 0 ? t0 - kBack : 0;
    const int c_hi = t0 + tc < m ? t0 + tc : m;
    const int stride = tc + kBack;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* res = lds + kExpDoubles;
    load_exp_table(lds, threadIdx.x);
    src.load_resident_range(res, stride, c_lo, c_hi, (int)threadIdx.x, (int)blockDim.x);
    const bool mine = w * kWave * R < n;
    if (mine) src.load_rows((w * kWave + lane) * R, n);    // rows past n: the far-away features whose score is exactly 0
    __syncthreads();
    if (!mine) return;
    double* __restrict__ out = pair_base + (int64_t)w * shape.strip_doubles() + lane;
    const int t1 = t0 + tc < t_end ? t0 + tc : t_end;
    for (int t = t0; t < t1; t++) {
        const int c = t - lane;
        if ((unsigned)c < (unsigned)m) {
            src.fetch_resident(res, stride, c - c_lo);
#pragma unroll
            for (int q = 0; q < R; q++) out[((int64_t)t * R + q) * kWave] = src.score(q, tab);
        } else {
#pragma unroll
            for (int q = 0; q < R; q++) out[((int64_t)t * R + q) * kWave] = 0.0;
        }
    }
}

__host__ __device__ inline size_t stage_lds_doubles(int col_doubles, int tc) {
    return kExpDoubles + (size_t)col_doubles * (tc + kWave - 1);
}

// tensor RBF of a pair's two structures / a node's two children (multiple_alignment.py:328-335)
template <int D, int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_stage_tensor(const PairDesc* __restrict__ pairs,
                                                                      const double* __restrict__ tensors, int d,
                                                                      double gamma, int tc, double* __restrict__ staged,
                                                                      const StagedShape shape) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.y];
    RbfTensor<R, D> src;
    src.rows_g = tensors + pd.off_i * d;
    src.cols_g = tensors + pd.off_j * d;
    src.d = d;
    src.neg_gamma = -gamma;
    stage_block<R>(src, pd.n, pd.m, tc, staged + (int64_t)blockIdx.y * shape.pair_doubles(), shape, lds);
}

// the same for tensors wider than 32 (RbfTensorAny: any stored width; dynamic LDS = exp table + d planes of the column window)
template <int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_stage_tensor_any(const PairDesc* __restrict__ pairs,
                                                                          const double* __restrict__ tensors, int d,
                                                                          double gamma, int tc, double* __restrict__ staged,
                                                                          const StagedShape shape) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.y];
    RbfTensorAny<R> src;
    src.rows_g = tensors + pd.off_i * d;
    src.cols_g = tensors + pd.off_j * d;
    src.d = d;
    src.neg_gamma = -gamma;
    stage_block<R>(src, pd.n, pd.m, tc, staged + (int64_t)blockIdx.y * shape.pair_doubles(), shape, lds);
}

// node score of the progressive alignment (multiple_alignment.py:204-210) in the frame of the node's seed superposition
template <int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_stage_node(const PairDesc* __restrict__ pairs,
                                                                      const double* __restrict__ coords,
                                                                      const double* __restrict__ weights,
                                                                      const NodeDesc* __restrict__ nodes,
                                                                      const Transform* __restrict__ xfs,
                                                                      double gamma_coords, double gamma_weight, int tc,
                                                                      double* __restrict__ staged, const StagedShape shape) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.y];
    const NodeDesc nd = nodes[blockIdx.y];
    RbfNode<R> src;
    src.xyz.rows_g = coords + pd.off_i * 3;
    src.xyz.cols_g = coords + pd.off_j * 3;
    src.xyz.xf = xfs + blockIdx.y;
    src.xyz.neg_gamma = -gamma_coords;
    src.w_rows = weights + pd.off_i;
    src.w_cols = weights + pd.off_j;
    src.mult1 = nd.mult1;
    src.mult2 = nd.mult2;
    src.neg_gamma_w = -gamma_weight;
    stage_block<R>(src, pd.n, pd.m, tc, staged + (int64_t)blockIdx.y * shape.pair_doubles(), shape, lds);
}

// node score of the progressive alignment with flexible=True (multiple_alignment.py:323-326 + :207-210): tensor RBF + weight RBF
template <int D, int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_stage_flex(const PairDesc* __restrict__ pairs,
                                                                      const double* __restrict__ tensors, int d,
                                                                      const double* __restrict__ weights,
                                                                      const NodeDesc* __restrict__ nodes, double gamma_tensor,
                                                                      double gamma_weight, int tc, double* __restrict__ staged,
                                                                      const StagedShape shape) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.y];
    const NodeDesc nd = nodes[blockIdx.y];
    RbfFlexNode<R, D> src;
    src.ten.rows_g = tensors + pd.off_i * d;
    src.ten.cols_g = tensors + pd.off_j * d;
    src.ten.d = d;
    src.ten.neg_gamma = -gamma_tensor;
    src.w_rows = weights + pd.off_i;
    src.w_cols = weights + pd.off_j;
    src.mult1 = nd.mult1;
    src.mult2 = nd.mult2;
    src.neg_gamma_w = -gamma_weight;
    stage_block<R>(src, pd.n, pd.m, tc, staged + (int64_t)blockIdx.y * shape.pair_doubles(), shape, lds);
}

// coordinate RBF of a pair in the frame of its seed superposition (multiple_alignment.py:158-170, Protein.score_function)
template <int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_stage_coords(const PairDesc* __restrict__ pairs,
                                                                        const double* __restrict__ coords,
                                                                        const Transform* __restrict__ xfs, double gamma,
                                                                        int tc, double* __restrict__ staged,
                                                                        const StagedShape shape) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.y];
    RbfCoords<R> src;
    src.rows_g = coords + pd.off_i * 3;
    src.cols_g = coords + pd.off_j * 3;
    src.xf = xfs + blockIdx.y;
    src.neg_gamma = -gamma;
    stage_block<R>(src, pd.n, pd.m, tc, staged + (int64_t)blockIdx.y * shape.pair_doubles(), shape, lds);
}

// Seed stage on staged scores: SW fill with one wave per strip, then
// traceback (wave 0) + seed Kabsch (the ordered sums by the whole workgroup), as the first half of k_pair_wide.
template <bool ZG, int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_seed_staged(const PairDesc* __restrict__ pairs,
                                                                     const double* __restrict__ coords, double sw_gap,
                                                                     int max_entries, const double* __restrict__ staged,
                                                                     const StagedShape shape, uint32_t* __restrict__ dirs,
                                                                     Transform* __restrict__ xf,
                                                                     double* __restrict__ seed_score) {
    extern __shared__ double lds[];
    CR_STAMP(0);
    const PairDesc pd = pairs[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SeedMax sm;
    AlignEnd unused;
    {
        const double* strip = staged + (int64_t)blockIdx.x * shape.pair_doubles() + (int64_t)w * shape.strip_doubles();
        const StripGeom geom = WidePlan<R>{0}.geom(w, pd.n);
        SweepParams prm{sw_gap, 0.0, 0.0};
        // (until round 5 the gap-0 seed of one or two rows per lane was a column sweep on an unskewed layout; the skewed sweep
        // without masks in its ramps and paced by progress words is faster at every size: 340 x 330 161 k -> 149 k cycles against
        // 196 k, 1 024 x 700 at two rows per lane 456 k against 597 k, tools/step_probe.hip)
        sweep_staged<R, kSwTrace | (ZG ? kZeroGap : 0)>(strip, pd.n, pd.m, prm, lds, dirs + pd.dirs_off, nullptr, sm, unused, geom);
    }
    // wave 0 walks (the others wait at the barrier); the position-ordered sums behind the walk are taken by everybody
    __shared__ int s_walk[4];
    uint32_t* const seed_list = reinterpret_cast<uint32_t*>(lds + kExpDoubles);
    double* const terms = lds + kExpDoubles + ((size_t)max_entries + 3) / 4 * 2;
    if (threadIdx.x < kWave) {
        CR_STAMP(1);
        int k, len;
        uint32_t fl;
        seed_walk<R, 1>(pd, dirs, sm, seed_list, 0, k, len, fl);
        if (threadIdx.x == 0) {
            s_walk[0] = k;
            s_walk[1] = len;
            s_walk[2] = (int)fl;
        }
        CR_STAMP(2);
    }
    __syncthreads();
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
        kabsch_team(coords + pd.off_i * 3, coords + pd.off_j * 3, seed_list + (cap - k), k, k, terms, terms + kSumTile * kMaxAcc + kSumSlack,
                    tr.c1, tr.c2, tr.R, t);
    }
    if (threadIdx.x == 0) {
        xf[blockIdx.x] = tr;
        seed_score[blockIdx.x] = sm.score;
    }
    CR_STAMP(3);
}

// Node stage on staged scores: affine DTW fill with one wave per strip, then the traceback (wave 0) and, by the whole
// workgroup, what node_finish does behind it: the superposition on the aligned positions and the merged node.
// FLEX: flexible=True in score and mean function -- no seed, no superposition, no coordinates: the node is its mean tensors and
// consensus weights (multiple_alignment.py:351-362); `coords`, `xfs` and `Xn_base` are not touched.
template <int R, bool FLEX = false>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_node_staged(const PairDesc* __restrict__ pairs, const double* coords,
                                                                     const double* tensors, int d, const double* weights,
                                                                     const NodeDesc* __restrict__ nodes,
                                                                     const Transform* __restrict__ xfs, double gap_open,
                                                                     double gap_extend, int max_entries,
                                                                     const double* __restrict__ staged, const StagedShape shape,
                                                                     uint32_t* __restrict__ bits_base,
                                                                     int32_t* __restrict__ aln_base, double* Xn_base,
                                                                     double* Tn_base, double* Wn_base,
                                                                     NodeOut* __restrict__ outs) {
    extern __shared__ double lds[];
    CR_STAMP(4);
    const PairDesc pd = pairs[blockIdx.x];
    const NodeDesc nd = nodes[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t* bits = bits_base + pd.bt_off;
    SeedMax unused;
    AlignEnd e;
    {
        SweepParams prm{0.0, gap_open, gap_extend};
        sweep_staged<R, kDtw>(staged + (int64_t)blockIdx.x * shape.pair_doubles() + (int64_t)w * shape.strip_doubles(), pd.n, pd.m,
                              prm, lds, nullptr, bits, unused, e, WidePlan<R>{0}.geom(w, pd.n));
    }
    // wave 0 walks; superposition sums and the merged node by the whole workgroup (node_finish with all hands)
    __shared__ int s_walk[4];
    uint32_t* const arow = reinterpret_cast<uint32_t*>(lds + kExpDoubles);
    double* const terms = lds + kExpDoubles + ((size_t)max_entries + 3) / 4 * 2;
    const int cap = pd.n + pd.m;
    if (threadIdx.x < kWave) {
        CR_STAMP(5);
        int idx, k;
        dtw_walk<R>(pd.n, pd.m, max_entries, bits, e.start_layer, lds + kExpDoubles, aln_base + pd.aln_off, idx, k);
        if (threadIdx.x == 0) {
            s_walk[0] = idx;
            s_walk[1] = k;
        }
        CR_STAMP(6);
    }
    __syncthreads();
    const int idx = s_walk[0], k = s_walk[1], first = cap - idx;
    const uint32_t* ent = arow + first;
    const double* X1 = coords + pd.off_i * 3;
    const double* X2 = coords + pd.off_j * 3;
    const double* T1 = tensors + pd.off_i * d;
    const double* T2 = tensors + pd.off_j * d;
    const double* W1 = weights + pd.off_i;
    const double* W2 = weights + pd.off_j;
    double* Xn = Xn_base + nd.out_off * 3;
    double* Tn = Tn_base + nd.out_off * d;
    double* Wn = Wn_base + nd.out_off;
    uint32_t flags = FLEX ? 0u : xfs[blockIdx.x].flags;
    double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3];
    const bool superpose = !FLEX && k > 3;               // multiple_alignment.py:364
    if constexpr (!FLEX) {
        if (superpose) kabsch_team(X1, X2, ent, idx, k, terms, terms + kSumTile * kMaxAcc + kSumSlack, c1, c2, Rm, t);
        else flags |= 8u;
    }
    // Protein.mean_function (:351-381) and get_mean_weights (:73-82), one alignment column per thread
    for (int x = threadIdx.x; x < idx; x += blockDim.x) {
        const uint32_t u = ent[x];
        const uint32_t i = u & 0xffffu, j = u >> 16;
        const bool has1 = i != kGap16, has2 = j != kGap16;
        const int64_t o = first + x;
        if constexpr (!FLEX) {
            double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
            if (has1)
                for (int c = 0; c < 3; c++) a[c] = superpose ? X1[(int64_t)i * 3 + c] - c1[c] : X1[(int64_t)i * 3 + c];
            if (has2) {
                if (superpose) {
                    const double v[3] = {X2[(int64_t)j * 3] - c2[0], X2[(int64_t)j * 3 + 1] - c2[1], X2[(int64_t)j * 3 + 2] - c2[2]};
                    rot3(v, Rm, b);
                } else {
                    for (int c = 0; c < 3; c++) b[c] = X2[(int64_t)j * 3 + c];
                }
            }
            for (int c = 0; c < 3; c++) Xn[o * 3 + c] = !has1 ? b[c] : (!has2 ? a[c] : (a[c] + b[c]) / 2);
        }
        for (int c = 0; c < d; c++) {
            const double ta = has1 ? T1[(int64_t)i * d + c] : 0.0, tb = has2 ? T2[(int64_t)j * d + c] : 0.0;
            Tn[o * d + c] = !has1 ? tb : (!has2 ? ta : (ta + tb) / 2);
        }
        double wsum = 0.0;
        if (has1) wsum += W1[i];
        if (has2) wsum += W2[j];
        Wn[o] = wsum;
    }
    if (threadIdx.x == 0) {
        NodeOut no;
        no.len = idx;
        no.first = first;
        no.flags = flags;
        no.pad = 0;
        outs[blockIdx.x] = no;
    }
    CR_STAMP(7);
}

// Alignment stage of a pair on staged scores: SW score + affine DTW fill with one wave per strip, then the traceback (wave 0)
// and Kabsch, RMSD / coverage / TM (the whole workgroup), as the second half of k_pair_wide.  SCORES (gap 0 only): the SW
// score alone, as k_score_team.
template <bool ZG, bool SCORES, int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_align_staged(const PairDesc* __restrict__ pairs,
                                                                      const double* __restrict__ coords,
                                                                      const Transform* __restrict__ xf,
                                                                      const double* __restrict__ seed_score, double sw_gap,
                                                                      double gap_open, double gap_extend, int max_entries,
                                                                      const double* __restrict__ staged, const StagedShape shape,
                                                                      uint32_t* __restrict__ bits, int32_t* __restrict__ aln,
                                                                      PairResult* __restrict__ res, const HostOut hout) {
    extern __shared__ double lds[];
    CR_STAMP(4);
    const PairDesc pd = pairs[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const double* strip = staged + (int64_t)blockIdx.x * shape.pair_doubles() + (int64_t)w * shape.strip_doubles();
    const StripGeom geom = WidePlan<R>{0}.geom(w, pd.n);
    SeedMax unused;
    AlignEnd e;
    SweepParams prm{sw_gap, gap_open, gap_extend};
    if constexpr (SCORES) sweep_staged<R, kSwScore | kZeroGap>(strip, pd.n, pd.m, prm, lds, nullptr, nullptr, unused, e, geom);
    else sweep_staged<R, kSwScore | kDtw | (ZG ? kZeroGap : 0)>(strip, pd.n, pd.m, prm, lds, nullptr, bits + pd.bt_off, unused, e, geom);
    PairResult r;
    r.sw = e.sw;
    r.dtw_score = SCORES ? 0.0 : e.dtw_score;
#pragma unroll
    for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
    for (int x = 0; x < 3; x++) r.t[x] = 0.0;
    r.rmsd = r.coverage = r.tm = 0.0;
    r.aln_len = r.aln_start = 0;
    r.flags = 0;
    if constexpr (!SCORES) {
        // wave 0 walks (the others wait at the barrier); Kabsch and the metrics by the whole workgroup, as k_pair_wide
        __shared__ int s_walk[4];
        uint32_t* const arow = reinterpret_cast<uint32_t*>(lds + kExpDoubles);
        double* const terms = lds + kExpDoubles + ((size_t)max_entries + 3) / 4 * 2;
        const int cap = pd.n + pd.m;
        if (threadIdx.x < kWave) {
            CR_STAMP(5);
            int idx, k;
            dtw_walk<R>(pd.n, pd.m, max_entries, bits + pd.bt_off, e.start_layer, lds + kExpDoubles, aln + pd.aln_off, idx, k);
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
    r.seed_score = seed_score[blockIdx.x];
    r.seed_len = xf[blockIdx.x].seed_len;
    r.flags |= xf[blockIdx.x].flags;
    if (threadIdx.x == 0) {
        res[blockIdx.x] = r;
        if (!SCORES && hout.res) hout.res[hout.dst(blockIdx.x)] = r;
    }
    CR_STAMP(7);
}

// Explicit score matrix S[seq1[i], seq2[j]] (dynamic_time_warping.py:24-26,79) into the skewed step order: the
// single-call dtw_align / smith_waterman(_score) drop-ins then run the multi-wave staged sweep instead of one wave
// (index sequences, alphabet mode included: the gather happens here, once).
template <int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_stage_explicit(const int32_t* __restrict__ seq1, int n,
                                                                        const int32_t* __restrict__ seq2, int m,
                                                                        const double* __restrict__ S, int64_t s_cols, int tc,
                                                                        double* __restrict__ staged, const StagedShape shape) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w * kWave * R >= n) return;
    int64_t rowoff[R];
#pragma unroll
    for (int q = 0; q < R; q++) {
        const int row = (w * kWave + lane) * R + q;
        rowoff[q] = row < n ? (int64_t)seq1[row] * s_cols : (int64_t)-1;
    }
    double* __restrict__ out = staged + (int64_t)w * shape.strip_doubles() + lane;
    const int t0 = (int)blockIdx.x * tc;
    // (every block of 16 steps a strip runs through, zeros where a lane's column is outside [0, m): as stage_block)
    const int t_end = (m + kWave - 1 + kStagedBlock - 1) / kStagedBlock * kStagedBlock;
    const int t1 = t0 + tc < t_end ? t0 + tc : t_end;
    for (int t = t0; t < t1; t++) {
        const int c = t - lane;
        if ((unsigned)c < (unsigned)m) {
            const int64_t col = seq2[c];
#pragma unroll
            for (int q = 0; q < R; q++) out[((int64_t)t * R + q) * kWave] = rowoff[q] >= 0 ? S[rowoff[q] + col] : 0.0;
        } else {
#pragma unroll
            for (int q = 0; q < R; q++) out[((int64_t)t * R + q) * kWave] = 0.0;
        }
    }
}

struct StagedTrace {          // = cr::TraceOut of cr_dropins.h (defined behind this header)
    int32_t len, start;
};

// The DP of the explicit-matrix drop-ins on staged scores: one workgroup, one wave per strip; WALK: dtw_align's
// traceback by wave 0 on the register-resident decision blocks (rows back to front in aln[0 .. cap), aln[cap .. 2 cap)).
template <int R, int MODE, bool WALK>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_explicit_staged(int n, int m, SweepParams prm,
                                                                         const double* __restrict__ staged,
                                                                         const StagedShape shape, uint32_t* __restrict__ dirs,
                                                                         uint32_t* __restrict__ bits, SeedMax* __restrict__ seed,
                                                                         AlignEnd* __restrict__ end, int max_entries,
                                                                         int32_t* __restrict__ aln, StagedTrace* __restrict__ tout) {
    extern __shared__ double lds[];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SeedMax sm;
    AlignEnd ae;
    ae.sw = ae.dtw_score = 0.0;
    ae.start_layer = ae.pad = 0;
    sweep_staged<R, MODE, false>(staged + (int64_t)w * shape.strip_doubles(), n, m, prm, lds, dirs, bits, sm, ae, WidePlan<R>{0}.geom(w, n));
    if (threadIdx.x >= kWave) return;                  // wave 0 goes on alone (wave_sync, no s_barrier from here on)
    if (threadIdx.x == 0) {
        if constexpr ((MODE & kSwTrace) != 0) *seed = sm;
        if constexpr ((MODE & (kSwScore | kDtw)) != 0) *end = ae;
    }
    if constexpr (WALK) {
        int len, pairs;
        dtw_walk<R>(n, m, max_entries, bits, ae.start_layer, lds, aln, len, pairs);
        if (threadIdx.x == 0) {
            tout->len = len;
            tout->start = n + m - len;
        }
    }
}

}  // namespace cr

#ifndef CR_KERNELS_TEMPLATES_ONLY      // the launchers (cr_api.hip)
namespace {

constexpr int kStageSteps = 16;            // steps of every strip per staging workgroup, at least

// Steps per staging workgroup: 16 while that makes at most ~2 000 workgroups (a tree level, a handful of pairs: as many
// CUs as possible), up to 64 for longer lists (every workgroup pays for its exp table, its rows and its column window
// before it forms a score: 496 pairs of 150 staged in 37 us with 16 steps per workgroup)
inline int stage_steps(int64_t count, int steps_total) {
    const int64_t groups16 = count * ((steps_total + kStageSteps - 1) / kStageSteps);
    return kStageSteps * (int)std::min<int64_t>(4, std::max<int64_t>(1, groups16 / 2048));
}

// f(std::integral_constant<int, R>) for the shape's rows per lane
template <class F>
int by_rows(int r, F&& f) {
    switch (r) {
        case 1: return f(std::integral_constant<int, 1>{});
        case 2: return f(std::integral_constant<int, 2>{});
        case 3: return f(std::integral_constant<int, 3>{});
        default: return f(std::integral_constant<int, 4>{});
    }
}

inline cr::StagedShape staged_shape(int n_bound, int m_bound) {
    cr::StagedShape s;
    s.r = std::min(cr::kStagedMaxR, (n_bound + cr::kStagedMaxWaves * cr::kWave - 1) / (cr::kStagedMaxWaves * cr::kWave));
    s.r = std::max(s.r, 1);
    s.waves = (n_bound + cr::kWave * s.r - 1) / (cr::kWave * s.r);
    s.steps = cr::staged_steps(m_bound);
    return s;
}


template <int D>
int launch_stage_tensor_d(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, double* staged, const cr::StagedShape shape) {
    const int steps = ck.m_max + cr::kWave - 1, tc = stage_steps(ck.count, steps);
    const size_t lds = sizeof(double) * cr::stage_lds_doubles(D, tc);
    const unsigned chunks = (unsigned)((steps + tc - 1) / tc);
    auto go = [&](auto kernel) -> int {
        CR_LAUNCH(kernel, dim3(chunks, (unsigned)ck.count), dim3(shape.waves * cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
                  b->pairs.p + ck.first, b->tensors.p, (int)b->d, prm.gamma_tensor, tc, staged, shape);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    return by_rows(shape.r, [&](auto rt) {
        constexpr int R = decltype(rt)::value;
        return go(cr::k_stage_tensor<D, R>);
    });
}

int launch_stage_tensor(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, double* staged, const cr::StagedShape shape) {
    switch (b->d_pad) {
        case 4: return launch_stage_tensor_d<4>(b, ck, prm, staged, shape);
        case 8: return launch_stage_tensor_d<8>(b, ck, prm, staged, shape);
        case 10: return launch_stage_tensor_d<10>(b, ck, prm, staged, shape);
        case 16: return launch_stage_tensor_d<16>(b, ck, prm, staged, shape);
        case 24: return launch_stage_tensor_d<24>(b, ck, prm, staged, shape);
        case 32: return launch_stage_tensor_d<32>(b, ck, prm, staged, shape);
        default: break;
    }
    if (b->d_pad <= 32) return fail(CR_ERR_ARGUMENT, "unsupported tensor width");
    // wider than the register-resident providers: the run-time-width staging kernel, 16 steps per workgroup (the LDS holds d planes
    // of 16 + 63 columns)
    const int steps = ck.m_max + cr::kWave - 1, tc = kStageSteps;
    const size_t lds = sizeof(double) * cr::stage_lds_doubles((int)b->d, tc);
    const unsigned chunks = (unsigned)((steps + tc - 1) / tc);
    auto go = [&](auto kernel) -> int {
        int rc = allow_lds(kernel, lds);
        if (rc) return rc;
        CR_LAUNCH(kernel, dim3(chunks, (unsigned)ck.count), dim3(shape.waves * cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
                  b->pairs.p + ck.first, b->tensors.p, (int)b->d, prm.gamma_tensor, tc, staged, shape);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    return by_rows(shape.r, [&](auto rt) { return go(cr::k_stage_tensor_any<decltype(rt)::value>); });
}

int launch_seed_staged(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, const double* staged, const cr::StagedShape shape) {
    const int entries = std::min(ck.n_max, ck.m_max);
    const size_t fill = cr::sweep_staged_lds_doubles<cr::kSwTrace>(shape.waves);
    const size_t lds = sizeof(double) * std::max(fill, (size_t)cr::kExpDoubles + cr::trace_team_lds_doubles(entries));
    auto go = [&](auto kernel) -> int {
        int rc = allow_lds(kernel, lds);
        if (rc) return rc;
        CR_LAUNCH(kernel, dim3((unsigned)ck.count), dim3(shape.waves * cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream, b->pairs.p + ck.first,
                  b->coords.p, prm.sw_gap, entries, staged, shape, b->dirs.p, b->xf.p + ck.first, b->seed_score.p + ck.first);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    return by_rows(shape.r, [&](auto rt) {
        constexpr int R = decltype(rt)::value;
        return prm.sw_gap == 0.0 ? go(cr::k_seed_staged<true, R>) : go(cr::k_seed_staged<false, R>);
    });
}

// the pair batch (cr_batch_run on a list short enough to be latency bound, cr_batch_set_pairs): coordinate scores in the
// frame of the seed superposition, then the alignment stage (or, scores only with gap 0, the SW score alone)
int launch_stage_coords(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, double* staged, const cr::StagedShape shape) {
    const int steps = ck.m_max + cr::kWave - 1, tc = stage_steps(ck.count, steps);
    const size_t lds = sizeof(double) * cr::stage_lds_doubles(cr::RbfCoords<1>::kColDoubles, tc);
    const unsigned chunks = (unsigned)((steps + tc - 1) / tc);
    auto go = [&](auto kernel) -> int {
        CR_LAUNCH(kernel, dim3(chunks, (unsigned)ck.count), dim3(shape.waves * cr::kWave), lds,
                  b->launch_stream ? b->launch_stream : b->ctx->stream, b->pairs.p + ck.first, b->coords.p, b->xf.p + ck.first,
                  prm.gamma_coords, tc, staged, shape);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    return by_rows(shape.r, [&](auto rt) { return go(cr::k_stage_coords<decltype(rt)::value>); });
}

int launch_align_staged(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, const double* staged, const cr::StagedShape shape,
                        bool scores) {
    const int entries = ck.max_aln;
    const bool zg = prm.sw_gap == 0.0;
    const size_t fill = scores ? cr::sweep_staged_lds_doubles<cr::kSwScore>(shape.waves) : cr::sweep_staged_lds_doubles<cr::kSwScore | cr::kDtw>(shape.waves);
    const size_t lds = sizeof(double) * std::max(fill, scores ? (size_t)0 : (size_t)cr::kExpDoubles + cr::trace_team_lds_doubles(entries));
    auto go = [&](auto kernel) -> int {
        int rc = allow_lds(kernel, lds);
        if (rc) return rc;
        CR_LAUNCH(kernel, dim3((unsigned)ck.count), dim3(shape.waves * cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
                  b->pairs.p + ck.first, b->coords.p, b->xf.p + ck.first, b->seed_score.p + ck.first, prm.sw_gap, prm.gap_open,
                  prm.gap_extend, entries, staged, shape, b->bits.p, b->aln.p, b->res.p + ck.first, host_out_for(b, ck));
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    return by_rows(shape.r, [&](auto rt) {
        constexpr int R = decltype(rt)::value;
        if (scores) return go(cr::k_align_staged<true, true, R>);
        return zg ? go(cr::k_align_staged<true, false, R>) : go(cr::k_align_staged<false, false, R>);
    });
}

int launch_stage_node(hipStream_t stream, int count, int m_max, const cr::PairDesc* pairs, const double* coords,
                      const double* weights, const cr::NodeDesc* nodes, const cr::Transform* xf, const cr_params& prm,
                      double gamma_weight, double* staged, const cr::StagedShape shape) {
    const int steps = m_max + cr::kWave - 1, tc = stage_steps(count, steps);
    const size_t lds = sizeof(double) * cr::stage_lds_doubles(cr::RbfNode<1>::kColDoubles, tc);
    const unsigned chunks = (unsigned)((steps + tc - 1) / tc);
    auto go = [&](auto kernel) -> int {
        CR_LAUNCH(kernel, dim3(chunks, (unsigned)count), dim3(shape.waves * cr::kWave), lds, stream, pairs, coords, weights, nodes, xf,
                  prm.gamma_coords, gamma_weight, tc, staged, shape);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    return by_rows(shape.r, [&](auto rt) { return go(cr::k_stage_node<decltype(rt)::value>); });
}

// flexible=True: the node scores (tensor RBF + consensus-weight RBF) of a level in the skewed step order
template <int D>
int launch_stage_flex_d(hipStream_t stream, int count, int m_max, const cr::PairDesc* pairs, const double* tensors, int d,
                        const double* weights, const cr::NodeDesc* nodes, const cr_params& prm, double gamma_weight, double* staged,
                        const cr::StagedShape shape) {
    const int steps = m_max + cr::kWave - 1, tc = stage_steps(count, steps);
    const size_t lds = sizeof(double) * cr::stage_lds_doubles(D + 1, tc);
    const unsigned chunks = (unsigned)((steps + tc - 1) / tc);
    auto go = [&](auto kernel) -> int {
        int rc = allow_lds(kernel, lds);
        if (rc) return rc;
        CR_LAUNCH(kernel, dim3(chunks, (unsigned)count), dim3(shape.waves * cr::kWave), lds, stream, pairs, tensors, d, weights, nodes,
                  prm.gamma_tensor, gamma_weight, tc, staged, shape);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    return by_rows(shape.r, [&](auto rt) { return go(cr::k_stage_flex<D, decltype(rt)::value>); });
}

int launch_stage_flex(hipStream_t stream, int count, int m_max, int d_pad, const cr::PairDesc* pairs, const double* tensors, int d,
                      const double* weights, const cr::NodeDesc* nodes, const cr_params& prm, double gamma_weight, double* staged,
                      const cr::StagedShape shape) {
    switch (d_pad) {
        case 4: return launch_stage_flex_d<4>(stream, count, m_max, pairs, tensors, d, weights, nodes, prm, gamma_weight, staged, shape);
        case 8: return launch_stage_flex_d<8>(stream, count, m_max, pairs, tensors, d, weights, nodes, prm, gamma_weight, staged, shape);
        case 10: return launch_stage_flex_d<10>(stream, count, m_max, pairs, tensors, d, weights, nodes, prm, gamma_weight, staged, shape);
        case 16: return launch_stage_flex_d<16>(stream, count, m_max, pairs, tensors, d, weights, nodes, prm, gamma_weight, staged, shape);
        case 24: return launch_stage_flex_d<24>(stream, count, m_max, pairs, tensors, d, weights, nodes, prm, gamma_weight, staged, shape);
        case 32: return launch_stage_flex_d<32>(stream, count, m_max, pairs, tensors, d, weights, nodes, prm, gamma_weight, staged, shape);
        default: return fail(CR_ERR_ARGUMENT, "unsupported tensor width");
    }
}

int launch_node_staged(hipStream_t stream, int count, int entries, const cr::PairDesc* pairs, const double* coords,
                       const double* tensors, int d, const double* weights, const cr::NodeDesc* nodes, const cr::Transform* xf,
                       const cr_params& prm, const double* staged, const cr::StagedShape shape, uint32_t* bits, int32_t* aln,
                       double* xn, double* tn, double* wn, cr::NodeOut* out, bool flexible = false) {
    const size_t lds = sizeof(double) * std::max(cr::sweep_staged_lds_doubles<cr::kDtw>(shape.waves),
                                                 (size_t)cr::kExpDoubles + cr::trace_team_lds_doubles(entries));
    auto go = [&](auto kernel) -> int {
        int rc = allow_lds(kernel, lds);
        if (rc) return rc;
        CR_LAUNCH(kernel, dim3((unsigned)count), dim3(shape.waves * cr::kWave), lds, stream, pairs, coords, tensors, d, weights, nodes,
                  xf, prm.gap_open, prm.gap_extend, entries, staged, shape, bits, aln, xn, tn, wn, out);
        CR_HIP(hipGetLastError());
        return CR_OK;
    };
    return by_rows(shape.r, [&](auto rt) {
        constexpr int R = decltype(rt)::value;
        return flexible ? go(cr::k_node_staged<R, true>) : go(cr::k_node_staged<R, false>);
    });
}

}  // namespace
#endif
