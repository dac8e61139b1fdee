// The kernels: k_seed / k_align / k_score, tree nodes, team and wide kernels, result packing, level planning, MSA metrics.
// Part of cr_kernels.h (included there, inside namespace cr, in this order: cr_providers.h, cr_sweep.h, cr_sweep_cols.h,
// cr_sweep_wide.h, cr_trace.h, cr_pair_kernels.h); not a header of its own.

// ---------------------------------------------------------------------------------------------
// Batch kernels: two launches per batch, one wave per pair, each a fill followed by its traceback
// in the same wave (the latency-bound walk of one wave hides under the FP64 fill of its neighbours).
// LDS (doubles): [0,kExpDoubles) exp table | union { ring + strip hand-off rows , entries + window/scratch }.
// ---------------------------------------------------------------------------------------------

// Make this wave's own decision words (plain global stores) visible to its own later loads.
CR_D void drain_stores() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    wave_sync();
}

// Stages 1+2: tensor RBF + SW fill (multiple_alignment.py:328-335), then traceback + seed Kabsch.
// (the column sweep holds R * D row features and little else: three waves per SIMD -- 168 VGPRs -- up to 50 of them,
// two up to 100, one for the widest tensors)
template <int R, int D, bool ZG>
__global__ __launch_bounds__(kWave, (ZG && R * D <= 50) ? 3 : (R * D <= 100 ? 2 : 1)) void k_seed(const PairDesc* __restrict__ pairs,
                                               const double* __restrict__ tensors, int d,
                                               const double* __restrict__ coords, double gamma, double sw_gap,
                                               int max_entries, uint32_t* __restrict__ dirs,
                                               double* __restrict__ hand, Transform* __restrict__ xf,
                                               double* __restrict__ seed_score) {
    extern __shared__ double lds[];
    CR_STAMP(0);
    const PairDesc pd = pairs[blockIdx.x];
    SeedMax sm;
    AlignEnd unused;
    {
        RbfTensor<R, D> src;
        src.rows_g = tensors + pd.off_i * d;
        src.cols_g = tensors + pd.off_j * d;
        src.d = d;
        src.neg_gamma = -gamma;
        SweepParams prm{sw_gap, 0.0, 0.0};
        if constexpr (ZG) sweep_cols<R, D>(src, pd.n, pd.m, lds, dirs + pd.dirs_off, hand + pd.hand_off, sm);
        else sweep<R, kSwTrace>(src, pd.n, pd.m, prm, lds, dirs + pd.dirs_off, nullptr, hand + pd.hand_off, sm, unused);
    }
    drain_stores();
    CR_STAMP(1);
    Transform tr;
    seed_trace<R, ZG ? 0 : 1>(pd, max_entries, coords, dirs, sm, lds + kExpDoubles, tr);
    if (threadIdx.x == 0) {
        xf[blockIdx.x] = tr;
        seed_score[blockIdx.x] = sm.score;
    }
    CR_STAMP(3);
}

// Stages 3+4: coordinate RBF on the seed-superposed frames + SW score + affine DTW fill
// (multiple_alignment.py:347-349, :164, :263-275), then traceback + Kabsch + metrics.
template <int R, bool ZG>
__global__ __launch_bounds__(kWave, 4) void k_align(const PairDesc* __restrict__ pairs,
                                                const double* __restrict__ coords,
                                                const Transform* __restrict__ xf,
                                                const double* __restrict__ seed_score, double gamma,
                                                double sw_gap, double gap_open, double gap_extend,
                                                int max_entries, uint32_t* __restrict__ bits,
                                                double* __restrict__ hand, int32_t* __restrict__ aln,
                                                PairResult* __restrict__ res, const HostOut hout) {
    extern __shared__ double lds[];
    CR_STAMP(4);
    const PairDesc pd = pairs[blockIdx.x];
    SeedMax unused;
    AlignEnd e;
    {
        RbfCoords<R> src;
        src.rows_g = coords + pd.off_i * 3;
        src.cols_g = coords + pd.off_j * 3;
        src.xf = xf + blockIdx.x;
        src.neg_gamma = -gamma;
        SweepParams prm{sw_gap, gap_open, gap_extend};
        sweep<R, kSwScore | kDtw | (ZG ? kZeroGap : 0)>(src, pd.n, pd.m, prm, lds, nullptr, bits + pd.bt_off,
                                                         hand + pd.hand_off, unused, e);
    }
    drain_stores();
    CR_STAMP(5);
    PairResult r;
    align_trace<R>(pd, max_entries, coords, bits, e, lds + kExpDoubles, aln, r, hout);
    r.seed_score = seed_score[blockIdx.x];
    r.seed_len = xf[blockIdx.x].seed_len;
    r.flags |= xf[blockIdx.x].flags;
    if (threadIdx.x == 0) {
        res[blockIdx.x] = r;
        if (hout.res) hout.res[hout.dst(blockIdx.x)] = r;
    }
    CR_STAMP(7);
}

// Stage 3 alone: coordinate RBF on the seed-superposed frames + smith_waterman_score (multiple_alignment.py:347-349,
// :164) -- the P x P matrix entry of a pair without its pairwise alignment (sw_gap == 0; cr_batch_run_scores).
template <int R>
__global__ __launch_bounds__(kWave) void k_score(const PairDesc* __restrict__ pairs, const double* __restrict__ coords,
                                                const Transform* __restrict__ xf,
                                                const double* __restrict__ seed_score, double gamma,
                                                double* __restrict__ hand, PairResult* __restrict__ res) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.x];
    RbfCoords<R> src;
    src.rows_g = coords + pd.off_i * 3;
    src.cols_g = coords + pd.off_j * 3;
    src.xf = xf + blockIdx.x;
    src.neg_gamma = -gamma;
    const double sw = sweep_cols_score<R>(src, pd.n, pd.m, lds, hand + pd.hand_off);
    if (threadIdx.x == 0) {
        PairResult r;
        r.sw = sw;
        r.dtw_score = 0.0;
#pragma unroll
        for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
        for (int x = 0; x < 3; x++) r.t[x] = 0.0;
        r.rmsd = r.coverage = r.tm = 0.0;
        r.seed_score = seed_score[blockIdx.x];
        r.aln_len = r.aln_start = 0;
        r.seed_len = xf[blockIdx.x].seed_len;
        r.flags = xf[blockIdx.x].flags;
        res[blockIdx.x] = r;
    }
}

template <int RA, int RB>
__global__ __launch_bounds__(kWideMaxWaves* kWave) void k_score_team(const PairDesc* __restrict__ pairs,
                                                                    const double* __restrict__ coords,
                                                                    const Transform* __restrict__ xf,
                                                                    const double* __restrict__ seed_score, double gamma,
                                                                    int nA, PairResult* __restrict__ res) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const WidePlan<RA, RB> plan{nA};
    const StripGeom geom = plan.geom(w, pd.n);
    double sw = 0.0;
    auto fill = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;
        RbfCoords<R> src;
        src.rows_g = coords + pd.off_i * 3;
        src.cols_g = coords + pd.off_j * 3;
        src.xf = xf + blockIdx.x;
        src.neg_gamma = -gamma;
        sw = sweep_cols_score_team<R>(src, pd.n, pd.m, lds, geom);
    };
    if (plan.wave_in_a(w)) fill(std::integral_constant<int, RA>{});
    else if constexpr (RA != RB) fill(std::integral_constant<int, RB>{});
    if (threadIdx.x == 0) {
        PairResult r;
        r.sw = sw;
        r.dtw_score = 0.0;
#pragma unroll
        for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
        for (int x = 0; x < 3; x++) r.t[x] = 0.0;
        r.rmsd = r.coverage = r.tm = 0.0;
        r.seed_score = seed_score[blockIdx.x];
        r.aln_len = r.aln_start = 0;
        r.seed_len = xf[blockIdx.x].seed_len;
        r.flags = xf[blockIdx.x].flags;
        res[blockIdx.x] = r;
    }
}

// One node of progressive alignment (multiple_alignment.py:193-234), after k_seed has produced the seed
// superposition of the two children: node score -> affine DTW fill -> traceback -> Protein.mean_function
// (:351-381: tensors averaged column by column, coordinates averaged after superposing on the aligned
// positions) and get_mean_weights (:73-82).  One wave.  Outputs have cap = n + m rows, valid from `first`.
struct NodeOut {
    int32_t len, first;
    uint32_t flags;
    int32_t pad;
};

// Per-node launch arguments: the multipliers of multiple_alignment.py:199-202 and where the node goes.
struct NodeDesc {
    double mult1, mult2;
    int64_t out_off;         // residue offset of this node's cap-sized output region in Xn / Tn / Wn
};

// One wave per tree node; blockIdx.x indexes pairs / nodes / xf / out.  The children are read from
// coords / tensors / weights at pd.off_i, pd.off_j; the node is written to Xn / Tn / Wn at out_off (the
// output arrays may be the input arrays: a level of the guide tree appends to the arena it reads from).
template <int R>
CR_D void node_finish(const PairDesc& pd, const NodeDesc& nd, const Transform* xf, const AlignEnd& e, const double* coords,
                      const double* tensors, int d, const double* weights, int max_entries, const uint32_t* __restrict__ bits,
                      int32_t* __restrict__ aln, double* lds, double* Xn, double* Tn, double* Wn, NodeOut* out);

template <int R, bool TEAM>
CR_D void node_body(const PairDesc* __restrict__ pairs, const double* coords,
                                               const double* tensors, int d, const double* weights,
                                               const NodeDesc* __restrict__ nodes,
                                               const Transform* __restrict__ xfs, double gamma_coords,
                                               double gamma_weight, double gap_open, double gap_extend,
                                               int max_entries, uint32_t* __restrict__ bits_base,
                                               double* __restrict__ hand_base, int32_t* __restrict__ aln_base,
                                               double* Xn_base, double* Tn_base, double* Wn_base,
                                               NodeOut* __restrict__ outs) {
    extern __shared__ double lds[];
    CR_STAMP(4);
    const PairDesc pd = pairs[blockIdx.x];
    const NodeDesc nd = nodes[blockIdx.x];
    const Transform* xf = xfs + blockIdx.x;
    const double mult1 = nd.mult1, mult2 = nd.mult2;
    uint32_t* bits = bits_base + pd.bt_off;
    double* hand = hand_base + pd.hand_off;
    int32_t* aln = aln_base + pd.aln_off;
    double* Xn = Xn_base + nd.out_off * 3;
    double* Tn = Tn_base + nd.out_off * d;
    double* Wn = Wn_base + nd.out_off;
    NodeOut* out = outs + blockIdx.x;
    SeedMax unused;
    AlignEnd e;
    {
        RbfNode<R> src;
        src.xyz.rows_g = coords + pd.off_i * 3;
        src.xyz.cols_g = coords + pd.off_j * 3;
        src.xyz.xf = xf;
        src.xyz.neg_gamma = -gamma_coords;
        src.w_rows = weights + pd.off_i;
        src.w_cols = weights + pd.off_j;
        src.mult1 = mult1;
        src.mult2 = mult2;
        src.neg_gamma_w = -gamma_weight;
        SweepParams prm{0.0, gap_open, gap_extend};
        // one wave per strip: the wide sweep (all columns of the node resident in LDS, a barrier every 8 steps instead of
        // every step, the scores one column ahead with 1 or 2 rows per lane)
        if constexpr (TEAM) sweep_wide<R, kDtw>(src, pd.n, pd.m, prm, lds, 8, nullptr, bits, unused, e,
                                                WidePlan<R>{0}.geom(__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), pd.n));
        else sweep<R, kDtw>(src, pd.n, pd.m, prm, lds, nullptr, bits, hand, unused, e);
    }
    if constexpr (TEAM) {
        if (threadIdx.x >= kWave) return;              // wave 0 goes on alone (wave_sync, no s_barrier from here on)
    } else {
        drain_stores();
    }
    CR_STAMP(5);
    node_finish<R>(pd, nd, xf, e, coords, tensors, d, weights, max_entries, bits, aln, lds, Xn, Tn, Wn, out);
    CR_STAMP(7);
}

// The part of a node behind its fill (one wave): DTW traceback, superposition on the aligned positions, the merged node.
template <int R>
CR_D void node_finish(const PairDesc& pd, const NodeDesc& nd, const Transform* xf, const AlignEnd& e, const double* coords,
                      const double* tensors, int d, const double* weights, int max_entries, const uint32_t* __restrict__ bits,
                      int32_t* __restrict__ aln, double* lds, double* Xn, double* Tn, double* Wn, NodeOut* out) {
    const int lane = threadIdx.x;
    double* tl = lds + kExpDoubles;
    const int cap = pd.n + pd.m;
    int idx, k;
    dtw_walk<R>(pd.n, pd.m, max_entries, bits, e.start_layer, tl, aln, idx, k);
    CR_STAMP(6);
    const int first = cap - idx;
    const uint32_t* ent = reinterpret_cast<const uint32_t*>(tl) + first;
    double* scratch = tl + ((size_t)max_entries + 3) / 4 * 2;
    const double* X1 = coords + pd.off_i * 3;
    const double* X2 = coords + pd.off_j * 3;
    const double* T1 = tensors + pd.off_i * d;
    const double* T2 = tensors + pd.off_j * d;
    const double* W1 = weights + pd.off_i;
    const double* W2 = weights + pd.off_j;
    uint32_t flags = xf->flags;
    double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3];
    const bool superpose = k > 3;                        // multiple_alignment.py:364
    if (superpose) kabsch_ordered(X1, X2, ent, idx, k, lane, scratch, c1, c2, Rm, t);
    else flags |= 8u;
    for (int x = lane; x < idx; x += kWave) {
        const uint32_t u = ent[x];
        const uint32_t i = u & 0xffffu, j = u >> 16;
        const bool has1 = i != kGap16, has2 = j != kGap16;
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
        const int64_t o = first + x;
        for (int c = 0; c < 3; c++) Xn[o * 3 + c] = !has1 ? b[c] : (!has2 ? a[c] : (a[c] + b[c]) / 2);
        for (int c = 0; c < d; c++) {
            const double ta = has1 ? T1[(int64_t)i * d + c] : 0.0, tb = has2 ? T2[(int64_t)j * d + c] : 0.0;
            Tn[o * d + c] = !has1 ? tb : (!has2 ? ta : (ta + tb) / 2);
        }
        double wsum = 0.0;
        if (has1) wsum += W1[i];
        if (has2) wsum += W2[j];
        Wn[o] = wsum;
    }
    if (lane == 0) {
        NodeOut no;
        no.len = idx;
        no.first = first;
        no.flags = flags;
        no.pad = 0;
        *out = no;
    }
}

template <int R>
__global__ __launch_bounds__(kWave) void k_node(const PairDesc* __restrict__ pairs, const double* coords, const double* tensors,
                                               int d, const double* weights, const NodeDesc* __restrict__ nodes,
                                               const Transform* __restrict__ xfs, double gamma_coords,
                                               double gamma_weight, double gap_open, double gap_extend,
                                               int max_entries, uint32_t* __restrict__ bits_base,
                                               double* __restrict__ hand_base, int32_t* __restrict__ aln_base,
                                               double* Xn_base, double* Tn_base, double* Wn_base,
                                               NodeOut* __restrict__ outs) {
    node_body<R, false>(pairs, coords, tensors, d, weights, nodes, xfs, gamma_coords, gamma_weight, gap_open, gap_extend,
                        max_entries, bits_base, hand_base, aln_base, Xn_base, Tn_base, Wn_base, outs);
}

template <int R>
__global__ __launch_bounds__(kTeamWaves* kWave) void k_node_team(const PairDesc* __restrict__ pairs, const double* coords, const double* tensors,
                                               int d, const double* weights, const NodeDesc* __restrict__ nodes,
                                               const Transform* __restrict__ xfs, double gamma_coords,
                                               double gamma_weight, double gap_open, double gap_extend,
                                               int max_entries, uint32_t* __restrict__ bits_base,
                                               double* __restrict__ hand_base, int32_t* __restrict__ aln_base,
                                               double* Xn_base, double* Tn_base, double* Wn_base,
                                               NodeOut* __restrict__ outs) {
    node_body<R, true>(pairs, coords, tensors, d, weights, nodes, xfs, gamma_coords, gamma_weight, gap_open, gap_extend,
                        max_entries, bits_base, hand_base, aln_base, Xn_base, Tn_base, Wn_base, outs);
}

// Results of a batch in the CALLER's pair order and the caller's layout, produced on the device so that the host side
// of cr_batch_fetch is two plain copies: out_res[order[k]] = res[k]; out_aln[order[k]][0..1][0..stride) = the two
// alignment rows of launch slot k, left-aligned, padded with -2 (the rows sit back-to-front in `aln`, PairResult has
// their start and length).  One wave per pair.  T = int32_t or int64_t; order == nullptr: identity.
template <class T>
__global__ __launch_bounds__(kWave) void k_pack_results(const PairDesc* __restrict__ pairs,
                                                       const PairResult* __restrict__ res,
                                                       const int32_t* __restrict__ order,
                                                       const int32_t* __restrict__ aln, int64_t stride,
                                                       PairResult* __restrict__ out_res, T* __restrict__ out_aln) {
    const int k = blockIdx.x;
    const int lane = threadIdx.x;
    const int dst = order ? order[k] : k;
    const PairDesc pd = pairs[k];
    const PairResult r = res[k];
    if (out_res && lane == 0) out_res[dst] = r;
    if (!out_aln) return;
    const int cap = pd.n + pd.m;
    const int32_t* a1 = aln + pd.aln_off + r.aln_start;
    const int32_t* a2 = a1 + cap;
    T* o1 = out_aln + (int64_t)dst * 2 * stride;
    T* o2 = o1 + stride;
    for (int64_t x = lane; x < stride; x += kWave) {
        o1[x] = x < r.aln_len ? (T)a1[x] : (T)-2;
        o2[x] = x < r.aln_len ? (T)a2[x] : (T)-2;
    }
}

// The pair descriptors of a list over structures of EQUAL length, built on the device from the caller's (i, j) list: every
// pair has the same scratch footprint, so the offsets are arithmetic (per_chunk pairs share one scratch region after the
// other).  130 816 pairs: 1 MB of indices go up instead of 7.3 MB of descriptors, and the host never builds them.
template <class Dummy = void>
__global__ void k_make_pairs_uniform_t(const int32_t* __restrict__ ij, const int64_t* __restrict__ offsets, int n, int64_t dw, int64_t bw,
                                       int64_t hand_per, int64_t per_chunk, PairDesc* __restrict__ out, int64_t npairs) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const int64_t local = p % per_chunk;
    PairDesc pd;
    pd.n = pd.m = n;
    pd.off_i = offsets[ij[2 * p]];
    pd.off_j = offsets[ij[2 * p + 1]];
    pd.dirs_off = local * dw;
    pd.bt_off = local * bw;
    pd.aln_off = p * 4 * (int64_t)n;
    pd.hand_off = local * hand_per;
    out[p] = pd;
}
constexpr auto k_make_pairs_uniform = k_make_pairs_uniform_t<>;

// out[order[k]] = res[k].sw: the scores of a batch whose launch order differs from the caller's pair order
template <class Dummy = void>
__global__ void k_scatter_sw_t(const PairResult* __restrict__ res, const int32_t* __restrict__ order,
                               double* __restrict__ out, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[order[k]] = res[k].sw;
}
constexpr auto k_scatter_sw = k_scatter_sw_t<>;
// the same for the flags
template <class Dummy = void>
__global__ void k_scatter_flags_t(const PairResult* __restrict__ res, const int32_t* __restrict__ order,
                                  uint32_t* __restrict__ out, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[order[k]] = res[k].flags;
}
constexpr auto k_scatter_flags = k_scatter_flags_t<>;

// ---------------------------------------------------------------------------------------------
// Device-side planning of one level of the guide tree (cr_progressive.h, the launch sequence without host
// round trips).  One thread: (1) commits the previous level -- length and arena offset of every node it produced --,
// (2) lays out this level: PairDesc / NodeDesc of every node from its children's lengths, decision-scratch offsets by
// running sums, output rows appended to the arena.  Lengths beyond the bound the launches were sized for set
// *overflow and are clamped (the host then repeats the tree with the level-by-level path).
// ---------------------------------------------------------------------------------------------
struct PlanNode {
    int32_t c1, c2, id, pad;     // children and own node id
    double mult1, mult2;
};

template <class Dummy = void>
__global__ void k_plan_level_t(const PlanNode* __restrict__ prev, int prev_count, const NodeDesc* __restrict__ prev_desc,
                               const NodeOut* __restrict__ prev_out, const PlanNode* __restrict__ cur, int count, int R,
                               int bound, int64_t aln_base, int64_t* __restrict__ len, int64_t* __restrict__ off,
                               int64_t* __restrict__ used, PairDesc* __restrict__ pairs, NodeDesc* __restrict__ nodes,
                               int32_t* __restrict__ overflow) {
    // one workgroup: the global reads and writes are spread over the threads, the running sums are taken by thread 0
    // over LDS copies of the lengths
    extern __shared__ int32_t plan_nm[];                     // [count][2]
    for (int x = threadIdx.x; x < prev_count; x += blockDim.x) {
        len[prev[x].id] = prev_out[x].len;
        off[prev[x].id] = prev_desc[x].out_off + prev_out[x].first;
    }
    __threadfence_block();
    __syncthreads();                                   // (four waves: a level of more than 64 nodes is spread over all of them)
    for (int x = threadIdx.x; x < count; x += blockDim.x) {
        int64_t n = len[cur[x].c1], m = len[cur[x].c2];
        if (n > bound || m > bound || n < 1 || m < 1) {
            *overflow = 1;
            n = n > bound ? bound : (n < 1 ? 1 : n);
            m = m > bound ? bound : (m < 1 ? 1 : m);
        }
        plan_nm[2 * x] = (int32_t)n;
        plan_nm[2 * x + 1] = (int32_t)m;
        PairDesc pd;
        pd.n = (int32_t)n;
        pd.m = (int32_t)m;
        pd.off_i = off[cur[x].c1];
        pd.off_j = off[cur[x].c2];
        pd.dirs_off = pd.bt_off = pd.aln_off = pd.hand_off = 0;
        pairs[x] = pd;
        nodes[x].mult1 = cur[x].mult1;
        nodes[x].mult2 = cur[x].mult2;
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t dirs_off = 0, bt_off = 0, aln_off = aln_base, rows = *used;
        for (int x = 0; x < count; x++) {
            const int n = plan_nm[2 * x], m = plan_nm[2 * x + 1];
            pairs[x].dirs_off = dirs_off;
            pairs[x].bt_off = bt_off;
            pairs[x].aln_off = aln_off;
            nodes[x].out_off = rows;
            dirs_off += (int64_t)strips_of(n, R) * tblocks(m, 16) * R * kWave;
            bt_off += (int64_t)strips_of(n, R) * tblocks(m, 8) * R * kWave;
            aln_off += 2 * (int64_t)(n + m);
            rows += n + m;
        }
        *used = rows;
    }
}
constexpr auto k_plan_level = k_plan_level_t<>;

// Team versions of k_seed and k_node for launches with few blocks (progressive alignment levels, small pair
// lists): kTeamWaves waves sweep the strips of one pair concurrently (sweep_team); wave 0 then runs the same
// traceback / Kabsch / mean code as the single-wave kernels.  Requires strips_of(n, R) <= kTeamWaves.
template <int R, int D, bool ZG>
__global__ __launch_bounds__(kTeamWaves* kWave) void k_seed_team(const PairDesc* __restrict__ pairs,
                                                                const double* __restrict__ tensors, int d,
                                                                const double* __restrict__ coords, double gamma,
                                                                double sw_gap, int max_entries,
                                                                uint32_t* __restrict__ dirs,
                                                                Transform* __restrict__ xf,
                                                                double* __restrict__ seed_score) {
    extern __shared__ double lds[];
    CR_STAMP(0);
    const PairDesc pd = pairs[blockIdx.x];
    SeedMax sm;
    AlignEnd unused;
    {
        RbfTensor<R, D> src;
        src.rows_g = tensors + pd.off_i * d;
        src.cols_g = tensors + pd.off_j * d;
        src.d = d;
        src.neg_gamma = -gamma;
        SweepParams prm{sw_gap, 0.0, 0.0};
        if constexpr (ZG) sweep_cols_team<R, D>(src, pd.n, pd.m, lds, dirs + pd.dirs_off, sm,
                                                WidePlan<R>{0}.geom(__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), pd.n));
        else sweep_team<R, kSwTrace>(src, pd.n, pd.m, prm, lds, dirs + pd.dirs_off, nullptr, sm, unused);
    }
    if (threadIdx.x >= kWave) return;                  // wave 0 goes on alone (wave_sync, no s_barrier from here on)
    CR_STAMP(1);
    Transform tr;
    seed_trace<R, ZG ? 0 : 1>(pd, max_entries, coords, dirs, sm, lds + kExpDoubles, tr);
    if (threadIdx.x == 0) {
        xf[blockIdx.x] = tr;
        seed_score[blockIdx.x] = sm.score;
    }
    CR_STAMP(3);
}

template <int R, bool ZG>
__global__ __launch_bounds__(kTeamWaves* kWave) void k_align_team(const PairDesc* __restrict__ pairs,
                                                                 const double* __restrict__ coords,
                                                                 const Transform* __restrict__ xf,
                                                                 const double* __restrict__ seed_score, double gamma,
                                                                 double sw_gap, double gap_open, double gap_extend,
                                                                 int max_entries, uint32_t* __restrict__ bits,
                                                                 int32_t* __restrict__ aln, PairResult* __restrict__ res, const HostOut hout) {
    extern __shared__ double lds[];
    CR_STAMP(4);
    const PairDesc pd = pairs[blockIdx.x];
    SeedMax unused;
    AlignEnd e;
    {
        RbfCoords<R> src;
        src.rows_g = coords + pd.off_i * 3;
        src.cols_g = coords + pd.off_j * 3;
        src.xf = xf + blockIdx.x;
        src.neg_gamma = -gamma;
        SweepParams prm{sw_gap, gap_open, gap_extend};
        sweep_team<R, kSwScore | kDtw | (ZG ? kZeroGap : 0)>(src, pd.n, pd.m, prm, lds, nullptr, bits + pd.bt_off, unused, e);
    }
    if (threadIdx.x >= kWave) return;                  // wave 0 goes on alone (wave_sync, no s_barrier from here on)
    CR_STAMP(5);
    PairResult r;
    align_trace<R>(pd, max_entries, coords, bits, e, lds + kExpDoubles, aln, r, hout);
    r.seed_score = seed_score[blockIdx.x];
    r.seed_len = xf[blockIdx.x].seed_len;
    r.flags |= xf[blockIdx.x].flags;
    if (threadIdx.x == 0) {
        res[blockIdx.x] = r;
        if (hout.res) hout.res[hout.dst(blockIdx.x)] = r;
    }
    CR_STAMP(7);
}

// Wide versions (sweep_wide): up to kWideMaxWaves waves per pair, columns resident in LDS, a barrier every
// `sync_every` steps.  Requires strips_of(n, R) <= blockDim.x / 64 and the resident columns to fit the LDS.
template <int RA, int RB, int D, bool ZG>
__global__ __launch_bounds__(kWideMaxWaves* kWave) void k_seed_wide(const PairDesc* __restrict__ pairs,
                                                                   const double* __restrict__ tensors, int d,
                                                                   const double* __restrict__ coords, double gamma,
                                                                   double sw_gap, int max_entries, int sync_every, int nA,
                                                                   uint32_t* __restrict__ dirs,
                                                                   Transform* __restrict__ xf,
                                                                   double* __restrict__ seed_score) {
    extern __shared__ double lds[];
    CR_STAMP(0);
    const PairDesc pd = pairs[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const WidePlan<RA, RB> plan{nA};
    const StripGeom geom = plan.geom(w, pd.n);
    SeedMax sm;
    AlignEnd unused;
    // every wave runs the sweep instantiated for ITS strip's rows per lane; both have the same barriers
    auto fill = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;
        RbfTensor<R, D> src;
        src.rows_g = tensors + pd.off_i * d;
        src.cols_g = tensors + pd.off_j * d;
        src.d = d;
        src.neg_gamma = -gamma;
        SweepParams prm{sw_gap, 0.0, 0.0};
        if constexpr (ZG) sweep_cols_team<R, D>(src, pd.n, pd.m, lds, dirs + pd.dirs_off, sm, geom);
        else sweep_wide<R, kSwTrace>(src, pd.n, pd.m, prm, lds, sync_every, dirs + pd.dirs_off, nullptr, sm, unused, geom);
    };
    if (plan.wave_in_a(w)) fill(std::integral_constant<int, RA>{});
    else if constexpr (RA != RB) fill(std::integral_constant<int, RB>{});
    if (threadIdx.x >= kWave) return;                  // wave 0 goes on alone (wave_sync, no s_barrier from here on)
    CR_STAMP(1);
    Transform tr;
    seed_trace<RA, ZG ? 0 : 1, RB>(pd, max_entries, coords, dirs, sm, lds + kExpDoubles, tr, nA);
    if (threadIdx.x == 0) {
        xf[blockIdx.x] = tr;
        seed_score[blockIdx.x] = sm.score;
    }
    CR_STAMP(3);
}

// Both stages of a pair in ONE launch of the wide layout: seed fill -> (wave 0) seed walk + Kabsch -> align fill (or the
// score sweep alone, SCORES) -> (wave 0) DTW walk + Kabsch + metrics.  With one pair per CU (one GPU's share of a sharded
// long-chain family) two launches meant that every CU waited for the slowest pair of the seed launch before any of
// them started its alignment fill, and a launch gap on top: 252 pairs of 1200 x 1200 took 2.46 ms where the phases of
// the median pair add up to 2.32.  The seed superposition reaches the second fill through LDS.
template <int RA, int RB, int D, bool ZG, bool SCORES>
__global__ __launch_bounds__(kWideMaxWaves* kWave) void k_pair_wide(const PairDesc* __restrict__ pairs,
                                                                   const double* __restrict__ tensors, int d,
                                                                   const double* __restrict__ coords, double gamma_tensor,
                                                                   double gamma_coords, double sw_gap, double gap_open,
                                                                   double gap_extend, int seed_entries, int align_entries,
                                                                   int sync_every, int nA, uint32_t* __restrict__ dirs,
                                                                   uint32_t* __restrict__ bits, Transform* __restrict__ xf,
                                                                   double* __restrict__ seed_score, int32_t* __restrict__ aln,
                                                                   PairResult* __restrict__ res, const HostOut hout) {
    extern __shared__ double lds[];
    __shared__ Transform s_tr;
    CR_STAMP(0);
    const PairDesc pd = pairs[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const WidePlan<RA, RB> plan{nA};
    const StripGeom geom = plan.geom(w, pd.n);
    SeedMax sm;
    AlignEnd e;
    {
        AlignEnd unused;
        auto fill = [&](auto rtag) {
            constexpr int R = decltype(rtag)::value;
            RbfTensor<R, D> src;
            src.rows_g = tensors + pd.off_i * d;
            src.cols_g = tensors + pd.off_j * d;
            src.d = d;
            src.neg_gamma = -gamma_tensor;
            SweepParams prm{sw_gap, 0.0, 0.0};
            if constexpr (ZG) sweep_cols_team<R, D>(src, pd.n, pd.m, lds, dirs + pd.dirs_off, sm, geom);
            else sweep_wide<R, kSwTrace>(src, pd.n, pd.m, prm, lds, sync_every, dirs + pd.dirs_off, nullptr, sm, unused, geom);
        };
        if (plan.wave_in_a(w)) fill(std::integral_constant<int, RA>{});
        else if constexpr (RA != RB) fill(std::integral_constant<int, RB>{});
    }
    // wave 0 walks (the others wait at the barrier); the position-ordered sums behind the walk are taken by everybody
    __shared__ int s_walk[4];
    uint32_t* const seed_list = reinterpret_cast<uint32_t*>(lds + kExpDoubles);
    double* const seed_terms = lds + kExpDoubles + ((size_t)seed_entries + 3) / 4 * 2;
    if (threadIdx.x < kWave) {
        CR_STAMP(1);
        int k, len;
        uint32_t fl;
        seed_walk<RA, ZG ? 0 : 1, RB>(pd, dirs, sm, seed_list, nA, k, len, fl);
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
                sw_only = sweep_cols_score_team<R>(src, pd.n, pd.m, lds, geom);
            } else {
                SweepParams prm{sw_gap, gap_open, gap_extend};
                sweep_wide<R, kSwScore | kDtw | (ZG ? kZeroGap : 0)>(src, pd.n, pd.m, prm, lds, sync_every, nullptr, bits + pd.bt_off, unused, e, geom);
            }
        };
        if (plan.wave_in_a(w)) fill(std::integral_constant<int, RA>{});
        else if constexpr (RA != RB) fill(std::integral_constant<int, RB>{});
    }
    CR_STAMP(5);
    PairResult r;
    r.sw = SCORES ? sw_only : e.sw;
    r.dtw_score = SCORES ? 0.0 : e.dtw_score;
#pragma unroll
    for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
    for (int x = 0; x < 3; x++) r.t[x] = 0.0;
    r.rmsd = r.coverage = r.tm = 0.0;
    r.aln_len = r.aln_start = 0;
    r.flags = 0;
    if constexpr (!SCORES) {
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
        if (!SCORES && hout.res) hout.res[hout.dst(blockIdx.x)] = r;
    }
    CR_STAMP(7);
}

#ifndef CR_KERNELS_TEMPLATES_ONLY   // the one non-template kernel: defined in cr_api.hip's translation unit only
// Pairwise RMSD / coverage / TM matrices of a finished multiple alignment (make_rmsd_coverage_tm_matrix,
// multiple_alignment.py:1000-1055).  msa: int32 [P][W] residue indices, -1 = gap.  One wave per pair i<j
// (blockIdx.x enumerates them row-major).  superpose != 0: Kabsch per pair first (superpose_first=False);
// otherwise the coordinates are compared as they are.  out: [npairs][4] = rmsd, coverage, tm, k.
__global__ __launch_bounds__(kWave) void k_msa_metrics(const double* __restrict__ coords,
                                                      const int64_t* __restrict__ offsets,
                                                      const int32_t* __restrict__ msa, int P, int W, int superpose,
                                                      const int32_t* __restrict__ pairs, double* __restrict__ out) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int i = pairs[2 * blockIdx.x], j = pairs[2 * blockIdx.x + 1];
    uint32_t* ent = reinterpret_cast<uint32_t*>(lds);
    double* scratch = lds + ((size_t)W + 3) / 4 * 2;
    int kloc = 0;
    for (int x = lane; x < W; x += kWave) {
        const int a = msa[(int64_t)i * W + x], b = msa[(int64_t)j * W + x];
        const bool pair = a != -1 && b != -1;
        ent[x] = pair ? pack_entry(a, b) : pack_entry(-1, -1);
        kloc += pair ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) kloc += __shfl_xor(kloc, off);
    wave_sync();
    const int k = kloc;
    const double* Xi = coords + offsets[i] * 3;
    const double* Xj = coords + offsets[j] * 3;
    const int n = (int)(offsets[i + 1] - offsets[i]), m = (int)(offsets[j + 1] - offsets[j]);
    double rmsd = 0.0, tm = 0.0;
    if (k >= 3) {
        if (superpose) {
            double c1[3], c2[3], Rm[9], t[3];
            kabsch_ordered(Xi, Xj, ent, W, k, lane, scratch, c1, c2, Rm, t);
            rmsd_tm_ordered<true>(Xi, Xj, ent, W, k, n, m, Rm, t, lane, scratch, rmsd, tm);
        } else {
            rmsd_tm_ordered<false>(Xi, Xj, ent, W, k, n, m, nullptr, nullptr, lane, scratch, rmsd, tm);
        }
    }
    if (lane == 0) {
        out[4 * (int64_t)blockIdx.x + 0] = rmsd;
        out[4 * (int64_t)blockIdx.x + 1] = (double)k / (double)W;
        out[4 * (int64_t)blockIdx.x + 2] = tm;
        out[4 * (int64_t)blockIdx.x + 3] = (double)k;
    }
}
#endif
