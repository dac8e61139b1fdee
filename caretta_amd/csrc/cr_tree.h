// The whole guide tree of a progressive alignment (multiple_alignment.py:172-253) in ONE persistent launch, without a barrier
// between tree levels.  Included by cr_api.hip behind cr_staged.h (kernel), launcher in cr_progressive.h.
//
// Round 5 ran a tree LEVEL by level: per level a planning launch, a score-staging launch, the seed sweeps, a second staging
// launch and the node sweeps -- five launches, each level waiting for its slowest node, 17 levels for 128 structures of which
// the top eight hold ONE node each.  Here a workgroup takes the next node of the tree (tickets in level order, so the
// children of a node always hold smaller tickets), waits for its two children's "done" words, and does everything the five
// launches did for that node itself: the node's descriptors from its children's lengths, the tensor scores in the skewed
// step order (stage_whole: the provider code of the staging kernels), the Smith-Waterman sweep + walk + seed superposition
// (seed_staged_body), the node scores in the seed's frame, the affine sweep + walk + merged node (node_staged_body); then it
// publishes the node.  Sub-trees advance independently; nothing waits for a level.  Every value is the value the level-wise
// launches form: the same device functions on the same inputs.
//
// Hand-over between workgroups (possibly on different XCDs): a node's arrays, its length and offset are plain stores
// followed by a device-scope release fence by every wave, a barrier, and ONE device-scope store of the done word; a parent
// polls the done words of its children with device-scope loads and passes a device-scope acquire fence before it reads.
// Deadlock: the grid is at most one workgroup per CU (all resident); a waiting workgroup waits for smaller tickets only,
// which are held by resident workgroups or finished.  A poll that does not end gives up (abort word) and the host runs the
// tree level by level instead.
// Storage is static, sized by the launch's length bound (1.5 x the longest leaf): node x of the plan writes its rows into the
// arena at base + 2 bound x and its alignment rows at 4 bound x; decision words and staged scores belong to the WORKGROUP.
#pragma once

namespace cr {

struct TreeCtl {
    uint32_t next;           // tickets handed out
    uint32_t abort;          // a poll gave up
    uint32_t overflow;       // a node outgrew the bound
    uint32_t pad;
};

constexpr uint32_t kTreeSpinLimit = 1u << 22;            // polls of ~1 us

CR_D uint32_t tree_load_u32(const uint32_t* p) { return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CR_D void tree_store_u32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int D, int R, bool FLEX>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_tree_resident(
    const PlanNode* __restrict__ plan, const int num_nodes, const int num_leaves, const int bound, TreeCtl* ctl, uint32_t* done,
    int64_t* len, int64_t* off, const int64_t arena_base, double* coords, double* tensors, const int d, double* weights,
    const double gamma_tensor, const double gamma_coords, const double gamma_weight, const double sw_gap, const double gap_open,
    const double gap_extend, double* __restrict__ staged, const StagedShape shape, const int tc_tensor, const int tc_node,
    uint32_t* __restrict__ dirs, const int64_t dirs_words, uint32_t* __restrict__ bits, const int64_t bits_words,
    int32_t* __restrict__ aln, NodeOut* __restrict__ outs, const int dbg) {
    extern __shared__ double lds[];
    __shared__ int s_ticket[2];
    __shared__ Transform s_tr;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double* const my_staged = staged + (int64_t)blockIdx.x * shape.pair_doubles();
    const double* const my_strip = my_staged + (int64_t)w * shape.strip_doubles();
    for (;;) {
        // ---- the next node of the plan, its children finished ------------------------------------------------------------
        if (threadIdx.x == 0) {
            const uint32_t t = __hip_atomic_fetch_add(&ctl->next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int ok = t < (uint32_t)num_nodes ? 1 : 0;
            if (ok) {
                const PlanNode pn = plan[t];
                uint32_t spins = 0;
                for (int side = 0; side < 2 && ok; side++) {
                    const int c = side == 0 ? pn.c1 : pn.c2;
                    if (c < num_leaves) continue;
                    while (tree_load_u32(done + c) == 0u) {
                        __builtin_amdgcn_s_sleep(16);
                        if (++spins > kTreeSpinLimit || ((spins & 255u) == 0u && tree_load_u32(&ctl->abort) != 0u)) {
                            tree_store_u32(&ctl->abort, 1u);
                            ok = 0;
                            break;
                        }
                    }
                }
            }
            s_ticket[0] = (int)t;
            s_ticket[1] = ok;
        }
        __syncthreads();
        const int x = s_ticket[0];
        if (!s_ticket[1]) return;                         // (whole workgroup) no node left, or the launch was given up
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the children's arrays, lengths and offsets
        const PlanNode pn = plan[x];
        int64_t n64 = len[pn.c1], m64 = len[pn.c2];
        if (n64 > bound || m64 > bound || n64 < 1 || m64 < 1) {
            if (threadIdx.x == 0) tree_store_u32(&ctl->overflow, 1u);
            n64 = n64 > bound ? bound : (n64 < 1 ? 1 : n64);
            m64 = m64 > bound ? bound : (m64 < 1 ? 1 : m64);
        }
        PairDesc pd;
        pd.n = (int32_t)n64;
        pd.m = (int32_t)m64;
        pd.off_i = off[pn.c1];
        pd.off_j = off[pn.c2];
        pd.dirs_off = (int64_t)blockIdx.x * dirs_words;
        pd.bt_off = (int64_t)blockIdx.x * bits_words;
        pd.aln_off = (int64_t)x * 4 * bound;
        pd.hand_off = 0;
        NodeDesc nd;
        nd.mult1 = pn.mult1;
        nd.mult2 = pn.mult2;
        nd.out_off = arena_base + (int64_t)x * 2 * bound;
        uint32_t seed_flags = 0;
        if constexpr (!FLEX) {
            // ---- tensor scores -> Smith-Waterman sweep, walk, seed superposition (multiple_alignment.py:328-345) ----------
            {
                RbfTensor<R, D> src;
                src.rows_g = tensors + pd.off_i * d;
                src.cols_g = tensors + pd.off_j * d;
                src.d = d;
                src.neg_gamma = -gamma_tensor;
                if (!(dbg & 1)) stage_whole<R>(src, pd.n, pd.m, tc_tensor, my_staged, shape, lds);
            }
            SeedMax sm;
            Transform tr;
            tr.flags = kFlagSeedSkipped;
            tr.seed_len = 0;
            if (dbg & 2) {
            } else if (sw_gap == 0.0) seed_staged_body<true, R>(pd, coords, sw_gap, bound, my_strip, dirs, lds, sm, tr);
            else seed_staged_body<false, R>(pd, coords, sw_gap, bound, my_strip, dirs, lds, sm, tr);
            seed_flags = tr.flags;
            __syncthreads();                              // (the seed stage's LDS is read no more)
            if (threadIdx.x == 0) s_tr = tr;
            __syncthreads();
            // ---- node scores in the seed's frame (:204-210) ---------------------------------------------------------------
            RbfNode<R> src;
            src.xyz.rows_g = coords + pd.off_i * 3;
            src.xyz.cols_g = coords + pd.off_j * 3;
            src.xyz.xf = &s_tr;
            src.xyz.neg_gamma = -gamma_coords;
            src.w_rows = weights + pd.off_i;
            src.w_cols = weights + pd.off_j;
            src.mult1 = nd.mult1;
            src.mult2 = nd.mult2;
            src.neg_gamma_w = -gamma_weight;
            if (!(dbg & 1)) stage_whole<R>(src, pd.n, pd.m, tc_node, my_staged, shape, lds);
        } else {
            // flexible=True (:323-326, :351-362): the node score is the tensor RBF + the consensus-weight RBF, no seed
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
            stage_whole<R>(src, pd.n, pd.m, tc_tensor, my_staged, shape, lds);
        }
        // ---- affine sweep, walk, superposition on the aligned columns, the merged node (:211-234, :351-381) ------------------
        NodeOut no;
        no.len = pd.n > pd.m ? pd.n : pd.m;
        no.first = pd.n + pd.m - no.len;
        no.flags = 0;
        no.pad = 0;
        if (!(dbg & 4)) node_staged_body<R, FLEX>(pd, nd, coords, tensors, d, weights, seed_flags, gap_open, gap_extend, 2 * bound, my_strip, bits, aln, coords,
                                  tensors, weights, lds, no);
        if (threadIdx.x == 0) {
            outs[x] = no;
            len[pn.id] = no.len;
            off[pn.id] = nd.out_off + no.first;
        }
        // ---- publish ----------------------------------------------------------------------------------------------------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // every wave: its rows of the node (and thread 0's record) leave this XCD
        __syncthreads();
        if (threadIdx.x == 0) tree_store_u32(done + pn.id, 1u);
    }
}

}  // namespace cr
