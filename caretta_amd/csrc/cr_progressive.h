// Whole-tree progressive alignment (MultipleAlignment.progressive_align, multiple_alignment.py:172-253) with
// every node resident in HBM.  Included at the end of cr_api.hip (after cr_dropins.h).
//
// The guide tree is cut into LEVELS: a node's level is 1 + the larger level of its children, leaves are level
// 0.  Nodes of one level are independent (:193-234 reads only the two children), so a level is one k_seed
// launch plus one k_node launch with one wave per node.  Node lengths are data dependent, so the host reads the
// level's NodeOut records (and its alignment rows) back before it lays out the next level: one small
// device->host copy per level, no other traffic until the caller fetches results.
//
// Arena: coordinates / tensors / weights of all nodes live in three growing device arrays.  Leaves occupy
// the front in input order; every internal node reserves cap = n + m rows (k_node fills them back to front) and
// ends up at [out_off + first, out_off + first + len).

struct cr_progressive {
    cr_context* ctx = nullptr;
    int64_t P = 0, d = 0;
    cr_batch scratch;                        // k_seed launch state: arena (coords, tensors) + decision scratch
    DevBuf<double> weights;                  // arena, one double per row
    DevBuf<double> staged;                   // scores of one level formed ahead of its sweeps (cr_staged.h)
    int64_t used = 0, capacity = 0;          // arena rows
    DevBuf<cr::NodeDesc> d_nodes;
    DevBuf<cr::NodeOut> d_outs;
    PinnedBuf<cr::PairDesc> p_pairs;         // page-locked staging of the per-level uploads and read-backs
    PinnedBuf<cr::NodeDesc> p_nodes;
    PinnedBuf<cr::NodeOut> p_outs;
    PinnedBuf<int32_t> p_rows;
    std::vector<int64_t> off, len;           // per node id (0 .. 2P-2): arena row offset, rows
    std::vector<int64_t> child1, child2, level, members;   // per node id
    std::vector<uint32_t> flags;             // per internal node k
    std::vector<std::vector<int32_t>> aln;   // per internal node k: row 1 then row 2, `len` entries each
    int64_t levels = 0;
    uint32_t any_flags = 0;
    bool flexible = false;                   // flexible=True in score and mean function: tensors and consensus weights only
};

namespace {

// grow a device array to `rows * width` elements, keeping its first `keep_rows * width`
int grow_keep(DevBuf<double>& buf, int64_t rows, int64_t keep_rows, int64_t width, hipStream_t stream) {
    DevBuf<double> bigger;
    CR_HIP(bigger.ensure((size_t)(rows * width) + 64));      // slack: sweep_cols reads padded feature rows
    if (keep_rows > 0 && buf.p)
        CR_HIP(hipMemcpyAsync(bigger.p, buf.p, sizeof(double) * (size_t)(keep_rows * width), hipMemcpyDeviceToDevice, stream));
    CR_HIP(hipStreamSynchronize(stream));
    std::swap(buf.p, bigger.p);
    std::swap(buf.n, bigger.n);
    std::swap(buf.cls, bigger.cls);
    std::swap(buf.dev, bigger.dev);
    return CR_OK;
}

int arena_reserve(cr_progressive* h, int64_t rows) {
    if (rows <= h->capacity) return CR_OK;
    const int64_t want = std::max(rows, h->capacity + h->capacity / 2);
    int rc = grow_keep(h->scratch.coords, want, h->used, 3, h->ctx->stream);
    if (!rc) rc = grow_keep(h->scratch.tensors, want, h->used, h->d, h->ctx->stream);
    if (!rc) rc = grow_keep(h->weights, want, h->used, 1, h->ctx->stream);
    if (!rc) h->capacity = want;
    return rc;
}

// one level of the tree: the internal nodes `ids` (node ids), children complete
int run_level(cr_progressive* h, const std::vector<int64_t>& ids, const cr_params& prm, double gamma_weight) {
    cr_batch& b = h->scratch;
    hipStream_t stream = h->ctx->stream;
    const size_t count = ids.size();
    CR_HIP(h->p_pairs.ensure(count));
    CR_HIP(h->p_nodes.ensure(count));
    CR_HIP(h->p_outs.ensure(count));
    cr::PairDesc* pairs = h->p_pairs.p;
    cr::NodeDesc* nodes = h->p_nodes.p;
    int n_max = 0, m_max = 0, cap_max = 0;
    for (int64_t id : ids) n_max = std::max<int>(n_max, (int)h->len[(size_t)h->child1[(size_t)id]]);
    // few blocks per launch: the team kernels (kTeamWaves waves per node) whenever the rows fit their strips
    // (with at most 192 rows one strip of a single-wave kernel beats three 64-row strips of the team, cr_batch_set_pairs)
    const bool team = n_max > 3 * cr::kWave && n_max <= 5 * cr::kTeamWaves * cr::kWave && !g_cfg.no_team;
    const int R = team ? (n_max + cr::kTeamWaves * cr::kWave - 1) / (cr::kTeamWaves * cr::kWave) : rows_per_lane(n_max);
    int64_t dirs_off = 0, bt_off = 0, aln_off = 0, hand_off = 0, rows = h->used;
    for (size_t x = 0; x < count; x++) {
        const int64_t id = ids[x], c1 = h->child1[(size_t)id], c2 = h->child2[(size_t)id];
        cr::PairDesc& pd = pairs[x];
        pd.n = (int)h->len[(size_t)c1];
        pd.m = (int)h->len[(size_t)c2];
        CR_REQUIRE(pd.n + pd.m <= cr::kMaxLength, "tree node longer than 65534 columns");
        pd.off_i = h->off[(size_t)c1];
        pd.off_j = h->off[(size_t)c2];
        pd.dirs_off = dirs_off;
        pd.bt_off = bt_off;
        pd.aln_off = aln_off;
        pd.hand_off = hand_off;
        dirs_off += (int64_t)cr::strips_of(pd.n, R) * cr::tblocks(pd.m, 16) * R * cr::kWave;
        bt_off += (int64_t)cr::strips_of(pd.n, R) * cr::tblocks(pd.m, 8) * R * cr::kWave;
        aln_off += 2 * (int64_t)(pd.n + pd.m);
        if (cr::strips_of(pd.n, R) > 1) hand_off += 3 * (int64_t)pd.m;
        m_max = std::max(m_max, pd.m);
        cap_max = std::max(cap_max, pd.n + pd.m);
        // multiple_alignment.py:199-202: each side is weighted by the OTHER side's share of the members
        const double total = (double)(h->members[(size_t)c1] + h->members[(size_t)c2]);
        nodes[x].mult1 = (double)h->members[(size_t)c2] / (2.0 * total);
        nodes[x].mult2 = (double)h->members[(size_t)c1] / (2.0 * total);
        nodes[x].out_off = rows;
        rows += pd.n + pd.m;
    }
    int rc = arena_reserve(h, rows);
    if (rc) return rc;
    CR_HIP(b.pairs.ensure(count));
    CR_HIP(b.dirs.ensure((size_t)dirs_off));
    CR_HIP(b.bits.ensure((size_t)bt_off));
    CR_HIP(b.hand.ensure((size_t)hand_off));
    CR_HIP(b.aln.ensure((size_t)aln_off));
    CR_HIP(b.xf.ensure(count));
    CR_HIP(b.seed_score.ensure(count));
    CR_HIP(h->d_nodes.ensure(count));
    CR_HIP(h->d_outs.ensure(count));
    CR_UPLOAD(h->ctx, b.pairs.p, pairs, sizeof(cr::PairDesc) * count);
    CR_UPLOAD(h->ctx, h->d_nodes.p, nodes, sizeof(cr::NodeDesc) * count);
    b.r_seed = b.r_align = R;
    const cr_batch::Chunk ck{0, (int64_t)count, n_max, m_max, cap_max};
    rc = team ? launch_seed_team(R, &b, ck, prm) : launch_seed_r(R, &b, ck, prm);
    if (rc) return rc;
    rc = team ? launch_node_team(R, stream, (int)count, n_max, m_max, cap_max, b.pairs.p, b.coords.p, b.tensors.p, (int)h->d,
                                 h->weights.p, h->d_nodes.p, b.xf.p, prm, gamma_weight, b.bits.p, b.hand.p, b.aln.p, b.coords.p,
                                 b.tensors.p, h->weights.p, h->d_outs.p)
              : launch_node(R, stream, (int)count, n_max, m_max, cap_max, b.pairs.p, b.coords.p, b.tensors.p, (int)h->d,
                     h->weights.p, h->d_nodes.p, b.xf.p, prm, gamma_weight, b.bits.p, b.hand.p, b.aln.p, b.coords.p,
                     b.tensors.p, h->weights.p, h->d_outs.p);
    if (rc) return rc;
    CR_HIP(h->p_rows.ensure((size_t)aln_off));
    cr::NodeOut* outs = h->p_outs.p;
    int32_t* rows_host = h->p_rows.p;
    CR_DOWNLOAD(h->ctx, outs, h->d_outs.p, sizeof(cr::NodeOut) * count);
    CR_DOWNLOAD(h->ctx, rows_host, b.aln.p, sizeof(int32_t) * (size_t)aln_off);
    CR_HIP(hipStreamSynchronize(stream));
    for (size_t x = 0; x < count; x++) {
        const int64_t id = ids[x], k = id - h->P;
        const cr::NodeOut& no = outs[x];
        const int64_t cap = pairs[x].n + pairs[x].m;
        h->len[(size_t)id] = no.len;
        h->off[(size_t)id] = nodes[x].out_off + no.first;
        h->flags[(size_t)k] = no.flags;
        h->any_flags |= no.flags;
        std::vector<int32_t>& a = h->aln[(size_t)k];
        a.resize((size_t)(2 * no.len));
        const int32_t* src = rows_host + pairs[x].aln_off;
        std::copy(src + no.first, src + no.first + no.len, a.begin());
        std::copy(src + cap + no.first, src + cap + no.first + no.len, a.begin() + no.len);
    }
    h->used = rows;
    return CR_OK;
}

// The whole tree without host round trips.  Launch shapes (rows per lane, LDS, scratch) are sized for a length
// bound of 1.5 x the longest leaf; a one-thread planning kernel per level (cr::k_plan_level) turns the lengths the
// previous level produced into this level's descriptors on the device.  Returns 1 when the bound does not apply
// (staged scores: bound <= 2048; team kernels: 192 < bound <= 1280) or a node outgrew it: the caller then runs the
// level-by-level path.
int run_tree_planned(cr_progressive* h, const std::vector<std::vector<int64_t>>& by_level, const cr_params& prm,
                     double gamma_weight, const char** why) {
    cr_batch& b = h->scratch;
    hipStream_t stream = h->ctx->stream;
    const int64_t P = h->P, total = h->used, num_nodes = P - 1;
    int64_t longest = 0;
    for (int64_t s = 0; s < P; s++) longest = std::max(longest, h->len[(size_t)s]);
    const int bound = (int)std::min<int64_t>(cr::kStagedMaxRows, (longest * 3 + 1) / 2 + 8);
    *why = "the longest structure exceeds the 2048 columns the resident tree is sized for";
    if (longest > bound) return 1;
    *why = "CARETTA_NO_TEAM is set";
    if (g_cfg.no_team) return 1;
    int64_t widest_level = 0;
    for (int64_t lv = 1; lv <= h->levels; lv++) widest_level = std::max<int64_t>(widest_level, (int64_t)by_level[(size_t)lv].size());
    // Scores formed by their own launches (cr_staged.h) while up to four rows per lane fit the 8 waves of its workgroups
    // (2048 rows) and the widest level's scores fit a tenth of the device memory; CARETTA_STAGED=0: the fused kernels
    const cr::StagedShape shape = staged_shape(bound, bound);
    bool staged = bound <= cr::kStagedMaxRows && widest_level <= 65535 && g_cfg.staged;
    if (staged) {
        size_t free_b = 0, total_b = 0;
        CR_HIP(hipMemGetInfo(&free_b, &total_b));
        staged = (double)widest_level * (double)shape.pair_doubles() * sizeof(double) <= (double)total_b / 10.0;
    }
    *why = !g_cfg.staged ? "CARETTA_STAGED=0" : widest_level > 65535 ? "a tree level of more than 65535 nodes"
                                                               : "the staged scores of the widest tree level exceed a tenth of the device memory";
    if (h->flexible && !staged) return 1;                    // (flexible=True runs on staged scores only)
    if (!staged && (bound <= 3 * cr::kWave || bound > 5 * cr::kTeamWaves * cr::kWave)) return 1;     // (the four-wave team kernels: 193 .. 1280 rows)
    const int R = staged ? shape.r : (bound + cr::kTeamWaves * cr::kWave - 1) / (cr::kTeamWaves * cr::kWave);

    // static plan: every internal node in level order
    std::vector<cr::PlanNode> plan;
    std::vector<int64_t> start((size_t)h->levels + 2, 0), aln_base((size_t)h->levels + 2, 0);
    int64_t max_count = 0;
    for (int64_t lv = 1; lv <= h->levels; lv++) {
        start[(size_t)lv] = (int64_t)plan.size();
        for (int64_t id : by_level[(size_t)lv]) {
            const int64_t c1 = h->child1[(size_t)id], c2 = h->child2[(size_t)id];
            const double tot = (double)(h->members[(size_t)c1] + h->members[(size_t)c2]);
            plan.push_back(cr::PlanNode{(int32_t)c1, (int32_t)c2, (int32_t)id, 0, (double)h->members[(size_t)c2] / (2.0 * tot),
                                        (double)h->members[(size_t)c1] / (2.0 * tot)});
        }
        const int64_t count = (int64_t)by_level[(size_t)lv].size();
        max_count = std::max(max_count, count);
        aln_base[(size_t)lv + 1] = aln_base[(size_t)lv] + count * 4 * (int64_t)bound;
    }
    start[(size_t)h->levels + 1] = (int64_t)plan.size();
    const int64_t aln_total = aln_base[(size_t)h->levels + 1];
    int rc = arena_reserve(h, total + num_nodes * 2 * (int64_t)bound);
    if (rc) return rc;
    const int64_t dirs_words = max_count * cr::strips_of(bound, R) * cr::tblocks(bound, 16) * R * cr::kWave;
    const int64_t bits_words = max_count * cr::strips_of(bound, R) * cr::tblocks(bound, 8) * R * cr::kWave;
    // the tree's small tables in ONE device block and ONE upload: plan | len | off | used | overflow (every separate small
    // copy from or to pageable memory costs the host ~25 us: seventeen of them were 0.5 ms of a 5 ms tree)
    const size_t nids = (size_t)(2 * P - 1);
    const size_t o_len = (sizeof(cr::PlanNode) * plan.size() + 15) / 16 * 16, o_off = o_len + sizeof(int64_t) * nids;
    const size_t o_used = o_off + sizeof(int64_t) * nids, o_over = o_used + sizeof(int64_t), meta_bytes = o_over + sizeof(int64_t);
    DevBuf<char> d_meta;
    CR_HIP(d_meta.ensure(meta_bytes));
    struct {
        cr::PlanNode* p;
    } const d_plan{reinterpret_cast<cr::PlanNode*>(d_meta.p)};
    struct I64 {
        int64_t* p;
    };
    const I64 d_len{reinterpret_cast<int64_t*>(d_meta.p + o_len)}, d_off{reinterpret_cast<int64_t*>(d_meta.p + o_off)},
        d_used{reinterpret_cast<int64_t*>(d_meta.p + o_used)};
    struct {
        int32_t* p;
    } const d_overflow{reinterpret_cast<int32_t*>(d_meta.p + o_over)};
    CR_HIP(b.pairs.ensure((size_t)num_nodes));
    CR_HIP(b.xf.ensure((size_t)num_nodes));
    CR_HIP(b.seed_score.ensure((size_t)num_nodes));
    CR_HIP(h->d_nodes.ensure((size_t)num_nodes));
    CR_HIP(h->d_outs.ensure((size_t)num_nodes));
    CR_HIP(b.dirs.ensure((size_t)dirs_words));
    CR_HIP(b.bits.ensure((size_t)bits_words));
    CR_HIP(b.aln.ensure((size_t)aln_total));
    if (staged) CR_HIP(h->staged.ensure((size_t)(max_count * shape.pair_doubles())));
    {
        std::vector<char> meta(meta_bytes, 0);
        std::memcpy(meta.data(), plan.data(), sizeof(cr::PlanNode) * plan.size());
        std::memcpy(meta.data() + o_len, h->len.data(), sizeof(int64_t) * nids);
        std::memcpy(meta.data() + o_off, h->off.data(), sizeof(int64_t) * nids);
        std::memcpy(meta.data() + o_used, &total, sizeof(int64_t));
        // (through the context's page-locked ring whatever its size: the vector dies at the end of this block)
        char* stage = nullptr;
        if (meta_bytes <= kRingSlot) {
            if ((rc = ring_slot(h->ctx, 0, &stage))) return rc;
            std::memcpy(stage, meta.data(), meta_bytes);
            CR_HIP(hipMemcpyAsync(d_meta.p, stage, meta_bytes, hipMemcpyHostToDevice, stream));
            CR_HIP(hipEventRecord(h->ctx->ring_ev[0], stream));
            h->ctx->ring_busy[0] = true;
        } else {
            CR_UPLOAD(h->ctx, d_meta.p, meta.data(), meta_bytes);
            CR_HIP(hipStreamSynchronize(stream));
        }
    }
    b.r_seed = b.r_align = R;
    for (int64_t lv = 1; lv <= h->levels + 1; lv++) {
        const int64_t first = start[(size_t)lv], count = lv <= h->levels ? start[(size_t)lv + 1] - first : 0;
        const int64_t pfirst = lv > 1 ? start[(size_t)lv - 1] : 0, pcount = lv > 1 ? first - pfirst : 0;
        CR_LAUNCH(cr::k_plan_level, dim3(1), dim3(256), sizeof(int32_t) * 2 * (size_t)std::max<int64_t>(count, 1), stream, d_plan.p + pfirst, (int)pcount, h->d_nodes.p + pfirst,
                           h->d_outs.p + pfirst, d_plan.p + first, (int)count, R, bound, aln_base[(size_t)std::min(lv, h->levels)],
                           d_len.p, d_off.p, d_used.p, b.pairs.p + first, h->d_nodes.p + first, d_overflow.p);
        CR_HIP(hipGetLastError());
        if (count == 0) break;
        const cr_batch::Chunk ck{first, count, bound, bound, 2 * bound};
        if (h->flexible) {
            // flexible=True (multiple_alignment.py:323-326, :351-362): no seed stage -- node scores = tensor RBF + weight RBF
            // (every CU) -> DTW sweep + node (mean tensors and weights)
            if ((rc = launch_stage_flex(stream, (int)count, bound, b.d_pad, b.pairs.p + first, b.tensors.p, (int)h->d, h->weights.p,
                                        h->d_nodes.p + first, prm, gamma_weight, h->staged.p, shape)))
                return rc;
            if ((rc = launch_node_staged(stream, (int)count, 2 * bound, b.pairs.p + first, b.coords.p, b.tensors.p, (int)h->d,
                                         h->weights.p, h->d_nodes.p + first, b.xf.p + first, prm, h->staged.p, shape, b.bits.p,
                                         b.aln.p, b.coords.p, b.tensors.p, h->weights.p, h->d_outs.p + first, /*flexible=*/true)))
                return rc;
            continue;
        }
        if (staged) {
            // scores (every CU) -> SW sweep + seed superposition -> node scores in that frame (every CU) -> DTW sweep + node
            if ((rc = launch_stage_tensor(&b, ck, prm, h->staged.p, shape))) return rc;
            if ((rc = launch_seed_staged(&b, ck, prm, h->staged.p, shape))) return rc;
            if ((rc = launch_stage_node(stream, (int)count, bound, b.pairs.p + first, b.coords.p, h->weights.p, h->d_nodes.p + first,
                                        b.xf.p + first, prm, gamma_weight, h->staged.p, shape)))
                return rc;
            if ((rc = launch_node_staged(stream, (int)count, 2 * bound, b.pairs.p + first, b.coords.p, b.tensors.p, (int)h->d,
                                         h->weights.p, h->d_nodes.p + first, b.xf.p + first, prm, h->staged.p, shape, b.bits.p,
                                         b.aln.p, b.coords.p, b.tensors.p, h->weights.p, h->d_outs.p + first)))
                return rc;
            continue;
        }
        if ((rc = launch_seed_team(R, &b, ck, prm))) return rc;
        if ((rc = launch_node_team(R, stream, (int)count, bound, bound, 2 * bound, b.pairs.p + first, b.coords.p, b.tensors.p,
                                   (int)h->d, h->weights.p, h->d_nodes.p + first, b.xf.p + first, prm, gamma_weight, b.bits.p,
                                   b.hand.p, b.aln.p, b.coords.p, b.tensors.p, h->weights.p, h->d_outs.p + first)))
            return rc;
    }
    // one read-back for the whole tree: four copies into the context's page-locked landing area, ONE wait
    std::vector<cr::NodeOut> outs((size_t)num_nodes);
    std::vector<cr::NodeDesc> descs((size_t)num_nodes);
    std::vector<int64_t> len(nids), off(nids);
    std::vector<int32_t> rows_host((size_t)aln_total);
    int32_t overflow = 0;
    int64_t used = 0;
    {
        const size_t b_outs = sizeof(cr::NodeOut) * (size_t)num_nodes, b_descs = sizeof(cr::NodeDesc) * (size_t)num_nodes;
        const size_t b_tail = meta_bytes - o_len, b_rows = sizeof(int32_t) * (size_t)aln_total;
        const size_t a_descs = (b_outs + 15) / 16 * 16, a_tail = a_descs + (b_descs + 15) / 16 * 16, a_rows = a_tail + (b_tail + 15) / 16 * 16;
        void* land_v = nullptr;
        if ((rc = host_landing(h->ctx, a_rows + b_rows, &land_v))) return rc;
        char* land = static_cast<char*>(land_v);
        CR_HIP(hipMemcpyAsync(land, h->d_outs.p, b_outs, hipMemcpyDeviceToHost, stream));
        CR_HIP(hipMemcpyAsync(land + a_descs, h->d_nodes.p, b_descs, hipMemcpyDeviceToHost, stream));
        CR_HIP(hipMemcpyAsync(land + a_tail, d_meta.p + o_len, b_tail, hipMemcpyDeviceToHost, stream));
        if (b_rows) CR_HIP(hipMemcpyAsync(land + a_rows, b.aln.p, b_rows, hipMemcpyDeviceToHost, stream));
        CR_HIP(hipStreamSynchronize(stream));
        std::memcpy(outs.data(), land, b_outs);
        std::memcpy(descs.data(), land + a_descs, b_descs);
        std::memcpy(len.data(), land + a_tail, sizeof(int64_t) * nids);
        std::memcpy(off.data(), land + a_tail + (o_off - o_len), sizeof(int64_t) * nids);
        std::memcpy(&used, land + a_tail + (o_used - o_len), sizeof(int64_t));
        std::memcpy(&overflow, land + a_tail + (o_over - o_len), sizeof(int32_t));
        if (b_rows) std::memcpy(rows_host.data(), land + a_rows, b_rows);
    }
    *why = "a tree node outgrew the launch bound (1.5 x the longest structure, at most 2048 columns)";
    if (overflow) return 1;
    h->len = len;
    h->off = off;
    h->used = used;
    for (int64_t lv = 1; lv <= h->levels; lv++) {
        int64_t aln_off = aln_base[(size_t)lv];                 // the planning kernel's running offsets, replayed
        for (int64_t x = start[(size_t)lv]; x < start[(size_t)lv + 1]; x++) {
            const cr::PlanNode& pn = plan[(size_t)x];
            const int64_t k = pn.id - P, cap = len[(size_t)pn.c1] + len[(size_t)pn.c2];
            const cr::NodeOut& no = outs[(size_t)x];
            h->flags[(size_t)k] = no.flags;
            h->any_flags |= no.flags;
            std::vector<int32_t>& a = h->aln[(size_t)k];
            a.resize((size_t)(2 * no.len));
            const int32_t* src = rows_host.data() + aln_off;
            std::copy(src + no.first, src + no.first + no.len, a.begin());
            std::copy(src + cap + no.first, src + cap + no.first + no.len, a.begin() + no.len);
            aln_off += 2 * cap;
        }
    }
    return CR_OK;
}

}  // namespace

extern "C" {

static int progressive_align_impl(cr_context* ctx, const double* coords, const double* tensors, const int64_t* offsets, int64_t P,
                                  int64_t d, const uint64_t* tree, int64_t tree_rows, const cr_params* params,
                                  double consensus_weight, double gamma_weight, bool flexible, cr_progressive** out);

int cr_progressive_align(cr_context* ctx, const double* coords, const double* tensors, const int64_t* offsets,
                         int64_t P, int64_t d, const uint64_t* tree, int64_t tree_rows, const cr_params* params,
                         double consensus_weight, double gamma_weight, cr_progressive** out) {
    CR_REQUIRE(coords != nullptr, "null input");
    return progressive_align_impl(ctx, coords, tensors, offsets, P, d, tree, tree_rows, params, consensus_weight, gamma_weight, false, out);
}

int cr_progressive_align_flexible(cr_context* ctx, const double* tensors, const int64_t* offsets, int64_t P, int64_t d,
                                  const uint64_t* tree, int64_t tree_rows, const cr_params* params, double consensus_weight,
                                  double gamma_weight, cr_progressive** out) {
    return progressive_align_impl(ctx, nullptr, tensors, offsets, P, d, tree, tree_rows, params, consensus_weight, gamma_weight, true, out);
}

static int progressive_align_impl(cr_context* ctx, const double* coords, const double* tensors, const int64_t* offsets, int64_t P,
                                  int64_t d, const uint64_t* tree, int64_t tree_rows, const cr_params* params,
                                  double consensus_weight, double gamma_weight, bool flexible, cr_progressive** out) {
    CR_REQUIRE(out != nullptr, "null out");
    *out = nullptr;
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE((coords || flexible) && tensors && offsets && tree && params, "null input");
    CR_REQUIRE(P >= 2 && tree_rows == 2 * P - 3, "tree must have 2P-3 rows");
    CR_REQUIRE(d >= 1 && padded_width(d) != 0, "tensor width > 192 is not supported by this build");
    CR_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
    for (int64_t s = 0; s < P; s++)
        CR_REQUIRE(offsets[s + 1] > offsets[s] && offsets[s + 1] - offsets[s] <= cr::kMaxLength,
                   "every structure needs 1 .. 65534 residues");
    const cr_params prm = *params;
    // (flexible=True never forms a coordinate score: multiple_alignment.py:323-326 returns the tensor matrix alone)
    CR_REQUIRE(gamma_ok(prm.gamma_tensor) && (flexible || gamma_ok(prm.gamma_coords)) && std::isfinite(prm.gap_open) && std::isfinite(prm.gap_extend) &&
                   std::isfinite(prm.sw_gap),
               "parameters must be finite, gamma_tensor and gamma_coords >= 1e-290 (below that every score is exactly 1.0)");
    CR_REQUIRE(std::isfinite(gamma_weight) && gamma_weight >= 0.0 && std::isfinite(consensus_weight),
               "gamma_weight must be finite and >= 0, consensus_weight finite");
    const int64_t total = offsets[P];
    if (!flexible) CR_REQUIRE(all_finite(coords, (size_t)total * 3), "coordinates contain NaN or infinity");
    CR_REQUIRE(all_finite(tensors, (size_t)total * (size_t)d), "tensors contain NaN or infinity");

    // tree -> children of every internal node (ids P .. 2P-2), validated
    const int64_t num_ids = 2 * P - 1;
    cr_progressive* h = new (std::nothrow) cr_progressive();
    if (!h) return fail(CR_ERR_MEMORY, "out of host memory");
    struct Guard {
        cr_progressive* h;
        ~Guard() { delete h; }
    } guard{h};
    h->ctx = ctx;
    h->P = P;
    h->d = d;
    h->flexible = flexible;
    h->scratch.ctx = ctx;
    h->scratch.P = P;
    h->scratch.d = d;
    h->scratch.d_pad = padded_width(d);
    h->off.assign((size_t)num_ids, 0);
    h->len.assign((size_t)num_ids, 0);
    h->child1.assign((size_t)num_ids, -1);
    h->child2.assign((size_t)num_ids, -1);
    h->level.assign((size_t)num_ids, 0);
    h->members.assign((size_t)num_ids, 1);
    h->flags.assign((size_t)(P - 1), 0);
    h->aln.resize((size_t)(P - 1));
    std::vector<char> used_as_child((size_t)num_ids, 0);
    auto claim = [&](uint64_t c, int64_t parent) -> bool {
        if (c >= (uint64_t)parent || used_as_child[(size_t)c]) return false;
        used_as_child[(size_t)c] = 1;
        return true;
    };
    for (int64_t k = 0; k < P - 1; k++) {
        const int64_t id = P + k;
        uint64_t c1, c2;
        if (k < P - 2) {
            c1 = tree[(2 * k) * 2];
            c2 = tree[(2 * k + 1) * 2];
            CR_REQUIRE(tree[(2 * k) * 2 + 1] == (uint64_t)id && tree[(2 * k + 1) * 2 + 1] == (uint64_t)id,
                       "tree rows 2x, 2x+1 must both name node P + x as the parent");
        } else {
            c1 = tree[(tree_rows - 1) * 2];
            c2 = tree[(tree_rows - 1) * 2 + 1];
        }
        CR_REQUIRE(claim(c1, id) && claim(c2, id), "tree: child id out of order or joined twice");
        h->child1[(size_t)id] = (int64_t)c1;
        h->child2[(size_t)id] = (int64_t)c2;
        h->level[(size_t)id] = 1 + std::max(h->level[(size_t)c1], h->level[(size_t)c2]);
        h->members[(size_t)id] = h->members[(size_t)c1] + h->members[(size_t)c2];
        h->levels = std::max(h->levels, h->level[(size_t)id]);
    }
    CR_REQUIRE(h->members[(size_t)(num_ids - 1)] == P, "tree does not join every structure");

    // leaves into the arena
    for (int64_t s = 0; s < P; s++) {
        h->off[(size_t)s] = offsets[s];
        h->len[(size_t)s] = offsets[s + 1] - offsets[s];
    }
    h->used = total;
    if ((rc = arena_reserve(h, total + total / 2 + 2 * cr::kWave))) return rc;
    {
        std::vector<double> w((size_t)total, consensus_weight);
        hipStream_t st = ctx->stream;
        if (!flexible) {
            rc = upload_async(ctx, h->scratch.coords.p, coords, sizeof(double) * (size_t)total * 3);
            if (rc) return rc;
        }
        rc = upload_async(ctx, h->scratch.tensors.p, tensors, sizeof(double) * (size_t)(total * d));
        if (rc) return rc;
        rc = upload_async(ctx, h->weights.p, w.data(), sizeof(double) * (size_t)total);
        if (rc) return rc;
        CR_HIP(hipStreamSynchronize(st));
    }
    std::vector<std::vector<int64_t>> by_level((size_t)h->levels + 1);
    for (int64_t id = P; id < num_ids; id++) by_level[(size_t)h->level[(size_t)id]].push_back(id);
    const char* why = "CARETTA_SYNC_LEVELS is set";
    rc = g_cfg.sync_levels ? 1 : run_tree_planned(h, by_level, prm, gamma_weight, &why);
    if (rc < 0) return rc;
    if (rc == 1 && flexible)                         // (CR_ERR_STATE = "not served by the device path": the host module then walks the tree itself)
        return fail(CR_ERR_STATE, std::string("flexible progressive alignment is not served by the resident-tree path here: ") + why);
    if (rc == 1) {                                  // not applicable, or a node outgrew the bound: level by level
        h->used = total;
        h->any_flags = 0;
        for (int64_t lv = 1; lv <= h->levels; lv++)
            if ((rc = run_level(h, by_level[(size_t)lv], prm, gamma_weight))) return rc;
    }
    guard.h = nullptr;
    *out = h;
    return CR_OK;
}

int cr_progressive_sizes(cr_progressive* h, int64_t sizes[5]) {
    CR_REQUIRE(h != nullptr && sizes != nullptr, "null argument");
    int64_t sum = 0;
    for (int64_t id = h->P; id < 2 * h->P - 1; id++) sum += h->len[(size_t)id];
    sizes[0] = h->len[(size_t)(2 * h->P - 2)];
    sizes[1] = h->P - 1;
    sizes[2] = sum;
    sizes[3] = h->levels;
    sizes[4] = (int64_t)h->any_flags;
    return CR_OK;
}

int cr_progressive_fetch_msa(cr_progressive* h, int64_t* msa) {
    CR_REQUIRE(h != nullptr && msa != nullptr, "null argument");
    // top down: colmap[id][x] = column of node id shown in final column x, or -1 (multiple_alignment.py:218-229
    // re-indexes every member row bottom up; composing the maps from the root gives the same rows)
    const int64_t root = 2 * h->P - 2, L = h->len[(size_t)root];
    std::vector<std::vector<int32_t>> colmap((size_t)(2 * h->P - 1));
    colmap[(size_t)root].resize((size_t)L);
    for (int64_t x = 0; x < L; x++) colmap[(size_t)root][(size_t)x] = (int32_t)x;
    for (int64_t id = root; id >= h->P; id--) {
        const std::vector<int32_t>& mine = colmap[(size_t)id];
        const std::vector<int32_t>& a = h->aln[(size_t)(id - h->P)];
        const int64_t ln = h->len[(size_t)id];
        for (int side = 0; side < 2; side++) {
            const int64_t c = side == 0 ? h->child1[(size_t)id] : h->child2[(size_t)id];
            std::vector<int32_t>& cm = colmap[(size_t)c];
            cm.resize((size_t)L);
            const int32_t* row = a.data() + side * ln;
            for (int64_t x = 0; x < L; x++) cm[(size_t)x] = mine[(size_t)x] < 0 ? -1 : row[mine[(size_t)x]];
        }
        colmap[(size_t)id] = std::vector<int32_t>();
    }
    for (int64_t s = 0; s < h->P; s++)
        for (int64_t x = 0; x < L; x++) msa[s * L + x] = colmap[(size_t)s][(size_t)x];
    return CR_OK;
}

int cr_progressive_node_table(cr_progressive* h, int64_t* table) {
    CR_REQUIRE(h != nullptr && table != nullptr, "null argument");
    for (int64_t k = 0; k < h->P - 1; k++) {
        const int64_t id = h->P + k;
        int64_t* t = table + 6 * k;
        t[0] = h->child1[(size_t)id];
        t[1] = h->child2[(size_t)id];
        t[2] = h->len[(size_t)id];
        t[3] = h->level[(size_t)id];
        t[4] = (int64_t)h->flags[(size_t)k];
        t[5] = h->members[(size_t)id];
    }
    return CR_OK;
}

int cr_progressive_fetch_nodes(cr_progressive* h, int64_t* aln, double* coords, double* tensors, double* weights) {
    CR_REQUIRE(h != nullptr, "null argument");
    int rc = set_device(h->ctx);
    if (rc) return rc;
    if (aln) {
        int64_t o = 0;
        for (int64_t k = 0; k < h->P - 1; k++)
            for (int32_t v : h->aln[(size_t)k]) aln[o++] = v;
    }
    if (!coords && !tensors && !weights) return CR_OK;
    // the arena comes back in one copy per array; nodes are sliced out on the host
    std::vector<double> host;
    auto slice = [&](const double* dev, int64_t width, double* dst) -> int {
        host.resize((size_t)(h->used * width));
        int drc = download(h->ctx, host.data(), dev, sizeof(double) * (size_t)(h->used * width));
        if (drc) return drc;
        int64_t o = 0;
        for (int64_t id = h->P; id < 2 * h->P - 1; id++) {
            const int64_t cnt = h->len[(size_t)id] * width;
            std::memcpy(dst + o, host.data() + h->off[(size_t)id] * width, sizeof(double) * (size_t)cnt);
            o += cnt;
        }
        return CR_OK;
    };
    CR_HIP(hipStreamSynchronize(h->ctx->stream));
    CR_REQUIRE(!(coords && h->flexible), "a flexible progressive alignment has no node coordinates");
    if (coords && (rc = slice(h->scratch.coords.p, 3, coords))) return rc;
    if (tensors && (rc = slice(h->scratch.tensors.p, h->d, tensors))) return rc;
    if (weights && (rc = slice(h->weights.p, 1, weights))) return rc;
    return CR_OK;
}

int cr_progressive_destroy(cr_progressive* h) {
    if (!h) return CR_OK;
    (void)hipSetDevice(h->ctx->device);
    delete h;
    return CR_OK;
}

}  // extern "C"
