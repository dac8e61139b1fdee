// Tracebacks on the packed decisions (Walker), position-ordered sums, Kabsch, RMSD / TM; the walks of both stages.
// Part of cr_kernels.h (included there, inside namespace cr, in this order: cr_providers.h, cr_sweep.h, cr_sweep_cols.h,
// cr_sweep_wide.h, cr_trace.h, cr_pair_kernels.h); not a header of its own.

CR_D uint32_t lookup_bits(const uint32_t* __restrict__ words, int R, int TB, int per_word_log2, int bits,
                          int row, int col) {
    const int s = row / (kWave * R);
    const int rem = row - s * kWave * R;
    const int l = rem / R;
    const int q = rem - l * R;
    const int t = col + l;
    const uint32_t w = words[((int64_t)(s * TB + (t >> per_word_log2)) * R + q) * kWave + l];
    return (w >> ((t & ((1 << per_word_log2) - 1)) * bits)) & ((1u << bits) - 1u);
}

// dynamic_time_warping.py:90-144 _get_dtw_alignment on packed decisions.  Writes the alignment rows
// back-to-front into a1/a2[cap-1 .. cap-len] and returns len.
CR_D int dtw_traceback(const uint32_t* __restrict__ w, int R, int TB, int n, int m, int dir,
                       int32_t* __restrict__ a1, int32_t* __restrict__ a2, int cap) {
    int idx = 0;
    int guard = 3 * cap + 8;
    while (!(n == 0 && m == 0) && guard-- > 0) {
        if (m == 0) {
            n--; idx++;
            a1[cap - idx] = n; a2[cap - idx] = -1;
        } else if (n == 0) {
            m--; idx++;
            a1[cap - idx] = -1; a2[cap - idx] = m;
        } else {
            const uint32_t nib = lookup_bits(w, R, TB, 3, 4, n - 1, m - 1);
            if (dir == 0) {
                dir = nib & 1u;
                n--; idx++;
                a1[cap - idx] = n; a2[cap - idx] = -1;
            } else if (dir == 1) {
                dir = (nib >> 1) & 3u;
                if (dir == 1) {
                    n--; m--; idx++;
                    a1[cap - idx] = n; a2[cap - idx] = m;
                }
            } else {
                dir = ((nib >> 3) & 1u) + 1;
                m--; idx++;
                a1[cap - idx] = -1; a2[cap - idx] = m;
            }
        }
    }
    return idx;
}

// dynamic_time_warping.py:249-278: smith_waterman traceback with gap entries, back-to-front.
CR_D int sw_traceback(const uint32_t* __restrict__ w, int R, int TB, int i, int j,
                      int32_t* __restrict__ a1, int32_t* __restrict__ a2, int cap) {
    int idx = 0;
    while (i > 0 && j > 0) {
        const uint32_t code = lookup_bits(w, R, TB, 4, 2, i - 1, j - 1);
        if (code == 0) break;
        idx++;
        if (code == 1) {
            i--; j--;
            a1[cap - idx] = i; a2[cap - idx] = j;
        } else if (code == 2) {
            j--;
            a1[cap - idx] = -1; a2[cap - idx] = j;
        } else {
            i--;
            a1[cap - idx] = i; a2[cap - idx] = -1;
        }
    }
    return idx;
}

struct PairResult {          // per-pair scalar outputs, device and host layout
    double sw, dtw_score;
    double R[9], t[3];
    double rmsd, coverage, tm;
    double seed_score;
    int32_t aln_len, aln_start;
    int32_t seed_len;
    uint32_t flags;
};

// Results that leave the device from the kernel that produces them (cr_batch_run_stream_i32): page-locked host arrays
// in the caller's layout, written over PCIe by the wave that finished the pair -- 256-byte coalesced stores, posted, under
// the fills of the other waves -- so that the download costs no time after the last kernel.  All null: nothing streamed.
struct HostOut {
    int32_t* aln;            // [npairs][2][stride], rows left-aligned (cr_batch_fetch_i32's layout without the -2 padding)
    int64_t stride;
    PairResult* res;         // [npairs]
    const int32_t* order;    // launch slot -> index in the caller's pair list (null: identity)
    int32_t first;           // launch slot of block 0 of this launch
    int32_t pad;
    CR_D int dst(int block) const { return order ? order[first + block] : first + block; }
};

// ---------------------------------------------------------------------------------------------
// Traceback + superposition stages.  ONE WAVE PER PAIR.
//
// The walk is a single logical thread, so it is written wave-uniform (every lane carries the same
// state; the compiler keeps it in SGPRs); its decision lookups come out of a register-resident block of words
// (Walker, below).  Emitted alignment columns go to LDS as packed (i, j) 16-bit pairs and are written to
// HBM at the end with coalesced stores.  The aligned positions are then gathered 64 at a time by
// all lanes, per-position terms are computed in parallel, and the sums are taken by one lane per
// accumulator IN POSITION ORDER out of LDS, so every sum has the reference's (numba's) sequential
// rounding.  Gap columns contribute +0.0 terms, which never change a running sum that started at
// +0.0 (such a sum can not be -0.0).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxAcc = 9;              // accumulators summed in order (3x3 correlation matrix)
constexpr uint32_t kGap16 = 0xffffu;    // -1 in a packed 16-bit alignment entry
constexpr int kMaxLength = 65534;       // longest structure the packed entries can index

CR_D uint32_t pack_entry(int i, int j) { return ((uint32_t)i & 0xffffu) | ((uint32_t)j << 16); }

// ---------------------------------------------------------------------------------------------
// Walk-side view of the packed decisions of one pair (BITS = 2: SW, 16 steps per word; BITS = 4: DTW, 8 per word).
//
// A walk is one logical thread chasing a chain of dependent lookups, so everything that can be taken off that
// chain is: the wave keeps, in ONE VGPR, the decision words of a block of kBlockRows consecutive DP rows x kBlockWords
// consecutive words per row (lane 4a + w: row r0 - a, word (c0 + lane_of_row) / steps_per_word - w), gathered straight
// from L2/HBM with a single global load; a lookup inside the block is one v_readlane (no memory access), and the block
// covers every path that leaves the anchor cell (r0, c0) going up, diagonally, or up to ~25 (DTW) / ~50 (SW) columns
// to the left per row.  Whole DIAGONAL RUNS are resolved at once: every lane tests the cell of its row on the diagonal
// through the current cell, one ballot gives the run length, and the run's alignment entries are emitted by the lanes
// in parallel -- on structural alignments most columns are aligned pairs, so the walk advances by up to 16 cells per
// iteration.  Row bookkeeping (strip, fill lane, row slot) is wave-uniform and lives in SGPRs.
// SKEW = 1: words written by the time-skewed sweeps (time step of a cell = column + fill lane); SKEW = 0: words of
// the column sweep (time step = column).
// ---------------------------------------------------------------------------------------------
template <int R, int BITS, int SKEW = 1, int RB = R>
struct Walker {
    static constexpr int kLog = BITS == 2 ? 4 : 3;                 // log2(steps per word)
    static constexpr int kStepMask = (1 << kLog) - 1;
    static constexpr uint32_t kFieldMask = (1u << BITS) - 1u;
    static constexpr int kBlockRows = 16, kBlockWords = 4;
    static constexpr bool kMixed = RB != R;    // strips [0, nA) have R rows per lane, the others RB (WidePlan)
    const uint32_t* __restrict__ words;
    int TB, nA;
    int ax, wx;               // per lane: row offset and word slot held by this lane
    uint32_t blk;             // per lane: the word
    int lax;                  // per lane: fill lane of this lane's row
    int r0, c0, bs, amax;     // block key (wave-uniform): anchor cell, strip (-1: empty), deepest row offset held
    int s, l, q;              // position of the current row (wave-uniform): strip, fill lane, row slot
    int rs, base, slot0;      // of strip s (wave-uniform): rows per lane, first row, row slots before it
    CR_D void init(const uint32_t* __restrict__ w, int tb, int lane, int na = 0) {
        words = w;
        TB = tb;
        nA = na;
        ax = lane >> 2;
        wx = lane & 3;
        blk = 0;
        lax = 0;
        r0 = c0 = 0;
        bs = -1;                  // no strip: the first lookup fills the block
        amax = -1;
        s = l = q = 0;
        rs = R;
        base = slot0 = 0;
    }
    CR_D void set_row(int row) {
        if (!kMixed || row < nA * (kWave * R)) {
            s = row / (kWave * R);
            const int rem = row - s * (kWave * R);
            l = rem / R;
            q = rem - l * R;
            rs = R;
            base = s * (kWave * R);
            slot0 = s * R;
        } else {
            const int x = row - nA * (kWave * R);
            const int sb = x / (kWave * RB);
            const int rem = x - sb * (kWave * RB);
            l = rem / RB;
            q = rem - l * RB;
            s = nA + sb;
            rs = RB;
            base = nA * (kWave * R) + sb * (kWave * RB);
            slot0 = nA * R + sb * RB;
        }
    }
    // the word of lane (ax, wx) in the block of the current strip anchored at (r, c); `la_out`: the fill lane of its row
    CR_D uint32_t load_block(int r, int c, int& la_out) const {
        const int rel = r - ax - base;                    // this lane's row, relative to the strip
        const bool rv = rel >= 0;
        const int relc = rv ? rel : 0;
        const int la = (!kMixed || rs == R) ? relc / R : relc / RB;
        const int qa = relc - la * rs;
        la_out = la;
        const int tb = ((c + la * SKEW) >> kLog) - wx;
        return (rv && tb >= 0) ? words[((int64_t)slot0 * TB + (int64_t)tb * rs + qa) * kWave + la] : 0u;
    }
    // (Requesting the block above along the diagonal while the walk crosses this one was measured: 225 k -> 218 k cycles
    // per 1200-row walk, and 2 % more time for the headline kernels -- a walk step is bound by its ~100 dependent scalar
    // instructions, not by the load; not kept.)
    CR_D void refill(int r, int c) {
        r0 = r;
        c0 = c;
        bs = s;
        amax = r - base < kBlockRows - 1 ? r - base : kBlockRows - 1;
        blk = load_block(r, c, lax);
    }
    // decision field of cell (r, c); (s, l, q) must be the position of row r
    CR_D uint32_t get(int r, int c) {
        int a = r0 - r;
        int w = ((c0 + l * SKEW) >> kLog) - ((c + l * SKEW) >> kLog);
        if (!(s == bs && a <= amax && w < kBlockWords)) {
            refill(r, c);
            a = 0;
            w = 0;
        }
        const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)blk, a * 4 + w);
        return (word >> (((c + l * SKEW) & kStepMask) * BITS)) & kFieldMask;
    }
    // Number of consecutive cells (r - k, c - k * DC), k = 0, 1, ..., whose decision field satisfies `pred`, as far as the
    // block holds them (DC = 1: a diagonal run, DC = 0: a vertical one).  `more`: the cell behind the run is in the block
    // too (so the run ended because that cell's field does not satisfy `pred`, not because the block did).
    template <int DC, class Pred>
    CR_D int run_up(int r, int c, Pred pred, bool& more) {
        const int a_cur = r0 - r;
        const int k = ax - a_cur;
        const int col = c - k * DC;
        const int t = col + lax * SKEW;
        const int wneed = ((c0 + lax * SKEW) >> kLog) - (t >> kLog);
        const uint32_t f = (blk >> ((t & kStepMask) * BITS)) & kFieldMask;
        const bool have = k >= 0 && ax <= amax && col >= 0 && wneed == wx;
        uint64_t mh = __ballot(have), mk = __ballot(have && pred(f));
        mh = (mh | (mh >> 1) | (mh >> 2) | (mh >> 3)) & 0x1111111111111111ull;     // bit 4a: row a's cell is held
        mk = (mk | (mk >> 1) | (mk >> 2) | (mk >> 3)) & 0x1111111111111111ull;     // bit 4a: ... and continues the run
        const uint64_t stop = ~(mk >> (4 * a_cur)) & 0x1111111111111111ull;
        const int rows = __builtin_amdgcn_readfirstlane(stop ? (__builtin_ctzll(stop) >> 2) : 16);
        more = a_cur + rows < 16 && ((mh >> (4 * (a_cur + rows))) & 1ull);
        return rows;
    }
    template <class Pred>
    CR_D int diag_run(int r, int c, Pred diag) {
        bool more;
        return run_up<1>(r, c, diag, more);
    }
    // The same along the row: cells (r, c - k), k = 0, 1, ... (a horizontal gap run).  Lane k looks at cell k: the word it
    // needs is one of the four the block holds for row r and comes over with one ds_bpermute.  (s, l, q) must be the
    // position of row r and the block must hold (r, c) -- the caller has just read it.
    template <class Pred>
    CR_D int run_left(int r, int c, Pred pred, bool& more) {
        const int lane = ax * 4 + wx;
        const int a_cur = r0 - r;
        const int top = (c0 + l * SKEW) >> kLog;              // newest word the block holds for this row
        const int col = c - lane;
        const int t = col + l * SKEW;
        const int w = top - (t >> kLog);
        const bool have = col >= 0 && w < kBlockWords;
        const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute((a_cur * 4 + (have ? w : 0)) * 4, (int)blk);
        const uint32_t f = (word >> ((t & kStepMask) * BITS)) & kFieldMask;
        const uint64_t mh = __ballot(have), mk = __ballot(have && pred(f));
        const int cells = __builtin_amdgcn_readfirstlane(~mk ? __builtin_ctzll(~mk) : 64);
        more = cells < 64 && ((mh >> cells) & 1ull);
        return cells;
    }
};

// Sum `count` per-position term vectors in position order.  term(e, out[NACC]) is evaluated by the
// lane that owns position e; lane a < NACC returns sum_e term(e)[a] accumulated e = 0, 1, 2, ...
// (exactly the rounding sequence of a sequential loop).  `scratch` = 64 * NACC doubles of LDS.
template <int NACC, class TermFn>
CR_D double ordered_sums(int count, int lane, double* scratch, TermFn term) {
    double acc = 0.0;
    for (int base = 0; base < count; base += kWave) {
        const int e = base + lane;
        if (e < count) {
            double tv[NACC];
            term(e, tv);
#pragma unroll
            for (int a = 0; a < NACC; a++) scratch[lane * NACC + a] = tv[a];
        }
        wave_sync();
        const int cnt = count - base < kWave ? count - base : kWave;
        if (lane < NACC) {
#pragma unroll 8
            for (int x = 0; x < cnt; x++) acc += scratch[x * NACC + lane];
        }
        wave_sync();
    }
    return acc;
}

// The coordinates of one alignment column (a packed entry): both residues, or pair = false for a gap column (then the
// values are those of residue 0 and must not be used).
struct ColumnXYZ {
    double a[3], b[3];       // residue of X_i, residue of X_j
    bool pair;
};

// Issue the loads of column e (clamped: every lane loads, lanes past `count` get pair = false) -- no arithmetic on the
// loaded values here, so the wait for them sits at their first use.
CR_D ColumnXYZ load_column(const double* __restrict__ Xi, const double* __restrict__ Xj, const uint32_t* entries, int e, int count) {
    ColumnXYZ c;
    const bool in = e < count;
    const uint32_t u = entries[in ? e : 0];
    const uint32_t i = u & 0xffffu, j = u >> 16;
    c.pair = in && i != kGap16 && j != kGap16;
    const double* v1 = Xi + (int64_t)(c.pair ? i : 0) * 3;
    const double* v2 = Xj + (int64_t)(c.pair ? j : 0) * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        c.a[k] = v1[k];
        c.b[k] = v2[k];
    }
    return c;
}

// Sum per-column term vectors over `count` packed entries in position order.  term(column, out[NACC]) is evaluated by the
// lane that owns the column; lane a < NACC returns sum_e term(e)[a] accumulated e = 0, 1, 2, ... (exactly the rounding
// sequence of a sequential loop).  `scratch` = 64 * NACC doubles of LDS.  The coordinates of the NEXT 64 columns are
// requested before the 64 dependent additions of the current ones, so that the gather's trip to L2 / HBM (about as long
// as the chain) is hidden behind it -- a lone wave per SIMD (one pair per CU) has nobody else to hide it.
template <int NACC, class TermFn>
CR_D double ordered_sums(const double* __restrict__ Xi, const double* __restrict__ Xj, const uint32_t* entries, int count,
                         int lane, double* scratch, TermFn term) {
    double acc = 0.0;
    ColumnXYZ cur = load_column(Xi, Xj, entries, lane, count);
    for (int base = 0; base < count; base += kWave) {
        const int e = base + lane;
        if (e < count) {
            double tv[NACC];
            term(cur, tv);
#pragma unroll
            for (int a = 0; a < NACC; a++) scratch[lane * NACC + a] = tv[a];
        }
        wave_sync();
        const ColumnXYZ nxt = load_column(Xi, Xj, entries, e + kWave, count);
        const int cnt = count - base < kWave ? count - base : kWave;
        if (lane < NACC) {
#pragma unroll 8
            for (int x = 0; x < cnt; x++) acc += scratch[x * NACC + lane];
        }
        wave_sync();
        cur = nxt;
    }
    return acc;
}

// Kabsch over `count` packed alignment entries of which `k` are aligned pairs
// (superposition_functions.py:7-35), every sum in position order.  Results in all lanes.
CR_D void kabsch_ordered(const double* __restrict__ Xi, const double* __restrict__ Xj, const uint32_t* entries,
                         int count, int k, int lane, double* scratch, double* c1, double* c2, double* R, double* t) {
    // column means (helper.py:46-53): lanes 0-2 sum X_i columns, lanes 3-5 X_j columns
    const double msum = ordered_sums<6>(Xi, Xj, entries, count, lane, scratch, [&](const ColumnXYZ& c, double* out) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            out[a] = c.pair ? c.a[a] : 0.0;
            out[3 + a] = c.pair ? c.b[a] : 0.0;
        }
    });
    const double mean = msum / (double)k;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        c1[a] = lane_value(mean, a);
        c2[a] = lane_value(mean, 3 + a);
    }
    // correlation matrix C = (X_j - c2)^T (X_i - c1)  (superposition_functions.py:26-27)
    const double csum = ordered_sums<9>(Xi, Xj, entries, count, lane, scratch, [&](const ColumnXYZ& col, double* out) {
        const double a[3] = {col.b[0] - c2[0], col.b[1] - c2[1], col.b[2] - c2[2]};
        const double b[3] = {col.a[0] - c1[0], col.a[1] - c1[1], col.a[2] - c1[2]};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) out[3 * r + c] = col.pair ? a[r] * b[c] : 0.0;
    });
    double C[9];
#pragma unroll
    for (int a = 0; a < 9; a++) C[a] = lane_value(csum, a);
    kabsch_from_correlation(C, c1, c2, R, t);     // every lane computes the same 3x3 SVD
}

// LDS carve-up of a traceback stage: [entries: max_entries words][sum scratch]
__host__ __device__ inline size_t trace_lds_doubles(int /*R*/, int max_entries) {
    return ((size_t)max_entries + 3) / 4 * 2 + (size_t)kWave * kMaxAcc;
}

// Stage 2: SW traceback on the stored decisions, common positions, seed Kabsch
// (dynamic_time_warping.py:249-278, helper.py:13-42, superposition_functions.py:39-60).
// Wave-uniform; `lds` is this stage's LDS.  Returns the transform in every lane.
// The walk of stage 2 alone: the aligned pairs into plist[cap - k, cap) (cap = min(n, m)), their number, the length of the
// local alignment, kFlagSeedAllZero.  One wave.
template <int R, int SKEW = 1, int RB = R>
CR_D void seed_walk(const PairDesc& pd, const uint32_t* __restrict__ dirs, const SeedMax sm, uint32_t* plist, const int nA,
                    int& k_out, int& len_out, uint32_t& flags_out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int cap = pd.n < pd.m ? pd.n : pd.m;
    uint32_t flags = 0;
    int k = 0, len = 0;
    if (sm.i == 0) {
        flags |= kFlagSeedAllZero;
    } else {
        Walker<R, 2, SKEW, RB> wk;
        wk.init(dirs + pd.dirs_off, SKEW ? tblocks(pd.m, 16) : (pd.m + 15) >> 4, lane, nA);
        // the walk is wave-uniform: pin its state to SGPRs so that it compiles to scalar code
        int i = __builtin_amdgcn_readfirstlane(sm.i), j = __builtin_amdgcn_readfirstlane(sm.j);
        wk.set_row(i - 1);
#pragma unroll 1
        while (i > 0 && j > 0) {
            const uint32_t code = wk.get(i - 1, j - 1);
            if (code == 0) break;
            if (code == 1) {                                     // a run of aligned pairs: all of it at once
                const int run = wk.diag_run(i - 1, j - 1, [](uint32_t f) { return f == 1u; });
                if (lane < run) plist[cap - k - 1 - lane] = pack_entry(i - 1 - lane, j - 1 - lane);
                k += run;
                len += run;
                i -= run;
                j -= run;
                if (i > 0) wk.set_row(i - 1);
            } else if (code == 2) {                              // a run of gaps along the row: all of it at once
                bool more;
                const int run = wk.run_left(i - 1, j - 1, [](uint32_t f) { return f == 2u; }, more);
                len += run;
                j -= run;
            } else {                                             // ... and along the column
                bool more;
                const int run = wk.template run_up<0>(i - 1, j - 1, [](uint32_t f) { return f == 3u; }, more);
                len += run;
                i -= run;
                if (i > 0) wk.set_row(i - 1);
            }
        }
    }
    k_out = k;
    len_out = len;
    flags_out = flags;
}

template <int R, int SKEW = 1, int RB = R>
CR_D void seed_trace(const PairDesc& pd, int max_entries, const double* __restrict__ coords,
                     const uint32_t* __restrict__ dirs, const SeedMax sm, double* lds, Transform& tr, const int nA = 0) {
    const int lane = threadIdx.x;
    uint32_t* plist = reinterpret_cast<uint32_t*>(lds);          // aligned pairs, filled back-to-front
    double* scratch = lds + ((size_t)max_entries + 3) / 4 * 2;   // 16-byte aligned, after the list
    const int cap = pd.n < pd.m ? pd.n : pd.m;
    uint32_t flags = 0;
    int k = 0, len = 0;
    seed_walk<R, SKEW, RB>(pd, dirs, sm, plist, nA, k, len, flags);
    wave_sync();
    CR_STAMP(2);
#pragma unroll
    for (int x = 0; x < 3; x++) tr.c1[x] = tr.c2[x] = 0.0;
#pragma unroll
    for (int x = 0; x < 9; x++) tr.R[x] = (x % 4 == 0) ? 1.0 : 0.0;
    if (k <= 3) {
        flags |= kFlagSeedSkipped;
    } else {
        double t[3];
        kabsch_ordered(coords + pd.off_i * 3, coords + pd.off_j * 3, plist + (cap - k), k, k, lane, scratch,
                       tr.c1, tr.c2, tr.R, t);
    }
    tr.flags = flags;
    tr.seed_len = len;
}

// get_rmsd (score_functions.py:15-19) and tm_score (multiple_alignment.py:59-70) over `count` packed entries of
// which `k` are aligned pairs, sums in position order: lane 0 sums the squared differences (three per
// position), lanes 1/2 the two TM sums.  MOVE: compare X_i with X_j @ R + t, else with X_j as it is.
template <bool MOVE>
CR_D void rmsd_tm_ordered(const double* __restrict__ Xi, const double* __restrict__ Xj, const uint32_t* ent,
                          int count, int k, int len1, int len2, const double* R, const double* t, int lane,
                          double* scratch, double& rmsd, double& tm) {
    const double d1 = 1.24 * (double)(len1 - 15) / 3.0 - 1.8;
    const double d2 = 1.24 * (double)(len2 - 15) / 3.0 - 1.8;
    double acc = 0.0;
    ColumnXYZ cur = load_column(Xi, Xj, ent, lane, count);
    for (int base = 0; base < count; base += kWave) {
        const int x = base + lane;
        if (x < count) {
            const bool pair = cur.pair;
            double mv[3] = {cur.b[0], cur.b[1], cur.b[2]};
            if constexpr (MOVE) {
                rot3(cur.b, R, mv);
                mv[0] = mv[0] + t[0];
                mv[1] = mv[1] + t[1];
                mv[2] = mv[2] + t[2];
            }
            const double e0 = cur.a[0] - mv[0], e1 = cur.a[1] - mv[1], e2 = cur.a[2] - mv[2];
            const double sg = (e0 + e1) + e2;
            const double q1 = sg / d1, q2 = sg / d2;
            scratch[lane * 5 + 0] = pair ? e0 * e0 : 0.0;
            scratch[lane * 5 + 1] = pair ? e1 * e1 : 0.0;
            scratch[lane * 5 + 2] = pair ? e2 * e2 : 0.0;
            scratch[lane * 5 + 3] = pair ? 1.0 / (1.0 + q1 * q1) : 0.0;
            scratch[lane * 5 + 4] = pair ? 1.0 / (1.0 + q2 * q2) : 0.0;
        }
        wave_sync();
        const ColumnXYZ nxt = load_column(Xi, Xj, ent, x + kWave, count);     // in flight during the chain below
        const int cnt = count - base < kWave ? count - base : kWave;
        if (lane == 0) {
            for (int y = 0; y < cnt; y++) {
                acc += scratch[y * 5 + 0];
                acc += scratch[y * 5 + 1];
                acc += scratch[y * 5 + 2];
            }
        } else if (lane < 3) {
#pragma unroll 8
            for (int y = 0; y < cnt; y++) acc += scratch[y * 5 + 2 + lane];
        }
        wave_sync();
        cur = nxt;
    }
    const double ss = lane_value(acc, 0), sum1 = lane_value(acc, 1), sum2 = lane_value(acc, 2);
    rmsd = sqrt(ss / (double)k);
    const double t1 = (1.0 / (double)len1) * sum1;
    const double t2 = (1.0 / (double)len2) * sum2;
    tm = t1 > t2 ? t1 : t2;
}

// ---------------------------------------------------------------------------------------------
// The same sums with the whole WORKGROUP at work (one pair per workgroup: the wide layout).  After a fill all waves of the
// workgroup are still there and wave 0 has walked: every thread forms the terms of its columns -- kSumTile columns per
// round, each thread gathering the coordinates of its own -- into LDS, then ONE thread per accumulator adds the round's
// terms in position order (the rounding sequence of the sequential loop, as above).  What is left on the critical path
// is the chain of dependent additions itself; the gathers of all columns are in flight together.
// `terms`: kSumTile * kMaxAcc doubles of LDS; `red`: 16 doubles.  Every thread of the workgroup must call these (they
// contain barriers); results in every thread.
// ---------------------------------------------------------------------------------------------
constexpr int kSumTile = 1024;
constexpr int kSumSlack = 8 * kMaxAcc;     // doubles behind the term tile that chain_sum may read (never add)

// acc + p[0] + p[stride] + ... + p[(cnt - 1) * stride], added in this order by ONE thread.  The chain of dependent
// additions is the critical path of a sum that has to round like a sequential loop; the LDS reads are kept off it: two
// register blocks of 8 in turn, each read one block ahead of its additions (the last read-ahead runs up to 8 elements
// past the end: read, never added).
CR_D double chain_sum(const double* p, int stride, int cnt, double acc) {
    int x = 0;
    if (cnt >= 16) {
        double a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = p[k * stride];
        for (; x + 16 <= cnt; x += 16) {
#pragma unroll
            for (int k = 0; k < 8; k++) b[k] = p[(x + 8 + k) * stride];
            __builtin_amdgcn_sched_barrier(0);          // (the scheduler would sink the reads below the adds)
#pragma unroll
            for (int k = 0; k < 8; k++) acc += a[k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = p[(x + 16 + k) * stride];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k++) acc += b[k];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (; x < cnt; x++) acc += p[x * stride];
    return acc;
}

template <int NACC, class TermFn>
CR_D void ordered_sums_team(const double* __restrict__ Xi, const double* __restrict__ Xj, const uint32_t* entries, int count,
                            double* terms, double* red, TermFn term) {
    const int tid = threadIdx.x, nth = blockDim.x;
    double acc = 0.0;
    for (int base = 0; base < count; base += kSumTile) {
        const int cnt = count - base < kSumTile ? count - base : kSumTile;
        for (int e = tid; e < cnt; e += nth) {
            const ColumnXYZ col = load_column(Xi, Xj, entries, base + e, count);
            double tv[NACC];
            term(col, tv);
#pragma unroll
            for (int a = 0; a < NACC; a++) terms[e * NACC + a] = tv[a];
        }
        __syncthreads();
        if (tid < NACC) acc = chain_sum(terms + tid, NACC, cnt, acc);
        __syncthreads();
    }
    if (tid < NACC) red[tid] = acc;
    __syncthreads();
}

CR_D void kabsch_team(const double* __restrict__ Xi, const double* __restrict__ Xj, const uint32_t* entries, int count, int k,
                      double* terms, double* red, double* c1, double* c2, double* R, double* t) {
    ordered_sums_team<6>(Xi, Xj, entries, count, terms, red, [&](const ColumnXYZ& c, double* out) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            out[a] = c.pair ? c.a[a] : 0.0;
            out[3 + a] = c.pair ? c.b[a] : 0.0;
        }
    });
#pragma unroll
    for (int a = 0; a < 3; a++) {
        c1[a] = red[a] / (double)k;
        c2[a] = red[3 + a] / (double)k;
    }
    __syncthreads();                                   // `red` is written again below
    ordered_sums_team<9>(Xi, Xj, entries, count, terms, red, [&](const ColumnXYZ& col, double* out) {
        const double a[3] = {col.b[0] - c2[0], col.b[1] - c2[1], col.b[2] - c2[2]};
        const double b[3] = {col.a[0] - c1[0], col.a[1] - c1[1], col.a[2] - c1[2]};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) out[3 * r + c] = col.pair ? a[r] * b[c] : 0.0;
    });
    double C[9];
#pragma unroll
    for (int a = 0; a < 9; a++) C[a] = red[a];
    __syncthreads();
    kabsch_from_correlation(C, c1, c2, R, t);          // every thread computes the same 3x3 SVD
}

template <bool MOVE>
CR_D void rmsd_tm_team(const double* __restrict__ Xi, const double* __restrict__ Xj, const uint32_t* ent, int count, int k,
                       int len1, int len2, const double* R, const double* t, double* terms, double* red, double& rmsd, double& tm) {
    const int tid = threadIdx.x, nth = blockDim.x;
    const double d1 = 1.24 * (double)(len1 - 15) / 3.0 - 1.8;
    const double d2 = 1.24 * (double)(len2 - 15) / 3.0 - 1.8;
    double acc = 0.0;
    for (int base = 0; base < count; base += kSumTile) {
        const int cnt = count - base < kSumTile ? count - base : kSumTile;
        for (int e = tid; e < cnt; e += nth) {
            const ColumnXYZ cur = load_column(Xi, Xj, ent, base + e, count);
            double mv[3] = {cur.b[0], cur.b[1], cur.b[2]};
            if constexpr (MOVE) {
                rot3(cur.b, R, mv);
                mv[0] = mv[0] + t[0];
                mv[1] = mv[1] + t[1];
                mv[2] = mv[2] + t[2];
            }
            const double e0 = cur.a[0] - mv[0], e1 = cur.a[1] - mv[1], e2 = cur.a[2] - mv[2];
            const double sg = (e0 + e1) + e2;
            const double q1 = sg / d1, q2 = sg / d2;
            // three regions: the squared differences (three per column, in the order they are added), the two TM sums
            terms[e * 3 + 0] = cur.pair ? e0 * e0 : 0.0;
            terms[e * 3 + 1] = cur.pair ? e1 * e1 : 0.0;
            terms[e * 3 + 2] = cur.pair ? e2 * e2 : 0.0;
            terms[3 * kSumTile + e] = cur.pair ? 1.0 / (1.0 + q1 * q1) : 0.0;
            terms[4 * kSumTile + kSumSlack + e] = cur.pair ? 1.0 / (1.0 + q2 * q2) : 0.0;
        }
        __syncthreads();
        if (tid == 0) acc = chain_sum(terms, 1, 3 * cnt, acc);
        else if (tid == 1) acc = chain_sum(terms + 3 * kSumTile, 1, cnt, acc);
        else if (tid == 2) acc = chain_sum(terms + 4 * kSumTile + kSumSlack, 1, cnt, acc);
        __syncthreads();
    }
    if (tid < 3) red[tid] = acc;
    __syncthreads();
    const double ss = red[0], sum1 = red[1], sum2 = red[2];
    rmsd = sqrt(ss / (double)k);
    const double t1 = (1.0 / (double)len1) * sum1;
    const double t2 = (1.0 / (double)len2) * sum2;
    tm = t1 > t2 ? t1 : t2;
    __syncthreads();
}

// LDS (doubles) of a trace stage whose sums are taken by the whole workgroup: entries | term tile | reduction slots
__host__ __device__ inline size_t trace_team_lds_doubles(int max_entries) {
    return ((size_t)max_entries + 3) / 4 * 2 + (size_t)kSumTile * kMaxAcc + kSumSlack + 16;
}

// DTW traceback (dynamic_time_warping.py:90-144) on the packed decisions: leaves the alignment columns
// as packed entries in lds[first .. cap) (cap = n + m), writes the rows to HBM (back-to-front in
// [aln, aln + 2*cap)), returns the number of columns and of aligned pairs.  Wave-uniform.
template <int R, int RB = R>
CR_D void dtw_walk(int n0, int m0, int max_entries, const uint32_t* __restrict__ w, int start_layer,
                   double* lds, int32_t* __restrict__ aln, int& len_out, int& pairs_out, const int nA = 0) {
    const int lane = threadIdx.x;
    uint32_t* arow = reinterpret_cast<uint32_t*>(lds);           // packed alignment columns, back-to-front
    const int cap = n0 + m0;
    Walker<R, 4, 1, RB> wk;
    wk.init(w, tblocks(m0, 8), lane, nA);
    // the walk is wave-uniform: pin its state to SGPRs so that it compiles to scalar code
    int n = __builtin_amdgcn_readfirstlane(n0), m = __builtin_amdgcn_readfirstlane(m0);
    int dir = __builtin_amdgcn_readfirstlane(start_layer), idx = 0, k = 0;
    wk.set_row(n - 1);
    int guard = cap + 8;      // every iteration consumes a cell: damaged decision words can never hang the device (the rows are then wrong, not endless)
#pragma unroll 1
    while (n > 0 && m > 0 && --guard >= 0) {
        const uint32_t nib = wk.get(n - 1, m - 1);
        // dynamic_time_warping.py:118-143.  In layer 1 the stored decision either keeps the walk on
        // the diagonal or switches layer at the SAME cell; the switch and the move it then makes in
        // layer 0 / 2 (which reads the same cell's decisions) are done in one iteration.
        int layer = dir;
        if (layer == 1) layer = (int)((nib >> 1) & 3u);
        if (layer == 1) {
            // every following cell of the diagonal whose layer-1 decision is "diagonal" belongs to the same run
            const int run = wk.diag_run(n - 1, m - 1, [](uint32_t f) { return ((f >> 1) & 3u) == 1u; });
            if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(n - 1 - lane, m - 1 - lane);
            idx += run;
            k += run;
            n -= run;
            m -= run;
            dir = 1;
            if (n > 0) wk.set_row(n - 1);
        } else if (layer == 0) {
            // The vertical gap layer (:122-127): every cell it passes is consumed and its bit 0 says whether the walk stays in
            // the layer.  A whole run at once: the leading cells of the column whose bit is 0, plus the cell that ends the
            // run (bit 1: back to layer 1) when the block holds it.
            bool more;
            int run = wk.template run_up<0>(n - 1, m - 1, [](uint32_t f) { return (f & 1u) == 0u; }, more);
            dir = more ? 1 : 0;
            run += more ? 1 : 0;
            if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(n - 1 - lane, -1);
            idx += run;
            n -= run;
            if (n > 0) wk.set_row(n - 1);
        } else {
            // the horizontal gap layer (:138-143): bit 3 set = stay in it
            bool more;
            int run = wk.run_left(n - 1, m - 1, [](uint32_t f) { return (f & 8u) != 0u; }, more);
            dir = more ? 1 : 2;
            run += more ? 1 : 0;
            if (lane < run) arow[cap - idx - 1 - lane] = pack_entry(-1, m - 1 - lane);
            idx += run;
            m -= run;
        }
    }
    // border runs (dynamic_time_warping.py:108-117): only one of n, m is still positive
    for (int x = lane; x < n; x += kWave) arow[cap - idx - 1 - x] = pack_entry(n - 1 - x, -1);
    for (int x = lane; x < m; x += kWave) arow[cap - idx - 1 - x] = pack_entry(-1, m - 1 - x);
    idx += n + m;
    wave_sync();
    const int first = cap - idx;
    int32_t* a1 = aln;                                           // alignment rows -> HBM, coalesced
    int32_t* a2 = a1 + cap;
    for (int x = first + lane; x < cap; x += kWave) {
        const uint32_t u = arow[x];
        const uint32_t i = u & 0xffffu, j = u >> 16;
        a1[x] = i == kGap16 ? -1 : (int)i;
        a2[x] = j == kGap16 ? -1 : (int)j;
    }
    len_out = idx;
    pairs_out = k;
}

// Stage 4: DTW traceback, common positions, Kabsch on the original coordinates, RMSD / coverage / TM
// (multiple_alignment.py:1033-1054, :59-70).  Wave-uniform.
// the alignment rows of this block's pair straight into the caller's page-locked array (one wave; `ent`: the idx packed
// columns).  Only the aln_len entries of each row cross the link: what lies behind them in the caller's array is not touched.
CR_D void stream_rows(const HostOut& hout, const uint32_t* ent, int idx, int lane) {
    if (!hout.aln) return;
    int32_t* o1 = hout.aln + (int64_t)hout.dst(blockIdx.x) * 2 * hout.stride;
    int32_t* o2 = o1 + hout.stride;
    for (int x = lane; x < idx; x += kWave) {
        const uint32_t u = ent[x];
        const int i = (u & 0xffffu) == kGap16 ? -1 : (int)(u & 0xffffu);
        const int j = (u >> 16) == kGap16 ? -1 : (int)(u >> 16);
        __builtin_nontemporal_store(i, o1 + x);
        __builtin_nontemporal_store(j, o2 + x);
    }
}

template <int R, int RB = R>
CR_D void align_trace(const PairDesc& pd, int max_entries, const double* __restrict__ coords,
                      const uint32_t* __restrict__ bits, const AlignEnd e, double* lds,
                      int32_t* __restrict__ aln, PairResult& r, const HostOut hout = HostOut{}, const int nA = 0) {
    const int lane = threadIdx.x;
    uint32_t* arow = reinterpret_cast<uint32_t*>(lds);
    double* scratch = lds + ((size_t)max_entries + 3) / 4 * 2;
    const int cap = pd.n + pd.m;
    int idx, k;
    dtw_walk<R, RB>(pd.n, pd.m, max_entries, bits + pd.bt_off, e.start_layer, lds, aln + pd.aln_off, idx, k, nA);
    CR_STAMP(6);
    const int first = cap - idx;
    stream_rows(hout, arow + first, idx, lane);
    r.sw = e.sw;
    r.dtw_score = e.dtw_score;
#pragma unroll
    for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
    for (int x = 0; x < 3; x++) r.t[x] = 0.0;
    r.rmsd = r.coverage = r.tm = 0.0;
    r.flags = 0;
    r.aln_len = idx;
    r.aln_start = first;
    if (k < 3) {
        r.flags |= kFlagMetricsSkipped;
    } else {
        const double* Xi = coords + pd.off_i * 3;
        const double* Xj = coords + pd.off_j * 3;
        const uint32_t* ent = arow + first;
        double c1[3], c2[3];
        kabsch_ordered(Xi, Xj, ent, idx, k, lane, scratch, c1, c2, r.R, r.t);
        rmsd_tm_ordered<true>(Xi, Xj, ent, idx, k, pd.n, pd.m, r.R, r.t, lane, scratch, r.rmsd, r.tm);
        r.coverage = (double)k / (double)idx;
    }
}
