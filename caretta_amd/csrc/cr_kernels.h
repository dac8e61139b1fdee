// gfx950 kernels of the pairwise alignment path.
//
// Two kinds of DP fill:
//  * COLUMN SWEEP (sweep_cols, sweep_cols_team, sweep_cols_score*): Smith-Waterman with gap 0 -- the reference's only
//    use of smith_waterman / smith_waterman_score in the pipeline -- is monotone along rows and columns, so the `up`
//    dependency of a column is a prefix maximum: all 64 lanes (R rows each) work on the SAME column every step, a DPP
//    max-scan (row_shr 1/2/4/8, row_bcast 15/31) resolves the dependency, the column's features are wave-uniform and come
//    through scalar loads.  No pipeline ramp, every value bit-identical to the cell-by-cell evaluation.
//  * TIME-SKEWED WAVEFRONT (sweep, sweep_team, sweep_wide): the 3-layer affine "DTW" subtracts rounded gap penalties along
//    both axes (no exact scan), so lane l owns R consecutive rows of the current strip and at step t fills column
//    c = t - l of those rows.  The values of the row above a lane's block arrive from lane l-1 by DPP (wave_shr:1), the
//    values to the left stay in registers, a strip's last row is handed to the next strip through LDS / HBM.
// The residue score S(i,j) is never materialised: the RBF is evaluated in the sweep from row features held in registers
// and column features streamed through SGPRs or a 128-column LDS ring.  Backtrack decisions are packed (2 bit/cell SW,
// 4 bit/cell DTW) in the order the sweep produces them and written with 256-byte coalesced stores; the traceback, Kabsch
// and metric phase follows in the same wave (Walker: wave-uniform walk on a register-resident block of decision words,
// whole diagonal runs per ballot; position-ordered cooperative sums).  Launches with few pairs use one WORKGROUP per
// pair, one wave per strip: four waves (sweep_team) or up to sixteen (sweep_wide, sweep_cols_team).
//
// Reference semantics: dynamic_time_warping.py (fills, tie-breaks), score_functions.py (RBF),
// superposition_functions.py (Kabsch), multiple_alignment.py:321-349, 1028-1054.
#pragma once

#include <type_traits>

#include "cr_math.h"

// Diagnostic build only (-DCR_STAMPS, tools/stamps.py): shader-clock stamps of the phases of the batch kernels,
// 8 slots per block (0-3 seed kernel: start, fill done, walk done, end; 4-7 the same for the align kernel).
#ifdef CR_STAMPS
static __device__ unsigned long long g_stamps[8192 * 8];
#define CR_STAMP(k)                                                                                  \
    do {                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CR_STAMP(k) \
    do {            \
    } while (0)
#endif

namespace cr {

constexpr int kWave = 64;
constexpr int kRing = 128;          // columns held in the LDS ring (two 64-column halves)
constexpr double kMinF64 = -0x1.fffffffffffffp+1023;  // np.finfo(float64).min, dynamic_time_warping.py:4
constexpr double kFarAway = 1e150;   // feature value of rows past the end: RBF score underflows to exactly 0

enum : uint32_t {
    kFlagSeedSkipped = 1u,      // <=3 seed positions: no superposition (multiple_alignment.py:337-342)
    kFlagMetricsSkipped = 2u,   // <3 aligned positions: no RMSD/TM (assert at :1034)
    kFlagSeedAllZero = 4u,      // tensor SW matrix all zero (the reference raises)
};

// One pair of the batch (device copy).
struct PairDesc {
    int32_t n, m;            // lengths of structure i (rows) and j (columns)
    int64_t off_i, off_j;    // residue offsets into the packed coordinate/tensor arrays
    int64_t dirs_off;        // word offset of this pair's SW decisions
    int64_t bt_off;          // word offset of this pair's DTW decisions
    int64_t aln_off;         // element offset of this pair's alignment rows (2 rows of n+m)
    int64_t hand_off;        // double offset of this pair's strip hand-off rows (3 planes of m; multi-strip pairs)
};

struct SeedMax {             // result of the tensor SW fill
    double score;
    int32_t i, j;            // 1-based DP coordinates of the first maximum in row-major order; 0 if none
};

struct Transform {           // seed superposition (superposition_functions.py:57-58)
    double c1[3], c2[3], R[9];
    uint32_t flags;
    int32_t seed_len;
};

struct AlignEnd {            // result of the coordinate fill
    double sw;               // smith_waterman_score
    double dtw_score;
    int32_t start_layer;
    int32_t pad;
};

// __syncthreads() for code that one wave runs on its own (every traceback / Kabsch / metric phase): the same
// fences, but a wave barrier instead of s_barrier.  In a 64-thread workgroup the compiler lowers __syncthreads() to
// exactly this; in the team kernels, where the other waves of the workgroup have already exited, it keeps wave 0 off
// the hardware barrier altogether instead of relying on s_barrier ignoring terminated waves.
CR_D void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Workgroup barrier for hand-offs that go through LDS only (the edge rings between the strips of a pair): waits for this
// wave's LDS operations and not for its global stores.  __syncthreads() carries a release fence, which on gfx950 is
// s_waitcnt vmcnt(0): every barrier of a sweep then waited for the decision words (and, with staged scores, for the
// score lines requested a block ahead) to reach memory -- 300 .. 900 cycles per barrier that nothing needs; the words
// are made visible to the traceback once, by the fence behind the sweep.
CR_D void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// value of `v` in lane `src_lane` (wave-uniform index), broadcast to every lane
CR_D double lane_value(double v, int src_lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                            __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

CR_HD int strips_of(int n, int R) { return (n + kWave * R - 1) / (kWave * R); }
CR_HD int tblocks(int m, int per_word) { return (m + kWave - 1 + per_word - 1) / per_word; }

// Where the strips of a one-wave-per-strip sweep lie.  The strips of one pair need not have the same number of rows per
// lane: a workgroup's waves are dealt round robin to the CU's four SIMDs, and e.g. 1200 rows as 7 strips of 3 rows per
// lane put 6 row slots on three SIMDs and 3 on the fourth, while (3,3,3,2,2,2,2,2) puts 5,5,5,4 -- the sweep advances
// at the pace of the fullest SIMD.  Strips [0, nA) have RA rows per lane, the others RB (RA == RB: all alike).
// "Row slot": one of a strip's R lane-rows; decision word of (strip, time block tb, row slot q, lane l) =
// ((slot0 * TB + tb * R + q) * 64 + l, slot0 = the row slots of all strips before it -- which for equal strips is the
// ((strip * TB + tb) * R + q) * 64 + l of the single-wave sweeps.
struct StripGeom {
    int nstrips;             // strips that hold rows of this pair
    int rowbase0;            // first row of this wave's strip
    int slot0;               // row slots of the strips before it
    int owner_wave, owner_lane, owner_q;   // where row n - 1 lives
};

template <int RA, int RB = RA>
struct WidePlan {
    int nA;                  // strips with RA rows per lane (ignored when RA == RB)
    CR_HD int rows_a() const { return nA * kWave * RA; }
    CR_HD bool in_a(int row) const { return RA == RB || row < rows_a(); }
    CR_HD bool wave_in_a(int w) const { return RA == RB || w < nA; }
    CR_HD int strips(int n) const {
        if (RA == RB || n <= rows_a()) return (n + kWave * RA - 1) / (kWave * RA);
        return nA + (n - rows_a() + kWave * RB - 1) / (kWave * RB);
    }
    CR_HD int slots(int n) const {                      // row slots of strips(n) strips
        const int st = strips(n);
        return (RA == RB || st <= nA) ? st * RA : nA * RA + (st - nA) * RB;
    }
    CR_HD StripGeom geom(int w, int n) const {
        StripGeom g;
        g.nstrips = strips(n);
        const bool a = wave_in_a(w);
        g.rowbase0 = a ? w * kWave * RA : rows_a() + (w - nA) * kWave * RB;
        g.slot0 = a ? w * RA : nA * RA + (w - nA) * RB;
        const int last = n - 1;
        if (in_a(last)) {
            g.owner_wave = last / (kWave * RA);
            const int rem = last - g.owner_wave * kWave * RA;
            g.owner_lane = rem / RA;
            g.owner_q = rem - g.owner_lane * RA;
        } else {
            const int x = last - rows_a();
            const int sb = x / (kWave * RB);
            g.owner_wave = nA + sb;
            const int rem = x - sb * kWave * RB;
            g.owner_lane = rem / RB;
            g.owner_q = rem - g.owner_lane * RB;
        }
        return g;
    }
};

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
                uint32_t code = (h == dg[q]) ? 1u : (h == lf) ? 2u : 3u;
                code = (h > 0.0) ? code : 0u;
                bool gt = h > st.rowmax[q];
                if constexpr (Src::kMaskRows) {
                    const bool rv = rowbase + q < n;
                    gt = gt & rv;
                    code = rv ? code : 0u;
                }
                st.swbits[q] |= code << sh2;
                if constexpr (Src::kMaskRows) st.rowmax[q] = gt ? h : st.rowmax[q];
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

// ---------------------------------------------------------------------------------------------
// The column sweep: Smith-Waterman with gap 0 on non-negative scores (the reference's only use of smith_waterman in
// the pipeline, multiple_alignment.py:332-335) WITHOUT the time skew.
//
// With gap = 0 and S >= 0 the recurrence H = max(0, diag + S, left, up) (dynamic_time_warping.py:234-238) makes H
// non-decreasing along rows and columns, and max is exact and associative, so for one column j
//     H[i][j] = max over i' <= i of B[i'][j],      B[i][j] = max(H[i-1][j-1] + S[i][j], H[i][j-1]),
// i.e. the `up` dependency is a PREFIX MAXIMUM down the column.  All 64 lanes (R rows each) therefore work on the SAME
// column in every step: B from the previous column's values (registers), a sequential scan down the lane's R rows, a
// 6-step DPP max-scan across the lanes (row_shr 1/2/4/8, row_bcast 15/31), one more max per cell.  A strip takes m
// steps instead of m + 63, no lane ever idles in a ramp, and the column's features are wave-uniform: they are read
// with scalar loads into SGPRs (no LDS ring, no per-step ds_reads).  Every value is bit-identical to the
// cell-by-cell evaluation; the decisions (h == diag + S, then h == left, else up; 0 when h == 0) and the row-major
// first maximum are taken from the same values: a row's maximum is its last value and its first position is the column
// of the row's last strict increase (h != left).
// Decision words: ((strip * TB + (j >> 4)) * R + q) * 64 + lane, TB = ceil(m / 16), bits (j & 15) * 2: the layout of
// the skewed sweeps with time step = column (Walker<R, 2, 0>).
// Strips after the first take the row above them (the previous strip's last row, one value per column) from `hand_g`,
// 64 columns per coalesced load.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
CR_D double scan_step(double v) {
    // lanes without a source lane read +0.0 (bound_ctrl), rows outside ROW_MASK keep the +0.0 they are given:
    // max(v, 0) = v for v >= 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    return vmax(v, __hiloint2double(hi, lo));
}

// inclusive prefix maximum over the 64 lanes of non-negative doubles
CR_D double wave_scan_max(double v) {
    v = scan_step<0x111, 0xf>(v);      // row_shr:1
    v = scan_step<0x112, 0xf>(v);      // row_shr:2
    v = scan_step<0x114, 0xf>(v);      // row_shr:4
    v = scan_step<0x118, 0xf>(v);      // row_shr:8
    v = scan_step<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = scan_step<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}

// rows per lane up to which the column sweep keeps two sets of column features (ColSweep::step; 32 structures x 150:
// k_seed 0.114 -> 0.110 ms -- with one wave per SIMD and three rows per lane the step is bound by the latency of its
// dependent FP64 chains, about 6 cycles per instruction, more than by the scalar loads)
template <int R>
constexpr bool kTwoColumnSets = R <= 3;

// Per-lane state of the column sweep and one column step.
template <int R, int D>
struct ColSweep {
    double hprev[R];          // H of this lane's rows, previous column
    double eprev;             // H of the row above them, previous column
    int rowfirst[R];          // column of each row's last strict increase
    uint32_t bits[R];         // decisions of the current word

    CR_D void reset() {
#pragma unroll
        for (int q = 0; q < R; q++) {
            hprev[q] = 0.0;
            rowfirst[q] = 0;
            bits[q] = 0;
        }
        eprev = 0.0;
    }
    // The column's features are wave-uniform: scalar loads into SGPRs (src.col).  A step first forms the R squared
    // distances -- the only readers of the features -- and then requests the NEXT column into the same registers, so the
    // load's latency hides behind the exp / DP / scan arithmetic of this step even with a single wave on the SIMD, and one
    // set of SGPRs suffices.  (Always D loads: the tensor array is allocated with D doubles of slack and the padded
    // features are zeroed by scalar selects -- conditional loads would cost a branch each.)
    template <bool FULL>
    CR_D void prefetch(RbfTensor<R, D>& src, int j) {
        prefetch_into<FULL>(src, j, src.col);
    }
    template <bool FULL>
    CR_D void prefetch_into(RbfTensor<R, D>& src, int j, double (&set)[D]) {
        const double* __restrict__ cg = src.cols_g;
        const int d = FULL ? D : src.d;
#pragma unroll
        for (int k = 0; k < D; k++) {
            const double v = cg[(int64_t)j * d + k];
            set[k] = (FULL || k < d) ? v : 0.0;
        }
    }
    // Column j (prefetch<FULL>(src, j) has been called; `jn` = the column to request now, any valid column).  FULL: the
    // stored tensor width equals D (no padded features).  `top`: H of the row above the strip in this column
    // (wave-uniform; only read when TOP).
    // SET 0: one set of feature registers, as described above.  SET 1 / 2 (few rows per lane: the arithmetic behind the
    // squared distances is too short to cover a scalar load that misses): two sets in turn -- column j is in set SET,
    // column jn is requested into the other one BEFORE anything else, so the load has the whole step to arrive.
    template <bool FULL, bool TOP, int SET = 0>
    CR_D void step(RbfTensor<R, D>& src, const ExpEntry* tab, int j, int jn, double top) {
        double acc[R];
        if constexpr (SET == 1) prefetch_into<FULL>(src, jn, src.col2);
        if constexpr (SET == 2) prefetch_into<FULL>(src, jn, src.col);
#pragma unroll
        for (int q = 0; q < R; q++) acc[q] = SET == 2 ? src.dist2_of(q, src.col2) : src.dist2_of(q, src.col);
        if constexpr (SET == 0) prefetch<FULL>(src, jn);
        double dg[R], p[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double sc = exp_tab<true>(src.neg_gamma * acc[q], tab);
            dg[q] = (q == 0 ? eprev : hprev[q - 1]) + sc;
            const double b = vmax(dg[q], hprev[q]);
            p[q] = q == 0 ? b : vmax(p[q - 1], b);
        }
        double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
        if constexpr (TOP) e = vmax(e, top);
        const int sh2 = (j & 15) * 2;
#pragma unroll
        for (int q = 0; q < R; q++) {
            const double h = vmax(p[q], e);
            // decision replayed by the traceback's equality tests (:255-277): diag, then left, else up
            const bool same = h == hprev[q];
            uint32_t code = (h == dg[q]) ? 1u : same ? 2u : 3u;
            code = (h > 0.0) ? code : 0u;
            bits[q] |= code << sh2;
            rowfirst[q] = same ? rowfirst[q] : j;          // column of the row's last strict increase
            hprev[q] = h;
        }
        eprev = e;
    }
    CR_D void flush(uint32_t* __restrict__ sw_dirs, int64_t base) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            sw_dirs[base + q * kWave] = bits[q];
            bits[q] = 0;
        }
    }
    // fold the rows' maxima (= last values) into a running best, rows ascending
    CR_D void fold(int rowbase, double& best_v, int& best_i, int& best_j) const {
#pragma unroll
        for (int q = 0; q < R; q++) {
            const bool gt = hprev[q] > best_v;
            best_v = gt ? hprev[q] : best_v;
            best_i = gt ? rowbase + q : best_i;
            best_j = gt ? rowfirst[q] : best_j;
        }
    }
};

CR_D void wave_first_max(double& best_v, int& best_i, int& best_j) {
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(best_v, off);
        int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
        bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
        best_v = take ? ov : best_v;
        best_i = take ? oi : best_i;
        best_j = take ? oj : best_j;
    }
}

// One wave, strips one after the other.
template <int R, int D>
CR_D void sweep_cols(RbfTensor<R, D>& src, const int n, const int m, double* lds, uint32_t* __restrict__ sw_dirs,
                     double* __restrict__ hand_g, SeedMax& seed_out) {
    const int lane = threadIdx.x;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    load_exp_table(lds, lane);
    __syncthreads();

    const int nstrips = strips_of(n, R);
    const int TB = (m + 15) >> 4;
    const bool full = src.d == D;
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    ColSweep<R, D> st;

    for (int s = 0; s < nstrips; s++) {
        const int rowbase = (s * kWave + lane) * R;
        src.load_rows(rowbase, n);
        st.reset();
        const bool hand_out = s + 1 < nstrips;
        auto run = [&](auto full_tag, auto top_tag) {
            constexpr bool FULL = decltype(full_tag)::value, TOP = decltype(top_tag)::value;
            double top_vec = 0.0;                // row above the strip, 64 columns per load (lane x: column j0 + x)
            st.template prefetch<FULL>(src, 0);
            auto column = [&](auto set_tag, int j) {
                constexpr int SET = decltype(set_tag)::value;
                if (TOP && (j & (kWave - 1)) == 0) top_vec = (j + lane < m) ? hand_g[j + lane] : 0.0;
                st.template step<FULL, TOP, SET>(src, tab, j, j + 1 < m ? j + 1 : j, TOP ? lane_value(top_vec, j & (kWave - 1)) : 0.0);
                if (hand_out && lane == kWave - 1) hand_g[j] = st.hprev[R - 1];
                if ((j & 15) == 15 || j == m - 1) st.flush(sw_dirs, ((int64_t)(s * TB + (j >> 4)) * R) * kWave + lane);
            };
            if constexpr (kTwoColumnSets<R>) {
#pragma unroll 1
                for (int j = 0; j < m; j += 2) {
                    column(std::integral_constant<int, 1>{}, j);
                    if (j + 1 < m) column(std::integral_constant<int, 2>{}, j + 1);
                }
            } else {
#pragma unroll 1
                for (int j = 0; j < m; j++) column(std::integral_constant<int, 0>{}, j);
            }
        };
        if (s == 0) {
            if (full) run(std::true_type{}, std::false_type{});
            else run(std::false_type{}, std::false_type{});
        } else {
            if (full) run(std::true_type{}, std::true_type{});
            else run(std::false_type{}, std::true_type{});
        }
        if (hand_out) {                        // the hand-off row: visible to this wave's loads in the next strip
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
        st.fold(rowbase, best_v, best_i, best_j);
    }
    wave_first_max(best_v, best_i, best_j);
    seed_out.score = best_v;
    seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
    seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    __syncthreads();                                   // the caller may reuse the LDS from here on
}

// One WORKGROUP per pair, one wave per strip, all strips in flight: strip s works on columns [c * B, (c + 1) * B) in
// phase c + s (B = kColChunk), i.e. only B columns behind the strip above it -- against 64 + in the skewed team sweeps.
// The row above a strip arrives through an LDS ring written by the previous strip's last lane (one double per column,
// two chunks deep); the waves meet at one barrier per phase.  LDS (doubles): exp table | NW rings of 2 * kColChunk |
// NW * 4 reduction slots.  Results in every lane of every wave.
constexpr int kColChunk = 8;

template <int R, int D>
CR_D void sweep_cols_team(RbfTensor<R, D>& src, const int n, const int m, double* lds,
                          uint32_t* __restrict__ sw_dirs, SeedMax& seed_out, const StripGeom geom) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* rings = lds + kExpDoubles;
    double* ring_out = rings + w * (2 * kColChunk);
    const double* ring_in = rings + (w > 0 ? w - 1 : 0) * (2 * kColChunk);
    double* red = rings + NW * (2 * kColChunk);
    load_exp_table(lds, threadIdx.x);

    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const int TB = (m + 15) >> 4;
    const bool mine = w < nstrips;
    const bool full = src.d == D;
    const int rowbase = geom.rowbase0 + lane * R;
    const bool hand_out = w + 1 < nstrips;
    ColSweep<R, D> st;
    st.reset();
    if (mine) src.load_rows(rowbase, n);
    const int chunks = (m + kColChunk - 1) / kColChunk;
    const int phases = chunks + nstrips - 1;

    auto run = [&](auto full_tag, auto top_tag) {
        constexpr bool FULL = decltype(full_tag)::value, TOP = decltype(top_tag)::value;
#pragma unroll 1
        for (int g = 0; g < phases; g++) {
            __syncthreads();                           // the chunk written in phase g - 1 is visible to the strip below
            const int c = g - w;
            if (!mine || c < 0 || c >= chunks) continue;
            const int j0 = c * kColChunk;
            const int jend = j0 + kColChunk < m ? j0 + kColChunk : m;
            // the row above the strip for this chunk: lane x holds column j0 + x
            double top_vec = 0.0;
            if (TOP && lane < kColChunk) top_vec = ring_in[(c & 1) * kColChunk + lane];
            if (c == 0) st.template prefetch<FULL>(src, 0);
#pragma unroll 1
            for (int j = j0; j < jend; j++) {
                st.template step<FULL, TOP>(src, tab, j, j + 1 < m ? j + 1 : j, TOP ? lane_value(top_vec, j - j0) : 0.0);
                if (hand_out && lane == kWave - 1) ring_out[(c & 1) * kColChunk + (j - j0)] = st.hprev[R - 1];
            }
            if (((jend - 1) & 15) == 15 || jend == m)                            // a decision word holds 16 columns
                st.flush(sw_dirs, ((int64_t)geom.slot0 * TB + (int64_t)((jend - 1) >> 4) * R) * kWave + lane);
        }
    };
    if (w == 0) {
        if (full) run(std::true_type{}, std::false_type{});
        else run(std::false_type{}, std::false_type{});
    } else {
        if (full) run(std::true_type{}, std::true_type{});
        else run(std::false_type{}, std::true_type{});
    }

    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    if (mine) st.fold(rowbase, best_v, best_i, best_j);
    wave_first_max(best_v, best_i, best_j);
    if (lane == 0) {
        red[w * 4 + 0] = best_v;
        red[w * 4 + 1] = (double)best_i;
        red[w * 4 + 2] = (double)best_j;
    }
    __threadfence();                                   // decision words of every wave visible to wave 0's walk
    __syncthreads();
    best_v = 0.0;
    best_i = best_j = 0x7fffffff;
    for (int x = 0; x < nstrips; x++) {
        const double ov = red[x * 4 + 0];
        const int oi = (int)red[x * 4 + 1], oj = (int)red[x * 4 + 2];
        const bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
        best_v = take ? ov : best_v;
        best_i = take ? oi : best_i;
        best_j = take ? oj : best_j;
    }
    seed_out.score = best_v;
    seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
    seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    __syncthreads();
}

__host__ __device__ inline size_t sweep_cols_team_lds_doubles(int waves) {
    return kExpDoubles + (size_t)waves * (2 * kColChunk + 4);
}

// ---------------------------------------------------------------------------------------------
// smith_waterman_score (gap 0) of a pair as a column sweep WITHOUT decisions: what MultipleAlignment.make_pairwise_matrix
// needs of a pair (multiple_alignment.py:164) -- the P x P matrix entry, no alignment.  Same recurrence and scan as
// sweep_cols; the provider's columns come through its LDS ring (RbfCoords transforms 64 columns per chunk with the seed
// superposition) and are read back with wave-uniform addresses (LDS broadcast).  np.max of the matrix is H[n][m]
// (monotone rows and columns).  One wave, strips one after the other; the row above a strip travels through `hand_g`.
// ---------------------------------------------------------------------------------------------
template <int R, class Src>
CR_D double sweep_cols_score(Src& src, const int n, const int m, double* lds, double* __restrict__ hand_g) {
    const int lane = threadIdx.x;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* ring = lds + kExpDoubles;
    load_exp_table(lds, lane);
    src.init_ring(ring, lane);
    __syncthreads();
    const int nstrips = strips_of(n, R);
    double hprev[R], eprev = 0.0;
    for (int s = 0; s < nstrips; s++) {
        const int rowbase = (s * kWave + lane) * R;
        src.load_rows(rowbase, n);
#pragma unroll
        for (int q = 0; q < R; q++) hprev[q] = 0.0;
        eprev = 0.0;
        const bool hand_out = s + 1 < nstrips;
        double top_vec = 0.0;
#pragma unroll 1
        for (int j = 0; j < m; j++) {
            if ((j & (kWave - 1)) == 0) {
                __syncthreads();
                src.load_chunk(ring, j >> 6, m, lane);
                if (s > 0) top_vec = (j + lane < m) ? hand_g[j + lane] : 0.0;
                __syncthreads();
            }
            src.fetch_col(ring, j & (kRing - 1));
            double p[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const double sc = src.score(q, tab);
                const double dg = (q == 0 ? eprev : hprev[q - 1]) + sc;
                const double b = vmax(dg, hprev[q]);
                p[q] = q == 0 ? b : vmax(p[q - 1], b);
            }
            double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
            if (s > 0) e = vmax(e, lane_value(top_vec, j & (kWave - 1)));
#pragma unroll
            for (int q = 0; q < R; q++) hprev[q] = vmax(p[q], e);
            eprev = e;
            if (hand_out && lane == kWave - 1) hand_g[j] = hprev[R - 1];
        }
        if (hand_out) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
    }
    // H[n][m]: row n - 1 lives in the last strip, lane ((n - 1) / R) % 64, slot (n - 1) % R
    const int qo = (n - 1) % R;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) v = (q == qo) ? hprev[q] : v;
    return lane_value(v, ((n - 1) / R) % kWave);
}

// The same with one wave per strip and all strips in flight (strip s works kColChunk columns behind strip s - 1, one
// barrier per phase, as sweep_cols_team): for pair lists too short to fill the chip with one wave per pair.  Every wave
// has its own column ring.  LDS (doubles): exp table | NW column rings | NW edge rings of 2 * kColChunk | NW slots.
template <int R, class Src>
CR_D double sweep_cols_score_team(Src& src, const int n, const int m, double* lds, const StripGeom geom) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* ring = lds + kExpDoubles + w * Src::kRingDoubles;
    double* edges = lds + kExpDoubles + NW * Src::kRingDoubles;
    double* edge_out = edges + w * (2 * kColChunk);
    const double* edge_in = edges + (w > 0 ? w - 1 : 0) * (2 * kColChunk);
    double* red = edges + NW * (2 * kColChunk);
    load_exp_table(lds, threadIdx.x);
    src.init_ring(ring, lane);
    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const bool mine = w < nstrips;
    const int rowbase = geom.rowbase0 + lane * R;
    const bool hand_out = w + 1 < nstrips;
    double hprev[R], eprev = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) hprev[q] = 0.0;
    if (mine) src.load_rows(rowbase, n);
    const int chunks = (m + kColChunk - 1) / kColChunk;
    const int phases = chunks + nstrips - 1;
#pragma unroll 1
    for (int g = 0; g < phases; g++) {
        __syncthreads();                               // the chunk written in phase g - 1 is visible to the strip below
        const int c = g - w;
        if (!mine || c < 0 || c >= chunks) continue;
        const int j0 = c * kColChunk;
        const int jend = j0 + kColChunk < m ? j0 + kColChunk : m;
        if ((j0 & (kWave - 1)) == 0) {                 // this wave's own ring: a wave-level fence is enough
            wave_sync();
            src.load_chunk(ring, j0 >> 6, m, lane);
            wave_sync();
        }
        double top_vec = 0.0;
        if (w > 0 && lane < kColChunk) top_vec = edge_in[(c & 1) * kColChunk + lane];
#pragma unroll 1
        for (int j = j0; j < jend; j++) {
            src.fetch_col(ring, j & (kRing - 1));
            double p[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const double sc = src.score(q, tab);
                const double dg = (q == 0 ? eprev : hprev[q - 1]) + sc;
                const double b = vmax(dg, hprev[q]);
                p[q] = q == 0 ? b : vmax(p[q - 1], b);
            }
            double e = wave_shr1(wave_scan_max(p[R - 1]), 0.0);
            if (w > 0) e = vmax(e, lane_value(top_vec, j - j0));
#pragma unroll
            for (int q = 0; q < R; q++) hprev[q] = vmax(p[q], e);
            eprev = e;
            if (hand_out && lane == kWave - 1) edge_out[(c & 1) * kColChunk + (j - j0)] = hprev[R - 1];
        }
    }
    const int qo = geom.owner_q;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) v = (q == qo) ? hprev[q] : v;
    if (w == geom.owner_wave && lane == geom.owner_lane) red[0] = v;
    __syncthreads();
    return red[0];
}

template <class Src>
__host__ __device__ inline size_t sweep_cols_score_team_lds_doubles(int waves) {
    return kExpDoubles + (size_t)waves * (Src::kRingDoubles + 2 * kColChunk) + 8;
}

// LDS doubles needed by a sweep of the given provider/mode for column count m and row count n
template <int R, int MODE, class Src>
__host__ __device__ inline size_t sweep_lds_doubles(int n_max, int m_max) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    size_t v = exp_doubles<Src>::value + Src::kRingDoubles;
    if (strips_of(n_max, R) > 1) v += (size_t)NB * (kWave + kRing);
    return v;
}

// ---------------------------------------------------------------------------------------------
// The team sweep: one WORKGROUP per pair, one wave per strip, all strips in flight at once.
// For launches with too few pairs to fill the chip (a level of the guide tree, a small pair list) the
// single-wave sweep is latency bound: one wave issues one instruction every few cycles and walks the
// strips one after the other.  Here strip s runs on wave s, kTeamDelay = 64 steps behind strip s-1 (the
// smallest lag: lane 63 of strip s-1 finishes column c one step before lane 0 of strip s needs it), and takes
// the row above it from an LDS ring that strip s-1's last lane fills; the waves meet at a barrier every step.
// (A 128-step lag needs no barrier beyond those of the column-chunk loads, but the longer pipeline costs more
// than the barriers: 15.8 vs 13.8 ms for the 17 levels of the 128 x 300 guide tree.)  Decision words use the
// same (strip, time block, row, lane) layout as the single-wave sweep, so the traceback code is shared.
// Results are returned in every lane of every wave.
// LDS (doubles): exp table | NW column rings | NW edge rings of NB * kEdgeRing | NW * 8 reduction slots.
// ---------------------------------------------------------------------------------------------
constexpr int kTeamDelay = kWave;
constexpr int kEdgeRing = 4 * kWave;
constexpr int kTeamWaves = 4;

template <int R, int MODE, class Src>
CR_D void sweep_team(Src& src, const int n, const int m, const SweepParams prm, double* lds,
                     uint32_t* __restrict__ sw_dirs, uint32_t* __restrict__ dtw_bits, SeedMax& seed_out,
                     AlignEnd& end_out) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* ring = lds + kExpDoubles + w * Src::kRingDoubles;
    double* edges = lds + kExpDoubles + NW * Src::kRingDoubles;
    double* edge_out = edges + w * (NB * kEdgeRing);
    const double* edge_in = edges + (w > 0 ? w - 1 : 0) * (NB * kEdgeRing);
    double* red = edges + NW * (NB * kEdgeRing);

    load_exp_table(lds, lane);
    src.init_ring(ring, lane);
    __syncthreads();

    const int nstrips = strips_of(n, R);                 // <= NW, guaranteed by the launcher
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;
    const bool mine = w < nstrips;
    const int rowbase = (w * kWave + lane) * R;
    const int rows_here = n - w * kWave * R;
    const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
    const int T = mine ? m + lanes_here - 1 : 0;

    DpState<R> st;
    st.sw_max = 0.0;
    if (mine) src.load_rows(rowbase, n);
    st.reset_column0(col0_m2);
#pragma unroll
    for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;

    const int G = kTeamDelay * (nstrips - 1) + m + kWave - 1;
    for (int g = 0; g < G; g++) {
        const int t = g - kTeamDelay * w;
        const bool live = mine && t >= 0 && t < T;
        const bool boundary = (g & (kWave - 1)) == 0;
        __syncthreads();                                   // edge values of step g-1 visible to the next strip
        if (boundary) {
            if (live) src.load_chunk(ring, t >> 6, m, lane);
            __syncthreads();
        }
        if (!live) continue;
        const int c = t - lane;
        const bool active = (unsigned)c < (unsigned)m;

        double h_top0 = 0.0, m0_top0 = col0_m2, m1_top0 = 0.0;
        if (w > 0 && lane == 0 && active) {
            if constexpr (SW) h_top0 = edge_in[c & (kEdgeRing - 1)];
            if constexpr (DTW) {
                m0_top0 = edge_in[(NB - 2) * kEdgeRing + (c & (kEdgeRing - 1))];
                m1_top0 = edge_in[(NB - 1) * kEdgeRing + (c & (kEdgeRing - 1))];
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
            if (w + 1 < nstrips && lane == kWave - 1) {
                if constexpr (SW) edge_out[c & (kEdgeRing - 1)] = st.h_left[R - 1];
                if constexpr (DTW) {
                    edge_out[(NB - 2) * kEdgeRing + (c & (kEdgeRing - 1))] = st.m0_left[R - 1];
                    edge_out[(NB - 1) * kEdgeRing + (c & (kEdgeRing - 1))] = st.m1_left[R - 1];
                }
            }
        }
        if constexpr (TRACE) {
            if ((t & 15) == 15 || t == T - 1) {
                const int64_t base = ((int64_t)(w * TB_SW + (t >> 4)) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    sw_dirs[base + q * kWave] = st.swbits[q];
                    st.swbits[q] = 0;
                }
            }
        }
        if constexpr (DTW) {
            if ((t & 7) == 7 || t == T - 1) {
                const int64_t base = ((int64_t)(w * TB_DTW + (t >> 3)) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    dtw_bits[base + q * kWave] = st.dtbits[q];
                    st.dtbits[q] = 0;
                }
            }
        }
    }

    // ---- per-wave results, then across the waves through LDS ----------------------------------------
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    if constexpr (TRACE) {
        if (mine) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const bool gt = st.rowmax[q] > best_v;
                best_v = gt ? st.rowmax[q] : best_v;
                best_i = gt ? rowbase + q : best_i;
                best_j = gt ? st.rowarg[q] : best_j;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(best_v, off);
            int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
            bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
    }
    double sw_max = mine ? st.sw_max : 0.0;
    if constexpr ((MODE & kSwScore) != 0) {
        for (int off = 32; off > 0; off >>= 1) sw_max = __builtin_fmax(sw_max, __shfl_xor(sw_max, off));
    }
    const int owner_wave = (n - 1) / (kWave * R);
    if (lane == 0) {
        red[w * 8 + 0] = best_v;
        red[w * 8 + 1] = (double)best_i;
        red[w * 8 + 2] = (double)best_j;
        red[w * 8 + 3] = sw_max;
    }
    if constexpr (DTW) {
        if (w == owner_wave) {
            const int owner = ((n - 1) / R) % kWave;
            const int qo = (n - 1) % R;
            double fin0 = 0.0, fin1 = 0.0, fin2 = 0.0;
#pragma unroll
            for (int q = 0; q < R; q++) {
                fin0 = (q == qo) ? st.m0_left[q] : fin0;
                fin1 = (q == qo) ? st.m1_left[q] : fin1;
                fin2 = (q == qo) ? st.m2_left[q] : fin2;
            }
            if (lane == owner) {
                red[w * 8 + 4] = fin0;
                red[w * 8 + 5] = fin1;
                red[w * 8 + 6] = fin2;
            }
        }
    }
    __threadfence();                                   // decision words of every wave visible to wave 0's walk
    __syncthreads();
    if constexpr (TRACE) {
        best_v = 0.0;
        best_i = best_j = 0x7fffffff;
        for (int x = 0; x < nstrips; x++) {
            const double ov = red[x * 8 + 0];
            const int oi = (int)red[x * 8 + 1], oj = (int)red[x * 8 + 2];
            const bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
        seed_out.score = best_v;
        seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
        seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    }
    if constexpr ((MODE & kSwScore) != 0 || DTW) {
        double smax = 0.0;
        for (int x = 0; x < nstrips; x++) smax = __builtin_fmax(smax, red[x * 8 + 3]);
        const double fin0 = red[owner_wave * 8 + 4], fin1 = red[owner_wave * 8 + 5], fin2 = red[owner_wave * 8 + 6];
        end_out.sw = smax;
        int idx = 0;
        double best = fin0;
        if (fin1 > best) { best = fin1; idx = 1; }
        if (fin2 > best) { best = fin2; idx = 2; }
        end_out.dtw_score = DTW ? best : 0.0;
        end_out.start_layer = idx;
        end_out.pad = 0;
    }
    __syncthreads();
}

template <int R, int MODE, class Src>
__host__ __device__ inline size_t sweep_team_lds_doubles(int waves) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    return kExpDoubles + (size_t)waves * (Src::kRingDoubles + NB * kEdgeRing + 8);
}

// ---------------------------------------------------------------------------------------------
// The wide sweep: one WORKGROUP of up to kWideMaxWaves waves per pair, one wave per strip, for pair lists that
// cannot fill the chip with one or four waves per pair (one GPU's share of a sharded long-chain family: 252 pairs
// of 1200 x 1200 on 256 CUs).  Differences from sweep_team:
//   * all m columns of the pair are RESIDENT in LDS (feature-major planes, loaded once by the whole workgroup with
//     coalesced reads): no per-wave column rings, no chunk loads, no chunk barriers, and the LDS cost does not
//     grow with the number of waves;
//   * strip s runs lag = 63 + B steps behind strip s-1 and the waves meet at a barrier every B steps only
//     (B = sync_every): a value written by strip s-1's last lane in global step g is read by strip s in step g + B,
//     and every window of B consecutive steps holds one barrier.  B = 1 is sweep_team's lock step; larger B lets the
//     waves of one SIMD drift and fill each other's issue gaps at the price of a (S - 1) * (B - 1) steps longer
//     pipeline.  Edge rings of kWideEdge entries per value: the writer is at most 2B - 1 columns ahead (B <= 32).
// Decision words use the same (strip, time block, row, lane) layout as the other sweeps (shared traceback).
// LDS (doubles): exp table | Src::kColDoubles planes of `stride` | NW edge rings of NB * kWideEdge | NW * 8.
// ---------------------------------------------------------------------------------------------
constexpr int kWideEdge = 64;
constexpr int kWideMaxWaves = 16;
constexpr int kWideMaxSync = 32;

// The end of a one-wave-per-strip sweep: per-wave results, then across the waves through LDS (as sweep_team).
// `red`: 8 doubles per wave.
template <int R, int MODE>
CR_D void wide_finish(const DpState<R>& st, const bool mine, const int w, const int lane, const int rowbase, const StripGeom geom,
                      double* red, SeedMax& seed_out, AlignEnd& end_out) {
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    const int nstrips = geom.nstrips;
    // ---- per-wave results, then across the waves through LDS (as sweep_team) -------------------------
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    if constexpr (TRACE) {
        if (mine) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const bool gt = st.rowmax[q] > best_v;
                best_v = gt ? st.rowmax[q] : best_v;
                best_i = gt ? rowbase + q : best_i;
                best_j = gt ? st.rowarg[q] : best_j;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(best_v, off);
            int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
            bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
    }
    double sw_max = mine ? st.sw_max : 0.0;
    if constexpr ((MODE & kSwScore) != 0) {
        for (int off = 32; off > 0; off >>= 1) sw_max = __builtin_fmax(sw_max, __shfl_xor(sw_max, off));
    }
    const int owner_wave = geom.owner_wave;
    if (lane == 0) {
        red[w * 8 + 0] = best_v;
        red[w * 8 + 1] = (double)best_i;
        red[w * 8 + 2] = (double)best_j;
        red[w * 8 + 3] = sw_max;
    }
    if constexpr (DTW) {
        if (w == owner_wave) {
            const int owner = geom.owner_lane;
            const int qo = geom.owner_q;
            double fin0 = 0.0, fin1 = 0.0, fin2 = 0.0;
#pragma unroll
            for (int q = 0; q < R; q++) {
                fin0 = (q == qo) ? st.m0_left[q] : fin0;
                fin1 = (q == qo) ? st.m1_left[q] : fin1;
                fin2 = (q == qo) ? st.m2_left[q] : fin2;
            }
            if (lane == owner) {
                red[w * 8 + 4] = fin0;
                red[w * 8 + 5] = fin1;
                red[w * 8 + 6] = fin2;
            }
        }
    }
    __threadfence();                                   // decision words of every wave visible to wave 0's walk
    __syncthreads();
    if constexpr (TRACE) {
        best_v = 0.0;
        best_i = best_j = 0x7fffffff;
        for (int x = 0; x < nstrips; x++) {
            const double ov = red[x * 8 + 0];
            const int oi = (int)red[x * 8 + 1], oj = (int)red[x * 8 + 2];
            const bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
        seed_out.score = best_v;
        seed_out.i = best_v > 0.0 ? best_i + 1 : 0;
        seed_out.j = best_v > 0.0 ? best_j + 1 : 0;
    }
    if constexpr ((MODE & kSwScore) != 0 || DTW) {
        double smax = 0.0;
        for (int x = 0; x < nstrips; x++) smax = __builtin_fmax(smax, red[x * 8 + 3]);
        const double fin0 = red[owner_wave * 8 + 4], fin1 = red[owner_wave * 8 + 5], fin2 = red[owner_wave * 8 + 6];
        end_out.sw = smax;
        int idx = 0;
        double best = fin0;
        if (fin1 > best) { best = fin1; idx = 1; }
        if (fin2 > best) { best = fin2; idx = 2; }
        end_out.dtw_score = DTW ? best : 0.0;
        end_out.start_layer = idx;
        end_out.pad = 0;
    }
    __syncthreads();
}

template <int R, int MODE, class Src>
CR_D void sweep_wide(Src& src, const int n, const int m, const SweepParams prm, double* lds, const int sync_every,
                     uint32_t* __restrict__ sw_dirs, uint32_t* __restrict__ dtw_bits, SeedMax& seed_out,
                     AlignEnd& end_out, const StripGeom geom) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    const int stride = m;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* res = lds + kExpDoubles;
    double* edges = res + (size_t)Src::kColDoubles * stride;
    double* edge_out = edges + w * (NB * kWideEdge);
    const double* edge_in = edges + (w > 0 ? w - 1 : 0) * (NB * kWideEdge);
    double* red = edges + NW * (NB * kWideEdge);

    load_exp_table(lds, threadIdx.x);
    src.load_resident(res, stride, m, (int)threadIdx.x, (int)blockDim.x);

    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;
    const bool mine = w < nstrips;
    const int rowbase = geom.rowbase0 + lane * R;
    const int rows_here = n - geom.rowbase0;
    const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
    const int T = mine ? m + lanes_here - 1 : 0;
    const int lag = kWave - 1 + sync_every;

    DpState<R> st;
    st.sw_max = 0.0;
    if (mine) src.load_rows(rowbase, n);
    st.reset_column0(col0_m2);
#pragma unroll
    for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;

    // Few rows per lane: a step is one long chain of dependent FP64 instructions (squared distance -> exp -> recurrences,
    // ~10 cycles each for a wave that has its SIMD to itself) with nothing to interleave.  The scores do not depend on the
    // recurrence, so they are formed ONE COLUMN AHEAD: the chain of column c + 1's scores runs beside the recurrence of
    // column c, and the step becomes issue-bound.  (Every lane's first column is column 0: its scores are formed here.)
    constexpr bool AHEAD = R <= 2;
    double sc_cur[R];
    __syncthreads();                                       // the resident columns and the exp table are complete (every
                                                           // wave, whatever its rows per lane: barriers must pair up)
    if constexpr (AHEAD) {
        src.fetch_resident(res, stride, 0);
#pragma unroll
        for (int q = 0; q < R; q++) sc_cur[q] = src.score(q, tab);
    }
    const int G = lag * (nstrips - 1) + m + kWave - 1;
    int until_sync = 0;
    for (int g = 0; g < G; g++) {
        if (until_sync == 0) {
            lds_barrier();                                 // edge values of the last B steps visible to the next strip
            until_sync = sync_every;
        }
        until_sync--;
        const int t = g - lag * w;
        const bool live = mine && t >= 0 && t < T;
        if (!live) continue;
        const int c = t - lane;
        const bool active = (unsigned)c < (unsigned)m;

        double h_top0 = 0.0, m0_top0 = col0_m2, m1_top0 = 0.0;
        if (w > 0 && lane == 0 && active) {
            if constexpr (SW) h_top0 = edge_in[c & (kWideEdge - 1)];
            if constexpr (DTW) {
                m0_top0 = edge_in[(NB - 2) * kWideEdge + (c & (kWideEdge - 1))];
                m1_top0 = edge_in[(NB - 1) * kWideEdge + (c & (kWideEdge - 1))];
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
            if constexpr (AHEAD) {
                double sc_next[R];
                src.fetch_resident(res, stride, c + 1 < m ? c + 1 : c);
#pragma unroll
                for (int q = 0; q < R; q++) sc_next[q] = src.score(q, tab);
                dp_column<R, MODE>(src, st, prm, tab, c, rowbase, n, sh2, sh4, h_top, m0_top, m1_top, sc_cur);
#pragma unroll
                for (int q = 0; q < R; q++) sc_cur[q] = sc_next[q];
            } else {
                src.fetch_resident(res, stride, c);
                dp_column<R, MODE>(src, st, prm, tab, c, rowbase, n, sh2, sh4, h_top, m0_top, m1_top);
            }
            if (w + 1 < nstrips && lane == kWave - 1) {
                if constexpr (SW) edge_out[c & (kWideEdge - 1)] = st.h_left[R - 1];
                if constexpr (DTW) {
                    edge_out[(NB - 2) * kWideEdge + (c & (kWideEdge - 1))] = st.m0_left[R - 1];
                    edge_out[(NB - 1) * kWideEdge + (c & (kWideEdge - 1))] = st.m1_left[R - 1];
                }
            }
        }
        if constexpr (TRACE) {
            if ((t & 15) == 15 || t == T - 1) {
                const int64_t base = ((int64_t)geom.slot0 * TB_SW + (int64_t)(t >> 4) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    sw_dirs[base + q * kWave] = st.swbits[q];
                    st.swbits[q] = 0;
                }
            }
        }
        if constexpr (DTW) {
            if ((t & 7) == 7 || t == T - 1) {
                const int64_t base = ((int64_t)geom.slot0 * TB_DTW + (int64_t)(t >> 3) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    dtw_bits[base + q * kWave] = st.dtbits[q];
                    st.dtbits[q] = 0;
                }
            }
        }
    }

    wide_finish<R, MODE>(st, mine, w, lane, rowbase, geom, red, seed_out, end_out);
}

template <int MODE, class Src>
__host__ __device__ inline size_t sweep_wide_lds_doubles(int waves, int m_max) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    return kExpDoubles + (size_t)Src::kColDoubles * m_max + (size_t)waves * (NB * kWideEdge + 8);
}

// ---------------------------------------------------------------------------------------------
// The wide sweep on scores that another launch has already formed (cr_staged.h): ONE row per lane up to 320 rows (five
// strips), then two, three, four (up to 2048 rows; blocks of 8 steps from three rows on).
//
// When a launch has few workgroups -- a level of the progressive alignment, a short pair list -- the fused kernels are
// bound by the instruction issue of the few waves that hold the recurrence, and 50 of a seed step's 59 instructions (30
// of an alignment step's 49) are the score, which does not depend on the recurrence at all.  A staging launch forms the
// scores on every CU of the chip in the SAME arithmetic (the provider's own score()), and this sweep is left with the
// recurrence.
// Layout of one strip (64 rows): element t * 64 + lane = S(row lane, column t - lane), t = 0 .. m + 62: the line a wave
// needs at step t is one coalesced 512-byte read.  The loop runs in blocks of kStagedBlock = 16 steps, unrolled: the
// block's 16 lines sit in registers, requested TWO blocks ahead (the scores are in L2 / MALL, 200 .. 900 cycles away;
// 48 lines in flight per wave); shifts and word boundaries of a block are fixed at compile time.  A strip follows the
// one above by at least five blocks (63 + 16 steps), paced by progress words (below).  The strip region has
// staged_steps(m_max) lines: the requests two blocks past the last step stay inside it.
// LDS (doubles): NW + 1 hand-off rings of NB * kStagedRing | NW * 8 | NW dumps | progress words.  Decision words: as every other skewed sweep.
// ---------------------------------------------------------------------------------------------
constexpr int kStagedBlock = 16;         // steps per block with one or two rows per lane; 8 with three or four (registers)
constexpr int kStagedMaxWaves = 8;       // the blocks of score lines in registers need more than the 128 VGPRs of a 16-wave
                                         // workgroup: 512 rows per row of a lane
constexpr int kStagedMaxR = 4;
constexpr int kStagedMaxRows = kStagedMaxWaves * kWave * kStagedMaxR;
CR_HD int staged_steps(int m_max) { return (m_max + kWave - 1 + kStagedBlock - 1) / kStagedBlock * kStagedBlock + 2 * kStagedBlock; }

template <int R, bool RBF = true>
struct StagedScore {                               // what dp_column sees: the scores of the lane's cells of this step
    static constexpr bool kNonNegative = RBF;      // RBF scores (the staging kernels write what the RBF providers return);
                                                   // explicit score matrices (cr_dropins.h) may hold anything
    static constexpr bool kMaskRows = !RBF;        // RBF: rows past n were staged as the exact zeros the RBF gives them
    double v[R];
    CR_D double score(int q, const ExpEntry*) const { return v[q]; }
};

// doubles per wave that take the hand-off writes of lanes 0 .. 62: dump[lane + plane * 64 + step], up to three planes
constexpr int kStagedRing = 128;          // steps a plane of a hand-off ring of the staged sweeps holds
constexpr int kStagedDump = kWave + 2 * kStagedRing + kStagedBlock;

// progress words of the staged sweeps (as cr_duo.h paces its strips): blocks a strip has completed
CR_D void staged_publish(int* word, int blocks_done) {
    asm volatile("" ::: "memory");       // (compiler: the hand-off values of the block are written first)
    *reinterpret_cast<volatile int*>(word) = blocks_done;
}
CR_D void staged_wait(const int* word, int need) {
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(word)) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");       // (compiler: hand-off values are read behind the word)
}

template <int R, int MODE, bool RBF = true>
CR_D void sweep_staged(const double* __restrict__ strip, const int n, const int m, const SweepParams prm, double* lds,
                       uint32_t* __restrict__ sw_dirs, uint32_t* __restrict__ dtw_bits, SeedMax& seed_out,
                       AlignEnd& end_out, const StripGeom geom) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);
    constexpr int B = R <= 2 ? kStagedBlock : 8;          // steps per block (the block's R * B score lines sit in registers)
    constexpr int LAGB = R <= 2 ? 5 : 9;                  // blocks a strip lags the one above: 80 / 72 steps (>= 63 + B)
    constexpr bool FAR = R == 1;                          // score lines two blocks ahead (R >= 2: one, the registers are taken)
    constexpr int RING = kStagedRing;                     // slots of a plane of a hand-off ring
    constexpr int PH = 0, PM0 = (NB - 2) * RING, PM1 = (NB - 1) * RING;             // planes of a ring
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int NW = (int)(blockDim.x >> 6);
    // Hand-off rings, indexed by the WRITER's step (t & 63: a block's 16 slots are contiguous): ring 0 holds the DP border
    // above row 0 (constants), ring w + 1 the last row of strip w.  Strip w reads ring w: no special case for the first.
    const double* ring_in = lds + w * (NB * RING);
    double* ring_out = lds + (w + 1) * (NB * RING);
    double* red = lds + (NW + 1) * (NB * RING);
    double* dump = red + NW * 8 + w * kStagedDump;
    int* prog = reinterpret_cast<int*>(red + NW * 8 + NW * kStagedDump);     // prog[w]: blocks strip w has completed

    const int nstrips = geom.nstrips;                    // <= NW, guaranteed by the launcher
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;
    const bool mine = w < nstrips;
    const int rowbase = geom.rowbase0 + lane * R;
    const int rows_here = n - geom.rowbase0;
    const int lanes_here = rows_here >= kWave * R ? kWave : (rows_here + R - 1) / R;
    const int my_blocks = mine ? (m + lanes_here - 1 + B - 1) / B : 0;
    // every ring starts as the DP border: ring 0 IS the border; in the others a lane 0 that is past its last column (the
    // ramps run unmasked) may read a slot its writer never reaches -- e.g. the one step of the writer's masked last block
    // in which its lane 63 is past the last column -- and must not find whatever the LDS held
    for (int x = threadIdx.x; x < (NW + 1) * NB * RING; x += blockDim.x) lds[x] = (DTW && (x / RING) % NB == NB - 2) ? col0_m2 : 0.0;
    if (threadIdx.x < kStagedMaxWaves) prog[threadIdx.x] = 0;

    DpState<R> st;
    st.sw_max = 0.0;
    st.reset_column0(col0_m2);
#pragma unroll
    for (int q = 0; q < R; q++) st.swbits[q] = st.dtbits[q] = 0;
    StagedScore<R, RBF> src;
    const double* __restrict__ line = strip + lane;      // line t: R sub-lines of 64 doubles (row slot q, lane)
    double cur[B][R], nxt[B][R], nx2[FAR ? B : 1][R];
    if (mine) {
#pragma unroll
        for (int k = 0; k < B; k++)
#pragma unroll
            for (int q = 0; q < R; q++) {
                nxt[k][q] = line[(k * R + q) * kWave];
                if constexpr (FAR) nx2[k][q] = line[((B + k) * R + q) * kWave];
            }
    }
    const int blocks_above = (m + kWave - 1 + B - 1) / B;    // blocks of the strip above (it has all 64 lanes)
    const bool hand_out = w + 1 < nstrips;
    // RAMPS WITHOUT MASKS.  A lane's column t - lane is outside [0, m) in the first 63 and the last 63 steps of its strip; a
    // block in which that happens for any lane runs the EXEC-masked step (269 against 206 cycles for the DTW, 250 against
    // 156 for the SW, tools/step_probe.hip) -- and with the strips 80 steps apart nearly every block of the WORKGROUP has
    // some strip in a ramp, so the whole fill ran at the masked step's pace.  The masks are not needed where the staging
    // kernels have written exact zeros for the columns outside [0, m) (cr_staged.h, stage_block) and the penalties are not
    // negative:
    //  * before its column 0 a lane then sits at a fixed point that its first real step cannot tell from the DP border:
    //    SW: h = max(0 + 0, 0, 0) = 0, decision code 0, no row maximum.  DTW: c1 = 0 + 0, the layer above gives
    //    m0 = max(m0' - extend, 0 - open) <= 0, m2 = max(0 - open, m2 - extend) = -open from the first such step on, so
    //    m1 = max(max(m0, 0), -open) = +0.0 = M[i][0][1]; the first real step reads m2 - extend, which is below 0 - open
    //    for -open as for the border's MIN - open (same maximum, same decision bit);
    //  * behind column m - 1 a lane's state is dead: its decision bits lie at positions no walk reads, the values it hands
    //    down belong to columns the strip below does not have; SW with gap 0 repeats the row's last value (no new row
    //    maximum), a global SW maximum only ever sees values of real cells again.  Two things do outlive the last column:
    //    the DTW layers of row n - 1 (the score) and, for an SW with a gap, each row's first maximum -- hence the LAST
    //    block of every strip stays masked (the owner of row n - 1 is the last lane of its strip to finish), and the SW
    //    trace with a gap keeps its masks altogether.
    // Explicit score matrices (RBF = false: any sign, caller's penalties) keep the masks.
    const bool unmasked = !kProbeMaskedRamps && RBF && !(TRACE && !(MODE & kZeroGap)) && prm.sw_gap >= 0.0 && prm.gap_open >= 0.0 &&
                          prm.gap_extend >= 0.0;
    // PACING.  The strips form a chain -- strip w needs, for its block tb, the last row of the strip above up to that
    // strip's step 16 tb + 15 + 63, i.e. its blocks up to tb + LAGB - 1 -- and used to advance together behind one
    // s_barrier per block: every block took what the slowest strip's block took.  Now every strip publishes the number of
    // blocks it has completed (an LDS word, written behind the block's hand-off values: the LDS serves a wave's requests in
    // order) and waits only for the strip above; a writer also waits until the strip below is past the values a block will
    // overwrite (the rings hold RING = 128 steps: five blocks of slack on top of the five of lag).
    lds_barrier();                                         // border ring and progress words
#pragma unroll 1
    for (int tb = 0; tb < my_blocks; tb++) {
        if (w > 0) staged_wait(prog + w - 1, tb + LAGB < blocks_above ? tb + LAGB : blocks_above);
        // (this block overwrites the values of block tb - RING / B, whose last one the strip below reads in its step
        // B (tb - RING / B) + B - 1 - 63)
        if (hand_out && tb >= RING / B + LAGB - 2) staged_wait(prog + w + 1, tb - (RING / B + LAGB - 3));
        const double* __restrict__ ahead = line + (int64_t)(tb + (FAR ? 2 : 1)) * (B * R * kWave);
#pragma unroll
        for (int k = 0; k < B; k++)
#pragma unroll
            for (int q = 0; q < R; q++) {
                cur[k][q] = nxt[k][q];
                if constexpr (FAR) {
                    nxt[k][q] = nx2[k][q];
                    nx2[k][q] = ahead[(k * R + q) * kWave];
                } else {
                    nxt[k][q] = ahead[(k * R + q) * kWave];
                }
            }
        // The row above the strip.  Lane 0's column at step t is t itself, written by the strip above at ITS step t + 63:
        // slot (t - 1) & 63.  Every lane reads it (one address: a broadcast) and hands it to the shift as lane 0's fill;
        // the read of step k + 1 is issued before the arithmetic of step k.
        // (the slots of a ring are the WRITER's steps mod RING; lane 0's column at step t is t, written above at step t + 63)
        const int q4 = (tb * B) & (RING - 1);              // this strip writes the slots q4 + k
        const int q4r = (tb * B + kWave) & (RING - 1);     // ... and reads slot0, then q4r + k - 1 for its step k >= 1
        const double* fills = ring_in + q4r - 1;           // step k >= 1: fills[k]
        const int slot0 = (tb * B + kWave - 1) & (RING - 1);
        // the strip's last row: lane 63 writes its values of step k to slot q4 + k of the ring, the other lanes write theirs
        // to a dump (one LDS instruction per step with no EXEC juggling)
        double* wr = (lane == kWave - 1 && w + 1 < nstrips) ? ring_out + q4 : dump + lane;
        double f_h = 0.0, f_m0 = 0.0, f_m1 = 0.0;
        if constexpr (SW) f_h = ring_in[PH + slot0];
        if constexpr (DTW) {
            f_m0 = ring_in[PM0 + slot0];
            f_m1 = ring_in[PM1 + slot0];
        }
        auto steps = [&](auto all_tag) {
            constexpr bool ALL = decltype(all_tag)::value;    // every lane's column of every step of the block is inside [0, m)
            static_for<0, B>([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                const int c = tb * B + k - lane;
                const bool active = ALL || (unsigned)c < (unsigned)m;
                double g_h = 0.0, g_m0 = 0.0, g_m1 = 0.0;
                if constexpr (k + 1 < B) {
                    if constexpr (SW) g_h = fills[PH + k + 1];
                    if constexpr (DTW) {
                        g_m0 = fills[PM0 + k + 1];
                        g_m1 = fills[PM1 + k + 1];
                    }
                }
                double h_top = 0.0, m0_top = 0.0, m1_top = 0.0;
                if constexpr (SW) h_top = wave_shr1(st.h_left[R - 1], f_h);
                if constexpr (DTW) {
                    m0_top = wave_shr1(st.m0_left[R - 1], f_m0);
                    m1_top = wave_shr1(st.m1_left[R - 1], f_m1);
                }
                if (active) {
#pragma unroll
                    for (int q = 0; q < R; q++) src.v[q] = cur[k][q];
                    dp_column<R, MODE>(src, st, prm, nullptr, c, rowbase, n, ((tb * B + k) & 15) * 2, (k & 7) * 4, h_top, m0_top, m1_top);
                    if constexpr (!kProbeNoDump) {
                        if constexpr (SW) wr[PH + k] = st.h_left[R - 1];
                        if constexpr (DTW) {
                            wr[PM0 + k] = st.m0_left[R - 1];
                            wr[PM1 + k] = st.m1_left[R - 1];
                        }
                    }
                }
                f_h = g_h;
                f_m0 = g_m0;
                f_m1 = g_m1;
                if constexpr (DTW) {
                    if ((k & 7) == 7 && tb * (B / 8) + (k >> 3) < TB_DTW) {
                        const int64_t base = ((int64_t)geom.slot0 * TB_DTW + (int64_t)(tb * (B / 8) + (k >> 3)) * R) * kWave + lane;
#pragma unroll
                        for (int q = 0; q < R; q++) {
                            dtw_bits[base + q * kWave] = st.dtbits[q];
                            st.dtbits[q] = 0;
                        }
                    }
                }
            });
        };
        if (unmasked ? tb != my_blocks - 1 : (tb * B >= kWave - 1 && tb * B + B - 1 < m)) steps(std::true_type{});
        else steps(std::false_type{});
        if (nstrips > 1 && lane == 0) staged_publish(prog + w, tb + 1);    // (the strip below follows it, the strip above must not lap it)
        if constexpr (TRACE) {
            if ((((tb + 1) * B) & 15) == 0 || tb == my_blocks - 1) {      // a decision word holds 16 steps
                const int64_t base = ((int64_t)geom.slot0 * TB_SW + (int64_t)((tb * B) >> 4) * R) * kWave + lane;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    sw_dirs[base + q * kWave] = st.swbits[q];
                    st.swbits[q] = 0;
                }
            }
        }
    }
    wide_finish<R, MODE>(st, mine, w, lane, rowbase, geom, red, seed_out, end_out);
}

template <int MODE>
__host__ __device__ inline size_t sweep_staged_lds_doubles(int waves) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    return (size_t)(waves + 1) * (NB * kStagedRing) + (size_t)waves * (8 + kStagedDump) + kStagedMaxWaves / 2;
}

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
#pragma unroll 1
    while (n > 0 && m > 0) {
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
    wave_sync();
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
    wave_sync();
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

}  // namespace cr
