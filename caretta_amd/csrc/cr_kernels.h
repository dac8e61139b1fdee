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

#include "cr_providers.h"       // RbfTensor, RbfCoords, RbfNode, RbfFlexNode, Explicit
#include "cr_sweep.h"           // DpState, dp_column, sweep
#include "cr_sweep_cols.h"      // sweep_cols, sweep_cols_team, sweep_cols_score(_team)
#include "cr_sweep_wide.h"      // sweep_team, wide_finish, sweep_wide, sweep_staged
#include "cr_trace.h"           // Walker, ordered sums, Kabsch, RMSD / TM, seed_walk, dtw_walk
#include "cr_pair_kernels.h"    // the kernels

}  // namespace cr
