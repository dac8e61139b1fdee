// gfx950 kernels of the pairwise alignment path.
//
// The DP fills (Smith-Waterman and the 3-layer affine "DTW") run as a STRIP-MINED, TIME-SKEWED
// WAVEFRONT: one 64-lane wave per structure pair, lane l owns R consecutive rows of the current
// strip, and at step t it fills column c = t - l of those rows.  The values of the row above a
// lane's block arrive from lane l-1 by DPP (wave_shr:1), the values to the left stay in registers,
// a strip's last row is handed to the next strip through LDS.  The residue score S(i,j) is never
// materialised: the RBF is evaluated in the sweep from row features held in registers and column
// features streamed through a 128-column LDS ring filled by coalesced HBM reads.  Backtrack
// decisions are packed (2 bit/cell SW, 4 bit/cell DTW) in the same skewed order and written with
// 256-byte coalesced stores; the tracebacks run afterwards, one lane per pair.
//
// Reference semantics: dynamic_time_warping.py (fills, tie-breaks), score_functions.py (RBF),
// superposition_functions.py (Kabsch), multiple_alignment.py:321-349, 1028-1054.
#pragma once

#include "cr_math.h"

namespace cr {

constexpr int kWave = 64;
constexpr int kRing = 128;          // columns held in the LDS ring (two 64-column halves)
constexpr double kMinF64 = -0x1.fffffffffffffp+1023;  // np.finfo(float64).min, dynamic_time_warping.py:4

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
    int64_t pos_off;         // element offset of this pair's seed position list (min(n,m))
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

CR_HD int strips_of(int n, int R) { return (n + kWave * R - 1) / (kWave * R); }
CR_HD int tblocks(int m, int per_word) { return (m + kWave - 1 + per_word - 1) / per_word; }

// ---------------------------------------------------------------------------------------------
// Score providers.  load_rows(): once per strip, lane-private row data into registers.
// load_chunk(): once per 64 steps, the next 64 columns into the LDS ring.  fetch_col(): once per
// step, this lane's column.  score(q): S(row q of this lane, current column).
// ---------------------------------------------------------------------------------------------

// exp(-gamma * sum_k (a_ik - b_jk)^2), k ascending (score_functions.py:7-11).  D is the padded
// width (zero padding adds exact zeros to the sum); `d` is the stored width.
template <int R, int D>
struct RbfTensor {
    const double* __restrict__ rows_g;   // (n, d)
    const double* __restrict__ cols_g;   // (m, d)
    int d;
    double neg_gamma;
    double row[R][D];
    double col[D];
    static constexpr int kRingDoubles = D * kRing;

    CR_D void load_rows(int rowbase, int n) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            int r = rowbase + q;
            r = r < n ? r : n - 1;
#pragma unroll
            for (int k = 0; k < D; k++) row[q][k] = (k < d) ? rows_g[(int64_t)r * d + k] : 0.0;
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
    CR_D double score(int q, const ExpEntry* tab) const {
        double df = row[q][0] - col[0];
        double acc = df * df;
#pragma unroll
        for (int k = 1; k < D; k++) {
            df = row[q][k] - col[k];
            acc = acc + df * df;
        }
        return exp_tab(neg_gamma * acc, tab);
    }
};

// Coordinate RBF on the seed-superposed frames: rows X_i - c1, columns (X_j - c2) @ R
// (superposition_functions.py:57-58), or the raw coordinates when the seed was skipped.
template <int R>
struct RbfCoords {
    const double* __restrict__ rows_g;   // (n, 3)
    const double* __restrict__ cols_g;   // (m, 3)
    const Transform* __restrict__ xf;
    double neg_gamma;
    double row[R][3];
    double col[3];
    static constexpr int kRingDoubles = 3 * kRing;

    CR_D void load_rows(int rowbase, int n) {
        const bool raw = xf->flags & kFlagSeedSkipped;
#pragma unroll
        for (int q = 0; q < R; q++) {
            int r = rowbase + q;
            r = r < n ? r : n - 1;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double v = rows_g[(int64_t)r * 3 + k];
                row[q][k] = raw ? v : v - xf->c1[k];
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
    CR_D double score(int q, const ExpEntry* tab) const {
        double dx = row[q][0] - col[0], dy = row[q][1] - col[1], dz = row[q][2] - col[2];
        double acc = (dx * dx + dy * dy) + dz * dz;
        return exp_tab(neg_gamma * acc, tab);
    }
};

// Explicit score matrix with index sequences: S[seq1[i], seq2[j]] (dynamic_time_warping.py:24-26,79).
template <int R>
struct Explicit {
    const double* __restrict__ S;
    const int32_t* __restrict__ seq1;
    const int32_t* __restrict__ seq2;
    int64_t s_cols;
    const double* rowp[R];
    int64_t colidx;
    static constexpr int kRingDoubles = 0;

    CR_D void load_rows(int rowbase, int n) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            int r = rowbase + q;
            r = r < n ? r : n - 1;
            rowp[q] = S + (int64_t)seq1[r] * s_cols;
        }
    }
    CR_D void init_ring(double*, int) {}
    CR_D void load_chunk(double*, int, int, int) {}
    CR_D void fetch_col(const double*, int) {}
    CR_D void set_col(int c, int m) { colidx = seq2[c < 0 ? 0 : (c < m ? c : m - 1)]; }
    CR_D double score(int q, const ExpEntry*) const { return rowp[q][colidx]; }
};

enum : int { kSwTrace = 1, kSwScore = 2, kDtw = 4 };

struct SweepParams {
    double sw_gap, gap_open, gap_extend;
};

// ---------------------------------------------------------------------------------------------
// The sweep.  One wave, one pair.  MODE selects the recurrences evaluated per cell:
//   kSwTrace : SW fill + 2-bit decisions + first maximum   (dynamic_time_warping.py:226-247)
//   kSwScore : SW fill, maximum only                        (dynamic_time_warping.py:205-222)
//   kDtw     : 3-layer affine fill + 4-bit decisions        (dynamic_time_warping.py:8-86,181-182)
// LDS layout (doubles): [0,32) exp table | ring | strip hand-off rows (nb * m, only if >1 strip).
// ---------------------------------------------------------------------------------------------
template <int R, int MODE, class Src>
CR_D void sweep(Src& src, const int n, const int m, const SweepParams prm, double* lds,
                uint32_t* __restrict__ sw_dirs, uint32_t* __restrict__ dtw_bits,
                SeedMax* seed_out, AlignEnd* end_out) {
    constexpr bool SW = (MODE & (kSwTrace | kSwScore)) != 0;
    constexpr bool TRACE = (MODE & kSwTrace) != 0;
    constexpr bool DTW = (MODE & kDtw) != 0;
    constexpr int NB = (SW ? 1 : 0) + (DTW ? 2 : 0);   // values handed from strip to strip per column
    const int lane = threadIdx.x;
    const ExpEntry* tab = reinterpret_cast<const ExpEntry*>(lds);
    double* ring = lds + 32;
    double* bnd = ring + Src::kRingDoubles;

    if (lane < 16) reinterpret_cast<ExpEntry*>(lds)[lane] = kExpTable[lane];
    src.init_ring(ring, lane);
    __syncthreads();

    const int nstrips = strips_of(n, R);
    const int T = m + kWave - 1;
    const int TB_SW = tblocks(m, 16), TB_DTW = tblocks(m, 8);
    const double col0_m2 = kMinF64 - prm.gap_open;      // M[i][0][2], M[0][j][0] (dynamic_time_warping.py:45,49)

    // first maximum of H in row-major order (smith_waterman, :241-247)
    double best_v = 0.0;
    int best_i = 0x7fffffff, best_j = 0x7fffffff;
    double sw_max = 0.0;
    double fin0 = 0.0, fin1 = 0.0, fin2 = 0.0;          // M[n][m][0..2], held by the owning lane

    for (int s = 0; s < nstrips; s++) {
        const int rowbase = (s * kWave + lane) * R;
        src.load_rows(rowbase, n);
        double h_left[R], m1_left[R], m2_left[R];
        uint32_t swbits[R], dtbits[R];
#pragma unroll
        for (int q = 0; q < R; q++) {
            h_left[q] = 0.0;
            m1_left[q] = 0.0;          // M[i][0][1] = 0
            m2_left[q] = col0_m2;      // M[i][0][2] = MIN - open
            swbits[q] = 0;
            dtbits[q] = 0;
        }
        double h_diag = 0.0, m1_diag = 0.0;             // row above, previous column
        double h_bot = 0.0, m0_bot = 0.0, m1_bot = 0.0; // this lane's last row, current column

        for (int t = 0; t < T; t++) {
            if ((t & (kWave - 1)) == 0) {
                __syncthreads();
                src.load_chunk(ring, t >> 6, m, lane);
                __syncthreads();
            }
            const int c = t - lane;
            const bool active = (unsigned)c < (unsigned)m;
            if constexpr (Src::kRingDoubles == 0) src.set_col(c, m);
            src.fetch_col(ring, c & (kRing - 1));

            // row above this lane's block: lane 0 reads the DP border (strip 0) or the hand-off row
            double h_top0 = 0.0, m0_top0 = col0_m2, m1_top0 = 0.0;   // M[0][j][0] = MIN - open, M[0][j][1] = 0
            if (s > 0 && lane == 0 && active) {
                if constexpr (SW) h_top0 = bnd[c];
                if constexpr (DTW) {
                    m0_top0 = bnd[(NB - 2) * m + c];
                    m1_top0 = bnd[(NB - 1) * m + c];
                }
            }
            double h_top = 0.0, m0_top = 0.0, m1_top = 0.0;
            if constexpr (SW) h_top = wave_shr1(h_bot, h_top0);
            if constexpr (DTW) {
                m0_top = wave_shr1(m0_bot, m0_top0);
                m1_top = wave_shr1(m1_bot, m1_top0);
            }

            double h_up = h_top, h_dg = h_diag;
            double m0_up = m0_top, m1_up = m1_top, m1_dg = m1_diag;
            const int sh2 = (t & 15) * 2, sh4 = (t & 7) * 4;
#pragma unroll
            for (int q = 0; q < R; q++) {
                const double sc = src.score(q, tab);
                const int row = rowbase + q;
                const bool valid = active && row < n;
                if constexpr (SW) {
                    // H = max(0, diag + S, left - gap, up - gap), first maximal argument
                    const double dg = h_dg + sc;
                    const double lf = h_left[q] - prm.sw_gap;
                    const double up = h_up - prm.sw_gap;
                    double h = 0.0;
                    h = dg > h ? dg : h;
                    h = lf > h ? lf : h;
                    h = up > h ? up : h;
                    if constexpr (TRACE) {
                        // decision replayed by the traceback's equality tests (:255-277)
                        uint32_t code = (h == 0.0) ? 0u : (h == dg) ? 1u : (h == lf) ? 2u : 3u;
                        swbits[q] |= (valid ? code : 0u) << sh2;
                        const bool better = valid && (h > best_v || (h == best_v && row < best_i));
                        best_v = better ? h : best_v;
                        best_j = better ? c : best_j;
                        best_i = better ? row : best_i;
                    } else {
                        sw_max = (valid && h > sw_max) ? h : sw_max;
                    }
                    h_dg = h_left[q];
                    h_up = h;
                    h_left[q] = active ? h : h_left[q];
                }
                if constexpr (DTW) {
                    const double lo0 = m0_up - prm.gap_extend;
                    const double lo1 = m1_up - prm.gap_open;
                    const bool b0 = lo1 > lo0;
                    const double m0 = b0 ? lo1 : lo0;
                    const double up0 = m1_left[q] - prm.gap_open;
                    const double up1 = m2_left[q] - prm.gap_extend;
                    const bool b2 = up1 > up0;
                    const double m2 = b2 ? up1 : up0;
                    const double c1 = m1_dg + sc;
                    uint32_t idx = 0;
                    double m1 = m0;
                    if (c1 > m1) { m1 = c1; idx = 1; }
                    if (m2 > m1) { m1 = m2; idx = 2; }
                    const uint32_t nib = (b0 ? 1u : 0u) | (idx << 1) | (b2 ? 8u : 0u);
                    dtbits[q] |= (valid ? nib : 0u) << sh4;
                    if (valid && row == n - 1 && c == m - 1) { fin0 = m0; fin1 = m1; fin2 = m2; }
                    m1_dg = m1_left[q];
                    m0_up = m0;
                    m1_up = m1;
                    m1_left[q] = active ? m1 : m1_left[q];
                    m2_left[q] = active ? m2 : m2_left[q];
                }
            }
            if constexpr (SW) {
                h_diag = active ? h_top : h_diag;
                h_bot = h_up;
            }
            if constexpr (DTW) {
                m1_diag = active ? m1_top : m1_diag;
                m0_bot = m0_up;
                m1_bot = m1_up;
            }
            if (s + 1 < nstrips && lane == kWave - 1 && active) {
                if constexpr (SW) bnd[c] = h_up;
                if constexpr (DTW) {
                    bnd[(NB - 2) * m + c] = m0_up;
                    bnd[(NB - 1) * m + c] = m1_up;
                }
            }
            if constexpr (TRACE) {
                if ((t & 15) == 15 || t == T - 1) {
                    const int64_t base = ((int64_t)(s * TB_SW + (t >> 4)) * R) * kWave + lane;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        sw_dirs[base + q * kWave] = swbits[q];
                        swbits[q] = 0;
                    }
                }
            }
            if constexpr (DTW) {
                if ((t & 7) == 7 || t == T - 1) {
                    const int64_t base = ((int64_t)(s * TB_DTW + (t >> 3)) * R) * kWave + lane;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        dtw_bits[base + q * kWave] = dtbits[q];
                        dtbits[q] = 0;
                    }
                }
            }
        }
    }

    // ---- wave reductions -------------------------------------------------------------------
    if constexpr (TRACE) {
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(best_v, off);
            int oi = __shfl_xor(best_i, off), oj = __shfl_xor(best_j, off);
            bool take = ov > best_v || (ov == best_v && (oi < best_i || (oi == best_i && oj < best_j)));
            best_v = take ? ov : best_v;
            best_i = take ? oi : best_i;
            best_j = take ? oj : best_j;
        }
        if (lane == 0) {
            SeedMax r;
            r.score = best_v;
            r.i = best_v > 0.0 ? best_i + 1 : 0;
            r.j = best_v > 0.0 ? best_j + 1 : 0;
            *seed_out = r;
        }
    }
    if constexpr ((MODE & kSwScore) != 0 || DTW) {
        if constexpr ((MODE & kSwScore) != 0) {
            for (int off = 32; off > 0; off >>= 1) {
                double ov = __shfl_xor(sw_max, off);
                sw_max = ov > sw_max ? ov : sw_max;
            }
        }
        const int owner = ((n - 1) / R) % kWave;       // lane that owns row n-1
        if (lane == owner) {
            AlignEnd e;
            e.sw = sw_max;
            int idx = 0;                               // np.argmax of the three layers at (n, m), :181-182
            double best = fin0;
            if (fin1 > best) { best = fin1; idx = 1; }
            if (fin2 > best) { best = fin2; idx = 2; }
            e.dtw_score = DTW ? best : 0.0;
            e.start_layer = idx;
            e.pad = 0;
            *end_out = e;
        }
    }
}

// LDS doubles needed by a sweep of the given provider/mode for column count m and row count n
template <int R, int MODE, class Src>
__host__ __device__ inline size_t sweep_lds_doubles(int n_max, int m_max) {
    constexpr int NB = ((MODE & (kSwTrace | kSwScore)) ? 1 : 0) + ((MODE & kDtw) ? 2 : 0);
    size_t v = 32 + Src::kRingDoubles;
    if (strips_of(n_max, R) > 1) v += (size_t)NB * m_max;
    return v;
}

// ---------------------------------------------------------------------------------------------
// Batch kernels
// ---------------------------------------------------------------------------------------------

// Stage 1: tensor RBF + SW fill (multiple_alignment.py:328-335).  One wave per pair.
template <int R, int D>
__global__ __launch_bounds__(kWave) void k_seed_fill(const PairDesc* __restrict__ pairs,
                                                    const double* __restrict__ tensors, int d,
                                                    double gamma, double sw_gap,
                                                    uint32_t* __restrict__ dirs, SeedMax* __restrict__ out) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.x];
    RbfTensor<R, D> src;
    src.rows_g = tensors + pd.off_i * d;
    src.cols_g = tensors + pd.off_j * d;
    src.d = d;
    src.neg_gamma = -gamma;
    SweepParams prm{sw_gap, 0.0, 0.0};
    sweep<R, kSwTrace>(src, pd.n, pd.m, prm, lds, dirs + pd.dirs_off, nullptr, out + blockIdx.x, nullptr);
}

// Stage 3: coordinate RBF on the seed-superposed frames + SW score + affine DTW fill
// (multiple_alignment.py:347-349, :164, :263-275).  One wave per pair.
template <int R>
__global__ __launch_bounds__(kWave) void k_align_fill(const PairDesc* __restrict__ pairs,
                                                     const double* __restrict__ coords,
                                                     const Transform* __restrict__ xf, double gamma,
                                                     double sw_gap, double gap_open, double gap_extend,
                                                     uint32_t* __restrict__ bits, AlignEnd* __restrict__ out) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.x];
    RbfCoords<R> src;
    src.rows_g = coords + pd.off_i * 3;
    src.cols_g = coords + pd.off_j * 3;
    src.xf = xf + blockIdx.x;
    src.neg_gamma = -gamma;
    SweepParams prm{sw_gap, gap_open, gap_extend};
    sweep<R, kSwScore | kDtw>(src, pd.n, pd.m, prm, lds, nullptr, bits + pd.bt_off, nullptr, out + blockIdx.x);
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

// sequential column means of gathered coordinates (helper.py:46-53 under numba: sum, then / k)
CR_D void gathered_means(const double* __restrict__ X, const int32_t* __restrict__ pos, int stride, int first,
                         int step, int k, double* c) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int x = 0, p = first; x < k; x++, p += step) {
        const double* v = X + (int64_t)pos[(int64_t)p * stride] * 3;
        s0 += v[0];
        s1 += v[1];
        s2 += v[2];
    }
    c[0] = s0 / (double)k;
    c[1] = s1 / (double)k;
    c[2] = s2 / (double)k;
}

// Stage 2: SW traceback on the stored decisions, common positions, seed Kabsch
// (dynamic_time_warping.py:249-278, helper.py:13-42, superposition_functions.py:39-60).
// One lane per pair.  Positions are written back-to-front while walking and consumed front-to-back
// so that every sum runs in the reference's order.
__global__ void k_seed_trace(const PairDesc* __restrict__ pairs, int npairs, int R,
                             const double* __restrict__ coords, const uint32_t* __restrict__ dirs,
                             const SeedMax* __restrict__ seed, int32_t* __restrict__ pos,
                             Transform* __restrict__ xf, double* __restrict__ seed_score) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const PairDesc pd = pairs[p];
    const SeedMax sm = seed[p];
    Transform tr;
    for (int x = 0; x < 3; x++) tr.c1[x] = tr.c2[x] = 0.0;
    for (int x = 0; x < 9; x++) tr.R[x] = (x % 4 == 0) ? 1.0 : 0.0;
    tr.flags = 0;
    tr.seed_len = 0;
    seed_score[p] = sm.score;
    const int cap = pd.n < pd.m ? pd.n : pd.m;
    int32_t* pp = pos + pd.pos_off * 2;          // pairs (i, j) interleaved
    int k = 0, len = 0;
    if (sm.i == 0) {
        tr.flags |= kFlagSeedAllZero;
    } else {
        const uint32_t* w = dirs + pd.dirs_off;
        const int TB = tblocks(pd.m, 16);
        int i = sm.i, j = sm.j;
        while (i > 0 && j > 0) {
            const uint32_t code = lookup_bits(w, R, TB, 4, 2, i - 1, j - 1);
            if (code == 0) break;
            if (code == 1) {
                i--; j--;
                k++;
                pp[2 * (cap - k)] = i;
                pp[2 * (cap - k) + 1] = j;
            } else if (code == 2) {
                j--;
            } else {
                i--;
            }
            len++;
        }
    }
    tr.seed_len = len;
    if (k <= 3) {
        tr.flags |= kFlagSeedSkipped;
    } else {
        const double* Xi = coords + pd.off_i * 3;
        const double* Xj = coords + pd.off_j * 3;
        const int first = cap - k;
        gathered_means(Xi, pp, 2, first, 1, k, tr.c1);
        gathered_means(Xj, pp + 1, 2, first, 1, k, tr.c2);
        double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int x = 0; x < k; x++) {
            const double* v1 = Xi + (int64_t)pp[2 * (first + x)] * 3;
            const double* v2 = Xj + (int64_t)pp[2 * (first + x) + 1] * 3;
            const double a[3] = {v2[0] - tr.c2[0], v2[1] - tr.c2[1], v2[2] - tr.c2[2]};
            const double b[3] = {v1[0] - tr.c1[0], v1[1] - tr.c1[1], v1[2] - tr.c1[2]};
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) C[3 * r + c] += a[r] * b[c];
        }
        double t[3];
        kabsch_from_correlation(C, tr.c1, tr.c2, tr.R, t);
    }
    xf[p] = tr;
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

// Stage 4: DTW traceback (dynamic_time_warping.py:90-144), common positions, Kabsch on the
// original coordinates, RMSD / coverage / TM (multiple_alignment.py:1033-1054, :59-70).
// One lane per pair.  Alignment rows are written back-to-front into [aln_off, aln_off + n + m).
__global__ void k_align_trace(const PairDesc* __restrict__ pairs, int npairs, int R,
                              const double* __restrict__ coords, const uint32_t* __restrict__ bits,
                              const AlignEnd* __restrict__ ends, const Transform* __restrict__ xf,
                              const double* __restrict__ seed_score, int32_t* __restrict__ aln,
                              PairResult* __restrict__ res) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const PairDesc pd = pairs[p];
    const AlignEnd e = ends[p];
    PairResult r;
    r.sw = e.sw;
    r.dtw_score = e.dtw_score;
    for (int x = 0; x < 9; x++) r.R[x] = 0.0;
    for (int x = 0; x < 3; x++) r.t[x] = 0.0;
    r.rmsd = r.coverage = r.tm = 0.0;
    r.seed_score = seed_score[p];
    r.seed_len = xf[p].seed_len;
    r.flags = xf[p].flags;
    const int cap = pd.n + pd.m;
    int32_t* a1 = aln + pd.aln_off;
    int32_t* a2 = a1 + cap;
    const uint32_t* w = bits + pd.bt_off;
    const int TB = tblocks(pd.m, 8);
    const int idx = dtw_traceback(w, R, TB, pd.n, pd.m, e.start_layer, a1, a2, cap);
    const int len = idx, first = cap - idx;
    r.aln_len = len;
    r.aln_start = first;
    // common positions in alignment order (helper.py:13-42)
    int k = 0;
    for (int x = first; x < cap; x++) k += (a1[x] != -1 && a2[x] != -1) ? 1 : 0;
    if (k < 3) {
        r.flags |= kFlagMetricsSkipped;
    } else {
        const double* Xi = coords + pd.off_i * 3;
        const double* Xj = coords + pd.off_j * 3;
        double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
        for (int x = first; x < cap; x++) {
            const int i = a1[x], j = a2[x];
            if (i != -1 && j != -1) {
                const double* v1 = Xi + (int64_t)i * 3;
                const double* v2 = Xj + (int64_t)j * 3;
                s1[0] += v1[0]; s1[1] += v1[1]; s1[2] += v1[2];
                s2[0] += v2[0]; s2[1] += v2[1]; s2[2] += v2[2];
            }
        }
        double c1[3], c2[3];
        for (int x = 0; x < 3; x++) { c1[x] = s1[x] / (double)k; c2[x] = s2[x] / (double)k; }
        double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int x = first; x < cap; x++) {
            const int i = a1[x], j = a2[x];
            if (i != -1 && j != -1) {
                const double* v1 = Xi + (int64_t)i * 3;
                const double* v2 = Xj + (int64_t)j * 3;
                const double a[3] = {v2[0] - c2[0], v2[1] - c2[1], v2[2] - c2[2]};
                const double b[3] = {v1[0] - c1[0], v1[1] - c1[1], v1[2] - c1[2]};
#pragma unroll
                for (int rr = 0; rr < 3; rr++)
#pragma unroll
                    for (int cc = 0; cc < 3; cc++) C[3 * rr + cc] += a[rr] * b[cc];
            }
        }
        kabsch_from_correlation(C, c1, c2, r.R, r.t);
        // get_rmsd (score_functions.py:15-19) and tm_score (multiple_alignment.py:59-70)
        const double d1 = 1.24 * (double)(pd.n - 15) / 3.0 - 1.8;
        const double d2 = 1.24 * (double)(pd.m - 15) / 3.0 - 1.8;
        double ss = 0.0, sum1 = 0.0, sum2 = 0.0;
        for (int x = first; x < cap; x++) {
            const int i = a1[x], j = a2[x];
            if (i != -1 && j != -1) {
                const double* v1 = Xi + (int64_t)i * 3;
                const double* v2 = Xj + (int64_t)j * 3;
                double mv[3];
                rot3(v2, r.R, mv);
                mv[0] += r.t[0]; mv[1] += r.t[1]; mv[2] += r.t[2];
                const double e0 = v1[0] - mv[0], e1 = v1[1] - mv[1], e2 = v1[2] - mv[2];
                ss += e0 * e0;
                ss += e1 * e1;
                ss += e2 * e2;
                const double sg = (e0 + e1) + e2;
                const double q1 = sg / d1, q2 = sg / d2;
                sum1 += 1.0 / (1.0 + q1 * q1);
                sum2 += 1.0 / (1.0 + q2 * q2);
            }
        }
        r.rmsd = sqrt(ss / (double)k);
        r.coverage = (double)k / (double)len;
        const double t1 = (1.0 / (double)pd.n) * sum1;
        const double t2 = (1.0 / (double)pd.m) * sum2;
        r.tm = t1 > t2 ? t1 : t2;
    }
    res[p] = r;
}

}  // namespace cr
