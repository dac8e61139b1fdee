// Neighbor joining of a SYMMETRIC distance matrix on the device (neighbor_joining.py:19-157), included by cr_api.hip
// after cr_dropins.h.  The guide tree is the one sequential step between the P x P matrix and the progressive
// alignment; at P = 512 the host implementation (cr_neighbor_joining) takes as long as half the matrix.
//
// Same values as the reference, bit for bit:
//  * a row sum is ONE lane's left-to-right sum over the current node order starting from +0.0 (numba's np.sum,
//    :117-118), formed once per iteration instead of once per (i, j);
//  * Q(i, j) = ((n - 2) * D[i][j] - sum_i) - sum_j is evaluated for every ORDERED pair, and the first minimum in
//    row-major order wins (:98-121, strict <): each lane scans its pairs in that order, lanes, waves and workgroups
//    are combined by (Q, i, j) lexicographically;
//  * the reference rebuilds the matrix with the new node FIRST and the survivors behind it in their old order
//    (:60-83).  Here nothing moves: the matrix has a row and a column for every node id (2P - 3 of them), and `order`
//    lists the node ids in the reference's order.
//
// Shape.  One persistent launch of up to kNjMaxGroups workgroups (a fraction of the chip: the work per join is O(n^2)
// loads and an n-long chain of dependent adds).  Per join:
//    A  row sums of the NEW order.  A wave owns a few rows; it loads them 64 columns per instruction (a row sum reads
//       the ROW, gathered through `order`), parks 256 columns per row in LDS and lets one lane per row add them in
//       order while the next 256 columns are in flight.  The new node's row does not exist yet: its entries
//       v_t = 0.5 * ((D[i][t] + D[j][t]) - D[i][j]) are computed where they are needed -- by the wave that owns the
//       new row (which also writes the new row and column) and, for column 0 of every other row, by that row's wave --
//       from rows i and j, which nobody writes.  Each sum is published as one 8-byte word.
//    B  every workgroup collects all row sums, every wave scans Q over its rows (matrix entries requested before the
//       sums arrive; the entry in the new column is the one the wave formed itself in A) and publishes its first
//       minimum as three 8-byte words.
//    C  every workgroup collects all minima and reduces them (same data, same order: same answer), updates its own
//       copy of `order`, and goes on with A.
// There is no barrier in the loop: a consumer polls the published words themselves.  Each word lives in three
// generations (join k uses generation k % 3); a producer writes its word of this generation and resets its word of
// the next one to a reserved NaN pattern, and a consumer spins on a word until it is not that pattern.  Everything
// that crosses workgroups -- the words, and every matrix entry, since rows keep getting entries appended by whoever
// makes the next node -- moves with device-scope loads and stores (written through to memory, read past the per-XCD
// L2), so the loop holds no cache write-back and no invalidate; a wave publishes its minimum only after its own
// stores have been acknowledged (s_waitcnt vmcnt(0)), and nobody starts the next A before it has every minimum:
// that is what orders the new row and column, and the resets, before their readers.  A join is then 3 - 4 dependent
// round trips to memory instead of 7 with two counter barriers.  A spin that does not end (lost workgroup) gives up
// after a bounded number of polls and the call returns an error.
// The matrix is symmetric and stays so (both triangles receive the same rounded value), which is what lets A read rows
// i and j where the reference reads columns; cr_neighbor_joining_device checks the symmetry of its input and hands
// anything else to the host implementation.
#pragma once

namespace cr {

constexpr int kNjGroupThreads = 256;
constexpr int kNjGroupWaves = kNjGroupThreads / 64;
constexpr int kNjMaxGroups = 64;
constexpr int kNjRowsPerWave = 8;          // rows of one wave (register staging: 4 doubles per row and lane)
constexpr int kNjChunk = 256;              // columns parked in LDS per step
constexpr int kNjStride = kNjChunk + 2;    // row pitch of the parking area: rows two bank pairs apart
constexpr int kNjMaxNodes = kNjMaxGroups * kNjGroupWaves * kNjRowsPerWave;      // 2048
constexpr unsigned kNjSpinLimit = 1u << 24;

struct NjGridState {
    unsigned arrived;      // monotone: += 1 per workgroup and barrier (the one barrier before the loop)
    unsigned abort;        // a workgroup gave up waiting (a lost workgroup must not hang the device)
};

// a published word that has not been written in this generation: a NaN no sum, Q value or matrix entry can be
// (sums and entries are finite -- checked on the host --, and a Q value that is NaN never wins a comparison, so it
// is never published)
constexpr unsigned long long kNjEmpty = 0x7ff8dead0badf00dull;
constexpr int kNjCandWords = 4;            // Q, D[i][j], (i << 32 | j), unused

inline size_t nj_lds_bytes(int P, int rows_per_wave) {
    return sizeof(double) * ((size_t)kNjGroupWaves * rows_per_wave * kNjStride + (size_t)P) + sizeof(int) * 2 * (size_t)P;
}

CR_D bool nj_before(double q, int i, int j, double q2, int i2, int j2) {
    return q < q2 || (q == q2 && (i < i2 || (i == i2 && j < j2)));
}

CR_D void nj_wave_first_min(double& q, double& d, int& i, int& j) {
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const double q2 = __shfl_xor(q, o), d2 = __shfl_xor(d, o);
        const int i2 = __shfl_xor(i, o), j2 = __shfl_xor(j, o);
        if (nj_before(q2, i2, j2, q, i, j)) {
            q = q2;
            d = d2;
            i = i2;
            j = j2;
        }
    }
}

CR_D unsigned long long nj_peek(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
CR_D void nj_post(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a matrix entry as the device sees it now (entries are appended to rows by other workgroups: a cached line may
// predate its newest entry)
CR_D double nj_entry(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
// every store this wave has issued has been acknowledged (the words are device-scope stores, written through: once
// acknowledged they are where a device-scope load finds them)
CR_D void nj_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
CR_D unsigned long long nj_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
CR_D double nj_value(unsigned long long b) { return __longlong_as_double((long long)b); }

// spin until the word has been published; false: gave up (and told everybody)
CR_D bool nj_await(const unsigned long long* p, NjGridState* gs, unsigned long long& v) {
    unsigned spins = 0;
    while ((v = nj_peek(p)) == kNjEmpty) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kNjSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(&gs->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// All workgroups of the launch meet here; what they wrote before is visible to all after.  false: somebody timed out.
CR_D bool nj_grid_sync(NjGridState* gs, unsigned& target, int* ok) {
    target += gridDim.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&gs->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool good = true;
        unsigned spins = 0;
        while (__hip_atomic_load(&gs->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kNjSpinLimit || __hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&gs->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *ok = good ? 1 : 0;
    }
    __syncthreads();
    return *ok != 0;
}

// Row sums of the order `order[0 .. n)`, published in rs_pub (and the words of the next generation reset).  JOIN:
// position 0 is the new node `order[0]`, not in memory yet, formed from rows pi and pj; the wave that owns position
// 0 also writes its row and column.  newcol[u] (lane 0): the entry of row u in the new node's column.
template <bool JOIN>
CR_D void nj_row_sums(double* D, int W, const int* order, int n, int rows_per_wave, double* stage, unsigned long long* rs_pub,
                      unsigned long long* rs_reset, int pi, int pj, double dij, double (&newcol)[kNjRowsPerWave]) {
    // (the wave index through readfirstlane: the compiler then knows that everything derived from it is wave-uniform and
    // branches on it with scalar jumps, which do not serialise the loads on either side)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_total = gridDim.x * kNjGroupWaves, gw = blockIdx.x * kNjGroupWaves + wave;
    if (gw >= n) return;                                    // no row for this wave (rows are dealt round robin)
    double* park = stage + (size_t)wave * rows_per_wave * kNjStride;
    const int my_rows = min(rows_per_wave, (n - gw + waves_total - 1) / waves_total);
    auto entry_of_new = [&](int p) { return 0.5 * ((nj_entry(D + (size_t)pi * W + p) + nj_entry(D + (size_t)pj * W + p)) - dij); };
    int prow[kNjRowsPerWave];
#pragma unroll
    for (int u = 0; u < kNjRowsPerWave; u++) prow[u] = u < my_rows ? __builtin_amdgcn_readfirstlane(order[gw + u * waves_total]) : 0;
    // the entry of every row in the new node's column (formed here; the copy in memory is being written by wave 0)
#pragma unroll
    for (int u = 0; u < kNjRowsPerWave; u++) newcol[u] = (JOIN && u < my_rows && !(gw == 0 && u == 0)) ? entry_of_new(prow[u]) : 0.0;
    double x[kNjRowsPerWave][4];
    // element t of the row at position gw + u * waves_total, for t = c0 + lane + 64 e.  No lane-dependent branch: every
    // lane loads (from a clamped column) and selects afterwards, so all the loads of a chunk are in flight together
    auto load = [&](int c0) {
#pragma unroll
        for (int u = 0; u < kNjRowsPerWave; u++)
            if (u < my_rows) {
                const bool new_row = JOIN && gw == 0 && u == 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int t = c0 + lane + 64 * e;
                    const int pt = order[min(t, n - 1)];
                    const double v = new_row ? entry_of_new(pt) : nj_entry(D + (size_t)prow[u] * W + pt);
                    x[u][e] = t < n ? v : 0.0;
                }
            }
        if (JOIN && c0 == 0 && lane == 0) {
#pragma unroll
            for (int u = 0; u < kNjRowsPerWave; u++) x[u][0] = newcol[u];
        }
    };
    double s = 0.0;
    load(0);
    for (int c0 = 0; c0 < n; c0 += kNjChunk) {
#pragma unroll
        for (int u = 0; u < kNjRowsPerWave; u++)
            if (u < my_rows) {
#pragma unroll
                for (int e = 0; e < 4; e++) park[u * kNjStride + lane + 64 * e] = x[u][e];
            }
        if (JOIN && gw == 0) {                              // the new node's row and column (its diagonal 0 included)
            const int fresh = prow[0];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int t = c0 + lane + 64 * e;
                if (t < n) {
                    const int pt = order[t];
                    nj_post(reinterpret_cast<unsigned long long*>(D + (size_t)fresh * W + pt), nj_bits(x[0][e]));
                    nj_post(reinterpret_cast<unsigned long long*>(D + (size_t)pt * W + fresh), nj_bits(x[0][e]));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (c0 + kNjChunk < n) load(c0 + kNjChunk);
        if (lane < my_rows) {
            // columns past n are parked as +0.0, and a sum that starts at +0.0 is never -0.0: adding them changes nothing,
            // so the chain runs in blocks of 16.  Two register blocks in turn, each read one block ahead of its adds (the
            // last read-ahead runs past the chunk into the next row or the array behind: read, never added); the same
            // number of reads is outstanding on every path into the loop, so the waits the compiler places are exact
            const double* row = park + lane * kNjStride;
            const int cnt = (min(kNjChunk, n - c0) + 15) & ~15;
            double a[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = row[k];
            for (int t = 0; t < cnt; t += 16) {
#pragma unroll
                for (int k = 0; k < 8; k++) b[k] = row[t + 8 + k];
                __builtin_amdgcn_sched_barrier(0);          // (the scheduler would sink the reads below the adds)
#pragma unroll
                for (int k = 0; k < 8; k++) s += a[k];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k++) a[k] = row[t + 16 + k];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k++) s += b[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // (the new row and column are device-scope stores still in flight here: nobody reads them before the next A, and
    // no workgroup gets there before it has this wave's minimum, which is sent after the stores are acknowledged)
    if (lane < my_rows) {
        nj_post(&rs_pub[gw + lane * waves_total], nj_bits(s));
        nj_post(&rs_reset[gw + lane * waves_total], kNjEmpty);
    }
}

__global__ __launch_bounds__(kNjGroupThreads) void k_neighbor_joining(const double* __restrict__ dense, double* D, int W, int P, int rows_per_wave,
                                                                      unsigned long long* rs_words, unsigned long long* cand_words,
                                                                      NjGridState* gs, unsigned long long* __restrict__ tree,
                                                                      double* __restrict__ bl, long long* prof) {
    extern __shared__ double nj_lds[];
    // prof (diagnostic, else null): shader-clock cycles workgroup 0 spent in A (row sums), collecting the sums, B (Q scan),
    // collecting the minima, C (reduction + new order)
    long long clk = 0;
    auto lap = [&](int k) {
        if (prof && blockIdx.x == 0 && threadIdx.x == 0) {
            const long long now = __builtin_readcyclecounter();
            if (k >= 0) prof[k] += now - clk;
            clk = now;
        }
    };
    double* stage = nj_lds;                                                 // [waves][rows_per_wave][kNjStride]
    double* rs = stage + (size_t)kNjGroupWaves * rows_per_wave * kNjStride; // row sum by position
    int* order = reinterpret_cast<int*>(rs + P);                            // node id (= matrix row) by position, two copies
    __shared__ double red_q[kNjGroupWaves], red_d[kNjGroupWaves];
    __shared__ int red_i[kNjGroupWaves], red_j[kNjGroupWaves], sync_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int waves_total = gridDim.x * kNjGroupWaves, gw = blockIdx.x * kNjGroupWaves + wave;
    const double inf = __builtin_inf();
    for (int t = tid; t < P; t += kNjGroupThreads) order[t] = t;
    int* order_new = order + P;
    int n = P, index = 0, node = P;
    unsigned target = 0;
    // the caller's P x P matrix into the (2P - 3)-wide layout (a strided upload from pageable memory goes row by row),
    // every published word empty
    for (int r = blockIdx.x; r < P; r += gridDim.x)
        for (int c = tid; c < P; c += kNjGroupThreads) D[(size_t)r * W + c] = dense[(size_t)r * P + c];
    for (int t = blockIdx.x * kNjGroupThreads + tid; t < 3 * P; t += gridDim.x * kNjGroupThreads) nj_post(&rs_words[t], kNjEmpty);
    for (int t = blockIdx.x * kNjGroupThreads + tid; t < 3 * waves_total * kNjCandWords; t += gridDim.x * kNjGroupThreads)
        nj_post(&cand_words[t], kNjEmpty);
    if (!nj_grid_sync(gs, target, &sync_ok)) return;
    lap(-1);
    double newcol[kNjRowsPerWave];
    int gen = 0;                                                            // generation of the published words: join % 3
    nj_row_sums<false>(D, W, order, n, rows_per_wave, stage, rs_words, rs_words + P, 0, 0, 0.0, newcol);
    bool joined = false;
    lap(0);
    while (n > 3) {
        const int gen_next = gen == 2 ? 0 : gen + 1;
        // B: first minimum of Q over this wave's rows (the rows it summed).  The matrix entries are requested before
        // the row sums are collected, 256 columns ahead of the scan.
        const double nm2 = (double)(n - 2);
        double bq = inf, bd = 0.0;
        int bi = 0x7fffffff, bj = 0x7fffffff;
        {
            const int my_rows = gw < n ? min(rows_per_wave, (n - gw + waves_total - 1) / waves_total) : 0;
            const double* rowp[kNjRowsPerWave];
#pragma unroll
            for (int u = 0; u < kNjRowsPerWave; u++)
                rowp[u] = D + (size_t)(u < my_rows ? __builtin_amdgcn_readfirstlane(order[gw + u * waves_total]) : 0) * W;
            double x[kNjRowsPerWave][4];
            auto load = [&](int c0) {
#pragma unroll
                for (int u = 0; u < kNjRowsPerWave; u++)
                    if (u < my_rows) {
#pragma unroll
                        for (int e = 0; e < 4; e++) x[u][e] = nj_entry(rowp[u] + order[min(c0 + lane + 64 * e, n - 1)]);
                    }
                // column 0 of a joined order is the node made in A: other waves' stores, not acquired yet -- the wave
                // formed the same value itself
                if (joined && c0 == 0 && lane == 0) {
#pragma unroll
                    for (int u = 0; u < kNjRowsPerWave; u++) x[u][0] = newcol[u];
                }
            };
            load(0);
            bool lost = false;
            const unsigned long long* words = rs_words + (size_t)gen * P;
            for (int t = tid; t < n; t += kNjGroupThreads) {
                unsigned long long v;
                lost |= !nj_await(&words[t], gs, v);
                rs[t] = nj_value(v);
            }
            if (__syncthreads_or(lost)) return;
            lap(1);
            // a lane sees (i, j) out of row-major order across rows and chunks, so ties are broken by (i, j) explicitly
            for (int c0 = 0; c0 < n; c0 += kNjChunk) {
                double y[kNjRowsPerWave][4];
#pragma unroll
                for (int u = 0; u < kNjRowsPerWave; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++) y[u][e] = x[u][e];
                if (c0 + kNjChunk < n) load(c0 + kNjChunk);
#pragma unroll
                for (int u = 0; u < kNjRowsPerWave; u++)
                    if (u < my_rows) {
                        const int i = gw + u * waves_total;
                        const double ri = rs[i];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int j = c0 + lane + 64 * e;
                            if (j < n && j != i) {
                                const double q = (nm2 * y[u][e] - ri) - rs[j];
                                if (nj_before(q, i, j, bq, bi, bj)) {
                                    bq = q;
                                    bd = y[u][e];
                                    bi = i;
                                    bj = j;
                                }
                            }
                        }
                    }
            }
        }
        nj_wave_first_min(bq, bd, bi, bj);
        if (lane == 0) {
            unsigned long long* mine = cand_words + ((size_t)gen * waves_total + gw) * kNjCandWords;
            unsigned long long* next = cand_words + ((size_t)gen_next * waves_total + gw) * kNjCandWords;
            // The resets of the next generation are posted FIRST and waited for together with this wave's other stores (the
            // new row and column if it made them, the row-sum resets of A): everything has been acknowledged before the
            // words that let the others go on are sent, so no consumer that has seen this minimum can still find the value
            // of join k - 2 in a word of generation k + 1.
            nj_post(&next[0], kNjEmpty);
            nj_post(&next[1], kNjEmpty);
            nj_post(&next[2], kNjEmpty);
            nj_stores_done();
            nj_post(&mine[0], nj_bits(bq));
            nj_post(&mine[1], nj_bits(bd));
            nj_post(&mine[2], ((unsigned long long)(unsigned)bi << 32) | (unsigned)bj);
        }
        lap(2);
        // C: the same reduction in every workgroup
        bq = inf;
        bd = 0.0;
        bi = bj = 0x7fffffff;
        bool lost = false;
        for (int w = tid; w < waves_total; w += kNjGroupThreads) {
            const unsigned long long* c = cand_words + ((size_t)gen * waves_total + w) * kNjCandWords;
            unsigned long long vq, vd, vij;
            lost |= !nj_await(&c[0], gs, vq);
            lost |= !nj_await(&c[1], gs, vd);
            lost |= !nj_await(&c[2], gs, vij);
            const int ci = (int)(unsigned)(vij >> 32), cj = (int)(unsigned)(vij & 0xffffffffu);
            if (nj_before(nj_value(vq), ci, cj, bq, bi, bj)) {
                bq = nj_value(vq);
                bd = nj_value(vd);
                bi = ci;
                bj = cj;
            }
        }
        nj_wave_first_min(bq, bd, bi, bj);
        if (lane == 0) {
            red_q[wave] = bq;
            red_d[wave] = bd;
            red_i[wave] = bi;
            red_j[wave] = bj;
        }
        if (__syncthreads_or(lost)) return;
        lap(3);
        bq = red_q[0];
        bd = red_d[0];
        bi = red_i[0];
        bj = red_j[0];
        for (int w = 1; w < kNjGroupWaves; w++)
            if (nj_before(red_q[w], red_i[w], red_j[w], bq, bi, bj)) {
                bq = red_q[w];
                bd = red_d[w];
                bi = red_i[w];
                bj = red_j[w];
            }
        int pi, pj;
        double dij;
        if (bi >= n) {                                      // nothing compared below +inf (inf input)
            bi = 0;
            bj = 1;
            pi = order[0];
            pj = order[1];
            dij = nj_entry(D + (size_t)pi * W + pj);
        } else {
            pi = order[bi];
            pj = order[bj];
            dij = bd;
        }
        if (blockIdx.x == 0 && tid == 0) {                  // _find_branch_length (:124-136)
            const double di = 0.5 * dij + (0.5 / nm2) * (rs[bi] - rs[bj]);
            tree[2 * index] = (unsigned long long)pi;
            tree[2 * index + 1] = (unsigned long long)node;
            bl[index] = di;
            tree[2 * index + 2] = (unsigned long long)pj;
            tree[2 * index + 3] = (unsigned long long)node;
            bl[index + 1] = dij - di;
        }
        // the new order: the new node, then the survivors (:60-83)
        for (int k = tid; k < n; k += kNjGroupThreads) {
            if (k == bi || k == bj) continue;
            order_new[1 + k - (k > bi ? 1 : 0) - (k > bj ? 1 : 0)] = order[k];
        }
        if (tid == 0) order_new[0] = node;
        __syncthreads();
        n--;
        lap(4);
        gen = gen_next;
        nj_row_sums<true>(D, W, order_new, n, rows_per_wave, stage, rs_words + (size_t)gen * P,
                          rs_words + (size_t)(gen == 2 ? 0 : gen + 1) * P, pi, pj, dij, newcol);
        joined = true;
        int* swap = order;
        order = order_new;
        order_new = swap;
        index += 2;
        node++;
        lap(0);
    }
    // the last three nodes in order (:85-94); the last new row and column were written by wave 0 of this workgroup
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        auto at = [&](int a, int b) { return nj_entry(D + (size_t)order[a] * W + order[b]); };
        const double s1 = ((0.0 + at(1, 0)) + at(1, 1)) + at(1, 2), s2 = ((0.0 + at(2, 0)) + at(2, 1)) + at(2, 2);
        const double d12 = at(1, 2);
        const double di = 0.5 * d12 + (0.5 / (double)(n - 2)) * (s1 - s2);
        tree[2 * index] = (unsigned long long)order[1];
        tree[2 * index + 1] = (unsigned long long)node;
        bl[index++] = di;
        tree[2 * index] = (unsigned long long)order[2];
        tree[2 * index + 1] = (unsigned long long)node;
        bl[index++] = d12 - di;
        tree[2 * index] = (unsigned long long)order[0];
        tree[2 * index + 1] = (unsigned long long)node;
        bl[index++] = 0.5 * ((at(1, 0) + at(2, 0)) - d12);
    }
}

}  // namespace cr

namespace {

// workgroups of the launch: enough waves for 2 rows each, at most kNjMaxGroups (CARETTA_NJ_GROUPS overrides, for
// calibration)
int nj_groups(int64_t P) {
    int g = (int)std::min<int64_t>(cr::kNjMaxGroups, std::max<int64_t>(1, (P + 2 * cr::kNjGroupWaves - 1) / (2 * cr::kNjGroupWaves)));
    if (g_cfg.nj_groups >= 1 && g_cfg.nj_groups <= cr::kNjMaxGroups) g = g_cfg.nj_groups;
    while ((int64_t)g * cr::kNjGroupWaves * cr::kNjRowsPerWave < P) g++;
    return g;
}

}  // namespace

extern "C" {

// neighbor_joining.neighbor_joining on the device (see the top of this file): the result equals cr_neighbor_joining's
// bit for bit.  A matrix that is not exactly symmetric, or has more than kNjMaxNodes rows, runs on the host
// implementation, whose row sums read rows.
int cr_neighbor_joining_device(cr_context* ctx, const double* D0, int64_t P, uint64_t* tree, double* bl) {
    CR_REQUIRE(D0 && tree && bl, "null argument");
    CR_REQUIRE(P >= 3, "neighbor joining needs at least 3 taxa");
    // the device kernel wants a symmetric matrix of finite entries (its row sums read rows, and a NaN is its mark for
    // "not published yet"); anything else is the host implementation's
    bool symmetric = P <= cr::kNjMaxNodes;
    for (int64_t i = 0; i < P && symmetric; i++)
        for (int64_t j = 0; j <= i; j++)
            if (std::memcmp(&D0[i * P + j], &D0[j * P + i], sizeof(double)) != 0 || !std::isfinite(D0[i * P + j])) {
                symmetric = false;
                break;
            }
    if (!symmetric) return cr_neighbor_joining(D0, P, tree, bl);
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    int rc = set_device(ctx);
    if (rc) return rc;
    const int groups = nj_groups(P);
    const int waves_total = groups * cr::kNjGroupWaves;
    const int rows_per_wave = (int)((P + waves_total - 1) / waves_total);
    const int64_t W = (2 * P - 3 + 15) / 16 * 16;            // a row and a column per node id
    const size_t lds = cr::nj_lds_bytes((int)P, rows_per_wave);
    rc = allow_lds(cr::k_neighbor_joining, lds);
    if (rc) return rc;
    // the workgroups wait for each other: all of them must fit on the device at once (a partition with few CUs may not
    // take them), else the host implementation
    {
        int per_cu = 0, cus = 0;
        CR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cr::k_neighbor_joining, cr::kNjGroupThreads, lds));
        CR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
        if ((int64_t)per_cu * cus < groups) return cr_neighbor_joining(D0, P, tree, bl);
    }
    const size_t rows = (size_t)(2 * P - 3);
    DevBuf<double> d, dense;
    DevBuf<unsigned long long> dwords;                     // published words: [3][P] row sums, [3][waves][kNjCandWords] minima
    DevBuf<unsigned long long> dout;                       // [tree 2 * rows | branch lengths rows | barrier state]
    DevBuf<long long> dprof;
    const bool profile = g_cfg.nj_profile;     // diagnostic: cycles per phase to stderr
    if (profile) {
        CR_HIP(dprof.ensure(8));
        CR_HIP(hipMemsetAsync(dprof.p, 0, 8 * sizeof(long long), ctx->stream));
    }
    CR_HIP(d.ensure((size_t)(W * W)));
    CR_HIP(dense.ensure((size_t)(P * P)));
    CR_HIP(dwords.ensure(3 * (size_t)P + 3 * (size_t)waves_total * cr::kNjCandWords));
    CR_HIP(dout.ensure(3 * rows + 1));
    unsigned long long* dtree = dout.p;
    double* dbl = reinterpret_cast<double*>(dout.p + 2 * rows);
    cr::NjGridState* dstate = reinterpret_cast<cr::NjGridState*>(dout.p + 3 * rows);
    const double ms_alloc = ms_since(t_start);
    CR_HIP(hipMemsetAsync(dstate, 0, sizeof(cr::NjGridState), ctx->stream));
    rc = upload_async(ctx, dense.p, D0, sizeof(double) * (size_t)(P * P));
    if (rc) return rc;
    double ms_copied = 0.0;
    if (profile) {
        CR_HIP(hipStreamSynchronize(ctx->stream));
        ms_copied = ms_since(t_start);
    }
    CR_LAUNCH(cr::k_neighbor_joining, dim3(groups), dim3(cr::kNjGroupThreads), lds, ctx->stream, dense.p, d.p, (int)W, (int)P,
              rows_per_wave, dwords.p, dwords.p + 3 * (size_t)P, dstate, dtree, dbl, profile ? dprof.p : nullptr);
    CR_HIP(hipGetLastError());
    double ms_kernel = 0.0;
    if (profile) {
        CR_HIP(hipStreamSynchronize(ctx->stream));
        ms_kernel = ms_since(t_start);
    }
    const double ms_enqueued = ms_since(t_start);
    // one copy of [tree | branch lengths | barrier state] into page-locked memory
    const size_t tree_bytes = sizeof(uint64_t) * 2 * rows, bl_bytes = sizeof(double) * rows;
    void* land = nullptr;
    rc = host_landing(ctx, tree_bytes + bl_bytes + sizeof(cr::NjGridState), &land);
    if (rc) return rc;
    CR_HIP(hipMemcpyAsync(land, dout.p, tree_bytes + bl_bytes + sizeof(cr::NjGridState), hipMemcpyDeviceToHost, ctx->stream));
    CR_HIP(hipStreamSynchronize(ctx->stream));
    cr::NjGridState state;
    std::memcpy(tree, land, tree_bytes);
    std::memcpy(bl, static_cast<char*>(land) + tree_bytes, bl_bytes);
    std::memcpy(&state, static_cast<char*>(land) + tree_bytes + bl_bytes, sizeof(state));
    if (profile) {
        long long c[8];
        CR_HIP(hipMemcpy(c, dprof.p, sizeof(c), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[nj P=%lld groups=%d] cycles: row sums %lld  collecting them %lld  Q scan %lld  collecting minima %lld  reduce+order %lld;"
                     " host ms: buffers %.3f  copied %.3f  kernel done %.3f  enqueued %.3f  done %.3f\n",
                     (long long)P, groups, c[0], c[1], c[2], c[3], c[4], ms_alloc, ms_copied, ms_kernel, ms_enqueued, ms_since(t_start));
    }
    if (state.abort) {
        // A workgroup did not arrive (the device is shared or busy and the workgroups were not co-resident): a scheduling
        // accident, not an error of the input.  The host implementation gives the same tree bit for bit;
        // CARETTA_NJ_DEVICE_STRICT=1 keeps the error (tests of the device kernel itself).
        if (g_cfg.nj_device_strict) return fail(CR_ERR_HIP, "neighbor joining: a workgroup of the persistent launch did not arrive");
        return cr_neighbor_joining(D0, P, tree, bl);
    }
    return CR_OK;
}

}  // extern "C"
