// libcaretta_hip: C ABI (include/caretta_hip.h) over the gfx950 kernels in cr_kernels.h.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.build()).
#include "../../include/caretta_hip.h"

#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <thread>
#include <functional>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "cr_config.h"
#include "cr_kernels.h"
#include "cr_ilp_instances.h"
#include "cr_duo.h"
#include "cr_trio.h"
#include "cr_duo_instances.h"
#include "cr_flexible.h"

// compiled in cr_kernels_ilp.hip with another instruction scheduler (the diagnostic stamps build is one translation
// unit: its stamp buffer is a static __device__ array)
#ifndef CR_STAMPS
#define CR_X(R, D, ZG) extern template CR_SEED_SIGNATURE(R, D, ZG)
CR_ILP_SEED_INSTANCES(CR_X)
#undef CR_X
#define CR_X(R, ZG) extern template CR_ALIGN_SIGNATURE(R, ZG)
CR_ILP_ALIGN_INSTANCES(CR_X)
#undef CR_X
#define CR_X(R, D, ZG) extern template CR_SEED_TEAM_SIGNATURE(R, D, ZG)
CR_ILP_SEED_TEAM_INSTANCES(CR_X)
#undef CR_X
#define CR_X(R) extern template CR_NODE_TEAM_SIGNATURE(R)
CR_ILP_NODE_TEAM_INSTANCES(CR_X)
#undef CR_X
#define CR_X(RA, RB, D, ZG) extern template CR_SEED_WIDE_SIGNATURE(RA, RB, D, ZG)
CR_ILP_SEED_WIDE_INSTANCES(CR_X)
#undef CR_X
#define CR_X(RA, RB, D, ZG, SC) extern template CR_PAIR_WIDE_SIGNATURE(RA, RB, D, ZG, SC)
CR_ILP_PAIR_WIDE_INSTANCES(CR_X)
#undef CR_X
#define CR_X(RA, RB, D, SC) extern template CR_PAIR_DUO_SIGNATURE(RA, RB, D, SC)
CR_DUO_INSTANCES(CR_X)
#undef CR_X
#define CR_X(R, D, SC) extern template CR_PAIR_TRIO_SIGNATURE(R, D, SC)
CR_TRIO_INSTANCES(CR_X)
#undef CR_X
#endif

namespace {

thread_local std::string g_err;
// calibration switches: read from the environment when the library is loaded, again only by cr_config_reload()
crcfg::Calibration g_cfg = crcfg::Calibration::from_env();
// Work may be in flight on the device since the last device-wide wait of this thread (set by every API entry and every
// kernel launch; DevBuf::release waits once and clears it, instead of once per buffer).
thread_local bool g_dirty = true;
thread_local bool g_no_trio = false;     // ... when a trio batch of at most kTeamPairLimit pairs meets sw_gap != 0 (the one-pair-per-CU layouts serve it better)
thread_local bool g_no_duo = false;      // set around cr_batch_set_pairs when a duo batch meets sw_gap != 0 (run_batch)
thread_local bool g_no_wide = false;     // set around cr_batch_set_pairs by callers whose second kernel has no wide version (cr_progressive_node)

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define CR_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(_e == hipErrorOutOfMemory ? CR_ERR_MEMORY : CR_ERR_HIP,                        \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                            \
    } while (0)

#define CR_REQUIRE(cond, msg)                                 \
    do {                                                      \
        if (!(cond)) return fail(CR_ERR_ARGUMENT, (msg));     \
    } while (0)

// Device blocks are recycled instead of returned to the driver: hipMalloc / hipFree cost tens of microseconds to
// milliseconds each, which is most of the wall time of a single-call drop-in and ~20 % of a 128-structure
// make_pairwise_matrix.  Blocks are binned by size class (8 steps per power of two); a block goes back to its bin
// only after the device has drained (every caller has synchronised its stream by then; the device-wide wait is the
// safety net hipFree used to provide).  CARETTA_NO_CACHE=1 turns the cache off; cr_device_trim() empties it.
struct BlockCache {
    static constexpr size_t kLargest = (size_t)16 << 30;
    std::mutex mu;
    std::map<size_t, std::vector<void*>> bins;
    size_t held = 0;
    // Bytes kept per device at most: a tenth of the device's memory, 24 GiB at the most (28.8 GB on a 288 GB MI355X ->
    // 24 GiB; a small partition keeps proportionally less), so that what other allocators of the process (PyTorch's)
    // cannot see stays a bounded fraction; CARETTA_CACHE_MB overrides it.  (The decision scratch of a 512-structure
    // batch is 8 GiB, and re-allocating blocks of that size from the driver costs 300-400 ms.)
    size_t cap = 0;                                       // 0: not determined yet (first give() on the device)
    size_t limit() {                                      // called with `mu` held and the device current
        if (cap) return cap;
        const long long mb = g_cfg.cache_mb;
        if (mb > 0) return cap = (size_t)mb << 20;
        size_t free_b = 0, total_b = 0;
        cap = (size_t)24 << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) cap = std::min(cap, total_b / 10);
        else (void)hipGetLastError();
        return cap;
    }
    static size_t size_class(size_t bytes) {          // 8 steps per power of two: at most 12.5 % over
        size_t c = 256;
        while (c * 2 < bytes) c *= 2;
        if (bytes <= c) return c;
        const size_t q = c / 8;
        return c + (bytes - c + q - 1) / q * q;
    }
    static bool enabled() { return !g_cfg.no_cache; }
    void* take(size_t cls) {
        std::lock_guard<std::mutex> lock(mu);
        auto it = bins.find(cls);
        if (it == bins.end() || it->second.empty()) return nullptr;
        void* p = it->second.back();
        it->second.pop_back();
        held -= cls;
        return p;
    }
    bool give(void* p, size_t cls) {
        std::lock_guard<std::mutex> lock(mu);
        if (cls > kLargest || held + cls > limit()) return false;
        bins[cls].push_back(p);
        held += cls;
        return true;
    }
    void trim() {
        std::lock_guard<std::mutex> lock(mu);
        for (auto& b : bins)
            for (void* p : b.second) (void)hipFree(p);
        bins.clear();
        held = 0;
    }
};

BlockCache& block_cache(int device) {
    static BlockCache* caches = new BlockCache[64];      // never destroyed: the HIP runtime may be gone by then
    return caches[device & 63];
}

// device buffer that releases itself
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t cls = 0;      // bytes of the underlying block (0: not from the cache)
    int dev = 0;
    bool borrowed = false;   // another DevBuf owns the block (the size classes of a ragged pair list share their parent's structures)
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void borrow(const DevBuf& o) {
        release();
        p = o.p;
        n = o.n;
        borrowed = true;
    }
    void release() {
        if (p && !borrowed) {
            bool kept = false;
            if (cls) {
                int cur = 0;
                (void)hipGetDevice(&cur);
                if (cur != dev) (void)hipSetDevice(dev);
                if (g_dirty || cur != dev) {             // one wait for all the buffers a call releases
                    (void)hipDeviceSynchronize();
                    g_dirty = false;
                }
                kept = block_cache(dev).give(p, cls);
                if (cur != dev) (void)hipSetDevice(cur);
            }
            if (!kept) (void)hipFree(p);
        }
        p = nullptr;
        n = 0;
        cls = 0;
        borrowed = false;
    }
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        if (BlockCache::enabled()) {
            (void)hipGetDevice(&dev);
            cls = BlockCache::size_class(bytes);
            p = static_cast<T*>(block_cache(dev).take(cls));
            if (p) {
                n = count;
                return hipSuccess;
            }
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), cls);
            if (e == hipErrorOutOfMemory) {              // give the cached blocks back and try once more
                block_cache(dev).trim();
                e = hipMalloc(reinterpret_cast<void**>(&p), cls);
            }
            if (e == hipSuccess) n = count;
            else { p = nullptr; cls = 0; }
            return e;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), bytes);
        if (e == hipSuccess) n = count;
        return e;
    }
};

// page-locked host staging buffer (device <-> host copies without the driver's bounce buffer)
template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
        const size_t want = std::max<size_t>(count + count / 2, 64);
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) n = want;
        return e;
    }
};

}  // namespace

struct cr_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // profiling: a ring of event lists, one list (1 + 2 * chunks events) per recorded run
    std::vector<std::vector<hipEvent_t>> ev;
    int slots = 0;
    int64_t runs_recorded = 0;
    // Side streams for batches whose pairs fall into several rows-per-lane groups: the groups are independent, and
    // run side by side they fill each other's partial last rounds (created on first use).
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> sync_ev;
    // copy stream of the pipelined run + fetch (cr_batch_run_fetch_i32): results of one part of the pair list travel to
    // the host while the kernels of the next part run (created on first use)
    // page-locked landing area for small results of single calls; grown on demand by host_landing()
    void* landing = nullptr;
    size_t landing_bytes = 0;
    // page-locked ring for transfers between the device and the caller's PAGEABLE memory (upload_async / download)
    void* ring = nullptr;
    hipEvent_t ring_ev[2] = {nullptr, nullptr};
    bool ring_busy[2] = {false, false};
};

struct cr_batch {
    cr_context* ctx = nullptr;
    int64_t P = 0, d = 0, total = 0;
    std::vector<int64_t> offsets;
    DevBuf<double> coords, tensors;
    DevBuf<int64_t> d_offsets;          // offsets on the device (pair descriptors of equal-length lists are built there)
    DevBuf<int32_t> d_ij;               // the caller's (i, j) list on the device (the same fast path)
    // pair list
    int64_t npairs = 0;
    int r_seed = 5, r_align = 5, d_pad = 0;
    bool team = false;                  // few pairs: one workgroup of kTeamWaves waves per pair (k_seed_team / k_align_team)
    int wide_sync = 0;                  // > 0: the wide kernels (one wave per strip, up to 16 waves per pair) with a barrier every wide_sync steps
    int r_b = 5, wide_na = 0;           // wide kernels: strips [0, wide_na) have r_seed rows per lane, the others r_b (r_b == r_seed: all alike)
    std::vector<int32_t> duo_ij;        // ... the caller's pair list (k_pair_duo is built for sw_gap == 0: another gap lays the list out again)
    std::vector<int32_t> relaid_ij;     // ... the list a gap-driven re-layout was made from: its upload may still be in flight, so it lives with the batch
    bool trio_few = false;              // ... chosen for a list the one-pair-per-CU layouts would take with a Smith-Waterman gap (laid out again then)
    bool trio = false;                  // the single-wave LAYOUT (5 rows per lane, one strip) on k_pair_trio (cr_trio.h): one wave of recurrences + two of scores per pair
    bool duo = false;                   // the wide LAYOUT on k_pair_duo (cr_duo.h): 2 .. 4 waves per pair paced by LDS progress words, several pairs per CU
    bool staged = false;                // scores formed by their own launches, sweeps with one row per lane (cr_staged.h)
    DevBuf<double> staged_scores;       // ... one chunk's scores
    int n_max = 0, m_max = 0;
    int64_t max_aln = 0;
    std::vector<cr::PairDesc> h_pairs;  // in LAUNCH order: most cells first (order[k] = index in the caller's list)
    std::vector<int32_t> order;
    bool reordered = false;             // false: order is the identity (equal costs), no un-permuting needed
    DevBuf<int32_t> d_order;
    DevBuf<cr::PairDesc> pairs;
    DevBuf<uint32_t> dirs, bits;
    DevBuf<double> hand;                // strip hand-off rows of multi-strip pairs (per chunk)
    DevBuf<int32_t> aln;
    DevBuf<cr::Transform> xf;
    DevBuf<double> seed_score;
    hipStream_t launch_stream = nullptr;  // stream of the next launch_seed / launch_align (null: the context's)
    cr::HostOut host_out{};               // cr_batch_run_stream_i32: page-locked arrays the align kernels write results into
    DevBuf<double> sw_stage;            // cr_batch_fetch_scores: the sw field gathered on the device
    DevBuf<cr::PairResult> res_packed;  // cr_batch_fetch*: results / alignment rows in the caller's order and layout
    DevBuf<int64_t> aln_packed;
    DevBuf<cr::PairResult> res;
    int64_t aln_elems = 0;
    double alg_bytes = 0.0, cells = 0.0;
    bool ran = false;
    bool scores_only = false;           // the last run was cr_batch_run_scores: no alignments, transforms or metrics to fetch
    // The decision scratch (dirs, bits) is sized for one CHUNK of the pair list and reused chunk after
    // chunk in stream order, so an arbitrarily long pair list runs in bounded HBM.
    struct Chunk {
        int64_t first, count;
        int n_max, m_max, max_aln;
        int r = 0;                       // rows per lane of this chunk's kernels (pairs are grouped by it)
        int lane = 0;                    // 0: the context's stream; k > 0: side stream k-1 (one lane per group)
    };
    std::vector<Chunk> chunks;
    // A RAGGED list is split into at most three size classes (cr_batch_set_pairs): each class is a batch of its own on
    // this batch's structures (coords / tensors / offsets borrowed), with its own kernel family, scratch and launch
    // sequence; its order map leads straight to the caller's pair indices, so results land in the caller's order.
    std::vector<cr_batch*> parts;
    bool is_part = false;
    std::vector<int32_t> part_global;   // (a part) the caller's index of every pair of this class, in list order
    int base_lane = 0;                  // a part's launches go to stream (base_lane + chunk lane) mod kGroupLanes of the context
    ~cr_batch() {
        for (cr_batch* c : parts) delete c;
    }
};

static_assert(sizeof(cr::PairResult) == sizeof(cr_pair_result), "device/host result layouts differ");

namespace {

int set_device(cr_context* ctx) {
    CR_REQUIRE(ctx != nullptr, "null context");
    CR_HIP(hipSetDevice(ctx->device));
    g_dirty = true;
    return CR_OK;
}

// at least `bytes` of page-locked host memory owned by the context
int host_landing(cr_context* ctx, size_t bytes, void** out) {
    if (ctx->landing_bytes < bytes) {
        if (ctx->landing) (void)hipHostFree(ctx->landing);
        ctx->landing = nullptr;
        ctx->landing_bytes = 0;
        const size_t want = std::max<size_t>(bytes, 64 * 1024);
        CR_HIP(hipHostMalloc(&ctx->landing, want, hipHostMallocDefault));
        ctx->landing_bytes = want;
    }
    *out = ctx->landing;
    return CR_OK;
}

// Copies between the device and memory the library does not own (the caller's arrays, host vectors).  Pageable memory
// handed to hipMemcpyAsync for a large copy is pinned by the runtime as a user pointer; when the host allocator later
// unmaps or recycles such pages the process's queues are held back while the mapping is revalidated -- measured as the
// first kernel of the NEXT call starting ~20 ms late (tools/stall_probe.py).  So nothing of 64 KB
// or more is ever handed over directly: it is staged through a page-locked ring
// of two slots owned by the context, filled / drained by memcpy (~10 GB/s) while the other slot is on the wire, so the
// device side of every transfer is a plain DMA from or to page-locked memory.  Page-locked caller memory
// (cr_host_alloc) and small blocks go straight through.  Uploads return with the last slots still in flight (the ring
// is guarded by events; the caller's buffer is free again on return); downloads return when the data is in `dst`.
constexpr size_t kRingSlot = (size_t)8 << 20;
constexpr size_t kStageFrom = (size_t)64 << 10;

bool is_page_locked(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();                                   // unknown to the runtime: ordinary pageable memory
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

int ring_slot(cr_context* ctx, int slot, char** out) {
    if (!ctx->ring) {
        CR_HIP(hipHostMalloc(&ctx->ring, 2 * kRingSlot, hipHostMallocDefault));
        for (hipEvent_t& e : ctx->ring_ev) CR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (ctx->ring_busy[slot]) {
        CR_HIP(hipEventSynchronize(ctx->ring_ev[slot]));
        ctx->ring_busy[slot] = false;
    }
    *out = static_cast<char*>(ctx->ring) + (size_t)slot * kRingSlot;
    return CR_OK;
}

int upload_async(cr_context* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes < kStageFrom || is_page_locked(src)) {
        CR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return CR_OK;
    }
    int slot = 0;
    for (size_t off = 0; off < bytes; off += kRingSlot, slot ^= 1) {
        const size_t len = std::min(kRingSlot, bytes - off);
        char* stage = nullptr;
        int rc = ring_slot(ctx, slot, &stage);
        if (rc) return rc;
        std::memcpy(stage, static_cast<const char*>(src) + off, len);
        CR_HIP(hipMemcpyAsync(static_cast<char*>(dst) + off, stage, len, hipMemcpyHostToDevice, ctx->stream));
        CR_HIP(hipEventRecord(ctx->ring_ev[slot], ctx->stream));
        ctx->ring_busy[slot] = true;
    }
    return CR_OK;
}

// device -> caller memory; complete on return (wait = false: a direct copy may still be in flight, the caller waits)
int download(cr_context* ctx, void* dst, const void* src, size_t bytes, bool wait = true) {
    if (bytes < kStageFrom || is_page_locked(dst)) {
        CR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (wait) CR_HIP(hipStreamSynchronize(ctx->stream));
        return CR_OK;
    }
    // slot k is on the wire while slot k - 1 is drained into dst
    size_t prev_off = 0, prev_len = 0;
    int slot = 0;
    char* prev_stage = nullptr;
    for (size_t off = 0; off < bytes; off += kRingSlot, slot ^= 1) {
        const size_t len = std::min(kRingSlot, bytes - off);
        char* stage = nullptr;
        int rc = ring_slot(ctx, slot, &stage);
        if (rc) return rc;
        CR_HIP(hipMemcpyAsync(stage, static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost, ctx->stream));
        CR_HIP(hipEventRecord(ctx->ring_ev[slot], ctx->stream));
        ctx->ring_busy[slot] = true;
        if (prev_stage) {
            CR_HIP(hipEventSynchronize(ctx->ring_ev[slot ^ 1]));
            ctx->ring_busy[slot ^ 1] = false;
            std::memcpy(static_cast<char*>(dst) + prev_off, prev_stage, prev_len);
        }
        prev_stage = stage;
        prev_off = off;
        prev_len = len;
    }
    if (prev_stage) {
        CR_HIP(hipEventSynchronize(ctx->ring_ev[slot ^ 1]));
        ctx->ring_busy[slot ^ 1] = false;
        std::memcpy(static_cast<char*>(dst) + prev_off, prev_stage, prev_len);
    }
    return CR_OK;
}

// the copy helpers as statements (return the error code of the enclosing entry point)
#define CR_UPLOAD(ctx_, dst, src, bytes)                               \
    do {                                                               \
        const int _rc = upload_async((ctx_), (dst), (src), (bytes));   \
        if (_rc) return _rc;                                           \
    } while (0)
#define CR_DOWNLOAD(ctx_, dst, src, bytes)                             \
    do {                                                               \
        const int _rc = download((ctx_), (dst), (src), (bytes), false);\
        if (_rc) return _rc;                                           \
    } while (0)
#define CR_DOWNLOAD_WAIT(ctx_, dst, src, bytes)                        \
    do {                                                               \
        const int _rc = download((ctx_), (dst), (src), (bytes), true); \
        if (_rc) return _rc;                                           \
    } while (0)

// every kernel launch of the library goes through this (see g_dirty)
#define CR_LAUNCH(...)                   \
    do {                                 \
        g_dirty = true;                  \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

template <class K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return fail(CR_ERR_ARGUMENT, "pair too long: the alignment columns of one pair (n + m <= 39000) must fit the 160 KiB LDS");
    if (bytes > 48 * 1024)
        CR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)bytes));
    return CR_OK;
}

template <int R, int D, bool ZG>
int launch_seed_zg(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    using Src = cr::RbfTensor<R, D>;
    const int entries = std::min(ck.n_max, ck.m_max);
    // gap 0: the column sweep needs the exp table only (no column ring, the strip hand-off goes through HBM)
    const size_t fill = ZG ? (size_t)cr::kExpDoubles : cr::sweep_lds_doubles<R, cr::kSwTrace, Src>(ck.n_max, ck.m_max);
    const size_t lds = sizeof(double) * std::max(fill, (size_t)cr::kExpDoubles + cr::trace_lds_doubles(R, entries));
    int rc = allow_lds(cr::k_seed<R, D, ZG>, lds);
    if (rc) return rc;
    CR_LAUNCH((cr::k_seed<R, D, ZG>), dim3((unsigned)ck.count), dim3(cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
                       b->pairs.p + ck.first, b->tensors.p, (int)b->d, b->coords.p, prm.gamma_tensor, prm.sw_gap,
                       entries, b->dirs.p, b->hand.p, b->xf.p + ck.first, b->seed_score.p + ck.first);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int R, int D>
int launch_seed(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    return prm.sw_gap == 0.0 ? launch_seed_zg<R, D, true>(b, ck, prm) : launch_seed_zg<R, D, false>(b, ck, prm);
}

template <int R>
int launch_seed_d(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    switch (b->d_pad) {
        case 4: return launch_seed<R, 4>(b, ck, prm);
        case 8: return launch_seed<R, 8>(b, ck, prm);
        case 10: return launch_seed<R, 10>(b, ck, prm);
        case 16: return launch_seed<R, 16>(b, ck, prm);
        case 24: return launch_seed<R, 24>(b, ck, prm);
        case 32: return launch_seed<R, 32>(b, ck, prm);
        default: return fail(CR_ERR_ARGUMENT, "unsupported tensor width");
    }
}

// cr::WidePlan with run-time rows per lane (the host's view: strips and row slots of a pair with n rows)
struct StripPlan {
    int ra, rb, na;
    int strips(int n) const {
        if (ra == rb || n <= na * cr::kWave * ra) return (n + cr::kWave * ra - 1) / (cr::kWave * ra);
        return na + (n - na * cr::kWave * ra + cr::kWave * rb - 1) / (cr::kWave * rb);
    }
    int slots(int n) const {
        const int st = strips(n);
        return (ra == rb || st <= na) ? st * ra : na * ra + (st - na) * rb;
    }
};
StripPlan plan_of(const cr_batch* b) { return StripPlan{b->r_seed, b->wide_sync ? b->r_b : b->r_seed, b->wide_sync ? b->wide_na : 0}; }

// the host arrays of a streamed run as this chunk's launch sees them
cr::HostOut host_out_for(const cr_batch* b, const cr_batch::Chunk& ck) {
    cr::HostOut h = b->host_out;
    h.first = (int32_t)ck.first;
    return h;
}

template <int R, bool ZG>
int launch_align_zg(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    using Src = cr::RbfCoords<R>;
    const int entries = ck.max_aln;
    const size_t fill = cr::sweep_lds_doubles<R, cr::kSwScore | cr::kDtw, Src>(ck.n_max, ck.m_max);
    const size_t lds = sizeof(double) * std::max(fill, (size_t)cr::kExpDoubles + cr::trace_lds_doubles(R, entries));
    int rc = allow_lds(cr::k_align<R, ZG>, lds);
    if (rc) return rc;
    CR_LAUNCH((cr::k_align<R, ZG>), dim3((unsigned)ck.count), dim3(cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
                       b->pairs.p + ck.first, b->coords.p, b->xf.p + ck.first, b->seed_score.p + ck.first,
                       prm.gamma_coords, prm.sw_gap, prm.gap_open, prm.gap_extend, entries, b->bits.p, b->hand.p, b->aln.p,
                       b->res.p + ck.first, host_out_for(b, ck));
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int R>
int launch_align(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    return prm.sw_gap == 0.0 ? launch_align_zg<R, true>(b, ck, prm) : launch_align_zg<R, false>(b, ck, prm);
}

template <int R>
int launch_score(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const size_t lds = sizeof(double) * (cr::kExpDoubles + cr::RbfCoords<R>::kRingDoubles);
    CR_LAUNCH(cr::k_score<R>, dim3((unsigned)ck.count), dim3(cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
              b->pairs.p + ck.first, b->coords.p, b->xf.p + ck.first, b->seed_score.p + ck.first, prm.gamma_coords, b->hand.p,
              b->res.p + ck.first);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int RA, int RB>
int launch_score_team(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const int waves = plan_of(b).strips(ck.n_max);
    const size_t lds = sizeof(double) * cr::sweep_cols_score_team_lds_doubles<cr::RbfCoords<RA>>(waves);
    CR_LAUNCH((cr::k_score_team<RA, RB>), dim3((unsigned)ck.count), dim3(waves * cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
              b->pairs.p + ck.first, b->coords.p, b->xf.p + ck.first, b->seed_score.p + ck.first, prm.gamma_coords, b->wide_na, b->res.p + ck.first);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

int launch_score_team_r(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    return R == 1 ? launch_score_team<1, 1>(b, ck, prm) : R == 2 ? launch_score_team<2, 2>(b, ck, prm) : R == 3 ? launch_score_team<3, 3>(b, ck, prm)
         : R == 4 ? launch_score_team<4, 4>(b, ck, prm) : launch_score_team<5, 5>(b, ck, prm);
}

int launch_score_r(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    return R == 1 ? launch_score<1>(b, ck, prm) : R == 2 ? launch_score<2>(b, ck, prm) : R == 3 ? launch_score<3>(b, ck, prm)
         : R == 4 ? launch_score<4>(b, ck, prm) : launch_score<5>(b, ck, prm);
}

// flexible=True (cr_flexible.h): the tensor score matrix's smith_waterman_score alone
template <int R, int D>
int launch_tensor_score(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const size_t lds = sizeof(double) * (cr::kExpDoubles + cr::RbfTensor<R, D>::kRingDoubles);
    CR_LAUNCH((cr::k_tensor_score<R, D>), dim3((unsigned)ck.count), dim3(cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
              b->pairs.p + ck.first, b->tensors.p, (int)b->d, prm.gamma_tensor, b->hand.p, b->res.p + ck.first);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int R>
int launch_tensor_score_d(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    switch (b->d_pad) {
        case 4: return launch_tensor_score<R, 4>(b, ck, prm);
        case 8: return launch_tensor_score<R, 8>(b, ck, prm);
        case 10: return launch_tensor_score<R, 10>(b, ck, prm);
        case 16: return launch_tensor_score<R, 16>(b, ck, prm);
        case 24: return launch_tensor_score<R, 24>(b, ck, prm);
        case 32: return launch_tensor_score<R, 32>(b, ck, prm);
        default: return fail(CR_ERR_ARGUMENT, "unsupported tensor width");
    }
}

// (rows per lane: at least the layout's, so that a pair that needs several strips here has its hand-off row in b->hand)
int launch_tensor_score_r(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    return R <= 2 ? launch_tensor_score_d<2>(b, ck, prm) : R == 3 ? launch_tensor_score_d<3>(b, ck, prm)
         : R == 4 ? launch_tensor_score_d<4>(b, ck, prm) : launch_tensor_score_d<5>(b, ck, prm);
}

// The fused kernels feed rows past the end of a structure features of 1e150 so that their RBF score underflows to
// exactly 0 (they then never win a maximum).  That needs gamma * 1e300 > 745; below 1e-290 every real score is
// exactly 1.0 anyway, and such a gamma is rejected rather than computed wrongly.
template <int R, bool ZG>
int launch_align_team_zg(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    using Src = cr::RbfCoords<R>;
    const int entries = ck.max_aln;
    const size_t lds = sizeof(double) * std::max(cr::sweep_team_lds_doubles<R, cr::kSwScore | cr::kDtw, Src>(cr::kTeamWaves),
                                                 (size_t)cr::kExpDoubles + cr::trace_lds_doubles(R, entries));
    int rc = allow_lds(cr::k_align_team<R, ZG>, lds);
    if (rc) return rc;
    CR_LAUNCH((cr::k_align_team<R, ZG>), dim3((unsigned)ck.count), dim3(cr::kTeamWaves * cr::kWave), lds, b->launch_stream ? b->launch_stream : b->ctx->stream,
                       b->pairs.p + ck.first, b->coords.p, b->xf.p + ck.first, b->seed_score.p + ck.first, prm.gamma_coords,
                       prm.sw_gap, prm.gap_open, prm.gap_extend, entries, b->bits.p, b->aln.p, b->res.p + ck.first, host_out_for(b, ck));
    CR_HIP(hipGetLastError());
    return CR_OK;
}

int launch_align_team(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const bool zg = prm.sw_gap == 0.0;
    switch (R) {
        case 1: return zg ? launch_align_team_zg<1, true>(b, ck, prm) : launch_align_team_zg<1, false>(b, ck, prm);
        case 2: return zg ? launch_align_team_zg<2, true>(b, ck, prm) : launch_align_team_zg<2, false>(b, ck, prm);
        case 3: return zg ? launch_align_team_zg<3, true>(b, ck, prm) : launch_align_team_zg<3, false>(b, ck, prm);
        case 4: return zg ? launch_align_team_zg<4, true>(b, ck, prm) : launch_align_team_zg<4, false>(b, ck, prm);
        default: return zg ? launch_align_team_zg<5, true>(b, ck, prm) : launch_align_team_zg<5, false>(b, ck, prm);
    }
}

}  // namespace

#include "cr_staged.h"      // scores formed by their own launches: kernels and launchers

namespace {

// ---- wide kernels: one wave per strip, up to kWideMaxWaves waves per pair, columns resident in LDS ----------
template <int RA, int RB, int D, bool ZG>
int launch_seed_wide_zg(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    using Src = cr::RbfTensor<RA, D>;
    const int entries = std::min(ck.n_max, ck.m_max);
    const int waves = plan_of(b).strips(ck.n_max);
    // gap 0: the column sweep (scalar column loads, no resident columns)
    const size_t fill = ZG ? cr::sweep_cols_team_lds_doubles(waves) : cr::sweep_wide_lds_doubles<cr::kSwTrace, Src>(waves, ck.m_max);
    const size_t lds = sizeof(double) * std::max(fill, (size_t)cr::kExpDoubles + cr::trace_lds_doubles(RA, entries));
    int rc = allow_lds(cr::k_seed_wide<RA, RB, D, ZG>, lds);
    if (rc) return rc;
    CR_LAUNCH((cr::k_seed_wide<RA, RB, D, ZG>), dim3((unsigned)ck.count), dim3(waves * cr::kWave), lds,
                       b->launch_stream ? b->launch_stream : b->ctx->stream, b->pairs.p + ck.first, b->tensors.p, (int)b->d,
                       b->coords.p, prm.gamma_tensor, prm.sw_gap, entries, b->wide_sync, b->wide_na, b->dirs.p, b->xf.p + ck.first,
                       b->seed_score.p + ck.first);
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int RA, int RB>
int launch_seed_wide_r(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const bool zg = prm.sw_gap == 0.0;
    switch (b->d_pad) {
        case 4: return zg ? launch_seed_wide_zg<RA, RB, 4, true>(b, ck, prm) : launch_seed_wide_zg<RA, RB, 4, false>(b, ck, prm);
        case 8: return zg ? launch_seed_wide_zg<RA, RB, 8, true>(b, ck, prm) : launch_seed_wide_zg<RA, RB, 8, false>(b, ck, prm);
        case 10: return zg ? launch_seed_wide_zg<RA, RB, 10, true>(b, ck, prm) : launch_seed_wide_zg<RA, RB, 10, false>(b, ck, prm);
        case 16: return zg ? launch_seed_wide_zg<RA, RB, 16, true>(b, ck, prm) : launch_seed_wide_zg<RA, RB, 16, false>(b, ck, prm);
        default: return fail(CR_ERR_ARGUMENT, "unsupported tensor width");
    }
}

int launch_seed_wide(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    if (b->r_b != b->r_seed) return launch_seed_wide_r<3, 2>(b, ck, prm);
    return R == 2 ? launch_seed_wide_r<2, 2>(b, ck, prm) : launch_seed_wide_r<3, 3>(b, ck, prm);
}

// both stages in one launch (k_pair_wide)
template <int RA, int RB, int D, bool ZG, bool SC>
int launch_pair_wide_t(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const int seed_entries = std::min(ck.n_max, ck.m_max), align_entries = ck.max_aln;
    const int waves = plan_of(b).strips(ck.n_max);
    const size_t seed_fill = ZG ? cr::sweep_cols_team_lds_doubles(waves) : cr::sweep_wide_lds_doubles<cr::kSwTrace, cr::RbfTensor<RA, D>>(waves, ck.m_max);
    // (the sums behind the walks are taken by the whole workgroup: a term tile next to the entries)
    const size_t second = SC ? cr::sweep_cols_score_team_lds_doubles<cr::RbfCoords<RA>>(waves)
                             : std::max(cr::sweep_wide_lds_doubles<cr::kSwScore | cr::kDtw, cr::RbfCoords<RA>>(waves, ck.m_max),
                                        (size_t)cr::kExpDoubles + cr::trace_team_lds_doubles(align_entries));
    const size_t lds = sizeof(double) * std::max(std::max(seed_fill, (size_t)cr::kExpDoubles + cr::trace_team_lds_doubles(seed_entries)), second);
    int rc = allow_lds(cr::k_pair_wide<RA, RB, D, ZG, SC>, lds);
    if (rc) return rc;
    CR_LAUNCH((cr::k_pair_wide<RA, RB, D, ZG, SC>), dim3((unsigned)ck.count), dim3(waves * cr::kWave), lds,
              b->launch_stream ? b->launch_stream : b->ctx->stream, b->pairs.p + ck.first, b->tensors.p, (int)b->d, b->coords.p,
              prm.gamma_tensor, prm.gamma_coords, prm.sw_gap, prm.gap_open, prm.gap_extend, seed_entries, align_entries, b->wide_sync,
              b->wide_na, b->dirs.p, b->bits.p, b->xf.p + ck.first, b->seed_score.p + ck.first, b->aln.p, b->res.p + ck.first,
              host_out_for(b, ck));
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int RA, int RB, int D>
int launch_pair_wide_d(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, bool scores) {
    if (prm.sw_gap != 0.0) return launch_pair_wide_t<RA, RB, D, false, false>(b, ck, prm);
    return scores ? launch_pair_wide_t<RA, RB, D, true, true>(b, ck, prm) : launch_pair_wide_t<RA, RB, D, true, false>(b, ck, prm);
}

template <int RA, int RB>
int launch_pair_wide_r(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, bool scores) {
    switch (b->d_pad) {
        case 4: return launch_pair_wide_d<RA, RB, 4>(b, ck, prm, scores);
        case 8: return launch_pair_wide_d<RA, RB, 8>(b, ck, prm, scores);
        case 10: return launch_pair_wide_d<RA, RB, 10>(b, ck, prm, scores);
        case 16: return launch_pair_wide_d<RA, RB, 16>(b, ck, prm, scores);
        default: return fail(CR_ERR_ARGUMENT, "unsupported tensor width");
    }
}

int launch_pair_wide(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, bool scores) {
    if (b->r_b != b->r_seed) return launch_pair_wide_r<3, 2>(b, ck, prm, scores);
    return b->r_seed == 2 ? launch_pair_wide_r<2, 2>(b, ck, prm, scores) : launch_pair_wide_r<3, 3>(b, ck, prm, scores);
}

// ---- mid-size pair lists: the wide layout on small workgroups paced by progress words (cr_duo.h); gap 0 only ----------
template <int RA, int RB, int D, bool SC>
int launch_pair_duo_t(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const int seed_entries = std::min(ck.n_max, ck.m_max), align_entries = ck.max_aln;
    const int waves = plan_of(b).strips(ck.n_max);
    const size_t seed = std::max(cr::duo_cols_lds_doubles(waves, ck.m_max), (size_t)cr::kExpDoubles + cr::trace_lds_doubles(RA, seed_entries));
    const size_t second = SC ? cr::duo_score_lds_doubles<cr::RbfCoords<RA>>(waves, ck.m_max)
                             : std::max(cr::duo_sweep_lds_doubles<cr::kSwScore | cr::kDtw, cr::RbfCoords<RA>>(waves, ck.m_max),
                                        (size_t)cr::kExpDoubles + cr::trace_lds_doubles(RA, align_entries));
    size_t lds = sizeof(double) * std::max(seed, second);
    if (g_cfg.mid_lds_kb > 0) lds = std::max(lds, (size_t)g_cfg.mid_lds_kb * 1024);   // calibration: pairs per CU
    int rc = allow_lds(cr::k_pair_duo<RA, RB, D, SC>, lds);
    if (rc) return rc;
    CR_LAUNCH((cr::k_pair_duo<RA, RB, D, SC>), dim3((unsigned)ck.count), dim3(waves * cr::kWave), lds,
              b->launch_stream ? b->launch_stream : b->ctx->stream, b->pairs.p + ck.first, b->tensors.p, (int)b->d, b->coords.p,
              prm.gamma_tensor, prm.gamma_coords, prm.gap_open, prm.gap_extend, seed_entries, align_entries, b->wide_na, b->dirs.p,
              b->bits.p, b->xf.p + ck.first, b->seed_score.p + ck.first, b->aln.p, b->res.p + ck.first, host_out_for(b, ck));
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int RA, int RB>
int launch_pair_duo_r(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, bool scores) {
    switch (b->d_pad) {
        case 4: return scores ? launch_pair_duo_t<RA, RB, 4, true>(b, ck, prm) : launch_pair_duo_t<RA, RB, 4, false>(b, ck, prm);
        case 8: return scores ? launch_pair_duo_t<RA, RB, 8, true>(b, ck, prm) : launch_pair_duo_t<RA, RB, 8, false>(b, ck, prm);
        case 10: return scores ? launch_pair_duo_t<RA, RB, 10, true>(b, ck, prm) : launch_pair_duo_t<RA, RB, 10, false>(b, ck, prm);
        case 16: return scores ? launch_pair_duo_t<RA, RB, 16, true>(b, ck, prm) : launch_pair_duo_t<RA, RB, 16, false>(b, ck, prm);
        default: return fail(CR_ERR_ARGUMENT, "unsupported tensor width");
    }
}

int launch_pair_duo(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, bool scores) {
    const int key = b->r_seed * 10 + b->r_b;
    switch (key) {
        case 11: return launch_pair_duo_r<1, 1>(b, ck, prm, scores);
        case 21: return launch_pair_duo_r<2, 1>(b, ck, prm, scores);
        case 22: return launch_pair_duo_r<2, 2>(b, ck, prm, scores);
        case 32: return launch_pair_duo_r<3, 2>(b, ck, prm, scores);
        case 33: return launch_pair_duo_r<3, 3>(b, ck, prm, scores);
        default: return fail(CR_ERR_STATE, "no k_pair_duo instance for this strip plan");
    }
}

// ---- k_pair_trio (cr_trio.h): gap 0, pairs of at most 320 rows, tensor widths padded to at most 16 ----------------------
// the padded width of the score waves' registers (cr_duo_instances.h), from the STORED width
int trio_width(int64_t d) { return d <= 4 ? 4 : d <= 8 ? 8 : d <= 10 ? 10 : d <= 12 ? 12 : d <= 16 ? 16 : 0; }

template <int R, int D, bool SC>
int launch_pair_trio_t(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    const int seed_entries = std::min(ck.n_max, ck.m_max), align_entries = ck.max_aln;
    size_t lds = sizeof(double) * std::max(cr::trio_lds_doubles<R>(ck.m_max),
                                           (size_t)cr::kExpDoubles + cr::trace_lds_doubles(R, SC ? seed_entries : align_entries));
    if (g_cfg.mid_lds_kb > 0) lds = std::max(lds, (size_t)g_cfg.mid_lds_kb * 1024);   // calibration: pairs per CU
    int rc = allow_lds(cr::k_pair_trio<R, D, SC>, lds);
    if (rc) return rc;
    // one wave of recurrences + two of scores; THREE of scores while the chip then still holds fewer than ~2 800 waves
    // (tools/c3_share.py: 508 pairs of 300 0.45 -> 0.41 ms, 678 pairs 0.56 -> 0.46; 1 016 pairs 0.55 either way)
    // (the matrix entries alone -- both stages are column sweeps -- also at 1 016 pairs: 0.407 -> 0.388 ms, profiles/r05/c3_stages.txt;
    // 1 024 pairs x 4 waves = every wave slot of the chip at four per SIMD)
    int waves = ck.count <= (SC ? 1024 : 700) ? 4 : 3;
    if (g_cfg.trio_waves) waves = std::min(std::max(g_cfg.trio_waves, 2), cr::kTrioMaxWaves);   // calibration
    int waves2 = waves;                                            // waves of the second stage (1 + its score waves)
    if (g_cfg.trio_waves2) waves2 = std::min(std::max(g_cfg.trio_waves2, 2), waves);            // calibration
    CR_LAUNCH((cr::k_pair_trio<R, D, SC>), dim3((unsigned)ck.count), dim3(waves * cr::kWave), lds,
              b->launch_stream ? b->launch_stream : b->ctx->stream, b->pairs.p + ck.first, b->tensors.p, (int)b->d, b->coords.p, prm.gamma_tensor,
              prm.gamma_coords, prm.gap_open, prm.gap_extend, seed_entries, align_entries, waves2 - 1, b->dirs.p, b->bits.p, b->xf.p + ck.first,
              b->seed_score.p + ck.first, b->aln.p, b->res.p + ck.first, host_out_for(b, ck));
    CR_HIP(hipGetLastError());
    return CR_OK;
}

template <int R>
int launch_pair_trio_r(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, bool scores) {
    switch (trio_width(b->d)) {
        case 4: return scores ? launch_pair_trio_t<R, 4, true>(b, ck, prm) : launch_pair_trio_t<R, 4, false>(b, ck, prm);
        case 8: return scores ? launch_pair_trio_t<R, 8, true>(b, ck, prm) : launch_pair_trio_t<R, 8, false>(b, ck, prm);
        case 10: return scores ? launch_pair_trio_t<R, 10, true>(b, ck, prm) : launch_pair_trio_t<R, 10, false>(b, ck, prm);
        case 12: return scores ? launch_pair_trio_t<R, 12, true>(b, ck, prm) : launch_pair_trio_t<R, 12, false>(b, ck, prm);
        case 16: return scores ? launch_pair_trio_t<R, 16, true>(b, ck, prm) : launch_pair_trio_t<R, 16, false>(b, ck, prm);
        default: return fail(CR_ERR_STATE, "no k_pair_trio instance for this tensor width");
    }
}

int launch_pair_trio(cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm, bool scores) {
    switch (b->r_seed) {
        case 2: return launch_pair_trio_r<2>(b, ck, prm, scores);
        case 3: return launch_pair_trio_r<3>(b, ck, prm, scores);
        case 4: return launch_pair_trio_r<4>(b, ck, prm, scores);
        case 5: return launch_pair_trio_r<5>(b, ck, prm, scores);
        default: return fail(CR_ERR_STATE, "no k_pair_trio instance for these rows per lane");
    }
}

int rows_per_lane(int n);                                    // (below, with the layout table)

int launch_seed_r(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    return R == 2 ? launch_seed_d<2>(b, ck, prm) : R == 3 ? launch_seed_d<3>(b, ck, prm)
         : R == 4 ? launch_seed_d<4>(b, ck, prm) : launch_seed_d<5>(b, ck, prm);
}

int launch_align_r(int R, cr_batch* b, const cr_batch::Chunk& ck, const cr_params& prm) {
    return R == 2 ? launch_align<2>(b, ck, prm) : R == 3 ? launch_align<3>(b, ck, prm)
         : R == 4 ? launch_align<4>(b, ck, prm) : launch_align<5>(b, ck, prm);
}

bool gamma_ok(double g) { return std::isfinite(g) && g >= 1e-290; }

// x * 0 is NaN exactly when x is NaN or infinite, and a NaN survives every later addition: eight independent sums, no
// branch in the loop (the compiler vectorises it; 4 MB of structures in ~0.1 ms instead of ~1 ms)
bool all_finite(const double* v, size_t count) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t x = 0;
    for (; x + 8 <= count; x += 8)
        for (int k = 0; k < 8; k++) acc[k] += v[x + k] * 0.0;
    for (; x < count; x++) acc[0] += v[x] * 0.0;
    double s = 0.0;
    for (int k = 0; k < 8; k++) s += acc[k];
    return !std::isnan(s);
}

constexpr int64_t kMaxTensorWidth = 192;

// widths the seed-fill kernel is instantiated for; narrower tensors are zero-padded in registers
int padded_width(int64_t d) {
    if (d <= 4) return 4;
    if (d <= 8) return 8;
    if (d <= 10) return 10;
    if (d <= 16) return 16;
    if (d <= 24) return 24;
    if (d <= 32) return 32;
    // wider tensors (multiple_alignment.py:312-331 takes any width): no register-resident provider -- the staged family only, whose
    // tensor scores come from the run-time-width staging kernel (k_stage_tensor_any); the LDS holds d planes of 79 columns
    if (d <= kMaxTensorWidth) return (int)d;
    return 0;
}

}  // namespace

// Shared body of the two fetch entry points: pack on the device (caller's order and layout), then copies into the
// caller's arrays -- straight DMA when those are page-locked (cr_host_alloc), through the context's page-locked ring
// otherwise (download()).
namespace {
template <class F>
int for_each_part(cr_batch* b, F&& f);      // (defined with the size classes, below)
}

template <class T>
int fetch_packed(cr_batch* b, cr_pair_result* results, T* aln, int64_t aln_stride) {
    CR_REQUIRE(b != nullptr, "null batch");
    if (!b->ran) return fail(CR_ERR_STATE, "cr_batch_fetch before cr_batch_run");
    if (b->scores_only) return fail(CR_ERR_STATE, "the last run was cr_batch_run_scores: only cr_batch_fetch_scores has results");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    hipStream_t st = b->ctx->stream;
    if (b->npairs == 0) {
        CR_HIP(hipStreamSynchronize(st));
        return CR_OK;
    }
    if (aln) CR_REQUIRE(aln_stride >= b->max_aln, "aln_stride smaller than the longest possible alignment");
    if (!results && !aln) {
        CR_HIP(hipStreamSynchronize(st));
        return CR_OK;
    }
    const size_t np = (size_t)b->npairs;
    const size_t aln_bytes = aln ? np * 2 * (size_t)aln_stride * sizeof(T) : 0;
    // (size classes: every class packs into the parent's arrays through its order map)
    const bool permute = b->reordered || !b->parts.empty();
    if (results && permute) CR_HIP(b->res_packed.ensure(np));
    if (aln) CR_HIP(b->aln_packed.ensure((aln_bytes + 7) / 8));
    if (aln || (results && permute)) {
        rc = for_each_part(b, [&](cr_batch* c) -> int {
            if (!c->npairs) return CR_OK;
            CR_LAUNCH(cr::k_pack_results<T>, dim3((unsigned)c->npairs), dim3(cr::kWave), 0, st, c->pairs.p, c->res.p,
                      c->reordered ? c->d_order.p : (const int32_t*)nullptr, c->aln.p, aln_stride,
                      (results && permute) ? b->res_packed.p : (cr::PairResult*)nullptr,
                      aln ? reinterpret_cast<T*>(b->aln_packed.p) : (T*)nullptr);
            CR_HIP(hipGetLastError());
            return CR_OK;
        });
        if (rc) return rc;
    }
    if (results) {
        rc = download(b->ctx, results, permute ? (const void*)b->res_packed.p : (const void*)b->res.p, sizeof(cr_pair_result) * np, false);
        if (rc) return rc;
    }
    if (aln) {
        rc = download(b->ctx, aln, b->aln_packed.p, aln_bytes, false);
        if (rc) return rc;
    }
    CR_HIP(hipStreamSynchronize(st));
    return CR_OK;
}

extern "C" {

const char* cr_last_error(void) { return g_err.c_str(); }
int cr_abi_version(void) { return CR_ABI_VERSION; }

int cr_device_count(int* count) {
    CR_REQUIRE(count != nullptr, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(CR_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return CR_OK;
}

int cr_device_trim(int device) {
    CR_REQUIRE(device >= 0 && device < 64, "bad device index");
    CR_HIP(hipSetDevice(device));
    CR_HIP(hipDeviceSynchronize());
    block_cache(device).trim();
    return CR_OK;
}

static int context_create(int device, void* stream, bool borrow, cr_context** out) {
    CR_REQUIRE(out != nullptr, "null out");
    *out = nullptr;
    int n = 0;
    CR_HIP(hipGetDeviceCount(&n));
    if (n <= 0) return fail(CR_ERR_HIP, "no HIP device visible: libcaretta_hip has no CPU fallback");
    CR_REQUIRE(device >= 0 && device < n, "device index out of range");
    CR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    CR_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(CR_ERR_HIP, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    cr_context* ctx = new (std::nothrow) cr_context();
    if (!ctx) return fail(CR_ERR_MEMORY, "out of host memory");
    ctx->device = device;
    if (borrow) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);     // may be 0: the legacy default stream
    } else {
        // a blocking stream: ordered with the legacy default stream that PyTorch and RCCL use (a non-blocking one
        // would let a collective queued after cr_batch_run read the scores while the kernels still write them)
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamDefault);
        if (e != hipSuccess) {
            delete ctx;
            return fail(CR_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return CR_OK;
}

int cr_context_create(int device, void* stream, cr_context** out) { return context_create(device, stream, stream != nullptr, out); }

int cr_context_create_on_stream(int device, void* stream, cr_context** out) { return context_create(device, stream, true, out); }

int cr_context_destroy(cr_context* ctx) {
    if (!ctx) return CR_OK;
    (void)hipSetDevice(ctx->device);
    // drain first: the staging ring and the landing area may still be the source or target of copies queued on the
    // context's streams, and nothing below may touch a stream after it has been destroyed
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& st : ctx->side) (void)hipStreamSynchronize(st);
    if (ctx->ring) (void)hipHostFree(ctx->ring);
    for (hipEvent_t e : ctx->ring_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->landing) (void)hipHostFree(ctx->landing);
    for (auto& l : ctx->ev)
        for (auto& e : l) (void)hipEventDestroy(e);
    for (auto& e : ctx->sync_ev) (void)hipEventDestroy(e);
    for (auto& st : ctx->side) (void)hipStreamDestroy(st);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CR_OK;
}

int cr_context_synchronize(cr_context* ctx) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_HIP(hipStreamSynchronize(ctx->stream));
    return CR_OK;
}

int cr_context_stream(cr_context* ctx, void** stream_out) {
    CR_REQUIRE(ctx && stream_out, "null argument");
    *stream_out = reinterpret_cast<void*>(ctx->stream);
    return CR_OK;
}

int cr_context_set_profiling(cr_context* ctx, int slots) {
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(slots >= 0 && slots <= 4096, "profiling slots must be in [0, 4096]");
    CR_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& l : ctx->ev)
        for (auto& e : l) (void)hipEventDestroy(e);
    ctx->ev.clear();
    ctx->runs_recorded = 0;
    ctx->ev.resize((size_t)slots);
    ctx->slots = slots;
    return CR_OK;
}

// ---------------------------------------------------------------------------------------------
// batch
// ---------------------------------------------------------------------------------------------
}  // extern "C"

// (check_finite = false: the caller has validated the arrays -- cr_multi_pairwise_scores does it once for all devices)
static int batch_create(cr_context* ctx, const double* coords, const double* tensors, const int64_t* offsets, int64_t num_structures,
                        int64_t d, bool check_finite, cr_batch** out);

extern "C" {

int cr_batch_create(cr_context* ctx, const double* coords, const double* tensors, const int64_t* offsets,
                    int64_t num_structures, int64_t d, cr_batch** out) {
    return batch_create(ctx, coords, tensors, offsets, num_structures, d, true, out);
}

}  // extern "C"

static int batch_create(cr_context* ctx, const double* coords, const double* tensors, const int64_t* offsets, int64_t num_structures,
                        int64_t d, bool check_finite, cr_batch** out) {
    CR_REQUIRE(out != nullptr, "null out");
    *out = nullptr;
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_REQUIRE(coords && tensors && offsets, "null input array");
    CR_REQUIRE(num_structures >= 1, "need at least one structure");
    CR_REQUIRE(d >= 1, "tensor width must be >= 1");
    CR_REQUIRE(padded_width(d) != 0, "tensor width > 192 is not supported by this build (the Python package runs such tensors through the per-function drop-ins)");
    CR_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
    for (int64_t s = 0; s < num_structures; s++) {
        CR_REQUIRE(offsets[s + 1] > offsets[s], "every structure needs at least one residue");
        CR_REQUIRE(offsets[s + 1] - offsets[s] <= cr::kMaxLength, "structure longer than 65534 residues");
    }
    if (check_finite) {
        CR_REQUIRE(all_finite(coords, (size_t)offsets[num_structures] * 3), "coordinates contain NaN or infinity");
        CR_REQUIRE(all_finite(tensors, (size_t)offsets[num_structures] * (size_t)d), "tensors contain NaN or infinity");
    }
    cr_batch* b = new (std::nothrow) cr_batch();
    if (!b) return fail(CR_ERR_MEMORY, "out of host memory");
    b->ctx = ctx;
    b->P = num_structures;
    b->d = d;
    b->d_pad = padded_width(d);
    b->total = offsets[num_structures];
    b->offsets.assign(offsets, offsets + num_structures + 1);
    hipError_t e = b->coords.ensure((size_t)b->total * 3);
    // (+ d_pad doubles of slack: the column sweep always reads a padded row of features, cr_sweep_cols.h sweep_cols)
    if (e == hipSuccess) e = b->tensors.ensure((size_t)b->total * d + (size_t)b->d_pad);
    if (e != hipSuccess) {
        delete b;
        return fail(e == hipErrorOutOfMemory ? CR_ERR_MEMORY : CR_ERR_HIP,
                    std::string("allocating structures: ") + hipGetErrorString(e));
    }
    if (e == hipSuccess) e = b->d_offsets.ensure((size_t)num_structures + 1);
    if (e != hipSuccess) {
        delete b;
        return fail(e == hipErrorOutOfMemory ? CR_ERR_MEMORY : CR_ERR_HIP, std::string("allocating structures: ") + hipGetErrorString(e));
    }
    int up = upload_async(ctx, b->coords.p, coords, sizeof(double) * b->total * 3);
    if (!up) up = upload_async(ctx, b->tensors.p, tensors, sizeof(double) * b->total * d);
    if (!up) up = upload_async(ctx, b->d_offsets.p, b->offsets.data(), sizeof(int64_t) * ((size_t)num_structures + 1));
    if (!up && hipStreamSynchronize(ctx->stream) != hipSuccess) up = fail(CR_ERR_HIP, "uploading structures");
    if (up) {
        delete b;
        return up;
    }
    *out = b;
    return CR_OK;
}

namespace {

#include "cr_layout.h"      // the layout table, choose_layout(), size classes, plan_list()

// the batch's layout flags from the chosen family (what the launch sequences of run_batch read)
void apply_layout(cr_batch* b, const Layout& l) {
    b->team = l.family == kFamTeam || l.family == kFamWide || l.family == kFamDuo || l.family == kFamStaged;
    b->wide_sync = (l.family == kFamWide || l.family == kFamDuo) ? l.wide_sync : 0;
    b->wide_na = b->wide_sync ? l.wide_na : 0;
    b->duo = l.family == kFamDuo;
    b->trio = l.family == kFamTrio;
    b->trio_few = b->trio && l.trio_few;
    b->staged = l.family == kFamStaged;
    b->r_seed = b->r_align = l.r_seed;
    b->r_b = b->wide_sync ? l.r_b : l.r_seed;
}

// One pair list, one layout.  `global` (size classes): the caller's index of every pair of this list -- the order map then
// leads from launch order straight to the caller's order.
int set_pairs_one(cr_batch* b, const int32_t* pairs, int64_t npairs, const int32_t* global, const LayoutMask mask) {
    int rc = CR_OK;
    b->npairs = npairs;
    b->ran = false;
    b->n_max = b->m_max = 0;
    for (int64_t p = 0; p < npairs; p++) {
        const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
        const int n = (int)(b->offsets[i + 1] - b->offsets[i]), m = (int)(b->offsets[j + 1] - b->offsets[j]);
        b->n_max = std::max(b->n_max, n);
        b->m_max = std::max(b->m_max, m);
    }
    {
        const Layout lay = choose_layout(b->n_max, b->m_max, b->d_pad, npairs, mask);
        if (!lay.ok)
            return fail(CR_ERR_ARGUMENT, "tensor width > 32 runs on staged scores only: at most 1 024 strips of 64 rows, 2 048 rows and 2 GiB of scores per "
                                         "pair list -- hand the list over in pieces");
        apply_layout(b, lay);
    }
    // (a duo or few-pair trio list is laid out again when a run comes with a Smith-Waterman gap: keep the list)
    b->duo_ij.clear();
    if (b->duo || b->trio_few) b->duo_ij.assign(pairs, pairs + 2 * npairs);
    // scratch budget per chunk (decision words); CARETTA_SCRATCH_MB overrides the 8 GiB default
    int64_t budget_words = (int64_t)8192 * 1024 * 1024 / 4;
    if (g_cfg.scratch_mb > 0) budget_words = (int64_t)g_cfg.scratch_mb * 1024 * 1024 / 4;
    // Structures of equal length (every BASELINE configuration): nothing to sort, every pair has the same footprint, and the
    // descriptors are built on the device from the (i, j) list (k_make_pairs_uniform) -- at 130 816 pairs the host side of
    // this call drops from 3.2 to under 1 ms.
    bool equal_lengths = npairs > 0 && !global;
    for (int64_t s = 1; s < b->P && equal_lengths; s++)
        equal_lengths = b->offsets[(size_t)s + 1] - b->offsets[(size_t)s] == b->offsets[1] - b->offsets[0];
    if (equal_lengths) {
        const int n = b->n_max, R = b->r_seed;
        const int64_t slots = b->wide_sync ? plan_of(b).slots(n) : cr::strips_of(n, R) * R;
        const int64_t dw = slots * cr::tblocks(n, 16) * cr::kWave, bw = slots * cr::tblocks(n, 8) * cr::kWave;
        const int64_t hand_per = cr::strips_of(n, R) > 1 ? 3 * (int64_t)n : 0;
        const int64_t per_chunk = std::max<int64_t>(1, std::min<int64_t>(npairs, budget_words / (dw + bw)));
        b->h_pairs.clear();
        b->order.clear();
        b->reordered = false;
        b->chunks.clear();
        for (int64_t first = 0; first < npairs; first += per_chunk) {
            cr_batch::Chunk ck{first, std::min(per_chunk, npairs - first), n, n, 2 * n};
            ck.r = R;
            ck.lane = 0;
            b->chunks.push_back(ck);
        }
        b->max_aln = 2 * (int64_t)n;
        b->aln_elems = npairs * 4 * (int64_t)n;
        const double nm = (double)n * n, npm = 2.0 * n;
        b->alg_bytes = (double)npairs * (8.0 * (3 + b->d) * npm + nm / 4 + nm / 2 + 16.0 * npm + 136.0);   // SURVEY.md 8(d) B_alg
        b->cells = (double)npairs * nm;
        hipError_t e = b->pairs.ensure((size_t)npairs);
        if (e == hipSuccess) e = b->dirs.ensure((size_t)(per_chunk * dw));
        if (e == hipSuccess) e = b->bits.ensure((size_t)(per_chunk * bw));
        if (e == hipSuccess) e = b->hand.ensure((size_t)(per_chunk * hand_per));
        if (e == hipSuccess) e = b->aln.ensure((size_t)b->aln_elems);
        if (e == hipSuccess) e = b->xf.ensure((size_t)npairs);
        if (e == hipSuccess) e = b->seed_score.ensure((size_t)npairs);
        if (e == hipSuccess) e = b->res.ensure((size_t)npairs);
        if (e == hipSuccess) e = b->d_ij.ensure((size_t)npairs * 2);
        if (e == hipSuccess && b->staged) e = b->staged_scores.ensure((size_t)(per_chunk * staged_shape(b->n_max, b->m_max).pair_doubles()));
        if (e != hipSuccess)
            return fail(e == hipErrorOutOfMemory ? CR_ERR_MEMORY : CR_ERR_HIP, std::string("allocating pair scratch: ") + hipGetErrorString(e));
        if ((rc = upload_async(b->ctx, b->d_ij.p, pairs, sizeof(int32_t) * 2 * (size_t)npairs))) return rc;
        CR_LAUNCH(cr::k_make_pairs_uniform, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, b->ctx->stream, b->d_ij.p, b->d_offsets.p, n, dw,
                  bw, hand_per, per_chunk, b->pairs.p, npairs);
        CR_HIP(hipGetLastError());
        return CR_OK;
    }
    // Launch order: the pairs with the most DP cells first, so that the last wave slots to drain hold short pairs
    // (waves are dispatched in block order; with ragged structures the caller's order would leave long pairs for
    // the tail).  Stable, so equal-length inputs keep the caller's order and nothing is permuted.
    b->h_pairs.resize((size_t)npairs);
    b->order.resize((size_t)npairs);
    for (int64_t p = 0; p < npairs; p++) b->order[(size_t)p] = (int32_t)p;
    std::vector<int64_t> cost_of((size_t)npairs);               // sort keys, computed once per pair
    std::vector<int8_t> group_of((size_t)npairs);
    {
        std::vector<int8_t> r_of_len;                             // rows_per_lane by row count
        for (int64_t p = 0; p < npairs; p++) {
            const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
            const int64_t n = b->offsets[i + 1] - b->offsets[i], m = b->offsets[j + 1] - b->offsets[j];
            cost_of[(size_t)p] = n * m;
            if ((int64_t)r_of_len.size() <= n) r_of_len.resize((size_t)n + 1, 0);
            if (!r_of_len[(size_t)n]) r_of_len[(size_t)n] = (int8_t)rows_per_lane((int)n);
            group_of[(size_t)p] = r_of_len[(size_t)n];
        }
    }
    auto cost = [&](int32_t p) { return cost_of[(size_t)p]; };
    // Pairs are also grouped by the rows per lane that suit their row count (one launch pair per group, long rows
    // first): a 90-residue structure in a 5-rows-per-lane kernel would use 18 of 64 lanes.
    const bool team_batch = b->team || b->trio;       // (trio: the single-wave layout, but ONE group of R rows per lane)
    const int team_r = b->r_seed;
    auto group = [&](int32_t p) { return team_batch ? team_r : (int)group_of[(size_t)p]; };
    const bool grouped = !g_cfg.keep_order;
    bool uniform = true;                             // equal keys everywhere: nothing to sort
    for (int64_t p = 1; p < npairs && uniform; p++)
        uniform = cost_of[(size_t)p] == cost_of[0] && group_of[(size_t)p] == group_of[0];
    if (grouped && !uniform)                         // (CARETTA_KEEP_ORDER, for measurements: one group, the caller's order)
        std::stable_sort(b->order.begin(), b->order.end(), [&](int32_t a, int32_t c) {
            const int ga = group(a), gc = group(c);
            return ga != gc ? ga > gc : cost(a) > cost(c);
        });
    b->reordered = global != nullptr;
    for (int64_t k = 0; k < npairs && !b->reordered; k++)
        if (b->order[(size_t)k] != k) b->reordered = true;
    int64_t dirs_off = 0, bt_off = 0, aln_off = 0, max_aln = 0, dirs_max = 0, bits_max = 0, hand_off = 0, hand_max = 0;
    // Each group owns a region of the decision scratch (its chunks run in order on the group's stream and reuse the
    // region; different groups run side by side); the budget is shared equally between the groups.
    int ngroups = 0;
    {
        int last = -1;
        for (int64_t p = 0; p < npairs; p++) {
            const int g = grouped ? group(b->order[(size_t)p]) : b->r_seed;
            if (g != last) ngroups++;
            last = g;
        }
    }
    budget_words /= std::max(ngroups, 1);
    int64_t dirs_base = 0, bt_base = 0, hand_base = 0;
    int lane = 0;
    double bytes = 0.0, cells = 0.0;
    b->chunks.clear();
    cr_batch::Chunk ck{0, 0, 0, 0, 0};
    for (int64_t p = 0; p < npairs; p++) {                     // p: position in launch order
        const int64_t orig = b->order[(size_t)p];
        int64_t i = pairs[2 * orig], j = pairs[2 * orig + 1];
        cr::PairDesc& pd = b->h_pairs[(size_t)p];
        pd.n = (int)(b->offsets[i + 1] - b->offsets[i]);
        pd.m = (int)(b->offsets[j + 1] - b->offsets[j]);
        pd.off_i = b->offsets[i];
        pd.off_j = b->offsets[j];
        const int R = grouped ? group((int32_t)orig) : b->r_seed;
        // row slots of the pair's strips (wide launches: the batch's strip plan) x time blocks x 64 lanes
        const int64_t slots = b->wide_sync ? plan_of(b).slots(pd.n) : cr::strips_of(pd.n, R) * R;
        const int64_t dw = slots * cr::tblocks(pd.m, 16) * cr::kWave;
        const int64_t bw = slots * cr::tblocks(pd.m, 8) * cr::kWave;
        if (ck.count > 0 && R != ck.r) {                       // next group: its own region and stream
            b->chunks.push_back(ck);
            ck = cr_batch::Chunk{p, 0, 0, 0, 0};
            dirs_base = dirs_off = dirs_max;
            bt_base = bt_off = bits_max;
            hand_base = hand_off = hand_max;
            lane = (lane + 1) % kGroupLanes;
        } else if (ck.count > 0 && (dirs_off - dirs_base) + (bt_off - bt_base) + dw + bw > budget_words) {
            b->chunks.push_back(ck);                           // same group, scratch budget used up: reuse the region
            ck = cr_batch::Chunk{p, 0, 0, 0, 0};
            dirs_off = dirs_base;
            bt_off = bt_base;
            hand_off = hand_base;
        }
        ck.r = R;
        ck.lane = lane;
        pd.dirs_off = dirs_off;
        pd.bt_off = bt_off;
        pd.aln_off = aln_off;
        pd.hand_off = hand_off;
        if (cr::strips_of(pd.n, R) > 1) hand_off += 3 * (int64_t)pd.m;
        hand_max = std::max(hand_max, hand_off);
        dirs_off += dw;
        bt_off += bw;
        dirs_max = std::max(dirs_max, dirs_off);
        bits_max = std::max(bits_max, bt_off);
        aln_off += 2 * (int64_t)(pd.n + pd.m);
        max_aln = std::max<int64_t>(max_aln, pd.n + pd.m);
        ck.count++;
        ck.n_max = std::max(ck.n_max, pd.n);
        ck.m_max = std::max(ck.m_max, pd.m);
        ck.max_aln = std::max(ck.max_aln, pd.n + pd.m);
        const double nm = (double)pd.n * pd.m, npm = (double)pd.n + pd.m;
        bytes += 8.0 * (3 + b->d) * npm + nm / 4 + nm / 2 + 16.0 * npm + 136.0;   // SURVEY.md 8(d) B_alg
        cells += nm;
    }
    if (ck.count > 0) b->chunks.push_back(ck);
    if (!b->chunks.empty()) b->r_seed = b->r_align = b->chunks[0].r;     // single-chunk callers (drop-ins) read these
    if (!b->wide_sync) b->r_b = b->r_seed;
    b->max_aln = max_aln;
    b->aln_elems = aln_off;
    b->alg_bytes = bytes;
    b->cells = cells;
    if (global)                                                 // launch order -> the CALLER's pair index
        for (int64_t k = 0; k < npairs; k++) b->order[(size_t)k] = global[b->order[(size_t)k]];
    hipError_t e = b->pairs.ensure((size_t)npairs);
    if (e == hipSuccess) e = b->dirs.ensure((size_t)dirs_max);
    if (e == hipSuccess) e = b->bits.ensure((size_t)bits_max);
    if (e == hipSuccess) e = b->hand.ensure((size_t)hand_max);
    if (e == hipSuccess) e = b->aln.ensure((size_t)aln_off);
    if (e == hipSuccess) e = b->xf.ensure((size_t)npairs);
    if (e == hipSuccess) e = b->seed_score.ensure((size_t)npairs);
    if (e == hipSuccess) e = b->res.ensure((size_t)npairs);
    if (e == hipSuccess && b->reordered) e = b->d_order.ensure((size_t)npairs);
    if (e == hipSuccess && b->staged) {
        int64_t most = 0;
        for (const cr_batch::Chunk& c : b->chunks) most = std::max(most, c.count);
        e = b->staged_scores.ensure((size_t)(most * staged_shape(b->n_max, b->m_max).pair_doubles()));
    }
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? CR_ERR_MEMORY : CR_ERR_HIP,
                    std::string("allocating pair scratch: ") + hipGetErrorString(e));
    // (no wait: both sources are members of the batch or already staged, and every consumer is queued on the same stream)
    if (b->reordered && (rc = upload_async(b->ctx, b->d_order.p, b->order.data(), sizeof(int32_t) * (size_t)npairs))) return rc;
    if (npairs && (rc = upload_async(b->ctx, b->pairs.p, b->h_pairs.data(), sizeof(cr::PairDesc) * (size_t)npairs))) return rc;
    return CR_OK;
}

// a size class of `parent`'s list as a batch of its own on the parent's structures
cr_batch* make_part(cr_batch* parent, int index) {
    cr_batch* c = new (std::nothrow) cr_batch();
    if (!c) return nullptr;
    c->ctx = parent->ctx;
    c->P = parent->P;
    c->d = parent->d;
    c->d_pad = parent->d_pad;
    c->total = parent->total;
    c->offsets = parent->offsets;
    c->coords.borrow(parent->coords);
    c->tensors.borrow(parent->tensors);
    c->d_offsets.borrow(parent->d_offsets);
    c->is_part = true;
    c->base_lane = index;
    return c;
}

void drop_parts(cr_batch* b) {
    for (cr_batch* c : b->parts) delete c;
    b->parts.clear();
}

// the batches that hold pair lists: the size classes of a split list, else the batch itself
template <class F>
int for_each_part(cr_batch* b, F&& f) {
    if (b->parts.empty()) return f(b);
    for (cr_batch* c : b->parts) {
        const int rc = f(c);
        if (rc) return rc;
    }
    return CR_OK;
}

}  // namespace

extern "C" {

int cr_config_reload(void) {
    g_cfg = crcfg::Calibration::from_env();
    return CR_OK;
}

// host only: no device, no context (tests/test_capi_cpu.py pins the layout table and the size classes with it)
int cr_plan_layout(const int64_t* offsets, int64_t num_structures, int64_t d, const int32_t* pairs, int64_t npairs, int32_t* class_of_pair,
                   int32_t* parts, int* nparts) {
    CR_REQUIRE(offsets != nullptr && parts != nullptr && nparts != nullptr, "null argument");
    CR_REQUIRE(num_structures >= 1 && npairs >= 0 && (npairs == 0 || pairs != nullptr), "bad pair list");
    CR_REQUIRE(d >= 1 && padded_width(d) != 0, "tensor width > 192 is not supported by this build (the Python package runs such tensors through the per-function drop-ins)");
    ListPlan plan;
    const int rc = plan_list(offsets, num_structures, padded_width(d), pairs, npairs, LayoutMask{}, true, plan);
    if (rc) return rc;
    if (!plan.whole.ok)
        return fail(CR_ERR_ARGUMENT, "tensor width > 32 runs on staged scores only: at most 1 024 strips of 64 rows, 2 048 rows and 2 GiB of scores per pair "
                                     "list -- hand the list over in pieces");
    int count = 0;
    auto put = [&](const Layout& l, int64_t n) {
        int32_t* row = parts + 5 * count++;
        row[0] = public_family(l);
        row[1] = l.r_seed;
        row[2] = l.wide_sync ? l.r_b : l.r_seed;
        row[3] = l.wide_sync ? l.wide_na : 0;
        row[4] = (int32_t)n;
    };
    int part_of_class[3] = {0, 0, 0};
    if (!plan.split) {
        put(plan.whole, npairs);
    } else {
        for (int c = 0; c < 3; c++)
            if (plan.in_class[c]) {
                part_of_class[c] = count;
                put(plan.of_class[c], plan.in_class[c]);
            }
    }
    if (class_of_pair)
        for (int64_t p = 0; p < npairs; p++) {
            const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
            class_of_pair[p] = plan.split ? part_of_class[size_class((int)(offsets[i + 1] - offsets[i]), (int)(offsets[j + 1] - offsets[j]))] : 0;
        }
    *nparts = count;
    return CR_OK;
}

int cr_batch_set_pairs(cr_batch* b, const int32_t* pairs, int64_t npairs) {
    CR_REQUIRE(b != nullptr, "null batch");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    CR_REQUIRE(npairs >= 0 && (npairs == 0 || pairs != nullptr), "bad pair list");
    CR_REQUIRE(npairs < (int64_t)std::numeric_limits<int32_t>::max(), "too many pairs for one batch");
    const LayoutMask mask{g_no_wide, g_no_trio, g_no_duo};
    // (parts of an earlier list may still be running: the blocks they give back wait for the device, DevBuf::release)
    drop_parts(b);
    ListPlan plan;
    if ((rc = plan_list(b->offsets.data(), b->P, b->d_pad, pairs, npairs, mask, !b->is_part, plan))) return rc;
    const int n_max = plan.n_max, m_max = plan.m_max;
    const int64_t* in_class = plan.in_class;
    const bool split = plan.split;
    if (!split) return set_pairs_one(b, pairs, npairs, nullptr, mask);
    // the parent keeps the totals; every class is a batch of its own
    b->npairs = npairs;
    b->ran = false;
    b->n_max = n_max;
    b->m_max = m_max;
    b->chunks.clear();
    b->h_pairs.clear();
    b->order.clear();
    b->reordered = false;
    b->team = b->duo = b->trio = b->trio_few = b->staged = false;
    b->wide_sync = b->wide_na = 0;
    b->max_aln = 0;
    b->aln_elems = 0;
    b->alg_bytes = b->cells = 0.0;
    std::vector<int32_t> list, global;
    for (int c = 0; c < 3; c++) {
        if (!in_class[c]) continue;
        list.clear();
        global.clear();
        for (int64_t p = 0; p < npairs; p++) {
            const int64_t i = pairs[2 * p], j = pairs[2 * p + 1];
            if (size_class((int)(b->offsets[i + 1] - b->offsets[i]), (int)(b->offsets[j + 1] - b->offsets[j])) != c) continue;
            list.push_back((int32_t)i);
            list.push_back((int32_t)j);
            global.push_back((int32_t)p);
        }
        cr_batch* part = make_part(b, (int)b->parts.size());
        if (!part) return fail(CR_ERR_MEMORY, "out of host memory");
        b->parts.push_back(part);
        part->part_global = global;
        if ((rc = set_pairs_one(part, list.data(), (int64_t)global.size(), part->part_global.data(), mask))) return rc;
        b->max_aln = std::max(b->max_aln, part->max_aln);
        b->aln_elems += part->aln_elems;
        b->alg_bytes += part->alg_bytes;
        b->cells += part->cells;
        // (the uploads of `list` are in flight at most until the stream's next wait; small lists are copied by the runtime
        // before hipMemcpyAsync returns, large ones went through the context's ring)
    }
    return CR_OK;
}

// k_pair_duo / k_pair_trio are gap-0 pipelines: a Smith-Waterman gap lays a duo or few-pair trio list out again without them.
// The re-layout may change the launch order (and so `reordered` / d_order): callers that hand the kernels an order map
// (cr_batch_run_stream_i32) call this BEFORE they build it.  The list it is made from stays with the batch -- set_pairs
// returns with its upload possibly still in flight.
static int relayout_for_gap(cr_batch* b, const cr_params& prm) {
    if (!((b->duo || (b->trio && b->trio_few)) && prm.sw_gap != 0.0)) return CR_OK;
    b->relaid_ij = std::move(b->duo_ij);
    b->duo_ij.clear();
    LayoutMask mask{g_no_wide, true, true};
    if (!b->is_part) {
        g_no_duo = g_no_trio = true;
        const int rc = cr_batch_set_pairs(b, b->relaid_ij.data(), (int64_t)(b->relaid_ij.size() / 2));
        g_no_duo = g_no_trio = false;
        return rc;
    }
    // a size class: the same pairs with the same caller indices, another family
    return set_pairs_one(b, b->relaid_ij.data(), b->npairs, b->part_global.data(), mask);
}

// the context's stream number k: 0 = its own, k > 0 = side stream k - 1 (created on first use)
static int lane_stream(cr_context* ctx, int k, hipStream_t* out) {
    if (k == 0) {
        *out = ctx->stream;
        return CR_OK;
    }
    while ((int)ctx->side.size() < k) {
        hipStream_t st;
        CR_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ctx->side.push_back(st);
    }
    *out = ctx->side[(size_t)k - 1];
    return CR_OK;
}

// the launch sequence of ONE pair list (a batch without size classes, or one class); `evl` / `evi`: profiling events
static int run_part(cr_batch* b, const cr_params& prm, bool scores_only, std::vector<hipEvent_t>* evl, size_t& evi) {
    cr_context* ctx = b->ctx;
    const bool prof = evl != nullptr;
    int rc = CR_OK;
    for (const cr_batch::Chunk& ck : b->chunks) {
        hipStream_t st = nullptr;
        if ((rc = lane_stream(ctx, (b->base_lane + ck.lane) % kGroupLanes, &st))) return rc;
        b->launch_stream = st;
        if (prof) (void)hipEventRecord((*evl)[evi++], st);
        if (b->staged) {
            // scores (every CU) -> SW sweep + seed superposition -> coordinate scores in that frame (every CU) -> SW score +
            // DTW sweep + metrics (or the SW score alone); the staged scores of a chunk are reused by the next in stream order
            const cr::StagedShape shape = staged_shape(b->n_max, b->m_max);
            rc = launch_stage_tensor(b, ck, prm, b->staged_scores.p, shape);
            if (!rc) rc = launch_seed_staged(b, ck, prm, b->staged_scores.p, shape);
            if (!rc && prof) (void)hipEventRecord((*evl)[evi++], st);
            if (!rc) rc = launch_stage_coords(b, ck, prm, b->staged_scores.p, shape);
            if (!rc) rc = launch_align_staged(b, ck, prm, b->staged_scores.p, shape, scores_only && prm.sw_gap == 0.0);
            if (!rc && prof) (void)hipEventRecord((*evl)[evi++], st);
            b->launch_stream = nullptr;
            if (rc) return rc;
            continue;
        }
        if (b->trio && prm.sw_gap == 0.0) {
            // three waves per pair, both stages in ONE launch (k_pair_trio); with a gap the same layout runs k_seed / k_align
            rc = launch_pair_trio(b, ck, prm, scores_only);
            if (!rc && prof) {
                (void)hipEventRecord((*evl)[evi++], st);
                (void)hipEventRecord((*evl)[evi++], st);
            }
            b->launch_stream = nullptr;
            if (rc) return rc;
            continue;
        }
        if (b->wide_sync) {
            // the wide layout: both stages of a pair in ONE launch (k_pair_wide / k_pair_duo) -- the stage split of the
            // events is (everything, 0)
            rc = b->duo ? launch_pair_duo(b, ck, prm, scores_only) : launch_pair_wide(b, ck, prm, scores_only && prm.sw_gap == 0.0);
            if (!rc && prof) {
                (void)hipEventRecord((*evl)[evi++], st);
                (void)hipEventRecord((*evl)[evi++], st);
            }
            b->launch_stream = nullptr;
            if (rc) return rc;
            continue;
        }
        rc = b->team ? launch_seed_team(ck.r, b, ck, prm) : launch_seed_r(ck.r, b, ck, prm);
        if (!rc && prof) (void)hipEventRecord((*evl)[evi++], st);
        if (!rc) {
            // scores only (gap 0): the column sweep without decisions -- one wave per pair, or one wave per strip for the
            // few-pair batches (team / wide layouts: strips_of(n_max, R) <= 16 waves)
            if (scores_only && prm.sw_gap == 0.0) rc = b->team ? launch_score_team_r(ck.r, b, ck, prm) : launch_score_r(ck.r, b, ck, prm);
            else rc = b->team ? launch_align_team(ck.r, b, ck, prm) : launch_align_r(ck.r, b, ck, prm);
        }
        if (!rc && prof) (void)hipEventRecord((*evl)[evi++], st);
        b->launch_stream = nullptr;
        if (rc) return rc;
    }
    return CR_OK;
}

// the streams a run of `b` uses beside the context's own (row-per-lane groups and size classes run side by side)
static int lanes_of(cr_batch* b) {
    int lanes = 1;
    (void)for_each_part(b, [&](cr_batch* c) {
        for (const cr_batch::Chunk& ck : c->chunks) lanes = std::max(lanes, (c->base_lane + ck.lane) % kGroupLanes + 1);
        return CR_OK;
    });
    return lanes;
}

static int fork_lanes(cr_context* ctx, int lanes_used) {
    if (lanes_used <= 1) return CR_OK;                            // fork: the side streams start after the work queued so far
    hipStream_t st;
    int rc = lane_stream(ctx, lanes_used - 1, &st);
    if (rc) return rc;
    while ((int)ctx->sync_ev.size() < lanes_used) {
        hipEvent_t e;
        CR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sync_ev.push_back(e);
    }
    CR_HIP(hipEventRecord(ctx->sync_ev[0], ctx->stream));
    for (int k = 1; k < lanes_used; k++) CR_HIP(hipStreamWaitEvent(ctx->side[(size_t)k - 1], ctx->sync_ev[0], 0));
    return CR_OK;
}

static int join_lanes(cr_context* ctx, int lanes_used) {
    for (int k = 1; k < lanes_used; k++) {
        CR_HIP(hipEventRecord(ctx->sync_ev[(size_t)k], ctx->side[(size_t)k - 1]));
        CR_HIP(hipStreamWaitEvent(ctx->stream, ctx->sync_ev[(size_t)k], 0));
    }
    return CR_OK;
}

// res[k].sw of every pair list of `b` into d_sw_out in the caller's pair order (on the context's stream)
static int scores_to_device(cr_batch* b, double* d_sw_out) {
    cr_context* ctx = b->ctx;
    return for_each_part(b, [&](cr_batch* c) -> int {
        if (!c->npairs) return CR_OK;
        if (c->reordered) {
            CR_LAUNCH(cr::k_scatter_sw, dim3((unsigned)((c->npairs + 255) / 256)), dim3(256), 0, ctx->stream, c->res.p, c->d_order.p, d_sw_out,
                      (int)c->npairs);
            CR_HIP(hipGetLastError());
        } else {
            // strided device-to-device copy of the first field of every PairResult
            CR_HIP(hipMemcpy2DAsync(d_sw_out, sizeof(double), c->res.p, sizeof(cr::PairResult), sizeof(double), (size_t)c->npairs,
                                    hipMemcpyDeviceToDevice, ctx->stream));
        }
        return CR_OK;
    });
}

// ... and the flags (cr_multi.h gathers both)
static int flags_to_device(cr_batch* b, uint32_t* d_flags_out) {
    cr_context* ctx = b->ctx;
    return for_each_part(b, [&](cr_batch* c) -> int {
        if (!c->npairs) return CR_OK;
        if (c->reordered) {
            CR_LAUNCH(cr::k_scatter_flags, dim3((unsigned)((c->npairs + 255) / 256)), dim3(256), 0, ctx->stream, c->res.p, c->d_order.p, d_flags_out,
                      (int)c->npairs);
            CR_HIP(hipGetLastError());
        } else {
            CR_HIP(hipMemcpy2DAsync(d_flags_out, sizeof(uint32_t), reinterpret_cast<const char*>(c->res.p) + offsetof(cr_pair_result, flags),
                                    sizeof(cr::PairResult), sizeof(uint32_t), (size_t)c->npairs, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return CR_OK;
    });
}

// `host` (cr_batch_run_stream_i32): page-locked arrays the alignment kernels write into; every list gets them with ITS order map
static int run_batch(cr_batch* b, const cr_params* params, double* d_sw_out, bool scores_only, const cr::HostOut* host = nullptr) {
    CR_REQUIRE(b != nullptr && params != nullptr, "null argument");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    if (b->npairs == 0) {
        b->ran = true;
        return CR_OK;
    }
    cr_context* ctx = b->ctx;
    const cr_params prm = *params;
    CR_REQUIRE(gamma_ok(prm.gamma_tensor) && gamma_ok(prm.gamma_coords),
               "gamma_tensor and gamma_coords must be finite and >= 1e-290 (below that every score is exactly 1.0)");
    CR_REQUIRE(std::isfinite(prm.gap_open) && std::isfinite(prm.gap_extend) && std::isfinite(prm.sw_gap),
               "gap penalties must be finite");
    // (before any order map is handed to a kernel)
    if ((rc = for_each_part(b, [&](cr_batch* c) { return relayout_for_gap(c, prm); }))) return rc;
    const bool prof = ctx->slots > 0;
    std::vector<hipEvent_t>* evl = nullptr;
    if (prof) {                                                   // three events per chunk: start, seed done, align done
        evl = &ctx->ev[(size_t)(ctx->runs_recorded % ctx->slots)];
        size_t need = 0;
        (void)for_each_part(b, [&](cr_batch* c) {
            need += 3 * c->chunks.size();
            return CR_OK;
        });
        while (evl->size() < need) {
            hipEvent_t e;
            CR_HIP(hipEventCreate(&e));
            evl->push_back(e);
        }
    }
    const int lanes_used = lanes_of(b);
    if ((rc = fork_lanes(ctx, lanes_used))) return rc;
    size_t evi = 0;
    rc = for_each_part(b, [&](cr_batch* c) -> int {
        if (host) {
            cr::HostOut h = *host;
            h.order = c->reordered ? c->d_order.p : nullptr;
            c->host_out = h;
        }
        const int r = run_part(c, prm, scores_only, evl, evi);
        c->host_out = cr::HostOut{};
        c->ran = r == CR_OK;
        c->scores_only = scores_only && prm.sw_gap == 0.0;
        return r;
    });
    if (rc) return rc;
    if ((rc = join_lanes(ctx, lanes_used))) return rc;
    if (d_sw_out && (rc = scores_to_device(b, d_sw_out))) return rc;
    if (prof) ctx->runs_recorded++;
    b->ran = true;
    b->scores_only = scores_only && prm.sw_gap == 0.0;
    return CR_OK;
}

int cr_batch_run(cr_batch* b, const cr_params* params, double* d_sw_out) { return run_batch(b, params, d_sw_out, false); }

int cr_batch_run_stream_i32(cr_batch* b, const cr_params* params, cr_pair_result* results, int32_t* aln, int64_t aln_stride,
                            double* d_sw_out) {
    CR_REQUIRE(b != nullptr && params != nullptr, "null argument");
    CR_REQUIRE(results != nullptr || aln != nullptr, "cr_batch_run_stream_i32 without a host array: use cr_batch_run");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    if (aln) CR_REQUIRE(aln_stride >= b->max_aln, "aln_stride smaller than the longest possible alignment");
    CR_REQUIRE((!results || is_page_locked(results)) && (!aln || is_page_locked(aln)),
               "cr_batch_run_stream_i32 needs page-locked host arrays (cr_host_alloc): the kernels write into them");
    cr::HostOut h{};
    if (results) CR_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h.res), results, 0));
    if (aln) CR_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h.aln), aln, 0));
    h.stride = aln_stride;
    // (the order map of every pair list is filled in by run_batch, behind a gap-driven re-layout)
    return run_batch(b, params, d_sw_out, false, &h);
}

int cr_batch_run_scores(cr_batch* b, const cr_params* params, double* d_sw_out) { return run_batch(b, params, d_sw_out, true); }

int cr_batch_run_tensor_scores(cr_batch* b, const cr_params* params, double* d_sw_out) {
    CR_REQUIRE(b != nullptr && params != nullptr, "null argument");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    CR_REQUIRE(gamma_ok(params->gamma_tensor), "gamma_tensor must be finite and >= 1e-290 (below that every score is exactly 1.0)");
    cr_context* ctx = b->ctx;
    rc = for_each_part(b, [&](cr_batch* c) -> int {
        for (const cr_batch::Chunk& ck : c->chunks) {              // (one stream: a single light launch per chunk)
            c->launch_stream = ctx->stream;
            const int r = launch_tensor_score_r(ck.r, c, ck, *params);
            c->launch_stream = nullptr;
            if (r) return r;
        }
        c->ran = true;
        c->scores_only = true;
        return CR_OK;
    });
    if (rc) return rc;
    if (b->npairs && d_sw_out && (rc = scores_to_device(b, d_sw_out))) return rc;
    b->ran = true;
    b->scores_only = true;
    return CR_OK;
}

int cr_batch_stage_ms(cr_batch* b, float ms[CR_NUM_STAGES], int* runs_averaged) {
    CR_REQUIRE(b != nullptr && ms != nullptr, "null argument");
    cr_context* ctx = b->ctx;
    if (ctx->slots == 0 || ctx->runs_recorded == 0) return fail(CR_ERR_STATE, "profiling not enabled or nothing recorded");
    int rc = set_device(ctx);
    if (rc) return rc;
    CR_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t n = std::min<int64_t>(ctx->runs_recorded, ctx->slots);
    double acc[CR_NUM_STAGES] = {};
    for (int64_t r = 0; r < n; r++) {
        const std::vector<hipEvent_t>& ev = ctx->ev[(size_t)r];
        size_t nchunks = 0;
        (void)for_each_part(b, [&](cr_batch* part) {
            nchunks += part->chunks.size();
            return CR_OK;
        });
        for (size_t c = 0; c < nchunks && 3 * c + 2 < ev.size(); c++) {
            for (int s = 0; s < CR_NUM_STAGES; s++) {
                float t = 0.f;
                CR_HIP(hipEventElapsedTime(&t, ev[3 * c + s], ev[3 * c + s + 1]));
                acc[s] += t;
            }
        }
    }
    for (int s = 0; s < CR_NUM_STAGES; s++) ms[s] = (float)(acc[s] / (double)n);
    if (runs_averaged) *runs_averaged = (int)n;
    return CR_OK;
}

int cr_batch_work(cr_batch* b, double* alg_bytes, double* cells) {
    CR_REQUIRE(b != nullptr, "null batch");
    if (alg_bytes) *alg_bytes = b->alg_bytes;
    if (cells) *cells = b->cells;
    return CR_OK;
}

int cr_batch_layout(cr_batch* b, int* family, int* rows_a, int* rows_b, int* strips_a) {
    CR_REQUIRE(b != nullptr, "null batch");
    if (!b->parts.empty()) {                         // a ragged list in size classes: cr_batch_part_layout has the families
        if (family) *family = CR_LAYOUT_CLASSES;
        if (rows_a) *rows_a = (int)b->parts.size();
        if (rows_b) *rows_b = 0;
        if (strips_a) *strips_a = 0;
        return CR_OK;
    }
    if (family) *family = b->staged ? CR_LAYOUT_STAGED : b->trio ? CR_LAYOUT_TRIO : b->duo ? CR_LAYOUT_DUO : b->wide_sync ? CR_LAYOUT_WIDE : b->team ? CR_LAYOUT_TEAM : CR_LAYOUT_SINGLE;
    if (rows_a) *rows_a = b->r_seed;
    if (rows_b) *rows_b = b->wide_sync ? b->r_b : b->r_seed;
    if (strips_a) *strips_a = b->wide_sync ? b->wide_na : 0;
    return CR_OK;
}

int cr_batch_part_layout(cr_batch* b, int part, int* family, int* rows_a, int* rows_b, int* strips_a, int64_t* npairs) {
    CR_REQUIRE(b != nullptr, "null batch");
    const int count = b->parts.empty() ? 1 : (int)b->parts.size();
    CR_REQUIRE(part >= 0 && part < count, "part index out of range");
    cr_batch* c = b->parts.empty() ? b : b->parts[(size_t)part];
    if (npairs) *npairs = c->npairs;
    return cr_batch_layout(c, family, rows_a, rows_b, strips_a);
}

int cr_batch_max_aln_len(cr_batch* b, int64_t* out) {
    CR_REQUIRE(b != nullptr && out != nullptr, "null argument");
    *out = b->max_aln;
    return CR_OK;
}

int cr_batch_fetch(cr_batch* b, cr_pair_result* results, int64_t* aln, int64_t aln_stride) {
    return fetch_packed<int64_t>(b, results, aln, aln_stride);
}

int cr_batch_fetch_i32(cr_batch* b, cr_pair_result* results, int32_t* aln, int64_t aln_stride) {
    return fetch_packed<int32_t>(b, results, aln, aln_stride);
}

int cr_host_alloc(size_t bytes, void** out) {
    CR_REQUIRE(out != nullptr, "null out");
    *out = nullptr;
    CR_HIP(hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault));
    return CR_OK;
}

int cr_host_free(void* p) {
    if (p) CR_HIP(hipHostFree(p));
    return CR_OK;
}

int cr_batch_fetch_scores(cr_batch* b, double* sw, uint32_t* flags) {
    CR_REQUIRE(b != nullptr, "null batch");
    if (!b->ran) return fail(CR_ERR_STATE, "cr_batch_fetch_scores before cr_batch_run");
    int rc = set_device(b->ctx);
    if (rc) return rc;
    if (b->npairs == 0) return CR_OK;
    // gather the fields on the device in the caller's pair order (8 + 4 bytes per pair instead of 160), then both land in
    // the context's page-locked area with one wait
    hipStream_t st = b->ctx->stream;
    const size_t np = (size_t)b->npairs;
    void* land = nullptr;
    rc = host_landing(b->ctx, np * (sizeof(double) + sizeof(uint32_t)), &land);
    if (rc) return rc;
    double* h_sw = static_cast<double*>(land);
    uint32_t* h_flags = reinterpret_cast<uint32_t*>(h_sw + np);
    DevBuf<uint32_t> stage;
    if (sw) {
        CR_HIP(b->sw_stage.ensure(np));
        if ((rc = scores_to_device(b, b->sw_stage.p))) return rc;
        CR_HIP(hipMemcpyAsync(h_sw, b->sw_stage.p, sizeof(double) * np, hipMemcpyDeviceToHost, st));
    }
    if (flags) {
        CR_HIP(stage.ensure(np));
        if ((rc = flags_to_device(b, stage.p))) return rc;
        CR_HIP(hipMemcpyAsync(h_flags, stage.p, sizeof(uint32_t) * np, hipMemcpyDeviceToHost, st));
    }
    CR_HIP(hipStreamSynchronize(st));
    if (sw) std::memcpy(sw, h_sw, sizeof(double) * np);
    if (flags) std::memcpy(flags, h_flags, sizeof(uint32_t) * np);
    return CR_OK;
}

#ifdef CR_STAMPS
// diagnostic build only: the per-wave stamps of k_pair_duo (cr_duo.h), 32 slots per block
int cr_debug_duo_stamps(unsigned long long* out, int blocks) {
    CR_HIP(hipDeviceSynchronize());
    CR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(cr::g_duo_stamps), sizeof(unsigned long long) * 32 * (size_t)blocks));
    return CR_OK;
}
// diagnostic build only: copy out (and clear) the phase stamps of the first `blocks` blocks
int cr_debug_stamps(unsigned long long* out, int blocks) {
    CR_HIP(hipDeviceSynchronize());
    CR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * (size_t)blocks));
    return CR_OK;
}
#endif

int cr_batch_destroy(cr_batch* b) {
    if (!b) return CR_OK;
    (void)hipSetDevice(b->ctx->device);
    // Everything that touches the batch's buffers runs on its context's streams: once those have drained the blocks can go
    // back to the cache without the device-wide wait DevBuf::release would otherwise make -- which would also wait for
    // the kernels of OTHER contexts (a second host thread preparing or running the next batch).
    bool drained = hipStreamSynchronize(b->ctx->stream) == hipSuccess;
    for (hipStream_t st : b->ctx->side) drained = hipStreamSynchronize(st) == hipSuccess && drained;
    if (drained) g_dirty = false;
    delete b;
    return CR_OK;
}

}  // extern "C"

#include "cr_dropins.h"
#include "cr_progressive.h"
#include "cr_explicit_batch.h"
#include "cr_nj_device.h"
#include "cr_multi.h"
