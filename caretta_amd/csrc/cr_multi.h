// All-vs-all score matrix over several GPUs of one node from ONE host process (included by cr_api.hip).
//
// The reference's caller of make_pairwise_matrix is a single process (multiple_alignment.py:497-501 inside
// align_from_structure_files), and the pair loop has no cross-pair dependency (:158-170).  So: one cr_context (device +
// stream) per GPU, the structures replicated on every GPU (16 MB at 512 x 300), the pair set dealt to the GPUs by
// cr_partition_pairs (the same deal caretta_amd/distributed.py makes for one-process-per-GPU runs), every GPU driven by
// its own host thread, no data-path collective -- and ONE grouped RCCL all-gather over xGMI (ncclGroupStart, one
// ncclAllGather per device on that device's stream, ncclGroupEnd; communicators from ncclCommInitAll) that leaves the
// whole score vector on every device; device 0's copy goes to the host.  ~1 MB at 512 structures: latency-bound.
// Results do not depend on the number of devices, bit for bit (no atomics, no cross-pair reductions).
//
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it, and a process that already has a
// librccl (PyTorch's) keeps using that one.
#pragma once

#include <dlfcn.h>

namespace {

struct RcclApi {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
constexpr int kNcclUint32 = 3, kNcclFloat64 = 8;     // ncclDataType_t (rccl.h)

RcclApi* rccl_api() {
    static RcclApi* api = [] {
        RcclApi* a = new RcclApi();
        // a copy that is already in the process first (PyTorch loads its own librccl.so, SONAME librccl.so.1), then the
        // caller's choice, then the loader's search path, then ROCm's default location
        std::vector<std::pair<std::string, int>> tries;
        tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_NOLOAD});
        if (const char* env = std::getenv("CARETTA_RCCL_LIB")) tries.push_back({env, RTLD_NOW | RTLD_GLOBAL});
        tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_GLOBAL});
        tries.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL});
        for (auto& t : tries) {
            a->handle = dlopen(t.first.c_str(), t.second);
            if (a->handle) break;
        }
        if (!a->handle) {
            a->error = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
            return a;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(a->handle, name);
            if (!p && a->error.empty()) a->error = std::string("librccl: missing symbol ") + name;
            return p;
        };
        a->CommInitAll = reinterpret_cast<decltype(a->CommInitAll)>(sym("ncclCommInitAll"));
        a->CommDestroy = reinterpret_cast<decltype(a->CommDestroy)>(sym("ncclCommDestroy"));
        a->GroupStart = reinterpret_cast<decltype(a->GroupStart)>(sym("ncclGroupStart"));
        a->GroupEnd = reinterpret_cast<decltype(a->GroupEnd)>(sym("ncclGroupEnd"));
        a->AllGather = reinterpret_cast<decltype(a->AllGather)>(sym("ncclAllGather"));
        a->GetErrorString = reinterpret_cast<decltype(a->GetErrorString)>(sym("ncclGetErrorString"));
        return a;
    }();
    return api;
}

#define CR_RCCL(api, expr)                                                                                   \
    do {                                                                                                     \
        const int _r = (expr);                                                                               \
        if (_r != 0) return fail(CR_ERR_HIP, std::string(#expr) + ": " + (api)->GetErrorString(_r));         \
    } while (0)

// The deal of the pair set: pair ids (row-major i < j) sorted by DP cell count, descending, stable on the id, and dealt
// round robin; a rank's ids ascending.  Equal lengths: id % world == rank.  (caretta_amd/distributed.py:partition_pairs)
void partition_pairs_host(const int64_t* lengths, int64_t P, int world, int rank, std::vector<int64_t>& idx) {
    const int64_t np = P * (P - 1) / 2;
    bool uniform = true;
    for (int64_t s = 1; s < P && uniform; s++) uniform = lengths[s] == lengths[0];
    idx.clear();
    if (uniform) {
        for (int64_t p = rank; p < np; p += world) idx.push_back(p);
        return;
    }
    std::vector<int64_t> cost((size_t)np), order((size_t)np);
    int64_t p = 0;
    for (int64_t i = 0; i < P; i++)
        for (int64_t j = i + 1; j < P; j++, p++) {
            cost[(size_t)p] = lengths[i] * lengths[j];
            order[(size_t)p] = p;
        }
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return cost[(size_t)a] > cost[(size_t)b]; });
    for (int64_t k = rank; k < np; k += world) idx.push_back(order[(size_t)k]);
    std::sort(idx.begin(), idx.end());
}

}  // namespace

namespace cr {
template <class Dummy = void>
__global__ void k_fill_f64_t(double* __restrict__ p, double v, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = v;
}
constexpr auto k_fill_f64 = k_fill_f64_t<>;
}  // namespace cr

struct cr_multi {
    std::vector<int> devices;
    std::vector<cr_context*> ctx;
    std::vector<void*> comm;                 // ncclComm_t per device (empty until the first collective)
    struct PerDevice {
        DevBuf<double> local, gathered;
        DevBuf<uint32_t> local_flags, gathered_flags;
    };
    std::vector<PerDevice*> dev;
    // The device list names a device twice (only accepted with CARETTA_MULTI_ALLOW_DUPLICATES=1; RCCL refuses such a
    // communicator): the shares are gathered with device copies instead.  This is how a one-GPU box runs the deal, the
    // host threads, the share layout and the scatter with MORE THAN ONE share (tests); it is not a product path.
    bool loopback = false;
    float last_ms[3] = {0.f, 0.f, 0.f};      // wall ms of the last call: compute (all devices), all-gather, download + scatter
};

extern "C" {

int cr_partition_pairs(const int64_t* lengths, int64_t P, int world, int rank, int64_t* idx_out, int64_t* count_out) {
    CR_REQUIRE(lengths && count_out, "null argument");
    CR_REQUIRE(P >= 1 && world >= 1 && rank >= 0 && rank < world, "bad partition arguments");
    std::vector<int64_t> idx;
    partition_pairs_host(lengths, P, world, rank, idx);
    *count_out = (int64_t)idx.size();
    if (idx_out) std::copy(idx.begin(), idx.end(), idx_out);
    return CR_OK;
}

int cr_multi_destroy(cr_multi* m) {
    if (!m) return CR_OK;
    for (size_t g = 0; g < m->ctx.size(); g++) {
        if (m->ctx[g]) {
            (void)hipSetDevice(m->devices[g]);
            (void)hipStreamSynchronize(m->ctx[g]->stream);
        }
    }
    if (!m->comm.empty()) {
        RcclApi* api = rccl_api();
        for (void* c : m->comm)
            if (c && api->CommDestroy) (void)api->CommDestroy(c);
    }
    for (size_t g = 0; g < m->dev.size(); g++) {
        (void)hipSetDevice(m->devices[g]);
        delete m->dev[g];
    }
    for (cr_context* c : m->ctx) (void)cr_context_destroy(c);
    delete m;
    return CR_OK;
}

int cr_multi_create(const int* devices, int ndev, cr_multi** out) {
    CR_REQUIRE(out != nullptr, "null out");
    *out = nullptr;
    int visible = 0;
    CR_HIP(hipGetDeviceCount(&visible));
    if (visible <= 0) return fail(CR_ERR_HIP, "no HIP device visible: libcaretta_hip has no CPU fallback");
    cr_multi* m = new (std::nothrow) cr_multi();
    if (!m) return fail(CR_ERR_MEMORY, "out of host memory");
    if (devices == nullptr || ndev <= 0) {
        for (int g = 0; g < visible; g++) m->devices.push_back(g);
    } else {
        const bool allow_twice = std::getenv("CARETTA_MULTI_ALLOW_DUPLICATES") != nullptr;
        for (int g = 0; g < ndev; g++) {
            const bool twice = std::count(devices, devices + g, devices[g]) != 0;
            if (devices[g] < 0 || devices[g] >= visible || (twice && !allow_twice)) {
                delete m;
                return fail(CR_ERR_ARGUMENT, "device list: indices must be distinct and visible");
            }
            m->loopback = m->loopback || twice;
            m->devices.push_back(devices[g]);
        }
    }
    for (int dv : m->devices) {
        cr_context* c = nullptr;
        const int rc = cr_context_create(dv, nullptr, &c);
        if (rc) {
            cr_multi_destroy(m);
            return rc;
        }
        m->ctx.push_back(c);
        m->dev.push_back(new cr_multi::PerDevice());
    }
    *out = m;
    return CR_OK;
}

int cr_multi_device_count(cr_multi* m, int* ndev) {
    CR_REQUIRE(m && ndev, "null argument");
    *ndev = (int)m->devices.size();
    return CR_OK;
}

int cr_multi_last_ms(cr_multi* m, float ms[3]) {
    CR_REQUIRE(m && ms, "null argument");
    for (int k = 0; k < 3; k++) ms[k] = m->last_ms[k];
    return CR_OK;
}

// scores f64[P(P-1)/2], flags u32[same] (may be NULL) in row-major i < j order.
int cr_multi_pairwise_scores(cr_multi* m, const double* coords, const double* tensors, const int64_t* offsets,
                             int64_t P, int64_t d, const cr_params* params, double* scores, uint32_t* flags) {
    CR_REQUIRE(m && coords && tensors && offsets && params && scores, "null argument");
    CR_REQUIRE(P >= 2, "need at least two structures");
    const int G = (int)m->devices.size();
    const int64_t np = P * (P - 1) / 2;
    const int64_t shard = (np + G - 1) / G;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) {
        return (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    std::vector<int64_t> lengths((size_t)P);
    for (int64_t s = 0; s < P; s++) lengths[(size_t)s] = offsets[s + 1] - offsets[s];
    // pair id -> (i, j), row-major (multiple_alignment.py:162-163)
    std::vector<int32_t> all_ij((size_t)np * 2);
    {
        int64_t p = 0;
        for (int64_t i = 0; i < P; i++)
            for (int64_t j = i + 1; j < P; j++, p++) {
                all_ij[(size_t)(2 * p)] = (int32_t)i;
                all_ij[(size_t)(2 * p + 1)] = (int32_t)j;
            }
    }
    std::vector<std::vector<int64_t>> owned((size_t)G);
    for (int g = 0; g < G; g++) partition_pairs_host(lengths.data(), P, G, g, owned[(size_t)g]);

    // ---- every device: its share of the pair set, driven by its own host thread -------------------------------
    std::vector<int> rcs((size_t)G, CR_OK);
    std::vector<std::string> errs((size_t)G);
    std::vector<cr_batch*> batches((size_t)G, nullptr);
    auto work = [&](int g) {
        auto run = [&]() -> int {
            cr_context* ctx = m->ctx[(size_t)g];
            cr_multi::PerDevice& pd = *m->dev[(size_t)g];
            int rc = set_device(ctx);
            if (rc) return rc;
            const std::vector<int64_t>& mine = owned[(size_t)g];
            std::vector<int32_t> ij(mine.size() * 2);
            for (size_t k = 0; k < mine.size(); k++) {
                ij[2 * k] = all_ij[(size_t)(2 * mine[k])];
                ij[2 * k + 1] = all_ij[(size_t)(2 * mine[k] + 1)];
            }
            CR_HIP(pd.local.ensure((size_t)shard));
            CR_HIP(pd.local_flags.ensure((size_t)shard));
            CR_HIP(pd.gathered.ensure((size_t)shard * G));
            CR_HIP(pd.gathered_flags.ensure((size_t)shard * G));
            // slots past this device's share stay NaN / 0 (a NaN that reaches the host is a collective fault)
            CR_LAUNCH(cr::k_fill_f64, dim3((unsigned)((shard + 255) / 256)), dim3(256), 0, ctx->stream, pd.local.p,
                      std::numeric_limits<double>::quiet_NaN(), shard);
            CR_HIP(hipGetLastError());
            CR_HIP(hipMemsetAsync(pd.local_flags.p, 0, sizeof(uint32_t) * (size_t)shard, ctx->stream));
            rc = cr_batch_create(ctx, coords, tensors, offsets, P, d, &batches[(size_t)g]);
            if (rc) return rc;
            cr_batch* b = batches[(size_t)g];
            rc = cr_batch_set_pairs(b, ij.data(), (int64_t)mine.size());
            if (rc) return rc;
            rc = cr_batch_run_scores(b, params, pd.local.p);
            if (rc) return rc;
            if (!mine.empty()) {                      // the flags, in the caller's pair order like the scores
                if (b->reordered) {
                    CR_LAUNCH(cr::k_scatter_flags, dim3((unsigned)((mine.size() + 255) / 256)), dim3(256), 0, ctx->stream, b->res.p,
                              b->d_order.p, pd.local_flags.p, (int)mine.size());
                    CR_HIP(hipGetLastError());
                } else {
                    CR_HIP(hipMemcpy2DAsync(pd.local_flags.p, sizeof(uint32_t),
                                            reinterpret_cast<const char*>(b->res.p) + offsetof(cr_pair_result, flags), sizeof(cr::PairResult),
                                            sizeof(uint32_t), mine.size(), hipMemcpyDeviceToDevice, ctx->stream));
                }
            }
            return CR_OK;
        };
        rcs[(size_t)g] = run();
        if (rcs[(size_t)g]) errs[(size_t)g] = g_err;
    };
    if (G == 1) {
        work(0);
    } else {
        std::vector<std::thread> threads;
        for (int g = 0; g < G; g++) threads.emplace_back(work, g);
        for (auto& t : threads) t.join();
    }
    auto cleanup = [&]() {
        for (int g = 0; g < G; g++)
            if (batches[(size_t)g]) {
                (void)cr_batch_destroy(batches[(size_t)g]);        // (waits for the device's stream)
                batches[(size_t)g] = nullptr;
            }
    };
    for (int g = 0; g < G; g++)
        if (rcs[(size_t)g]) {
            cleanup();
            return fail(rcs[(size_t)g], "device " + std::to_string(m->devices[(size_t)g]) + ": " + errs[(size_t)g]);
        }
    const bool timing = std::getenv("CARETTA_MULTI_TIMING") != nullptr;
    if (timing) {                                      // (the phases are only separable with a wait in between)
        for (int g = 0; g < G; g++) {
            (void)hipSetDevice(m->devices[(size_t)g]);
            (void)hipStreamSynchronize(m->ctx[(size_t)g]->stream);
        }
        m->last_ms[0] = ms_since(t0);
    }
    const auto t1 = std::chrono::steady_clock::now();

    // ---- one grouped all-gather: every device ends up with every share ------------------------------------------
    if (m->loopback) {
        // (test mode, see cr_multi::loopback) the same data movement with copies: wait for every share, then every
        // "device" collects all of them
        auto copy_all = [&]() -> int {
            for (int g = 0; g < G; g++) {
                CR_HIP(hipSetDevice(m->devices[(size_t)g]));
                CR_HIP(hipStreamSynchronize(m->ctx[(size_t)g]->stream));
            }
            for (int g = 0; g < G; g++)
                for (int r = 0; r < G; r++) {
                    CR_HIP(hipMemcpyAsync(m->dev[(size_t)g]->gathered.p + (size_t)r * shard, m->dev[(size_t)r]->local.p, sizeof(double) * (size_t)shard,
                                          hipMemcpyDeviceToDevice, m->ctx[(size_t)g]->stream));
                    CR_HIP(hipMemcpyAsync(m->dev[(size_t)g]->gathered_flags.p + (size_t)r * shard, m->dev[(size_t)r]->local_flags.p,
                                          sizeof(uint32_t) * (size_t)shard, hipMemcpyDeviceToDevice, m->ctx[(size_t)g]->stream));
                }
            return CR_OK;
        };
        const int rc_copy = copy_all();
        if (rc_copy) {
            cleanup();
            return rc_copy;
        }
    }
    RcclApi* api = m->loopback ? nullptr : rccl_api();
    if (api && !api->error.empty()) {
        cleanup();
        return fail(CR_ERR_HIP, api->error);
    }
    if (api && m->comm.empty()) {
        m->comm.assign((size_t)G, nullptr);
        const int r = api->CommInitAll(m->comm.data(), G, m->devices.data());
        if (r != 0) {
            m->comm.clear();
            cleanup();
            return fail(CR_ERR_HIP, std::string("ncclCommInitAll: ") + api->GetErrorString(r));
        }
    }
    auto gather = [&]() -> int {
        CR_RCCL(api, api->GroupStart());
        for (int g = 0; g < G; g++) {
            cr_multi::PerDevice& pd = *m->dev[(size_t)g];
            CR_RCCL(api, api->AllGather(pd.local.p, pd.gathered.p, (size_t)shard, kNcclFloat64, m->comm[(size_t)g], m->ctx[(size_t)g]->stream));
            CR_RCCL(api, api->AllGather(pd.local_flags.p, pd.gathered_flags.p, (size_t)shard, kNcclUint32, m->comm[(size_t)g],
                                        m->ctx[(size_t)g]->stream));
        }
        CR_RCCL(api, api->GroupEnd());
        return CR_OK;
    };
    int rc = api ? gather() : CR_OK;
    if (rc) {
        cleanup();
        return rc;
    }
    if (timing) {
        for (int g = 0; g < G; g++) {
            (void)hipSetDevice(m->devices[(size_t)g]);
            (void)hipStreamSynchronize(m->ctx[(size_t)g]->stream);
        }
        m->last_ms[1] = ms_since(t1);
    }
    const auto t2 = std::chrono::steady_clock::now();

    // ---- device 0's copy -> host, share order -> pair order ------------------------------------------------------
    auto collect = [&]() -> int {
        cr_context* ctx = m->ctx[0];
        int rc2 = set_device(ctx);
        if (rc2) return rc2;
        const size_t cnt = (size_t)shard * G;
        void* land = nullptr;
        rc2 = host_landing(ctx, cnt * (sizeof(double) + sizeof(uint32_t)), &land);
        if (rc2) return rc2;
        double* h_sw = static_cast<double*>(land);
        uint32_t* h_fl = reinterpret_cast<uint32_t*>(h_sw + cnt);
        CR_HIP(hipMemcpyAsync(h_sw, m->dev[0]->gathered.p, sizeof(double) * cnt, hipMemcpyDeviceToHost, ctx->stream));
        CR_HIP(hipMemcpyAsync(h_fl, m->dev[0]->gathered_flags.p, sizeof(uint32_t) * cnt, hipMemcpyDeviceToHost, ctx->stream));
        CR_HIP(hipStreamSynchronize(ctx->stream));
        for (int g = 0; g < G; g++) {
            const std::vector<int64_t>& mine = owned[(size_t)g];
            for (size_t k = 0; k < mine.size(); k++) {
                scores[mine[k]] = h_sw[(size_t)g * shard + k];
                if (flags) flags[mine[k]] = h_fl[(size_t)g * shard + k];
            }
        }
        return CR_OK;
    };
    rc = collect();
    // the other devices' streams: their part of the collective must be over before their buffers are reused
    for (int g = 1; g < G; g++) {
        (void)hipSetDevice(m->devices[(size_t)g]);
        (void)hipStreamSynchronize(m->ctx[(size_t)g]->stream);
    }
    cleanup();
    if (rc) return rc;
    for (int64_t p = 0; p < np; p++)
        if (std::isnan(scores[p])) return fail(CR_ERR_HIP, "all-gather left pair " + std::to_string(p) + " without a score");
    m->last_ms[2] = ms_since(t2);
    if (!timing) m->last_ms[0] = ms_since(t0);
    return CR_OK;
}

}  // extern "C"
