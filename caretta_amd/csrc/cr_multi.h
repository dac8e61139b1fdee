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
        if (!g_cfg.rccl_lib.empty()) tries.push_back({g_cfg.rccl_lib, RTLD_NOW | RTLD_GLOBAL});
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
    std::vector<void*> comm;                 // ncclComm_t per device, created by cr_multi_create (empty in loopback mode)
    // Everything a device needs for its share is KEPT between calls: the batch (device copies of the structures, pair
    // descriptors, scratch), the share's pair list, the gather buffers, the events, the host thread.  A call with the same
    // layout (P, d, offsets) uploads coordinates and tensors into the kept batch -- 16 MB at 512 x 300, cheaper than any
    // fingerprint of them -- and runs; another layout rebuilds the batch and the deal.
    struct PerDevice {
        DevBuf<double> local, gathered;
        DevBuf<uint32_t> local_flags, gathered_flags;
        cr_batch* batch = nullptr;
        std::vector<int64_t> owned;          // pair ids (row-major i < j) of this device's share, ascending
        std::vector<int32_t> ij;             // ... as (i, j): the source of the batch's pair-list upload, alive with the batch
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // on the device's stream: call start, share computed, gathered, (device 0) copied to the host
        int rc = CR_OK;
        std::string err;
    };
    std::vector<PerDevice*> dev;
    int64_t P = 0, d = 0;                    // layout the kept batches were made for
    std::vector<int64_t> offsets;
    // The device list names a device twice (only accepted with CARETTA_MULTI_ALLOW_DUPLICATES=1; RCCL refuses such a
    // communicator): the shares are gathered with device copies instead.  This is how a one-GPU box runs the deal, the
    // host threads, the share layout and the scatter with MORE THAN ONE share (tests); it is not a product path.
    bool loopback = false;
    float last_ms[3] = {0.f, 0.f, 0.f};      // last call: slowest device's share (events), all-gather (events), download (events) + scatter on the host (wall)
    // one host thread per device, parked between calls
    std::vector<std::thread> threads;
    std::vector<int> numa_node;         // per device: the NUMA node its host thread was pinned to (CARETTA_MULTI_NUMA=1), else -1
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<void(int)> job;
    uint64_t generation = 0;
    int pending = 0;
    int started = 0;                    // host threads that have pinned themselves and parked (cr_multi_create waits for all)
    bool stop = false;
};

namespace {

// run fn(g) for every device on that device's host thread (the caller's thread when there is one device), wait for all
void multi_run(cr_multi* m, const std::function<void(int)>& fn) {
    const int G = (int)m->devices.size();
    if (m->threads.empty()) {
        for (int g = 0; g < G; g++) fn(g);
        return;
    }
    std::unique_lock<std::mutex> lk(m->mu);
    m->job = fn;
    m->pending = G;
    m->generation++;
    m->cv_go.notify_all();
    m->cv_done.wait(lk, [&] { return m->pending == 0; });
    m->job = nullptr;
}

// CARETTA_MULTI_NUMA=1 (default off; for the first run on an 8-GPU node to A/B): the host thread that drives device `device`
// runs on the CPUs of the NUMA node the device hangs off (sysfs: numa_node of its PCI function, cpulist of that node), so
// that its staging copies and launch calls do not cross the socket interconnect.  Best effort: anything unreadable leaves
// the thread where it is.  Returns the node, or -1.
int pin_thread_to_device_numa(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (char* c = bus; *c; c++) *c = (char)std::tolower((unsigned char)*c);
    int node = -1;
    {
        const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f) return -1;
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node < 0) return -1;
    char list[4096] = {0};
    {
        const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f) return -1;
        const bool ok = std::fgets(list, (int)sizeof(list), f) != nullptr;
        std::fclose(f);
        if (!ok) return -1;
    }
    cpu_set_t allowed, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return -1;
    int any = 0;
    for (const char* c = list; *c;) {                      // "0-63,128-191"
        char* end = nullptr;
        const long lo = std::strtol(c, &end, 10);
        if (end == c) break;
        long hi = lo;
        c = end;
        if (*c == '-') {
            hi = std::strtol(c + 1, &end, 10);
            c = end;
        }
        for (long k = lo; k <= hi && k < CPU_SETSIZE; k++)
            if (CPU_ISSET((int)k, &allowed)) {             // (never outside what the cgroup / the caller's mask allows)
                CPU_SET((int)k, &want);
                any++;
            }
        while (*c == ',' || *c == ' ' || *c == '\n') c++;
    }
    if (!any) return -1;
    return pthread_setaffinity_np(pthread_self(), sizeof(want), &want) == 0 ? node : -1;
}

void multi_worker(cr_multi* m, int g) {
    {
        // pinned BEFORE cr_multi_create returns, and recorded under the mutex: cr_multi_numa_nodes reads from the caller's thread
        const int node = g_cfg.multi_numa ? pin_thread_to_device_numa(m->devices[(size_t)g]) : -1;
        std::lock_guard<std::mutex> lk(m->mu);
        m->numa_node[(size_t)g] = node;
        if (++m->started == (int)m->devices.size()) m->cv_done.notify_all();
    }
    uint64_t seen = 0;
    for (;;) {
        std::function<void(int)> fn;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_go.wait(lk, [&] { return m->stop || m->generation != seen; });
            if (m->stop) return;
            seen = m->generation;
            fn = m->job;
        }
        fn(g);
        {
            std::lock_guard<std::mutex> lk(m->mu);
            if (--m->pending == 0) m->cv_done.notify_all();
        }
    }
}

// the calling thread's current device, put back on every way out (the HIP runtime is shared with the caller: torch's
// current device must not change behind its back)
struct DeviceRestore {
    int dev = -1;
    DeviceRestore() {
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    }
    ~DeviceRestore() {
        if (dev >= 0) (void)hipSetDevice(dev);
    }
};

}  // namespace

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
    DeviceRestore restore;
    if (!m->threads.empty()) {
        {
            std::lock_guard<std::mutex> lk(m->mu);
            m->stop = true;
        }
        m->cv_go.notify_all();
        for (auto& t : m->threads) t.join();
    }
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
        if (m->dev[g]->batch) (void)cr_batch_destroy(m->dev[g]->batch);
        for (hipEvent_t e : m->dev[g]->ev)
            if (e) (void)hipEventDestroy(e);
        delete m->dev[g];
    }
    for (cr_context* c : m->ctx) (void)cr_context_destroy(c);
    delete m;
    return CR_OK;
}

int cr_multi_create(const int* devices, int ndev, cr_multi** out) {
    CR_REQUIRE(out != nullptr, "null out");
    *out = nullptr;
    DeviceRestore restore;
    int visible = 0;
    CR_HIP(hipGetDeviceCount(&visible));
    if (visible <= 0) return fail(CR_ERR_HIP, "no HIP device visible: libcaretta_hip has no CPU fallback");
    cr_multi* m = new (std::nothrow) cr_multi();
    if (!m) return fail(CR_ERR_MEMORY, "out of host memory");
    if (devices == nullptr || ndev <= 0) {
        for (int g = 0; g < visible; g++) m->devices.push_back(g);
    } else {
        const bool allow_twice = g_cfg.multi_allow_duplicates;
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
    const int G = (int)m->devices.size();
    for (int dv : m->devices) {
        cr_context* c = nullptr;
        int rc = cr_context_create(dv, nullptr, &c);
        if (rc) {
            cr_multi_destroy(m);
            return rc;
        }
        m->ctx.push_back(c);
        m->dev.push_back(new cr_multi::PerDevice());
        for (hipEvent_t& e : m->dev.back()->ev)
            if (hipEventCreate(&e) != hipSuccess) {
                cr_multi_destroy(m);
                return fail(CR_ERR_HIP, "creating events");
            }
    }
    // The communicators are part of the object: a box without a usable RCCL fails HERE, before any share is computed (the
    // caller then knows to stay on one device), and the first timed call does not pay for ncclCommInitAll.
    if (!m->loopback) {
        RcclApi* api = rccl_api();
        if (!api->error.empty()) {
            const std::string why = api->error;
            cr_multi_destroy(m);
            return fail(CR_ERR_HIP, why);
        }
        m->comm.assign((size_t)G, nullptr);
        const int r = api->CommInitAll(m->comm.data(), G, m->devices.data());
        if (r != 0) {
            m->comm.clear();
            cr_multi_destroy(m);
            return fail(CR_ERR_HIP, std::string("ncclCommInitAll: ") + api->GetErrorString(r));
        }
    }
    m->numa_node.assign((size_t)G, -1);
    if (G > 1) {
        for (int g = 0; g < G; g++) m->threads.emplace_back(multi_worker, m, g);
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv_done.wait(lk, [&] { return m->started == G; });
    }
    *out = m;
    return CR_OK;
}

int cr_multi_device_count(cr_multi* m, int* ndev) {
    CR_REQUIRE(m && ndev, "null argument");
    *ndev = (int)m->devices.size();
    return CR_OK;
}

int cr_multi_numa_nodes(cr_multi* m, int* nodes) {
    CR_REQUIRE(m && nodes, "null argument");
    std::lock_guard<std::mutex> lk(m->mu);
    for (size_t g = 0; g < m->devices.size(); g++) nodes[g] = g < m->numa_node.size() ? m->numa_node[g] : -1;
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
    CR_REQUIRE(d >= 1 && offsets[0] == 0, "bad layout");
    DeviceRestore restore;
    const int G = (int)m->devices.size();
    const int64_t np = P * (P - 1) / 2;
    const int64_t shard = (np + G - 1) / G;
    const int64_t total = offsets[P];
    // validated once, here, not once per device
    CR_REQUIRE(all_finite(coords, (size_t)total * 3), "coordinates contain NaN or infinity");
    CR_REQUIRE(all_finite(tensors, (size_t)total * (size_t)d), "tensors contain NaN or infinity");
    bool same_layout = m->P == P && m->d == d && m->offsets.size() == (size_t)P + 1 && std::equal(offsets, offsets + P + 1, m->offsets.begin());
    for (int g = 0; g < G && same_layout; g++) same_layout = m->dev[(size_t)g]->batch != nullptr;
    if (!same_layout) {
        std::vector<int64_t> lengths((size_t)P);
        for (int64_t s = 0; s < P; s++) lengths[(size_t)s] = offsets[s + 1] - offsets[s];
        // pair id -> (i, j), row-major (multiple_alignment.py:162-163)
        std::vector<int32_t> all_ij((size_t)np * 2);
        int64_t p = 0;
        for (int64_t i = 0; i < P; i++)
            for (int64_t j = i + 1; j < P; j++, p++) {
                all_ij[(size_t)(2 * p)] = (int32_t)i;
                all_ij[(size_t)(2 * p + 1)] = (int32_t)j;
            }
        for (int g = 0; g < G; g++) {
            cr_multi::PerDevice& pd = *m->dev[(size_t)g];
            partition_pairs_host(lengths.data(), P, G, g, pd.owned);
            pd.ij.resize(pd.owned.size() * 2);
            for (size_t k = 0; k < pd.owned.size(); k++) {
                pd.ij[2 * k] = all_ij[(size_t)(2 * pd.owned[k])];
                pd.ij[2 * k + 1] = all_ij[(size_t)(2 * pd.owned[k] + 1)];
            }
        }
        m->P = 0;                                       // (valid again once every device has its batch)
    }

    // ---- every device: its share of the pair set, driven by its own host thread -------------------------------
    multi_run(m, [&](int g) {
        cr_multi::PerDevice& pd = *m->dev[(size_t)g];
        auto run = [&]() -> int {
            cr_context* ctx = m->ctx[(size_t)g];
            int rc = set_device(ctx);
            if (rc) return rc;
            CR_HIP(hipEventRecord(pd.ev[0], ctx->stream));
            CR_HIP(pd.local.ensure((size_t)shard));
            CR_HIP(pd.local_flags.ensure((size_t)shard));
            CR_HIP(pd.gathered.ensure((size_t)shard * G));
            CR_HIP(pd.gathered_flags.ensure((size_t)shard * G));
            // slots past this device's share stay NaN / 0 (a NaN that reaches the host is a collective fault)
            CR_LAUNCH(cr::k_fill_f64, dim3((unsigned)((shard + 255) / 256)), dim3(256), 0, ctx->stream, pd.local.p,
                      std::numeric_limits<double>::quiet_NaN(), shard);
            CR_HIP(hipGetLastError());
            CR_HIP(hipMemsetAsync(pd.local_flags.p, 0, sizeof(uint32_t) * (size_t)shard, ctx->stream));
            if (!same_layout) {
                if (pd.batch) {
                    (void)cr_batch_destroy(pd.batch);
                    pd.batch = nullptr;
                }
                rc = batch_create(ctx, coords, tensors, offsets, P, d, /*check_finite=*/false, &pd.batch);
                if (rc) return rc;
                rc = cr_batch_set_pairs(pd.batch, pd.ij.data(), (int64_t)pd.owned.size());
                if (rc) return rc;
            } else {
                // same layout: the structures into the kept batch, in stream order before the kernels
                rc = upload_async(ctx, pd.batch->coords.p, coords, sizeof(double) * (size_t)total * 3);
                if (!rc) rc = upload_async(ctx, pd.batch->tensors.p, tensors, sizeof(double) * (size_t)total * (size_t)d);
                if (rc) return rc;
            }
            cr_batch* b = pd.batch;
            rc = cr_batch_run_scores(b, params, pd.local.p);
            if (rc) return rc;
            if (!pd.owned.empty()) {                  // the flags, in the caller's pair order like the scores
                rc = flags_to_device(b, pd.local_flags.p);
                if (rc) return rc;
            }
            CR_HIP(hipEventRecord(pd.ev[1], ctx->stream));
            return CR_OK;
        };
        pd.rc = run();
        pd.err = pd.rc ? g_err : std::string();
    });
    auto drain = [&]() {                                // a failed call: nothing in flight, batches rebuilt next time
        for (int g = 0; g < G; g++) {
            (void)hipSetDevice(m->devices[(size_t)g]);
            (void)hipStreamSynchronize(m->ctx[(size_t)g]->stream);
        }
        m->P = 0;
    };
    for (int g = 0; g < G; g++)
        if (m->dev[(size_t)g]->rc) {
            drain();
            return fail(m->dev[(size_t)g]->rc, "device " + std::to_string(m->devices[(size_t)g]) + ": " + m->dev[(size_t)g]->err);
        }
    m->P = P;
    m->d = d;
    m->offsets.assign(offsets, offsets + P + 1);

    // ---- one grouped all-gather: every device ends up with every share ------------------------------------------
    if (m->loopback) {
        // (test mode, see cr_multi::loopback) the same data movement with copies: wait for every share, then every
        // "device" collects all of them
        auto copy_all = [&]() -> int {
            for (int g = 0; g < G; g++) {
                CR_HIP(hipSetDevice(m->devices[(size_t)g]));
                CR_HIP(hipStreamSynchronize(m->ctx[(size_t)g]->stream));
            }
            for (int g = 0; g < G; g++) {
                CR_HIP(hipSetDevice(m->devices[(size_t)g]));
                for (int r = 0; r < G; r++) {
                    CR_HIP(hipMemcpyAsync(m->dev[(size_t)g]->gathered.p + (size_t)r * shard, m->dev[(size_t)r]->local.p, sizeof(double) * (size_t)shard,
                                          hipMemcpyDeviceToDevice, m->ctx[(size_t)g]->stream));
                    CR_HIP(hipMemcpyAsync(m->dev[(size_t)g]->gathered_flags.p + (size_t)r * shard, m->dev[(size_t)r]->local_flags.p,
                                          sizeof(uint32_t) * (size_t)shard, hipMemcpyDeviceToDevice, m->ctx[(size_t)g]->stream));
                }
                CR_HIP(hipEventRecord(m->dev[(size_t)g]->ev[2], m->ctx[(size_t)g]->stream));
            }
            return CR_OK;
        };
        const int rc_copy = copy_all();
        if (rc_copy) {
            drain();
            return rc_copy;
        }
    } else {
        RcclApi* api = rccl_api();
        // (a group that has been opened is always closed, whatever an ncclAllGather inside it returns: the librccl is the
        // one the caller's torch uses)
        int first_bad = 0;
        const char* what = "";
        int r = api->GroupStart();
        if (r != 0) {
            drain();
            return fail(CR_ERR_HIP, std::string("ncclGroupStart: ") + api->GetErrorString(r));
        }
        for (int g = 0; g < G && !first_bad; g++) {
            cr_multi::PerDevice& pd = *m->dev[(size_t)g];
            r = api->AllGather(pd.local.p, pd.gathered.p, (size_t)shard, kNcclFloat64, m->comm[(size_t)g], m->ctx[(size_t)g]->stream);
            if (r == 0) r = api->AllGather(pd.local_flags.p, pd.gathered_flags.p, (size_t)shard, kNcclUint32, m->comm[(size_t)g], m->ctx[(size_t)g]->stream);
            if (r != 0) {
                first_bad = r;
                what = "ncclAllGather: ";
            }
        }
        r = api->GroupEnd();
        if (r != 0 && !first_bad) {
            first_bad = r;
            what = "ncclGroupEnd: ";
        }
        if (first_bad) {
            drain();
            return fail(CR_ERR_HIP, std::string(what) + api->GetErrorString(first_bad));
        }
        for (int g = 0; g < G; g++) {
            (void)hipSetDevice(m->devices[(size_t)g]);
            (void)hipEventRecord(m->dev[(size_t)g]->ev[2], m->ctx[(size_t)g]->stream);
        }
    }
    auto t_scatter = std::chrono::steady_clock::now();

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
        CR_HIP(hipEventRecord(m->dev[0]->ev[3], ctx->stream));
        CR_HIP(hipStreamSynchronize(ctx->stream));
        t_scatter = std::chrono::steady_clock::now();
        for (int g = 0; g < G; g++) {
            const std::vector<int64_t>& mine = m->dev[(size_t)g]->owned;
            for (size_t k = 0; k < mine.size(); k++) {
                scores[mine[k]] = h_sw[(size_t)g * shard + k];
                if (flags) flags[mine[k]] = h_fl[(size_t)g * shard + k];
            }
        }
        return CR_OK;
    };
    int rc = collect();
    // the other devices' streams: their part of the collective is over before their buffers are reused by the next call
    for (int g = 1; g < G; g++) {
        (void)hipSetDevice(m->devices[(size_t)g]);
        (void)hipStreamSynchronize(m->ctx[(size_t)g]->stream);
    }
    if (rc) {
        drain();
        return rc;
    }
    for (int64_t p = 0; p < np; p++)
        if (std::isnan(scores[p])) return fail(CR_ERR_HIP, "all-gather left pair " + std::to_string(p) + " without a score");
    // the phases of this call, from the events on every device's stream (no host wait separates them)
    float compute = 0.f, gather = 0.f;
    for (int g = 0; g < G; g++) {
        (void)hipSetDevice(m->devices[(size_t)g]);
        float a = 0.f, b = 0.f;
        if (hipEventElapsedTime(&a, m->dev[(size_t)g]->ev[0], m->dev[(size_t)g]->ev[1]) == hipSuccess) compute = std::max(compute, a);
        if (hipEventElapsedTime(&b, m->dev[(size_t)g]->ev[1], m->dev[(size_t)g]->ev[2]) == hipSuccess) gather = std::max(gather, b);
    }
    float copy = 0.f;
    (void)hipSetDevice(m->devices[0]);
    (void)hipEventElapsedTime(&copy, m->dev[0]->ev[2], m->dev[0]->ev[3]);
    m->last_ms[0] = compute;
    m->last_ms[1] = gather;
    m->last_ms[2] = copy + (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scatter).count();
    return CR_OK;
}

}  // extern "C"
