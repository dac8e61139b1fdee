#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
// What the scalar store path sustains on gfx950: every wave stores `iters` x 16 bytes of SGPR data (s_store_dwordx4) to its
// own contiguous region -- the decision masks of a DP step would leave the wave this way.
__global__ void k_sstore(uint32_t* out, int iters) {
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    uint32_t* vbase = out + wave * (uint64_t)iters * 4;
    const uint64_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)vbase);
    const uint64_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)vbase >> 32));
    uint32_t* base = reinterpret_cast<uint32_t*>((bhi << 32) | blo);      // (wave-uniform: the address of a scalar store sits in SGPRs)
    uint64_t m0 = __ballot(threadIdx.x & 1), m1 = __ballot(threadIdx.x & 2);
    for (int t = 0; t < iters; t++) {
        m0 += t;
        m1 ^= m0;
        { typedef uint32_t U4 __attribute__((ext_vector_type(4))); U4 v4 = {(uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32)}; asm volatile("s_store_dwordx4 %0, %1, 0x0" ::"s"(v4), "s"(base + (uint64_t)t * 4) : "memory"); }
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
// the same bytes through the vector path: one 256-byte dword store per wave and 16 iterations' worth of data
__global__ void k_vstore(uint32_t* out, int iters) {
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    uint32_t* base = out + wave * (uint64_t)iters * 4;
    uint32_t v = threadIdx.x;
    for (int t = 0; t < iters; t += 16) {
        v = v * 3 + t;
        base[(uint64_t)t * 4 + (threadIdx.x & 63)] = v;
    }
}
int main() {
    const int blocks = 2048, threads = 256, iters = 4096;
    const size_t waves = (size_t)blocks * threads / 64, bytes = waves * iters * 16;
    uint32_t* d;
    hipMalloc(&d, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int which = 0; which < 2; which++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_sstore, dim3(blocks), dim3(threads), 0, 0, d, iters);
            else hipLaunchKernelGGL(k_vstore, dim3(blocks), dim3(threads), 0, 0, d, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %zu waves x %d x 16 B = %.1f MB in %.3f ms -> %.1f GB/s (%s)\n", which == 0 ? "s_store_dwordx4" : "global_store_dword (256 B per wave instruction)",
                   waves, iters, bytes / 1e6, ms, bytes / ms / 1e6, hipGetErrorString(hipGetLastError()));
        }
    std::vector<uint32_t> h(64);
    hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
    printf("first words: %08x %08x %08x %08x\n", h[0], h[1], h[2], h[3]);
    return 0;
}
