// Where do the waves of a launch land?  ./wave_placement.bin BLOCKS THREADS LDS_KB [SPIN_US]
// Every wave records HW_REG_HW_ID (wave, SIMD, CU, SH, SE) and HW_REG_XCC_ID while all of them are resident (each wave spins
// until SPIN_US have passed, so the launch's placement is the steady state of a kernel of that shape); the host prints
// how many SIMDs hold 0, 1, 2, ... waves and how the waves of one workgroup are spread over SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void k_where(unsigned* out, long long spin) {
    extern __shared__ double lds[];
    const long long t0 = __builtin_amdgcn_s_memtime();
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, offset 0, size 32
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
    lds[threadIdx.x] = 1.0;
    while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1016, threads = argc > 2 ? atoi(argv[2]) : 128, kb = argc > 3 ? atoi(argv[3]) : 17;
    const long long spin = (argc > 4 ? atoll(argv[4]) : 200) * 100;     // s_memtime ticks at 100 MHz
    const int wpb = threads / 64, waves = blocks * wpb;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 2 * waves);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_where), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_where, dim3(blocks), dim3(threads), kb * 1024, 0, d, spin);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * waves);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * waves, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_simd, per_cu, first_per_simd;
    int same_simd = 0;
    for (int w = 0; w < waves; w++) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_simd[(cukey << 2) | simd]++;
        per_cu[cukey]++;
        first_per_simd[(cukey << 2) | simd] += (w % wpb) == 0 ? 1 : 0;
        if (wpb > 1 && (w % wpb) == 1 && (((h[2 * (w - 1)] >> 4) & 3) == simd)) same_simd++;
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto& kv : per_simd) hist_simd[kv.second]++;
    for (auto& kv : per_cu) hist_cu[kv.second]++;
    printf("%d blocks x %d threads, %d KB LDS: %zu CUs, %zu SIMDs used\n  waves per SIMD:", blocks, threads, kb, per_cu.size(), per_simd.size());
    for (auto& kv : hist_simd) printf("  %d waves: %d SIMDs;", kv.first, kv.second);
    printf("\n  waves per CU:");
    for (auto& kv : hist_cu) printf("  %d waves: %d CUs;", kv.first, kv.second);
    if (wpb > 1) {                                       // kernels whose waves have different jobs (cr_trio.h): how many FIRST waves does a SIMD hold?
        std::map<int, int> hist_first;
        for (auto& kv : first_per_simd) hist_first[kv.second]++;
        printf("\n  first waves of a workgroup per SIMD:");
        for (auto& kv : hist_first) printf("  %d: %d SIMDs;", kv.first, kv.second);
    }
    if (wpb > 1) printf("\n  workgroups whose waves 0 and 1 share a SIMD: %d of %d", same_simd, blocks);
    printf("\n  first waves (hw_id): ");
    for (int w = 0; w < 8 && w < waves; w++) printf("%08x/%x ", h[2 * w], h[2 * w + 1] & 0xf);
    printf("\n");
    return 0;
}
