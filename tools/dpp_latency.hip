// Dependent-chain cost of the cross-lane shift the skewed sweeps make every step, ONE wave per SIMD on gfx950:
//   chain A: x = max(shift(x), b)        shift = two v_mov_b32_dpp (lo, hi) + one v_max_f64, each depending on the last
// for shift = wave_shr:1 (whole wave, what wave_shr1 uses), row_shr:1 (inside rows of 16 lanes), and no shift at all.
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_latency.hip -o tools/dpp_latency.bin && tools/dpp_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
template <int KIND>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, double a, double b) {
    double x = a + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            int lo = __double2loint(x), hi = __double2hiint(x);
            if (KIND == 1) {
                lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);      // wave_shr:1
                hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
            } else if (KIND == 2) {
                lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false);      // row_shr:1
                hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false);
            }
            double s = __hiloint2double(hi, lo);
            asm volatile("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(s), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256;
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 1024);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256 * waves_per_simd), 0, 0, out, cyc, 1.0, 0.5);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    printf("%-22s waves/SIMD %d : %.1f counter units per link (shift + max)\n", name, waves_per_simd, avg / (ITER * 16.0));
    hipFree(out);
    hipFree(cyc);
}
int main() {
    for (int w = 1; w <= 2; w++) {
        run<0>("max only", w);
        run<1>("wave_shr:1 + max", w);
        run<2>("row_shr:1 + max", w);
    }
    return 0;
}
