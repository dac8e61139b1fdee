// Dependent-issue latency of FP64-rate VALU instructions for ONE wave per SIMD on gfx950: 16 instructions per loop
// iteration dealt round robin to C independent chains (C = 1: every instruction waits for the one before).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_latency.hip -o /tmp/valu_latency && /tmp/valu_latency
// Prints shader cycles (s_memtime) per wave-instruction for C = 1, 2, 3, 4, 8 and W = 1, 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 8192
template <int C, int OP>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = a + threadIdx.x + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            double& r = x[u % C];
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r) : "v"(b));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r) : "v"(b));
            if (OP == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(r) : "v"(b));
            if (OP == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r) : "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int C, int OP>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256;                        // one workgroup per CU: 4 * W waves
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 1024);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<C, OP>), dim3(blocks), dim3(256 * waves_per_simd), 0, 0, out, cyc, 1.0, 1.0000001);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    printf("%-4s chains %d  waves/SIMD %d : %.2f cycles per wave-instruction (s_memtime units x?)\n", name, C, waves_per_simd, avg / (ITER * 16.0));
    hipFree(out);
    hipFree(cyc);
}
int main() {
    for (int w = 1; w <= 2; w++) {
        run<1, 0>("fma", w); run<2, 0>("fma", w); run<3, 0>("fma", w); run<4, 0>("fma", w); run<8, 0>("fma", w);
        run<1, 1>("add", w); run<2, 1>("add", w); run<4, 1>("add", w);
        run<1, 2>("max", w); run<2, 2>("max", w); run<4, 2>("max", w);
        run<1, 3>("mul", w); run<2, 3>("mul", w);
    }
    return 0;
}
