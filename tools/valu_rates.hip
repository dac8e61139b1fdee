// Issue-rate microbenchmark for the FP64-rate VALU instructions the sweep kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Each kernel runs ITER iterations of 16 independent instructions of one kind per wave; grid = 256 CUs x 4 SIMDs x W waves.
// Reports cycles per wave-instruction per SIMD (at the clock measured by s_memtime / wall).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 4096

#define KERNEL(NAME, ASM16)                                                             \
    __global__ __launch_bounds__(64) void NAME(double* out, double a, double b) {      \
        double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        double y = b;                                                                  \
        int i0 = threadIdx.x, i1 = i0 + 1;                                             \
        for (int it = 0; it < ITER; it++) { asm volatile(ASM16 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(i0), "+v"(i1) : "v"(y) : "vcc", "s20", "s21"); } \
        out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1; \
    }

#define KERNEL32(NAME, ASM16)                                                           \
    __global__ __launch_bounds__(64) void NAME(double* out, double a, double b) {      \
        int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        double y = b + threadIdx.x;                                                    \
        int i0 = threadIdx.x * 3, i1 = i0 + 1;                                         \
        for (int it = 0; it < ITER; it++) { asm volatile(ASM16 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(i0), "+v"(i1) : "v"(y) : "vcc", "s20", "s21"); } \
        out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1; \
    }

#define REP8(OP) OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7) OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)
#define S(x) #x
#define ADD(r) "v_add_f64 " S(r) ", " S(r) ", %10\n"
#define MUL(r) "v_mul_f64 " S(r) ", " S(r) ", %10\n"
#define FMA(r) "v_fma_f64 " S(r) ", " S(r) ", %10, %10\n"
#define MAX(r) "v_max_f64 " S(r) ", " S(r) ", %10\n"
#define CMP(r) "v_cmp_gt_f64 vcc, " S(r) ", %10\n"
#define LDEXP(r) "v_ldexp_f64 " S(r) ", " S(r) ", %8\n"
#define CND(r) "v_cndmask_b32 %8, %8, %9, vcc\n"
#define CNDI(r) "v_cndmask_b32 " S(r) ", " S(r) ", %9, vcc\n"
#define CND64(r) "v_cndmask_b32_e64 " S(r) ", " S(r) ", %9, s[20:21]\n"
#define CMP64(r) "v_cmp_gt_f64_e64 s[20:21], %10, %10\n"
#define CMPSEL(r) "v_cmp_gt_f64_e64 s[20:21], %10, %10\nv_cndmask_b32_e64 " S(r) ", " S(r) ", %9, s[20:21]\n"
#define LSHLOR(r) "v_lshl_or_b32 " S(r) ", %9, 3, " S(r) "\n"
#define AND32(r) "v_and_b32 " S(r) ", 15, " S(r) "\n"
#define ASHR(r) "v_ashrrev_i32 " S(r) ", 4, " S(r) "\n"
#define LSHLADD(r) "v_lshl_add_u32 %8, %8, 1, %9\n"
#define DPP(r) "v_mov_b32_dpp %8, %9 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define MOV64(r) "v_mov_b64 " S(r) ", %10\n"
#define ADDU(r) "v_add_u32 %8, %8, %9\n"

KERNEL(k_add, REP8(ADD))
KERNEL(k_mul, REP8(MUL))
KERNEL(k_fma, REP8(FMA))
KERNEL(k_max, REP8(MAX))
KERNEL(k_cmp, REP8(CMP))
KERNEL(k_ldexp, REP8(LDEXP))
KERNEL(k_cnd, REP8(CND))
KERNEL32(k_cndi, REP8(CNDI))
KERNEL32(k_cnd64, REP8(CND64))
KERNEL32(k_cmp64, REP8(CMP64))
KERNEL32(k_cmpsel, REP8(CMPSEL))
KERNEL32(k_lshlor, REP8(LSHLOR))
KERNEL32(k_and, REP8(AND32))
KERNEL32(k_ashr, REP8(ASHR))
KERNEL(k_lshladd, REP8(LSHLADD))
KERNEL(k_dpp, REP8(DPP))
KERNEL(k_mov64, REP8(MOV64))
KERNEL(k_addu, REP8(ADDU))

typedef void (*kern_t)(double*, double, double);

int main() {
    double* out;
    hipMalloc(&out, sizeof(double) * 64 * 256 * 4 * 8);
    struct { const char* name; kern_t k; } ks[] = {{"v_add_f64", k_add}, {"v_mul_f64", k_mul}, {"v_fma_f64", k_fma}, {"v_max_f64", k_max},
        {"v_cmp_gt_f64", k_cmp}, {"v_ldexp_f64", k_ldexp}, {"v_cndmask dep", k_cnd}, {"v_cndmask vcc", k_cndi}, {"v_cndmask_e64", k_cnd64}, {"v_cmp_e64 sgpr", k_cmp64}, {"cmp+cndmask x2", k_cmpsel}, {"v_lshl_or_b32", k_lshlor}, {"v_and_b32", k_and}, {"v_ashrrev_i32", k_ashr}, {"v_lshl_add_u32", k_lshladd},
        {"v_mov_b32_dpp", k_dpp}, {"v_mov_b64", k_mov64}, {"v_add_u32", k_addu}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-16s %8s %8s %8s   (ns per wave-instruction per SIMD; x clock GHz = cycles)\n", "instruction", "1w/SIMD", "2w/SIMD", "4w/SIMD");
    for (auto& k : ks) {
        printf("%-16s", k.name);
        for (int w : {1, 2, 4}) {
            int blocks = 256 * 4 * w;
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, out, 1.0, 1.0000001);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, out, 1.0, 1.0000001);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double per = (ms / 5) * 1e6 / ((double)ITER * 16 * w);   // ns per wave-instruction per SIMD
            printf(" %8.3f", per);
        }
        printf("\n");
    }
    return 0;
}
