// What a step of the staged sweeps (cr::sweep_staged, one row per lane: the fills of a level of
// the progressive alignment) costs, and what it would cost without its decision packing / its hand-off writes: ONE
// workgroup of eight waves on an otherwise idle chip, n rows (1 .. 8 strips) x m columns, scores from an L2-resident
// buffer, shader clock (s_memtime) around the sweep.
//   bash tools/step_probe.sh           (builds the four variants into tools/step_probe_*.bin; run them on the GPU box)
// -DCR_PROBE_NO_DECISIONS / -DCR_PROBE_NO_DUMP: see cr_kernels.h.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CR_KERNELS_TEMPLATES_ONLY
#include "../caretta_amd/csrc/cr_kernels.h"

using namespace cr;

template <int MODE, int R>
__global__ __launch_bounds__(kStagedMaxWaves* kWave) void k_probe(const double* __restrict__ staged, int n, int m, int64_t strip_doubles,
                                                               uint32_t* __restrict__ words, unsigned long long* cyc, double* out) {
    extern __shared__ double lds[];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SeedMax sm;
    AlignEnd e;
    sm.score = 0;
    sm.i = sm.j = 0;
    e.sw = e.dtw_score = 0;
    e.start_layer = 0;
    const StripGeom geom = WidePlan<R>{0}.geom(w, n);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (MODE == 0) {
        SweepParams prm{0.0, 1.0, 0.01};
        sweep_staged<R, kDtw>(staged + (int64_t)w * strip_doubles, n, m, prm, lds, nullptr, words, sm, e, geom);
    } else {
        SweepParams prm{0.0, 0.0, 0.0};
        sweep_staged<R, kSwTrace | kZeroGap>(staged + (int64_t)w * strip_doubles, n, m, prm, lds, words, nullptr, sm, e, geom);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        cyc[0] = t1 - t0;
        out[0] = sm.score + e.dtw_score + (double)sm.i + (double)e.start_layer;
    }
}

template <int MODE, int R = 1>
static void run(const char* name, int n, int m) {
    const int waves = kStagedMaxWaves;
    const int64_t strip_doubles = (int64_t)(staged_steps(m) + 64) * kWave * R;
    std::vector<double> h((size_t)strip_doubles * waves);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        v = (double)(x >> 11) * (1.0 / 9007199254740992.0);
    }
    // skewed layout: line t, row slot q, lane l = S(row, column t - l); exact zeros outside [0, m), as stage_block writes them
    for (int w = 0; w < waves; w++) {
            for (int64_t t = 0; t < strip_doubles / (kWave * R); t++)
                for (int q = 0; q < R; q++)
                    for (int l = 0; l < kWave; l++)
                        if (t - l < 0 || t - l >= m) h[(size_t)w * strip_doubles + ((size_t)t * R + q) * kWave + l] = 0.0;
    }
    double *d, *out;
    uint32_t* words;
    unsigned long long* cyc;
    hipMalloc(&d, h.size() * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&words, (size_t)waves * (m + 128) * kWave * 4 * R);
    hipMalloc(&cyc, 8);
    hipMalloc(&out, 8);
    size_t lds = (MODE == 0 ? sweep_staged_lds_doubles<kDtw>(waves) : sweep_staged_lds_doubles<kSwTrace>(waves)) * 8 + 4096;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<MODE, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL((k_probe<MODE, R>), dim3(1), dim3(waves * kWave), lds, 0, d, n, m, strip_doubles, words, cyc, out);
        hipDeviceSynchronize();
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        if (c < best) best = c;
    }
    const int strips = (n + 64 * R - 1) / (64 * R);
    const int steps = (R <= 2 ? 80 : 72) * (strips - 1) + (m + 63 + 15) / 16 * 16;
    printf("%-22s R %d n %4d (%d strips) m %4d : %8llu cycles, %4d steps, %6.1f cycles per step\n", name, R, n, strips, m, best, steps,
           (double)best / steps);
    hipFree(d);
    hipFree(words);
    hipFree(cyc);
    hipFree(out);
}

int main() {
#if defined(CR_PROBE_NO_DECISIONS) && defined(CR_PROBE_NO_DUMP)
    printf("== without decision packing, without hand-off writes\n");
#elif defined(CR_PROBE_NO_DECISIONS)
    printf("== without decision packing\n");
#elif defined(CR_PROBE_NO_DUMP)
    printf("== without hand-off writes\n");
#elif defined(CR_PROBE_MASKED_RAMPS)
    printf("== every block in which a lane is outside [0, m) with the EXEC-masked step (the library until round 5)\n");
#else
    printf("== as in the library\n");
#endif
#ifdef CR_PROBE_ROWS
    // rows per lane: where two (three) rows per lane and half (a third of) the strips overtake one row per lane
    for (int n : {128, 192, 256, 300, 340, 400, 448, 512}) {
        run<0, 1>("DTW skewed", n, 330);
        run<0, 2>("DTW skewed", n, 330);
        if (n > 256) run<0, 3>("DTW skewed", n, 330);
        run<1, 1>("SW gap 0 skewed", n, 330);
        run<1, 2>("SW gap 0 skewed", n, 330);
    }
    for (int n : {640, 704, 768, 900, 1024}) {
        run<0, 2>("DTW skewed", n, 700);
        run<0, 3>("DTW skewed", n, 700);
        run<1, 2>("SW gap 0 skewed", n, 700);
        run<1, 3>("SW gap 0 skewed", n, 700);
    }
    for (int n : {1100, 1300, 1536}) {
        run<0, 3>("DTW skewed", n, 1200);
        run<0, 4>("DTW skewed", n, 1200);
        run<1, 3>("SW gap 0 skewed", n, 1200);
        run<1, 4>("SW gap 0 skewed", n, 1200);
    }
#else
    const int ns[] = {64, 128, 256, 320, 340, 512};
    for (int n : ns) run<0>("DTW skewed", n, 330);
    for (int n : ns) run<1>("SW gap 0 skewed", n, 330);
    run<0>("DTW skewed", 64, 1200);
    run<1>("SW gap 0 skewed", 64, 1200);
#endif
    return 0;
}
