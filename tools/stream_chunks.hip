// How fast can MI355X feed MANY independent streams that each read their own row in small aligned chunks?
//   ./stream_chunks.bin [rows_per_matrix=300] [matrices=8128]
// The access pattern of the streaming explicit-matrix sweeps (cr_explicit_batch.h: sweep_stream): every lane of every wave owns
// one row (2400 bytes here) and walks along it; per visit it fetches CHUNK bytes, and between two visits of the same row all
// the other resident rows are visited once.  No arithmetic besides a checksum.  Prints the read rate for CHUNK = 64 .. 1024
// bytes (16 bytes per lane and load instruction, CHUNK / 16 lanes cooperate on a row's chunk) and 1, 2, 4 visits' loads in flight
// per wave, at 12 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CHUNK, int DEPTH>
__global__ __launch_bounds__(64) void k_streams(const double* __restrict__ S, int rows, int row_doubles, double* __restrict__ out) {
    constexpr int LPC = CHUNK / 16;                 // lanes per chunk
    constexpr int OWN = 64 / LPC;                   // rows served per load instruction
    constexpr int LOADS = 64 / OWN;                 // load instructions per visit of the wave's 64 rows
    const int lane = threadIdx.x;
    const int64_t base_row = (int64_t)blockIdx.x * rows;           // this wave's matrix
    double acc = 0.0;
    for (int strip = 0; strip * 64 < rows; strip++) {
        const int steps = (row_doubles * 8 + CHUNK - 1) / CHUNK;
        for (int v0 = 0; v0 < steps; v0 += DEPTH) {                 // DEPTH visits' loads in flight before the first is used
            double2 x[DEPTH][LOADS];
#pragma unroll
            for (int u = 0; u < DEPTH; u++)
#pragma unroll
                for (int y = 0; y < LOADS; y++) {
                    const int row = strip * 64 + y * OWN + lane / LPC;
                    const int off = (v0 + u) * (CHUNK / 8) + (lane % LPC) * 2;
                    x[u][y] = make_double2(0.0, 0.0);
                    if (row < rows && off + 1 < row_doubles) x[u][y] = *reinterpret_cast<const double2*>(S + (base_row + row) * row_doubles + off);
                }
#pragma unroll
            for (int u = 0; u < DEPTH; u++)
#pragma unroll
                for (int y = 0; y < LOADS; y++) acc += x[u][y].x + x[u][y].y;
        }
    }
    if (acc == 12345.678) out[blockIdx.x * 64 + lane] = acc;       // (keeps the loads alive)
}

template <int CHUNK, int DEPTH>
void run(const double* S, int rows, int mats, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 13 * 1024;                    // twelve waves per CU, as the sweep
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_streams<CHUNK, DEPTH>), dim3(mats), dim3(64), lds, 0, S, rows, rows, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = 8.0 * rows * rows * mats;
    printf("chunk %4d bytes, %d visits in flight: %.3f ms -> %.0f GB/s (%.3f of 8 TB/s)\n", CHUNK, DEPTH, best, bytes / best / 1e6, bytes / best / 1e6 / 8000);
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 300, mats = argc > 2 ? atoi(argv[2]) : 8128;
    const size_t n = (size_t)rows * rows * mats;
    double *S, *out;
    hipMalloc(&S, n * 8 + 4096);
    hipMalloc(&out, (size_t)mats * 64 * 8);
    hipMemset(S, 0, n * 8 + 4096);
    run<64, 1>(S, rows, mats, out);
    run<64, 2>(S, rows, mats, out);
    run<64, 4>(S, rows, mats, out);
    run<128, 1>(S, rows, mats, out);
    run<128, 2>(S, rows, mats, out);
    run<256, 1>(S, rows, mats, out);
    run<256, 2>(S, rows, mats, out);
    run<512, 1>(S, rows, mats, out);
    run<1024, 1>(S, rows, mats, out);
    return 0;
}
