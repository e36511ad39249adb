// Store-pattern microbenchmark for the (B*T, C = 40, F = 480) activation layout: the same 64-column x 40-channel x 4-row tile written
//   A: as the MFMA accumulator layout gives it (lane = channel li, 4 consecutive columns lk*4: one store instruction = 16 channel rows x 64 B)
//   B: row-contiguous (one store instruction = 4 channel rows x 256 B)
// build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256) void k(float* __restrict__ y, int rows, int tiles_per_wg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int F = 480, C = 40, tilesF = 8;
    int bid = blockIdx.x;
    const int ft = bid % tilesF; bid /= tilesF;
    const int f0 = ft * 64;
    for (int tile = 0; tile < tiles_per_wg; ++tile) {
        const long row = ((long)bid * tiles_per_wg + tile) * 4 + wave;
        if (row >= rows) return;
        const f32x4 v = {(float)row, (float)tid, 1.f, 2.f};
        if (PATTERN == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = j * 16 + li, f = f0 + i * 16 + lk * 4;
                    if (co < C && f + 3 < F) *reinterpret_cast<f32x4*>(y + (row * C + co) * F + f) = v;
                }
        } else {
#pragma unroll
            for (int c0 = 0; c0 < C; c0 += 4) {
                const int co = c0 + lk, f = f0 + li * 4;
                if (f + 3 < F) *reinterpret_cast<f32x4*>(y + (row * C + co) * F + f) = v;
            }
        }
    }
}

template <int PATTERN>
void run(float* y, int rows) {
    const int tiles_per_wg = 8;
    const int grid = ((rows / 4 + tiles_per_wg - 1) / tiles_per_wg) * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PATTERN>, dim3(grid), dim3(256), 0, 0, y, rows, tiles_per_wg);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k<PATTERN>, dim3(grid), dim3(256), 0, 0, y, rows, tiles_per_wg);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double bytes = (double)rows * 40 * 480 * 4;
    printf("pattern %c: %.3f ms  %.0f GB/s\n", PATTERN ? 'B' : 'A', ms, bytes / ms / 1e6);
}

int main() {
    const int rows = 64 * 1201;
    float* y;
    (void)hipMalloc(&y, (size_t)rows * 40 * 480 * 4);
    run<0>(y, rows); run<1>(y, rows); run<0>(y, rows); run<1>(y, rows);
    return 0;
}
