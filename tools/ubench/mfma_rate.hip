// Back-to-back v_mfma_f32_16x16x32_f16 issue rate: NACC independent accumulators per wave, W waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 acc[NACC];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int wgs_per_cu, int iters) {
    float* out;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 3 * NACC * wgs_per_cu;       // one wave of each workgroup per SIMD
    const double tflops = (double)grid * 4 * iters * 3 * NACC * 16384.0 / (ms * 1e-3) / 1e12;
    printf("NACC %2d  waves/SIMD %d: %.3f ms  %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)  %.0f TFLOP/s\n", NACC, wgs_per_cu, ms,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, tflops);
    hipFree(out);
}

int main() {
    run<12>(1, 20000); run<12>(2, 20000); run<4>(1, 60000); run<4>(2, 60000); run<2>(1, 100000); run<2>(2, 100000); run<1>(1, 100000);
    return 0;
}
