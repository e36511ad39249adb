// ds_read_b64_tr_b16 probe (gfx950): (1) which element lands where, (2) LDS cycles of the fragment-read pattern the row-streaming
// convolution uses: image [channel row][128 positions + pad] fp16, 16-lane group g of k-step s reads the 4-channel block 8 s + 4 r + g.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/tr_read.hip -o tools/ubench/tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) s16x4*)(p))

__global__ void semantics(unsigned short* out) {
    __shared__ unsigned short lds[64 * 160];
    for (int i = threadIdx.x; i < 64 * 160; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, q = l & 15, g = l >> 4;
    // group g: block of 4 rows (row = 4 g + q / 4, stride 160 elements), 16 columns starting at column 32
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(lds + (4 * g + q / 4) * 160 + 32 + (q % 4) * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

template <int STRIDE_B>      // bytes per channel row
__global__ void timing(unsigned long long* out, int iters, float* sink) {
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 132 * STRIDE_B / 4; i += blockDim.x) ((unsigned*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int l = threadIdx.x & 63, q = l & 15, g = l >> 4;
    const unsigned char* base = lds + (g * 4 + q / 4) * STRIDE_B + (q % 4) * 8;
    s16x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(base + ((8 * s + 4 * r) * 4 % 120) * STRIDE_B + m * 32 + (it & 1) * 128));
                    acc += v;
                }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc[0] == 12345 && acc[1] == 77) sink[0] = acc[2] + acc[3];
}

int main() {
    unsigned short* d; hipMalloc(&d, 512);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int g = l >> 4, q = l & 15;
            const int want = (4 * g + j) * 160 + 32 + q;     // lane q of group g receives column q of rows 4g .. 4g+3
            if (h[l * 4 + j] != want) { if (ok) printf("MISMATCH lane %d elem %d: got %d (row %d col %d) want %d\n", l, j, h[l * 4 + j], h[l * 4 + j] / 160, h[l * 4 + j] % 160 - 32, want); ok = 0; }
        }
    printf("tr16_b64 semantics (lane q of a 16-lane group gets column q of the group's 4-row block, element j = row j): %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    unsigned long long* t; hipMalloc(&t, 8 * 1024); float* sink; hipMalloc(&sink, 4);
    const int iters = 2000;
#define RUN(S)                                                                                                               \
    for (int waves = 4; waves <= 8; waves += 4) {                                                                            \
        hipLaunchKernelGGL(timing<S>, dim3(256), dim3(64 * waves), 132 * S + 512, 0, t, iters, sink);                          \
        hipDeviceSynchronize();                                                                                              \
        unsigned long long hh[256]; hipMemcpy(hh, t, 8 * 256, hipMemcpyDeviceToHost);                                        \
        double s = 0; for (int i = 0; i < 256; ++i) s += hh[i];                                                              \
        printf("row stride %4d B, %d waves/CU: %.2f clocks per wave-level tr read (%.1f per CU-level read slot)\n", S, waves, s / 256 / iters / 32, s / 256 / iters / 32 / waves); \
    }
    RUN(256) RUN(272) RUN(288) RUN(320) RUN(384)
    return 0;
}
