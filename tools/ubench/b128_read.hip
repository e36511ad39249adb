// ds_read_b128 fragment-read pattern of the MFMA 16x16x32 f16 operands (lane = (lr = row, lk): 16 bytes at row * STRIDE + 16 lk) and the
// 8-byte staging writes of csrc/a2s_linear.hip: LDS clocks per wave-level instruction as a function of the row stride / lane mapping.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/b128_read.hip -o tools/ubench/b128_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int STRIDE_B>
__global__ void read_timing(unsigned long long* out, int iters, unsigned* sink) {
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 128 * STRIDE_B / 4; i += blockDim.x) ((unsigned*)lds)[i] = i * 2654435761u;
    __syncthreads();
    const int l = threadIdx.x & 63, lr = l & 15, lk = l >> 4;
    const unsigned base = (unsigned)(unsigned long long)(lds) + lr * STRIDE_B + lk * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned a = base + (it & 1) * 64;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a + mt * 16 * STRIDE_B));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// 8-byte writes: MODE 0: lane -> (row = 4 (l & 31) + c, 8-byte slot = l >> 5) [the weight gradient's first mapping]; MODE 1: lane -> (row = 4 (l & 7) + c,
// slot = l >> 3); MODE 2: lane -> (row = l >> 4, slot = l & 15) [the forward's mapping: 16 lanes per row]
template <int STRIDE_B, int MODE>
__global__ void write_timing(unsigned long long* out, int iters) {
    extern __shared__ unsigned char lds[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    int row, slot;
    if (MODE == 0) { row = 4 * (l & 31); slot = (l >> 5) + 2 * w; }
    else if (MODE == 1) { row = 4 * (l & 7) + 32 * (w & 3); slot = (l >> 3) + 8 * (w >> 2); }
    else if (MODE == 3) { row = 4 * (l >> 4) + 16 * w; slot = l & 15; }          // 16 lanes fill one row's 128 bytes, 4 rows (4 apart) per instruction
    else if (MODE == 4) { row = 4 * (l & 3) + 16 * (w & 3); slot = (l >> 2); }    // 4 column quads x 16 row quads
    else { row = (l >> 4) + 4 * w; slot = l & 15; }
    const unsigned base = (unsigned)(unsigned long long)(lds) + row * STRIDE_B + slot * 8;
    const uint2 val = make_uint2(l, w);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("ds_write_b64 %0, %1" :: "v"(base + (MODE == 2 ? 32 * c : c) * STRIDE_B), "v"(val));
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned long long* t; hipMalloc(&t, 8 * 1024); unsigned* sink; hipMalloc(&sink, 4);
    const int iters = 4000;
    unsigned long long h[4];
#define RUNR(S)                                                                                                            \
    for (int waves = 1; waves <= 8; waves *= 2) {                                                                          \
        hipLaunchKernelGGL((read_timing<S>), dim3(1), dim3(64 * waves), 128 * S + 256, 0, t, iters, sink);                 \
        hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);                                                                         \
        printf("read  stride %4d B, %d waves: %.2f clocks per CU-level ds_read_b128 (ideal 8)\n", S, waves, (double)h[0] / (iters * 8.0 * waves)); \
    }
    RUNR(80) RUNR(144) RUNR(96) RUNR(160) RUNR(224) RUNR(288) RUNR(136) RUNR(152)
#define RUNW(S, M)                                                                                                         \
    { hipLaunchKernelGGL((write_timing<S, M>), dim3(1), dim3(256), 160 * 1024 - 512, 0, t, iters);                         \
      hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);                                                                           \
      printf("write stride %4d B, mode %d, 4 waves: %.2f clocks per CU-level ds_write_b64 (ideal 4)\n", S, M, (double)h[0] / (iters * 4.0 * 4)); }
    RUNW(144, 0) RUNW(144, 1) RUNW(144, 2) RUNW(144, 3) RUNW(144, 4)
    RUNW(160, 0) RUNW(160, 1) RUNW(160, 2) RUNW(160, 3) RUNW(160, 4)
    RUNW(176, 0) RUNW(176, 1) RUNW(176, 3) RUNW(136, 0) RUNW(136, 1) RUNW(152, 0) RUNW(152, 1)
    return 0;
}
