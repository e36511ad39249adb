// VQT front-end epilogue (reference utilities.get_VQT, utilities.py:240-254, which calls librosa.vqt offline).
// The transform itself is a framed complex GEMM on the matrix cores: frames are overlapping windows of the zero-padded waveform
// (A(n, m) = y[n*hop + m], a Hankel view expressed with the GEMM's row stride = hop), B is the precomputed kernel bank
// [taps][2*bins] (real | imaginary parts, each bin's Hann-windowed complex exponential centred in the tap axis), C = (frames, 2*bins).
// This file turns C into the network input: magnitude -> dB relative to the clip maximum (floor 1e-5, top_db 80) -> /80 + 1 in [0,1].
#include "a2s_common.h"

// per-block maximum of |C| over one clip's rows; partial[b][blk]
__global__ __launch_bounds__(256) void vqt_mag_max(const float* __restrict__ C, float* __restrict__ partial, long rows, int bins, int blocks_per_clip) {
    const int b = blockIdx.x / blocks_per_clip, blk = blockIdx.x % blocks_per_clip;
    const long n = rows * bins;
    const float* Cb = C + (long)b * rows * 2 * bins;
    __shared__ float red[16];
    float m = 0.f;
    for (long i = (long)blk * 256 + threadIdx.x; i < n; i += (long)blocks_per_clip * 256) {
        const long r = i / bins; const int k = (int)(i % bins);
        const float re = Cb[r * 2 * bins + k], im = Cb[r * 2 * bins + bins + k];
        m = fmaxf(m, re * re + im * im);
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = sqrtf(m);
}

// out[b, r, k] = clamp(20 log10(max(1e-5,|C|)) - 20 log10(max(1e-5, clipmax)), >= -top_db) / 80 + 1
__global__ void vqt_logmag(const float* __restrict__ C, const float* __restrict__ partial, float* __restrict__ out, long rows, int bins,
                           int blocks_per_clip, float top_db, long total) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long per = rows * bins;
        const int b = (int)(i / per); const long j = i % per;
        const long r = j / bins; const int k = (int)(j % bins);
        float mx = 0.f;
        for (int q = 0; q < blocks_per_clip; ++q) mx = fmaxf(mx, partial[(long)b * blocks_per_clip + q]);
        const float* Cb = C + (long)b * rows * 2 * bins;
        const float re = Cb[r * 2 * bins + k], im = Cb[r * 2 * bins + bins + k];
        const float mag = sqrtf(re * re + im * im);
        float db = 20.f * log10f(fmaxf(1e-5f, mag)) - 20.f * log10f(fmaxf(1e-5f, mx));
        db = fmaxf(db, -top_db);
        out[i] = (db + 80.f) / 80.f;          // exactly 0 at the floor and exactly 1 at the clip maximum (db/80 + 1 rounds to -1.5e-8 at -80 dB)
    }
}

int a2s_vqt_logmag_impl(hipStream_t st, const float* C, float* out, float* partial, int B, long rows, int bins, float top_db) {
    A2S_REQUIRE(C && out && partial, "vqt_logmag: null tensor");
    const int bpc = 64;
    hipLaunchKernelGGL(vqt_mag_max, dim3(B * bpc), dim3(256), 0, st, C, partial, rows, bins, bpc);
    A2S_CHECK_LAUNCH("vqt_mag_max");
    const long total = (long)B * rows * bins;
    hipLaunchKernelGGL(vqt_logmag, dim3(min((long)4096, (total + 255) / 256)), dim3(256), 0, st, C, partial, out, rows, bins, bpc, top_db, total);
    A2S_CHECK_LAUNCH("vqt_logmag");
    return A2S_OK;
}
