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


// ------------------------------------------------------------------------------------------- round 5
// (i) The decimator of the octave recursion as a kernel of its own.  Rounds 2-4 ran it as a framed GEMM with ONE output column (a 128 x 128
// tile for a 1-wide product): 8.2 of the front end's 10.0 ms per 64 clips (profiles/r05_vqt_kernel_stats.txt).  out[b][m] = sum_j ypad[b][2 m + j] h[j]:
// a workgroup stages 2 * 1024 + taps inputs in LDS and every thread forms 4 outputs; a lane reads the two inputs of a tap PAIR as one 8-byte
// LDS word (consecutive outputs are 8 bytes apart: conflict-free), the taps are wave-uniform loads.
#define DEC_OUT_PER_WG 1024
__global__ __launch_bounds__(256) void vqt_decimate(const float* __restrict__ ypad, long plen, const float* __restrict__ taps, int ntaps /* even */,
                                                    float* __restrict__ out, long n_out) {
    extern __shared__ __attribute__((aligned(16))) float win[];         // 2 * DEC_OUT_PER_WG + ntaps (+ 2) floats
    const int b = blockIdx.y;
    const long m0 = (long)blockIdx.x * DEC_OUT_PER_WG;
    const float* y = ypad + (long)b * plen;
    const int need = 2 * DEC_OUT_PER_WG + ntaps;
    for (int i = threadIdx.x; i < need; i += 256) { const long s = 2 * m0 + i; win[i] = s < plen ? y[s] : 0.f; }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < ntaps; j += 2) {
        const float h0 = taps[j], h1 = taps[j + 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float2 x = *reinterpret_cast<const float2*>(win + 2 * (threadIdx.x + 256 * u) + j);
            acc[u] = fmaf(x.x, h0, acc[u]);
            acc[u] = fmaf(x.y, h1, acc[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const long m = m0 + threadIdx.x + 256 * u; if (m < n_out) out[(long)b * n_out + m] = acc[u]; }
}
int a2s_vqt_decimate_impl(hipStream_t st, const float* ypad, long plen, const float* taps, int ntaps, float* out, long n_out, int B) {
    A2S_REQUIRE(ypad && taps && out && B > 0 && n_out > 0, "vqt_decimate: null tensor");
    A2S_REQUIRE(ntaps > 0 && ntaps % 2 == 0 && ntaps <= 4096, "vqt_decimate: the tap count must be even (pad with a zero) and <= 4096, got %d", ntaps);
    const size_t shm = sizeof(float) * (2 * DEC_OUT_PER_WG + ntaps + 2);
    hipLaunchKernelGGL(vqt_decimate, dim3((unsigned)a2s_cdiv(n_out, DEC_OUT_PER_WG), B), dim3(256), shm, st, ypad, plen, taps, ntaps, out, n_out);
    A2S_CHECK_LAUNCH("vqt_decimate");
    return A2S_OK;
}

// (ii) The log-magnitude epilogue over a response laid out OCTAVE BY OCTAVE: C (B, rows, n_oct * 2 * bpo), octave o (highest first) =
// [re (bpo) | im (bpo)] of bins lo_o .. lo_o + bpo - 1 with lo_o = bins - (o + 1) * bpo -- what ONE framed GEMM per octave against the
// (n_fft, 2 * bpo) bank writes (rounds 2-4: two GEMMs per octave, one per part).  One workgroup per (clip, row); the clip maximum is reduced
// once per workgroup, not once per element.
__global__ __launch_bounds__(256) void vqt_mag_max_oct(const float* __restrict__ C, float* __restrict__ partial, long rows, int bins, int bpo, int blocks_per_clip) {
    const int b = blockIdx.x / blocks_per_clip, blk = blockIdx.x % blocks_per_clip;
    const long n = rows * bins;
    const float* Cb = C + (long)b * rows * 2 * bins;
    __shared__ float red[16];
    float m = 0.f;
    for (long i = (long)blk * 256 + threadIdx.x; i < n; i += (long)blocks_per_clip * 256) {
        const long r = i / bins; const int k = (int)(i % bins);
        const int o = k / bpo, kk = k - o * bpo;                       // (any bin order serves the maximum)
        const float re = Cb[r * 2 * bins + o * 2 * bpo + kk], im = Cb[r * 2 * bins + o * 2 * bpo + bpo + kk];
        m = fmaxf(m, re * re + im * im);
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = sqrtf(m);
}
__global__ __launch_bounds__(256) void vqt_logmag_oct(const float* __restrict__ C, const float* __restrict__ partial, float* __restrict__ out, long rows, int bins,
                                                      int bpo, int blocks_per_clip, float top_db) {
    const int b = blockIdx.y;
    const long r = blockIdx.x;
    __shared__ float red[16];
    __shared__ float ref_db;
    float mx = 0.f;
    for (int q = threadIdx.x; q < blocks_per_clip; q += 256) mx = fmaxf(mx, partial[(long)b * blocks_per_clip + q]);
    mx = block_max(mx, red);
    if (threadIdx.x == 0) ref_db = 20.f * log10f(fmaxf(1e-5f, mx));
    __syncthreads();
    const float rdb = ref_db;
    const float* Cr = C + ((long)b * rows + r) * 2 * bins;
    const int n_oct = bins / bpo;
    for (int k = threadIdx.x; k < bins; k += 256) {
        const int o = n_oct - 1 - k / bpo, kk = k % bpo;               // octave 0 holds the HIGHEST bins
        const float re = Cr[o * 2 * bpo + kk], im = Cr[o * 2 * bpo + bpo + kk];
        const float mag = sqrtf(re * re + im * im);
        float db = 20.f * log10f(fmaxf(1e-5f, mag)) - rdb;
        db = fmaxf(db, -top_db);
        out[((long)b * rows + r) * bins + k] = (db + 80.f) / 80.f;
    }
}
int a2s_vqt_logmag_octaves_impl(hipStream_t st, const float* C, float* out, float* partial, int B, long rows, int bins, int bpo, float top_db) {
    A2S_REQUIRE(C && out && partial, "vqt_logmag_octaves: null tensor");
    A2S_REQUIRE(bpo > 0 && bins % bpo == 0, "vqt_logmag_octaves: bins (%d) must be a multiple of the bins per octave (%d)", bins, bpo);
    const int bpc = 64;
    hipLaunchKernelGGL(vqt_mag_max_oct, dim3(B * bpc), dim3(256), 0, st, C, partial, rows, bins, bpo, bpc);
    A2S_CHECK_LAUNCH("vqt_mag_max_oct");
    hipLaunchKernelGGL(vqt_logmag_oct, dim3((unsigned)rows, B), dim3(256), 0, st, C, partial, out, rows, bins, bpo, bpc, top_db);
    A2S_CHECK_LAUNCH("vqt_logmag_oct");
    return A2S_OK;
}
