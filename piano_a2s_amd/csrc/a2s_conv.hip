// ConvStack kernels (reference models.py:463-543): 3x3 convolutions (stride 1,
// zero pad 1, no bias) with BatchNorm (batch statistics) + ReLU, on activations laid out (B, T, C, F) fp32.
//
// Layout: the reference keeps NCHW = (B, C, T, F) and then does transpose(1,2).flatten(2) to get
// (B, T, C*F) rows for the 19200->256 Linear (models.py:537).  Here every activation lives as (B, T, C, F)
// from the start: the network input (B,1,T,F) is already that layout, each (b,t) row holds C planes of F
// contiguous floats (coalesced along F), and the last layer's output IS the (B*T, C*F) GEMM operand with the
// reference's column order c*F+f -- the 92 MB/clip transpose copy never exists.
//
// conv3x3_mfma: implicit GEMM on v_mfma_f32_16x16x4_f32.  M = 16 consecutive f positions, N = 16 output
//   channels, K = (dt, df, ci) with ci fastest (4 consecutive input channels per MFMA).  A workgroup owns
//   TR=4 rows x FT=32 columns x all Cout; input channels are streamed through LDS in chunks of 20 together
//   with their weight slice, so LDS stays at ~46 KB and 3 workgroups share a CU (one stages while others
//   multiply).  The previous layer's BatchNorm+ReLU is applied while staging (y = max(0, x*scale+shift)),
//   so post-activation tensors are never written; the epilogue emits per-workgroup per-channel sum / sum of
//   squares for THIS layer's batch statistics (reduced in a fixed order by bn_finalize -> deterministic).
// conv3x3_c1: the first layer (Cin = 1, K = 9) as a direct VALU kernel (0.7 % of the stack's flops).
#include "a2s_common.h"

#define CV_TR 4
#define CV_FT 32
#define CV_CK 20                 // input channels per LDS chunk
#define CV_RS 36                 // LDS row stride (FT + 2 halo, padded)
#define CV_PLANE 240             // LDS plane stride: (TR+2)*RS = 216 -> 240 (== 16 mod 32: conflict-free k pairs)
#define CV_WS 48                 // weight row stride in LDS (3 n-tiles of 16)

struct ConvArgs {
    const float* x;      // (B, T, Cin, F)   pre-activation of the previous layer (or the spectrogram)
    const float* w;      // (Cout, Cin, 3, 3) reference layout
    float* y;            // (B, T, Cout, F)  pre-BN conv output
    const float* in_scale; const float* in_shift;   // per input channel; null -> identity, no ReLU
    float* stat_partial; // [nblocks][Cout][2] sum, sumsq over the block's valid positions (null -> skip)
    int B, T, F, Cin, Cout;
    int flip;            // 1: use w as a transposed/flipped kernel (dgrad): w'[ci][co][2-dt][2-df]
};

template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_mfma(ConvArgs a) {
    constexpr int NT = (COUT + 15) / 16;               // n-tiles: 2 (Cout 20) or 3 (Cout 40)
    __shared__ __attribute__((aligned(16))) float lin[CV_CK * CV_PLANE];       // 19200 B
    __shared__ __attribute__((aligned(16))) float lw[9 * CV_CK * CV_WS];       // 34560 B
    __shared__ float red[4][NT * 16][2];

    const int tilesF = (a.F + CV_FT - 1) / CV_FT;
    const int tilesT = (a.T + CV_TR - 1) / CV_TR;
    int bid = blockIdx.x;
    const int ft = bid % tilesF; bid /= tilesF;
    const int tt = bid % tilesT; const int b = bid / tilesT;
    const int t0 = tt * CV_TR, f0 = ft * CV_FT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;

    // wave w owns output row t0+w: two m-tiles (f0..f0+15, f0+16..f0+31) x NT n-tiles
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < a.Cin; c0 += CV_CK) {
        __syncthreads();      // previous chunk fully consumed
        // ---- stage input planes [CK][TR+2][FT+2] with the producer's BN+ReLU folded in; zero = padding
        for (int e = tid; e < CV_CK * (CV_TR + 2) * (CV_FT + 2); e += 256) {
            const int fc = e % (CV_FT + 2);
            const int r = (e / (CV_FT + 2)) % (CV_TR + 2);
            const int c = e / ((CV_FT + 2) * (CV_TR + 2));
            const int t = t0 + r - 1, f = f0 + fc - 1, ci = c0 + c;
            float v = 0.f;
            if (t >= 0 && t < a.T && f >= 0 && f < a.F && ci < a.Cin) {
                v = a.x[(((long)b * a.T + t) * a.Cin + ci) * a.F + f];
                if (a.in_scale) v = fmaxf(v * a.in_scale[ci] + a.in_shift[ci], 0.f);
            }
            lin[c * CV_PLANE + r * CV_RS + fc] = v;
        }
        // ---- stage the weight slice as B[k][n], k = (dt*3+df)*CK + c, n = output channel (zero padded)
        for (int e = tid; e < 9 * CV_CK * CV_WS; e += 256) {
            const int n = e % CV_WS, k = e / CV_WS;
            const int c = k % CV_CK, tap = k / CV_CK;
            const int ci = c0 + c;
            float v = 0.f;
            if (n < COUT && ci < a.Cin) {
                if (!a.flip) v = a.w[((long)n * a.Cin + ci) * 9 + tap];
                else         v = a.w[((long)ci * COUT + n) * 9 + (8 - tap)];   // w'[n<-ci] flipped: dgrad
            }
            lw[k * CV_WS + n] = v;
        }
        __syncthreads();
        // ---- multiply: 9 taps x CK/4 k-steps
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dt = tap / 3, df = tap % 3;
#pragma unroll
            for (int cs = 0; cs < CV_CK / 4; ++cs) {
                const int c = cs * 4 + lk;
                const float* src = lin + c * CV_PLANE + (wave + dt) * CV_RS + df + li;
                const float a0 = src[0], a1 = src[16];
                const float* wsrc = lw + (tap * CV_CK + c) * CV_WS + li;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float bv = wsrc[j * 16];
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue.  C/D map: lane holds column n = li (channel), rows lk*4+r (f positions) of each tile.
    const int t = t0 + wave;
    const bool row_ok = t < a.T;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = j * 16 + li;
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = f0 + i * 16 + lk * 4;
            if (row_ok && co < COUT) {
                float* dst = a.y + (((long)b * a.T + t) * COUT + co) * a.F + f;
                if (f + 3 < a.F && (a.F % 4 == 0)) {
                    *reinterpret_cast<f32x4*>(dst) = acc[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s += acc[i][j][r]; s2 += acc[i][j][r] * acc[i][j][r]; }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (f + r < a.F) { dst[r] = acc[i][j][r]; s += acc[i][j][r]; s2 += acc[i][j][r] * acc[i][j][r]; }
                }
            }
        }
        if (a.stat_partial) {
            // reduce over the 4 lane groups (lk) -> lanes 0..15 hold the wave's per-channel sums
            s += __shfl_xor(s, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s += __shfl_xor(s, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) { red[wave][j * 16 + li][0] = s; red[wave][j * 16 + li][1] = s2; }
        }
    }
    if (a.stat_partial) {
        __syncthreads();
        if (tid < COUT) {
            float s = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[w][tid][0]; s2 += red[w][tid][1]; }
            a.stat_partial[((long)blockIdx.x * COUT + tid) * 2 + 0] = s;
            a.stat_partial[((long)blockIdx.x * COUT + tid) * 2 + 1] = s2;
        }
    }
}

// First layer: Cin = 1.  One thread per (b,t,f) position computes all Cout (<= 20) channels.
__global__ __launch_bounds__(256) void conv3x3_c1(ConvArgs a) {
    __shared__ float lw[20 * 9];
    __shared__ float red[4][20][2];
    for (int e = threadIdx.x; e < a.Cout * 9; e += 256) lw[e] = a.w[e];
    __syncthreads();
    const long pos = (long)blockIdx.x * 256 + threadIdx.x;
    const long npos = (long)a.B * a.T * a.F;
    const bool ok = pos < npos;
    const int f = ok ? (int)(pos % a.F) : 0;
    const int t = ok ? (int)((pos / a.F) % a.T) : 0;
    const int b = ok ? (int)(pos / ((long)a.F * a.T)) : 0;
    float v[9];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int df = 0; df < 3; ++df) {
            const int tt = t + dt - 1, ff = f + df - 1;
            float x = 0.f;
            if (ok && tt >= 0 && tt < a.T && ff >= 0 && ff < a.F) x = a.x[((long)b * a.T + tt) * a.F + ff];
            v[dt * 3 + df] = x;
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int co = 0; co < a.Cout; ++co) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s = fmaf(v[k], lw[co * 9 + k], s);
        if (ok) a.y[(((long)b * a.T + t) * a.Cout + co) * a.F + f] = s;
        if (a.stat_partial) {
            const float sv = ok ? s : 0.f;
            const float ws = wave_sum(sv), ws2 = wave_sum(sv * sv);
            if (lane == 0) { red[wave][co][0] = ws; red[wave][co][1] = ws2; }
        }
    }
    if (a.stat_partial) {
        __syncthreads();
        if (threadIdx.x < a.Cout) {
            float s = 0.f, s2 = 0.f;
            for (int w = 0; w < 4; ++w) { s += red[w][threadIdx.x][0]; s2 += red[w][threadIdx.x][1]; }
            a.stat_partial[((long)blockIdx.x * a.Cout + threadIdx.x) * 2 + 0] = s;
            a.stat_partial[((long)blockIdx.x * a.Cout + threadIdx.x) * 2 + 1] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------- BatchNorm
// Fixed-order reduction of the per-workgroup partials in double (the reference's CPU kernel accumulates
// batch statistics in double: at::acc_type<float> on CPU), then the affine the consumer applies on load.
struct BnFinalizeArgs {
    const float* partial; int nblocks; int C; double count;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var; long long* num_batches_tracked;   // updated when training
    float* mean; float* invstd;          // saved for backward
    float* scale; float* shift;          // y = x*scale + shift
    float eps, momentum; int training;
};

__global__ __launch_bounds__(256) void bn_finalize(BnFinalizeArgs a) {
    const int c = blockIdx.x;
    __shared__ double rs[256], rs2[256];
    double s = 0.0, s2 = 0.0;
    if (a.training) {
        for (int i = threadIdx.x; i < a.nblocks; i += 256) {
            s += (double)a.partial[((long)i * a.C + c) * 2 + 0];
            s2 += (double)a.partial[((long)i * a.C + c) * 2 + 1];
        }
    }
    rs[threadIdx.x] = s; rs2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rs2[threadIdx.x] += rs2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double mean, var;
        if (a.training) {
            mean = rs[0] / a.count;
            var = rs2[0] / a.count - mean * mean;          // biased; double keeps the cancellation harmless
            if (var < 0.0) var = 0.0;
            const double unb = a.count > 1.0 ? var * a.count / (a.count - 1.0) : var;
            a.running_mean[c] = (float)((1.0 - a.momentum) * a.running_mean[c] + a.momentum * mean);
            a.running_var[c] = (float)((1.0 - a.momentum) * a.running_var[c] + a.momentum * unb);
            if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
        } else {
            mean = a.running_mean[c]; var = a.running_var[c];
        }
        const double inv = 1.0 / sqrt(var + (double)a.eps);
        a.mean[c] = (float)mean; a.invstd[c] = (float)inv;
        const double sc = (double)a.gamma[c] * inv;
        a.scale[c] = (float)sc;
        a.shift[c] = (float)((double)a.beta[c] - mean * sc);
    }
}

// y = relu(x*scale[c] + shift[c]) over (rows, C, F) planes -- materialises the last ConvStack activation as
// the (B*T, C*F) operand of the 19200->256 GEMM.
__global__ void bn_relu_apply(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ scale,
                              const float* __restrict__ shift, long n, int C, int F) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)((i / F) % C);
        y[i] = fmaxf(x[i] * scale[c] + shift[c], 0.f);
    }
}

// Column statistics of a (rows, C) matrix (BatchNorm1d of the Linear output, reference models.py:505,539):
// partial sums per row block, same [nblocks][C][2] format as the conv epilogue.
__global__ __launch_bounds__(256) void col_stats_partial(const float* __restrict__ x, float* __restrict__ partial,
                                                         long rows, int C, int rows_per_block) {
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f, s2 = 0.f;
        for (long r = r0; r < r1; ++r) { const float v = x[r * C + c]; s += v; s2 += v * v; }
        partial[((long)blockIdx.x * C + c) * 2 + 0] = s;
        partial[((long)blockIdx.x * C + c) * 2 + 1] = s2;
    }
}

// y[r,c] = relu(x[r,c]*scale[c]+shift[c]) (* dropout mask/keep when mask != null)
__global__ void bn1d_relu_dropout(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ scale,
                                  const float* __restrict__ shift, const uint8_t* __restrict__ mask, float inv_keep,
                                  long n, int C) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % C);
        float v = fmaxf(x[i] * scale[c] + shift[c], 0.f);
        if (mask) v = mask[i] ? v * inv_keep : 0.f;
        y[i] = v;
    }
}

// ------------------------------------------------------------------------------------------- launchers
int a2s_conv3x3_impl(hipStream_t st, const float* x, const float* w, float* y, const float* in_scale,
                     const float* in_shift, float* stat_partial, int B, int T, int F, int Cin, int Cout, int flip) {
    A2S_REQUIRE(x && w && y, "conv3x3: null tensor");
    A2S_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3x3: scale/shift must come together");
    ConvArgs a{x, w, y, in_scale, in_shift, stat_partial, B, T, F, Cin, Cout, flip};
    if (Cin == 1) {
        A2S_REQUIRE(Cout <= 20 && !flip && !in_scale, "conv3x3: Cin=1 path supports Cout<=20, no flip, no input affine");
        hipLaunchKernelGGL(conv3x3_c1, dim3(a2s_cdiv((long)B * T * F, 256)), dim3(256), 0, st, a);
    } else {
        A2S_REQUIRE(Cin % 4 == 0 && Cin % CV_CK == 0, "conv3x3: Cin must be a multiple of %d", CV_CK);
        const int nblk = B * a2s_cdiv(T, CV_TR) * a2s_cdiv(F, CV_FT);
        if (Cout == 20) hipLaunchKernelGGL(conv3x3_mfma<20>, dim3(nblk), dim3(256), 0, st, a);
        else if (Cout == 40) hipLaunchKernelGGL(conv3x3_mfma<40>, dim3(nblk), dim3(256), 0, st, a);
        else A2S_FAIL(A2S_ERR_ARG, "conv3x3: Cout must be 20 or 40 (got %d)", Cout);
    }
    A2S_CHECK_LAUNCH("conv3x3");
    return A2S_OK;
}

int a2s_conv3x3_stat_blocks_impl(int B, int T, int F, int Cin) {
    return Cin == 1 ? a2s_cdiv((long)B * T * F, 256) : B * a2s_cdiv(T, CV_TR) * a2s_cdiv(F, CV_FT);
}

int a2s_bn_finalize_impl(hipStream_t st, const float* partial, int nblocks, int C, double count, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, long long* nbt, float* mean,
                         float* invstd, float* scale, float* shift, float eps, float momentum, int training) {
    A2S_REQUIRE(gamma && beta && running_mean && running_var && mean && invstd && scale && shift, "bn_finalize: null tensor");
    A2S_REQUIRE(!training || partial, "bn_finalize: training needs the statistics partials");
    BnFinalizeArgs a{partial, nblocks, C, count, gamma, beta, running_mean, running_var, nbt, mean, invstd, scale, shift, eps, momentum, training};
    hipLaunchKernelGGL(bn_finalize, dim3(C), dim3(256), 0, st, a);
    A2S_CHECK_LAUNCH("bn_finalize");
    return A2S_OK;
}

int a2s_bn_relu_apply_impl(hipStream_t st, const float* x, float* y, const float* scale, const float* shift, long n, int C, int F) {
    hipLaunchKernelGGL(bn_relu_apply, dim3(min((long)4096, (n + 255) / 256)), dim3(256), 0, st, x, y, scale, shift, n, C, F);
    A2S_CHECK_LAUNCH("bn_relu_apply");
    return A2S_OK;
}

int a2s_col_stats_impl(hipStream_t st, const float* x, float* partial, long rows, int C, int rows_per_block) {
    hipLaunchKernelGGL(col_stats_partial, dim3(a2s_cdiv(rows, rows_per_block)), dim3(256), 0, st, x, partial, rows, C, rows_per_block);
    A2S_CHECK_LAUNCH("col_stats_partial");
    return A2S_OK;
}

int a2s_bn1d_relu_dropout_impl(hipStream_t st, const float* x, float* y, const float* scale, const float* shift,
                               const uint8_t* mask, float inv_keep, long n, int C) {
    hipLaunchKernelGGL(bn1d_relu_dropout, dim3(min((long)4096, (n + 255) / 256)), dim3(256), 0, st, x, y, scale, shift, mask, inv_keep, n, C);
    A2S_CHECK_LAUNCH("bn1d_relu_dropout");
    return A2S_OK;
}
