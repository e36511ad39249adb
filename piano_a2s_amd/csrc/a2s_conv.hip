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
    // data-gradient launches only (template BNRED): the output g is the gradient wrt relu(bn(yl)) of the layer below; its BatchNorm
    // backward statistics  sum g', sum g' xhat  (g' = g where bn(yl) > 0) are accumulated in the epilogue and written to stat_partial
    // in the layout bn_bwd_finalize reads -- the separate statistics pass over (g, yl) disappears
    const float* yl; const float* yl_mean; const float* yl_invstd; const float* yl_scale; const float* yl_shift;
    // two-term fp16 split (conv3x3_split<.., 2>) only: device scalars holding max|x| (null: the operand is used unscaled) and max|w|
    const float* x_absmax; const float* w_absmax;
    float* out_absmax;   // [Cout] max |y| per output channel (atomic max of non-negative floats; zeroed by the launcher), or null
};

// ---- v2 geometry (round 1, after profiling: the first version spent more time staging than multiplying -- scalar loads with
// div/mod per element and a 34 KB weight slice re-derived from the (Cout,Cin,3,3) layout per 128 outputs).
//   * tile = 4 rows x 64 columns (4 m-tiles per wave): the weight slice is amortised over twice the outputs;
//   * weights are pre-packed ONCE per call into the LDS image [chunk][9*CK][48] (conv_pack_weights) -> straight 16-byte copies;
//   * the input tile is staged with 16-byte loads (interior) + 2 halo scalars per row; interior starts at column 4 of a 72-float
//     LDS row so the vector stores are aligned; plane stride 6*72 = 432 == 16 (mod 32) keeps the k-pair fragment reads conflict-free.
#define C2_FT 64
#define C2_RS 72
#define C2_PLANE (6 * C2_RS)
#define C2_WCHUNK (9 * CV_CK * CV_WS)      // floats per packed weight chunk

__global__ void conv_pack_weights(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int flip, int chunks) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= chunks * C2_WCHUNK) return;
    const int chunk = e / C2_WCHUNK, rem = e % C2_WCHUNK;
    const int n = rem % CV_WS, k = rem / CV_WS;
    const int c = k % CV_CK, tap = k / CV_CK;
    const int ci = chunk * CV_CK + c;
    float v = 0.f;
    if (n < Cout && ci < Cin) v = flip ? w[((long)ci * Cout + n) * 9 + (8 - tap)] : w[((long)n * Cin + ci) * 9 + tap];
    wp[e] = v;
}

// ---- v3 (round 1, after measuring that a workgroup spent ~2/3 of a chunk's wall time staging: every one of its ~17 global loads per
// thread was followed by its own LDS store, i.e. 17 serialised L2 round trips per chunk):
//   * a workgroup walks C3_TPW consecutive row tiles; the (tile, chunk) pairs form one pipeline of stages;
//   * the global loads of stage q+1 (input tile, weight slice if it changes) are issued into registers BEFORE the multiply of stage q
//     and committed to LDS after it -- one round trip per stage, hidden behind ~540 MFMAs per wave;
//   * with one input chunk (Cin = 20) the packed weights are staged once per workgroup; with two, consecutive tiles visit the chunks
//     in serpentine order (0,1 | 1,0 | 0,1 ...) so the weight slice changes every other stage only;
//   * batch-statistics partials are accumulated over the workgroup's tiles and written once.
#define C3_TPW 8
#define C3_WIT ((C2_WCHUNK / 4 + 255) / 256)                                  // float4 weight loads per thread and stage (9)
#define C3_XIT ((CV_CK * (CV_TR + 2) * (C2_FT / 4) + 255) / 256)              // float4 input loads per thread and stage (8)

template <int COUT, bool BNRED>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma(ConvArgs a, const float* __restrict__ wpack) {
    constexpr int NT = (COUT + 15) / 16;               // n-tiles: 2 (Cout 20) or 3 (Cout 40)
    __shared__ __attribute__((aligned(16))) float lin[CV_CK * C2_PLANE];       // 34560 B
    __shared__ __attribute__((aligned(16))) float lw[C2_WCHUNK];               // 34560 B
    __shared__ float red[4][NT * 16][2];

    const int tilesF = (a.F + C2_FT - 1) / C2_FT;
    const int tilesT = (a.T + CV_TR - 1) / CV_TR;
    const int groupsT = (tilesT + C3_TPW - 1) / C3_TPW;
    int bid = blockIdx.x;
    const int ft = bid % tilesF; bid /= tilesF;
    const int tg = bid % groupsT; const int b = bid / groupsT;
    const int f0 = ft * C2_FT;
    const int tile0 = tg * C3_TPW, ntile = min(C3_TPW, tilesT - tile0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const bool vec_ok = (a.F % 4 == 0) && (((uintptr_t)a.x & 15) == 0);
    const int nchunks = a.Cin / CV_CK;
    const int nstage = ntile * nchunks;

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[NT], st_s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { st_s[j] = 0.f; st_s2[j] = 0.f; }

    // stage q -> (tile, chunk): serpentine over the chunks so that consecutive stages share the weight slice where possible
    auto stage_tile = [&](int q) { return q / nchunks; };
    auto stage_chunk = [&](int q) { const int i = q / nchunks, c = q % nchunks; return (i & 1) ? nchunks - 1 - c : c; };

    f32x4 wreg[C3_WIT], xreg[C3_XIT];
    float hreg = 0.f;
    bool wreg_valid = false;
    auto issue = [&](int q, int resident_chunk) {
        const int ch = stage_chunk(q), t0 = (tile0 + stage_tile(q)) * CV_TR, c0 = ch * CV_CK;
        wreg_valid = ch != resident_chunk;
        if (wreg_valid) {
            const f32x4* wsrc4 = reinterpret_cast<const f32x4*>(wpack + (long)ch * C2_WCHUNK);
#pragma unroll
            for (int it = 0; it < C3_WIT; ++it) {
                const int e = tid + 256 * it;
                wreg[it] = e < C2_WCHUNK / 4 ? wsrc4[e] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int it = 0; it < C3_XIT; ++it) {
            const int e = tid + 256 * it;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < CV_CK * (CV_TR + 2) * (C2_FT / 4)) {
                const int j = e % (C2_FT / 4), row = e / (C2_FT / 4);
                const int r = row % (CV_TR + 2), c = row / (CV_TR + 2);
                const int t = t0 + r - 1, f = f0 + 4 * j, ci = c0 + c;
                if (t >= 0 && t < a.T && f < a.F) {
                    const float* src = a.x + (((long)b * a.T + t) * a.Cin + ci) * a.F + f;
                    if (vec_ok && f + 3 < a.F) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) if (f + qq < a.F) v[qq] = src[qq];
                    }
                }
            }
            xreg[it] = v;
        }
        if (tid < CV_CK * (CV_TR + 2) * 2) {           // halo columns f0-1 and f0+64
            const int side = tid & 1, row = tid >> 1;
            const int r = row % (CV_TR + 2), c = row / (CV_TR + 2);
            const int t = t0 + r - 1, f = side ? f0 + C2_FT : f0 - 1, ci = c0 + c;
            hreg = (t >= 0 && t < a.T && f >= 0 && f < a.F) ? a.x[(((long)b * a.T + t) * a.Cin + ci) * a.F + f] : 0.f;
        }
    };
    auto commit = [&](int q) {     // registers -> LDS; BN+ReLU of the producer folded in; zero = padding (of the ACTIVATED tensor)
        const int ch = stage_chunk(q), t0 = (tile0 + stage_tile(q)) * CV_TR, c0 = ch * CV_CK;
        if (wreg_valid) {
#pragma unroll
            for (int it = 0; it < C3_WIT; ++it) {
                const int e = tid + 256 * it;
                if (e < C2_WCHUNK / 4) reinterpret_cast<f32x4*>(lw)[e] = wreg[it];
            }
        }
#pragma unroll
        for (int it = 0; it < C3_XIT; ++it) {
            const int e = tid + 256 * it;
            if (e < CV_CK * (CV_TR + 2) * (C2_FT / 4)) {
                const int j = e % (C2_FT / 4), row = e / (C2_FT / 4);
                const int r = row % (CV_TR + 2), c = row / (CV_TR + 2);
                const int t = t0 + r - 1, f = f0 + 4 * j, ci = c0 + c;
                f32x4 v = xreg[it];
                if (a.in_scale && t >= 0 && t < a.T && f < a.F) {
                    const float sc = a.in_scale[ci], sh = a.in_shift[ci];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) v[qq] = (f + qq < a.F) ? fmaxf(v[qq] * sc + sh, 0.f) : 0.f;
                }
                *reinterpret_cast<f32x4*>(lin + c * C2_PLANE + r * C2_RS + 4 + 4 * j) = v;
            }
        }
        if (tid < CV_CK * (CV_TR + 2) * 2) {
            const int side = tid & 1, row = tid >> 1;
            const int r = row % (CV_TR + 2), c = row / (CV_TR + 2);
            const int t = t0 + r - 1, f = side ? f0 + C2_FT : f0 - 1, ci = c0 + c;
            float v = hreg;
            if (a.in_scale && t >= 0 && t < a.T && f >= 0 && f < a.F) v = fmaxf(v * a.in_scale[ci] + a.in_shift[ci], 0.f);
            lin[c * C2_PLANE + r * C2_RS + (side ? 4 + C2_FT : 3)] = v;
        }
    };

    int resident = -1;
    issue(0, resident);
    for (int q = 0; q < nstage; ++q) {
        __syncthreads();                      // previous stage fully consumed
        commit(q);
        resident = stage_chunk(q);
        __syncthreads();
        if (q + 1 < nstage) issue(q + 1, resident);      // in flight during the multiply below
        // ---- multiply: 9 taps x CK/4 k-steps, 4 m-tiles x NT n-tiles per wave
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dt = tap / 3, df = tap % 3;
#pragma unroll
            for (int cs = 0; cs < CV_CK / 4; ++cs) {
                const int c = cs * 4 + lk;
                const float* src = lin + c * C2_PLANE + (wave + dt) * C2_RS + 3 + df + li;
                float av[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = src[i * 16];
                const float* wsrc = lw + (tap * CV_CK + c) * CV_WS + li;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float bv = wsrc[j * 16];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[i][j], 0, 0, 0);
                }
            }
        }
        if ((q + 1) % nchunks != 0) continue;
        // ---- tile epilogue.  C/D map: lane holds column n = li (channel), rows lk*4+r (f positions) of each m-tile.
        const int t = (tile0 + stage_tile(q)) * CV_TR + wave;
        const bool row_ok = t < a.T;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = j * 16 + li;
            float bm = 0.f, bi = 0.f, bsc = 0.f, bsh = 0.f;
            f32x4 yv[4];
            if (BNRED && row_ok && co < COUT) {          // layer-below BatchNorm constants of this lane's channel + its yl values
                bm = a.yl_mean[co]; bi = a.yl_invstd[co]; bsc = a.yl_scale[co]; bsh = a.yl_shift[co];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = f0 + i * 16 + lk * 4;
                    const float* src = a.yl + (((long)b * a.T + t) * COUT + co) * a.F + f;
                    if (f + 3 < a.F && (a.F % 4 == 0)) yv[i] = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) yv[i][r] = (f + r < a.F) ? src[r] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = f0 + i * 16 + lk * 4;
                if (row_ok && co < COUT) {
                    float* dst = a.y + (((long)b * a.T + t) * COUT + co) * a.F + f;
                    const bool full = f + 3 < a.F && (a.F % 4 == 0);
                    if (full) *reinterpret_cast<f32x4*>(dst) = acc[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (!full && f + r >= a.F) continue;
                        const float v = acc[i][j][r];
                        if (!full) dst[r] = v;
                        if (BNRED) {
                            const float xv = yv[i][r];
                            const float gm = (xv * bsc + bsh > 0.f) ? v : 0.f;
                            st_s[j] += gm; st_s2[j] += gm * (xv - bm) * bi;
                        } else { st_s[j] += v; st_s2[j] += v * v; }
                    }
                }
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    if (a.stat_partial) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = st_s[j], s2 = st_s2[j];
            s += __shfl_xor(s, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s += __shfl_xor(s, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) { red[wave][j * 16 + li][0] = s; red[wave][j * 16 + li][1] = s2; }
        }
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

// ---- v4: the same fp32 convolution on the bf16 matrix pipes (16x the fp32-input MFMA rate).  Every fp32 operand is split
// EXACTLY into three bf16 terms x = x0 + x1 + x2 (8 significand bits each, by truncation; the residuals are exact in fp32), and a
// product is formed from the six term products of order >= 2^-16:  a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0), each exact in the
// fp32 accumulator (8 x 8 bits); the dropped terms are <= 3 * 2^-24 |a||b| -- the size of one fp32 rounding.  Same tile / pipeline
// as v3; what changes:
//   * K = (tap, 8-channel group): v_mfma_f32_16x16x32_bf16 takes 8 consecutive k per lane = the 8 channels of a group at one tap,
//     the four 16-lane groups of a k-step take four consecutive (tap, group) pairs of the chunk (pair p = tap * ncg + group);
//   * input chunk = 2 channel groups (16 channels; the last chunk of Cin = 20 / 40 has one), staged position-major:
//     lin[term][row 6][position 66][16 ch] bf16, so a lane's A fragment is ONE ds_read_b128 (32-byte position stride: conflict-free
//     within the hardware's 16-lane read groups);  a thread stages (row, position, group) items: 8 scalar loads (coalesced along
//     f, halo columns included -- no separate halo path), BatchNorm+ReLU, split, three 16-byte LDS stores;
//   * weights are pre-split and pre-packed per chunk as [term][pair slot 20][n COUT][8 ch] bf16 (conv_pack_weights_bf16x3).
#define C4_POS 66
#define C4_PSTR 32                              // bytes per position (2 groups x 8 bf16)
#define C4_ROWB (C4_POS * C4_PSTR)              // 2112
#define C4_INPL ((CV_TR + 2) * C4_ROWB)         // bytes per term plane of the input tile (12672)
#define C4_SLOTS 20                             // pair slots per chunk (5 k-steps x 4)
#define C4_XIT 4                                // staging items per thread and stage (792 items / 256)
#define C4_ITEMS ((CV_TR + 2) * C4_POS)         // items per channel group (396)

__host__ __device__ inline int c4_chunks(int Cin) { return ((Cin + 7) / 8 + 1) / 2; }
// Weight slots of a term plane: the even slots first, the odd ones from a 256-byte aligned base -- the two slots a hardware
// ds_read_b128 lane group touches (pairs p, p+1) then start on the same bank and their 8 + 8 lanes interleave without conflicts.
__host__ __device__ constexpr int c4_odd_base(int Cout) { return (C4_SLOTS / 2 * Cout * 16 + 255) / 256 * 256; }
__host__ __device__ constexpr int c4_wpl(int Cout) { return c4_odd_base(Cout) + C4_SLOTS / 2 * Cout * 16; }
__host__ __device__ inline size_t c4_chunk_bytes(int Cout, int terms = 3) { return ((size_t)terms * c4_wpl(Cout) + 4095) / 4096 * 4096; }   // whole LDS-DMA rounds

#ifndef C4_FRESH
#define C4_FRESH 0        // 1: sum each k-step's six products in a fresh accumulator (3x smaller element error, 15 % slower)
#endif
// max |w| of a small tensor by ONE workgroup (the fp16 two-term path scales the weights by a power of two: their second term would
// otherwise sit in fp16's subnormal range)
__global__ __launch_bounds__(256) void absmax_small(const float* __restrict__ w, int n, float* __restrict__ out) {
    __shared__ float red[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    m = block_max(m, red);
    if (threadIdx.x == 0) *out = m;
}

template <int TERMS>
__global__ void conv_pack_weights_split(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout, int flip,
                                        const float* __restrict__ w_absmax) {
    const int ncgs = (Cin + 7) / 8, nchunks = (ncgs + 1) / 2;
    const int per_term = C4_SLOTS * Cout * 8;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nchunks * per_term) return;
    const int chunk = e / per_term, rem = e % per_term;
    const int k8 = rem % 8, n = (rem / 8) % Cout, p = rem / (8 * Cout);
    const int ncg = min(2, ncgs - 2 * chunk);
    float v = 0.f;
    if (p < 9 * ncg) {
        const int tap = p / ncg, ci = (chunk * 2 + p % ncg) * 8 + k8;
        if (ci < Cin) v = flip ? w[((long)ci * Cout + n) * 9 + (8 - tap)] : w[((long)n * Cin + ci) * 9 + tap];
    }
#if C4_FRESH
    if ((p >> 2) & 1) v = -v;                         // odd k-steps accumulate the negated sum (c4_multiply subtracts it)
#endif
    const int wpl = c4_wpl(Cout) / 2;                 // in 16-bit elements
    unsigned short* dst = wp + (long)chunk * (c4_chunk_bytes(Cout, TERMS) / 2) + (p & 1) * (c4_odd_base(Cout) / 2) + ((p >> 1) * Cout + n) * 8 + k8;
    if (TERMS == 3) {
        unsigned t0, t1, t2;
        split3_pair(v, 0.f, t0, t1, t2);
        dst[0] = (unsigned short)t0; dst[wpl] = (unsigned short)t1; dst[2 * wpl] = (unsigned short)t2;      // pads are zeroed by the launcher
    } else {
        unsigned t0, t1;
        split2_pair_f16(ldexpf(v, pow2_scale_exp(*w_absmax, 13)), 0.f, t0, t1);
        dst[0] = (unsigned short)t0; dst[wpl] = (unsigned short)t1;
    }
}

#ifndef C4_MH
#define C4_MH 2          // m-tiles per pass of a k-step (2: half the A-fragment registers, B fragments read twice)
#endif
// two-term fp16 variant of c4_multiply: three products per k-step, smallest first.
//   * product-major order: each product runs over ALL MT x NT accumulators before the next product touches them again (an MFMA that
//     accumulates into the register its predecessor-but-one wrote would stall on the 8-pass latency of v_mfma_f32_16x16x32);
//   * the fragment reads are software-pipelined: the 2 (MT + NT) ds_read_b128 of k-step s+1 are issued BETWEEN the MFMAs of k-step s
//     (sched_group_barrier: one read, then three MFMAs, ...), into a second register set.  hipcc's own schedule issued every read right
//     before its first use (s_waitcnt lgkmcnt straight after the read, three to five times per k-step): with two waves per SIMD the
//     ~130-cycle LDS latency was exposed each time -- the multiply phase alone ran at 35 % matrix-pipe utilisation.
// MT = m-tiles computed (2: the last column tile of F = 480 holds 32 valid columns).
#ifndef C2_PIPE
#define C2_PIPE 1
#endif
template <int MT, int NT>
struct C2Frag { f16x8 a[2][MT], b[2][NT]; };
template <int MT, int NT, int COUT>
__device__ __forceinline__ void c2_load(C2Frag<MT, NT>& f, const unsigned char* __restrict__ lin, const unsigned char* __restrict__ lw, int aoff, int boff, int s) {
    constexpr int WPL = c4_wpl(COUT);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int i = 0; i < MT; ++i) f.a[sp][i] = *reinterpret_cast<const f16x8*>(lin + sp * C4_INPL + aoff + i * 16 * C4_PSTR);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int j = 0; j < NT; ++j) f.b[sp][j] = *reinterpret_cast<const f16x8*>(lw + sp * WPL + boff + s * 2 * COUT * 16 + j * 256);
}
template <int MT, int NT, bool PIN>
__device__ __forceinline__ void c2_products(const C2Frag<MT, NT>& f, f32x4 (&acc)[4][NT]) {
#define C4_PRODUCT(SA, SB)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                                           \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[SA][i], f.b[SB][j], acc[i][j], 0, 0, 0);
    // PIN (the last k-step, nothing to interleave): keep the three products apart -- left alone, hipcc groups the three MFMAs of each
    // accumulator back to back (a dependent chain: three times the 8-pass latency per accumulator)
    C4_PRODUCT(1, 0)
    if (PIN) __builtin_amdgcn_sched_barrier(0);
    C4_PRODUCT(0, 1)
    if (PIN) __builtin_amdgcn_sched_barrier(0);
    C4_PRODUCT(0, 0)
    if (PIN) __builtin_amdgcn_sched_barrier(0);
#undef C4_PRODUCT
}
template <int KS, int MT, int NT, int COUT>
__device__ __forceinline__ void c4_multiply_f16x2(const unsigned char* __restrict__ lin, const unsigned char* __restrict__ lw,
                                                  const int (&aoff)[KS], int boff, f32x4 (&acc)[4][NT]) {
    constexpr int NREAD = 2 * (MT + NT), NMFMA = 3 * MT * NT, PER = NMFMA / NREAD;        // MFMAs between two reads
#if !C2_PIPE
    // one register set (3 workgroups per CU: 168 registers): the other two waves of the SIMD cover the read latency
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        C2Frag<MT, NT> g;
        c2_load<MT, NT, COUT>(g, lin, lw, aoff[s], boff, s);
        c2_products<MT, NT, true>(g, acc);
    }
    return;
#endif
    C2Frag<MT, NT> f[2];
    c2_load<MT, NT, COUT>(f[0], lin, lw, aoff[0], boff, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) c2_load<MT, NT, COUT>(f[(s + 1) & 1], lin, lw, aoff[s + 1], boff, s + 1);
        if (s + 1 < KS) c2_products<MT, NT, false>(f[s & 1], acc);
        else { __builtin_amdgcn_sched_barrier(0); c2_products<MT, NT, true>(f[s & 1], acc); }
        if (s + 1 < KS) {
#pragma unroll
            for (int r = 0; r < NREAD; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  // one DS read
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                // PER MFMAs
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - PER * NREAD, 0);
        }
    }
}

template <int KS, int NT, int COUT>
__device__ __forceinline__ void c4_multiply(const unsigned char* __restrict__ lin, const unsigned char* __restrict__ lw,
                                            const int (&aoff)[KS], int boff, f32x4 (&acc)[4][NT], int nh) {
    constexpr int WPL = c4_wpl(COUT);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int h = 0; h < 4 / C4_MH; ++h) {
            if (h >= nh) break;                     // uniform: the right half of the last column tile of F = 480 lies outside the row
            bf16x8 av[3][C4_MH];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                for (int i = 0; i < C4_MH; ++i) av[sp][i] = *reinterpret_cast<const bf16x8*>(lin + sp * C4_INPL + aoff[s] + (C4_MH * h + i) * 16 * C4_PSTR);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 bv[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) bv[sp] = *reinterpret_cast<const bf16x8*>(lw + sp * WPL + boff + s * 2 * COUT * 16 + j * 256);
#if C4_FRESH
                // The six term products of this k-step (32 k values) are summed in a FRESH accumulator, smallest terms first, and added
                // to the running one once: the large accumulator is rounded once per 32 k (the fp32-input MFMA path rounds it 8 times),
                // and the 2^-8 / 2^-16 terms are never rounded against it.
                f32x4 t[C4_MH];
#pragma unroll
                for (int i = 0; i < C4_MH; ++i) t[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2][i], bv[0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#define C4_PRODUCT(SA, SB)                                                                                                   \
                _Pragma("unroll") for (int i = 0; i < C4_MH; ++i)                                                            \
                    t[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[SA][i], bv[SB], t[i], 0, 0, 0);
                C4_PRODUCT(1, 1) C4_PRODUCT(0, 2) C4_PRODUCT(1, 0) C4_PRODUCT(0, 1) C4_PRODUCT(0, 0)
#undef C4_PRODUCT
#pragma unroll
                for (int i = 0; i < C4_MH; ++i) acc[C4_MH * h + i][j] = (s & 1) ? acc[C4_MH * h + i][j] - t[i] : acc[C4_MH * h + i][j] + t[i];
#else
                // smallest terms first
#define C4_PRODUCT(SA, SB)                                                                                                   \
                _Pragma("unroll") for (int i = 0; i < C4_MH; ++i)                                                            \
                    acc[C4_MH * h + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[SA][i], bv[SB], acc[C4_MH * h + i][j], 0, 0, 0);
                C4_PRODUCT(2, 0) C4_PRODUCT(1, 1) C4_PRODUCT(0, 2) C4_PRODUCT(1, 0) C4_PRODUCT(0, 1) C4_PRODUCT(0, 0)
#undef C4_PRODUCT
#endif
            }
        }
    }
}

// TERMS = 3: every fp32 operand as three bf16 terms, six products (conv3x3_bf16x3 of round 1).  TERMS = 2: two fp16 terms, THREE
// products -- half the matrix-pipe work and two thirds of the LDS traffic for the same fp32-level result (a2s_common.h:
// split2_pair_f16); fp16's narrow exponent range is handled by exact power-of-two scales: the weights by 2^(13 - exponent of max|w|)
// (pack kernel), a gradient operand by 2^(12 - exponent of max|x|) (x_absmax, written by the kernel that produced it), activations
// unscaled; the accumulators are multiplied by the inverse power of two in the epilogue.
#ifdef C4_TRACE
// timing instrumentation (tools/conv_trace.py): shader-clock stamps of wave 0 of a few mid-grid workgroups, 8 stamps per stage
#define C4_TRACE_WGS 8
#define C4_TRACE_STAGES 24
__device__ unsigned long long c4_trace[C4_TRACE_WGS * C4_TRACE_STAGES * 8];
extern "C" int a2s_conv_trace_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c4_trace), sizeof(c4_trace));
}
#define C4_STAMP(k)                                                                                                           \
    do {                                                                                                                      \
        if (trace_wg >= 0 && q < C4_TRACE_STAGES && tid == 0) c4_trace[(trace_wg * C4_TRACE_STAGES + q) * 8 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define C4_STAMP(k) do {} while (0)
#endif
#ifndef C4_XCD_SWIZZLE
#define C4_XCD_SWIZZLE 1
#endif
#ifndef C4_WGS
#define C4_WGS 2          // workgroups per CU the two-term kernel is compiled for (3: 168 registers, 52.9 KB of LDS)
#endif
template <int COUT, bool BNRED, int TERMS>
__global__ __launch_bounds__(256, (TERMS == 2 ? C4_WGS : 2)) void conv3x3_split(ConvArgs a, const unsigned char* __restrict__ wpack) {
    constexpr int NT = (COUT + 15) / 16;
    constexpr int WPL = c4_wpl(COUT);                    // bytes per term plane of a packed weight chunk
    constexpr int WCH = (TERMS * WPL + 4095) / 4096 * 4096;  // chunk stride of the packed image: whole 4 x 1 KB LDS-DMA rounds
    // LDS copy of a chunk: whole 1 KB DMA pieces.  The B fragments of the last n-tile read up to 7 rows past a slot (channels 40..47): past the
    // last slot that is beyond this array -- whatever lies there only reaches accumulator columns >= COUT, which are never stored.
    constexpr int WLDS = (TERMS * WPL + 1023) / 1024 * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char lin[TERMS * C4_INPL];      // 38016 B (3 terms)
    __shared__ __attribute__((aligned(16))) unsigned char lw[WLDS];
    __shared__ float red[4][NT * 16][2];
    __shared__ __attribute__((aligned(16))) float lsc[48], lsh[48];                   // producer's BatchNorm scale / shift (0 beyond Cin)

    const int tilesF = (a.F + C2_FT - 1) / C2_FT;
    const int tilesT = (a.T + CV_TR - 1) / CV_TR;
    const int groupsT = (tilesT + C3_TPW - 1) / C3_TPW;
    // XCD-aware order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs (id % 8), each with its own L2.  Logical
    // tile L = (id % 8) * (n / 8) + id / 8 gives every XCD a contiguous run of tiles, so the F-tiles of a row group -- which share halo
    // columns and together read each 1920-byte activation row in full -- are resident on ONE XCD at the same time.
    int bid = blockIdx.x;
#if C4_XCD_SWIZZLE
    {
        const int per = (int)gridDim.x / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
#endif
    const int ft = bid % tilesF; bid /= tilesF;
    const int tg = bid % groupsT; const int b = bid / groupsT;
    const int f0 = ft * C2_FT;
    const int tile0 = tg * C3_TPW, ntile = min(C3_TPW, tilesT - tile0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int ncgs = (a.Cin + 7) / 8, nchunks = (ncgs + 1) / 2;
    const int nstage = ntile * nchunks;

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[NT], st_s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { st_s[j] = 0.f; st_s2[j] = 0.f; }
    if (tid < 48) {
        lsc[tid] = (a.in_scale && tid < a.Cin) ? a.in_scale[tid] : 0.f;
        lsh[tid] = (a.in_scale && tid < a.Cin) ? a.in_shift[tid] : 0.f;
    }
    float xscale = 1.f, unscale = 1.f;                   // TERMS == 2: exact power-of-two operand scale and its inverse (times the weights')
    if (TERMS == 2) {
        const int kx = a.x_absmax ? pow2_scale_exp(*a.x_absmax, 12) : 0;
        const int kw = pow2_scale_exp(*a.w_absmax, 13);
        xscale = ldexpf(1.f, kx);
        unscale = ldexpf(1.f, -(kx + kw));
    }

    // per-lane fragment offsets: pair p = 4 s + lk of the chunk (clamped: the padded slots carry zero weights)
    int aoff2[5], aoff1[3];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int p = min(4 * s + lk, 17), tap = p >> 1;
        aoff2[s] = ((wave + tap / 3) * C4_POS + tap % 3 + li) * C4_PSTR + (p & 1) * 16;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = min(4 * s + lk, 8);
        aoff1[s] = ((wave + tap / 3) * C4_POS + tap % 3 + li) * C4_PSTR;
    }
    const int boff = (lk & 1) * c4_odd_base(COUT) + ((lk >> 1) * COUT + li) * 16;

    auto stage_tile = [&](int q) { return q / nchunks; };
    auto stage_chunk = [&](int q) { const int i = q / nchunks, c = q % nchunks; return (i & 1) ? nchunks - 1 - c : c; };

    float xreg[C4_XIT][8];
    // packed weight chunk -> LDS by LDS-DMA (global_load_lds_dwordx4: a wave copies 1 KB, lane-linear, no registers, no ds_write):
    // issued right after the barrier that retires the previous stage, in flight under the commit's BatchNorm / split arithmetic
    auto load_weights = [&](int ch) {
        const unsigned char* wsrc = wpack + (long)ch * WCH;
#pragma unroll
        for (int it = 0; it < (WLDS / 1024 + 3) / 4; ++it) {
            const int off = (it * 4 + wave) * 1024;
            if (off < WLDS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + off + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(lw + off), 16, 0, 0);
        }
    };
    // Staging items of this thread, decoded ONCE: (row, position, group) packed in one register + the element offset relative to
    // the stage's (t0, first channel, f0) corner.  Inside the stage loop only "laundered" copies are used, so that nothing derived
    // from them is hoisted (the compiler otherwise keeps ~100 loop-invariant addresses in registers and spills around the multiply).
    int it_desc[C4_XIT], it_goff[C4_XIT];
#pragma unroll
    for (int it = 0; it < C4_XIT; ++it) {
        const int e = tid + 256 * it;
        const int cgl = e / C4_ITEMS, rp = e % C4_ITEMS;
        const int r = rp / C4_POS, pos = rp % C4_POS;
        it_desc[it] = r | (pos << 4) | (cgl << 12);
        it_goff[it] = ((r - 1) * a.Cin + cgl * 8) * a.F + pos - 1;
    }
    const int clip_elems = a.T * a.Cin * a.F;
    const float* __restrict__ xclip = a.x + (long)b * clip_elems;
    // raw buffer over this clip's input: a load whose byte offset falls outside [0, 4 clip_elems) returns 0 -- rows above / below the
    // clip, columns left of f = 0 on the first row -- so the addresses need no clamping (the clamped-index version spent ~6 VALU
    // instructions per scalar load, a third of the staging arithmetic); everything outside the tile's valid region is masked at commit
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xclip), 0, clip_elems * 4, 0x00020000);
    auto issue = [&](int q) {
        const int ch = stage_chunk(q), t0 = (tile0 + stage_tile(q)) * CV_TR;
        const int ncg = min(2, ncgs - 2 * ch);
        const int corner = (t0 * a.Cin + ch * 16) * a.F + f0;
#pragma unroll
        for (int it = 0; it < C4_XIT; ++it) {
            if (256 * it >= ncg * C4_ITEMS) break;                 // uniform: the one-group chunk has 396 items
            int g = it_goff[it];
            asm volatile("" : "+v"(g));
            g = (g + corner) * 4;
            const int plane = a.F * 4;
#pragma unroll
            for (int k = 0; k < 8; ++k) xreg[it][k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, g + k * plane, 0, 0));
        }
    };
    auto commit = [&](int q) {     // registers -> LDS; BN+ReLU of the producer folded in; zero = padding (of the ACTIVATED tensor)
        const int ch = stage_chunk(q), t0 = (tile0 + stage_tile(q)) * CV_TR;
        const int ncg = min(2, ncgs - 2 * ch);
        const unsigned sgn = (!C4_FRESH && (q & 1)) ? 0x80000000u : 0u;
#pragma unroll
        for (int it = 0; it < C4_XIT; ++it) {
            if (256 * it >= ncg * C4_ITEMS) break;
            int d = it_desc[it];
            asm volatile("" : "+v"(d));
            const int r = d & 15, pos = (d >> 4) & 255, cgl = d >> 12, rp = r * C4_POS + pos;
            if (cgl >= ncg) continue;
            const int t = t0 + r - 1, f = f0 + pos - 1, c0 = (ch * 2 + cgl) * 8;
            const bool ok = t >= 0 && t < a.T && f >= 0 && f < a.F;
            float v[8];
            if (a.in_scale) {                                     // lsc / lsh are zero beyond Cin: padded channels come out as 0
                const f32x4 sc0 = *reinterpret_cast<const f32x4*>(lsc + c0), sc1 = *reinterpret_cast<const f32x4*>(lsc + c0 + 4);
                const f32x4 sh0 = *reinterpret_cast<const f32x4*>(lsh + c0), sh1 = *reinterpret_cast<const f32x4*>(lsh + c0 + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = ok ? fmaxf(xreg[it][k] * sc0[k] + sh0[k], 0.f) : 0.f;
                    v[4 + k] = ok ? fmaxf(xreg[it][4 + k] * sc1[k] + sh1[k], 0.f) : 0.f;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (ok && c0 + k < a.Cin) ? xreg[it][k] : 0.f;
            }
            uint4 o[TERMS];
            if (TERMS == 3) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __uint_as_float(__float_as_uint(v[k]) ^ sgn);      // odd stages multiply -x (see the stage loop)
                split3_pair(v[0], v[1], o[0].x, o[1].x, o[TERMS - 1].x);
                split3_pair(v[2], v[3], o[0].y, o[1].y, o[TERMS - 1].y);
                split3_pair(v[4], v[5], o[0].z, o[1].z, o[TERMS - 1].z);
                split3_pair(v[6], v[7], o[0].w, o[1].w, o[TERMS - 1].w);
            } else {
                // the odd stages' sign rides on the power-of-two scale; activations (unscaled) are clamped to fp16's range in one
                // v_med3 (it only bites on absurd values), a gradient operand is already inside it by construction of its scale
                const float xs = sgn ? -xscale : xscale;
                if (a.in_scale) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_fmed3f(v[k] * xs, -65000.f, 65000.f);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] *= xs;
                }
                split2_pair_f16(v[0], v[1], o[0].x, o[1].x);
                split2_pair_f16(v[2], v[3], o[0].y, o[1].y);
                split2_pair_f16(v[4], v[5], o[0].z, o[1].z);
                split2_pair_f16(v[6], v[7], o[0].w, o[1].w);
            }
#pragma unroll
            for (int sp = 0; sp < TERMS; ++sp) *reinterpret_cast<uint4*>(lin + sp * C4_INPL + rp * C4_PSTR + cgl * 16) = o[sp];
        }
    };

    int resident = -1;
    bool acc_neg = false;
    const int nh = (f0 + C4_MH * 16 >= a.F) ? 1 : 4 / C4_MH;      // m-tile passes that hold any valid column (F = 480: 7.5 tiles of 64)
    const bool half_tile = f0 + 32 >= a.F;                        // two-term kernel: two m-tiles instead of four
    auto flip_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = -acc[i][j][r];
    };
#ifndef C4_PREFETCH
#define C4_PREFETCH 1     // issue the next stage's input loads before this stage's multiply (with three products per k-step the multiply no
#endif                    // longer covers the other workgroup's load round trip; round 1, six products: no gain)
#ifdef C4_TRACE
    const int trace_first = (int)gridDim.x / 2;
    const int trace_wg = ((int)blockIdx.x >= trace_first && (int)blockIdx.x < trace_first + C4_TRACE_WGS) ? (int)blockIdx.x - trace_first : -1;
#endif
    if (C4_PREFETCH && nstage > 0) issue(0);
    for (int q = 0; q < nstage; ++q) {
        C4_STAMP(0);
        __syncthreads();                      // previous stage fully consumed
        C4_STAMP(1);
        if (stage_chunk(q) != resident) load_weights(stage_chunk(q));
        if (!C4_PREFETCH) issue(q);
        commit(q);
        resident = stage_chunk(q);
        C4_STAMP(2);
        __syncthreads();
        C4_STAMP(3);
        if (C4_PREFETCH && q + 1 < nstage) issue(q + 1);
        C4_STAMP(4);
        // The bf16 matrix pipe truncates its internal sum toward -infinity (measured: mean error -3e-9 sum|a||b|, always negative, against
        // a random part of 5e-8) -- nothing for one output, but the BatchNorm sums over 10^6 positions see 70x their random error.  Odd
        // stages therefore accumulate the NEGATED sum (input negated while staging, accumulators flipped): the truncation then pushes
        // the value the other way, and a tile's stages / neighbouring tiles cancel.
        if (!C4_FRESH && acc_neg != bool(q & 1)) { flip_acc(); acc_neg = !acc_neg; }
        if (TERMS == 3) {
            if (ncgs - 2 * resident >= 2) c4_multiply<5, NT, COUT>(lin, lw, aoff2, boff, acc, nh);
            else c4_multiply<3, NT, COUT>(lin, lw, aoff1, boff, acc, nh);
        } else {
            if (half_tile) {
                if (ncgs - 2 * resident >= 2) c4_multiply_f16x2<5, 2, NT, COUT>(lin, lw, aoff2, boff, acc);
                else c4_multiply_f16x2<3, 2, NT, COUT>(lin, lw, aoff1, boff, acc);
            } else {
                if (ncgs - 2 * resident >= 2) c4_multiply_f16x2<5, 4, NT, COUT>(lin, lw, aoff2, boff, acc);
                else c4_multiply_f16x2<3, 4, NT, COUT>(lin, lw, aoff1, boff, acc);
            }
        }
        C4_STAMP(5);
        if ((q + 1) % nchunks != 0) continue;
        if (acc_neg) { flip_acc(); acc_neg = false; }
        if (TERMS == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] *= unscale;
        }
        // ---- tile epilogue.  C/D map: lane holds column n = li (channel), rows lk*4+r (f positions) of each m-tile.
        const int t = (tile0 + stage_tile(q)) * CV_TR + wave;
        const bool row_ok = t < a.T;
        int li_ = li, lk_ = lk;
        asm volatile("" : "+v"(li_), "+v"(lk_));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = j * 16 + li_;
            float bm = 0.f, bi = 0.f, bsc = 0.f, bsh = 0.f;
            f32x4 yv[4];
            if (BNRED && row_ok && co < COUT) {
                bm = a.yl_mean[co]; bi = a.yl_invstd[co]; bsc = a.yl_scale[co]; bsh = a.yl_shift[co];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = f0 + i * 16 + lk_ * 4;
                    const float* src = a.yl + (((long)b * a.T + t) * COUT + co) * a.F + f;
                    if (f + 3 < a.F && (a.F % 4 == 0)) yv[i] = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) yv[i][r] = (f + r < a.F) ? src[r] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = f0 + i * 16 + lk_ * 4;
                if (row_ok && co < COUT) {
                    float* dst = a.y + (((long)b * a.T + t) * COUT + co) * a.F + f;
                    const bool full = f + 3 < a.F && (a.F % 4 == 0);
                    if (full) *reinterpret_cast<f32x4*>(dst) = acc[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (!full && f + r >= a.F) continue;
                        const float v = acc[i][j][r];
                        if (!full) dst[r] = v;
                        if (BNRED) {
                            const float xv = yv[i][r];
                            const float gm = (xv * bsc + bsh > 0.f) ? v : 0.f;
                            st_s[j] += gm; st_s2[j] += gm * (xv - bm) * bi;
                        } else { st_s[j] += v; st_s2[j] += v * v; }
                    }
                }
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_sched_barrier(0);        // one n-tile at a time: hoisting every yl load spills the prefetched input
        }
        C4_STAMP(6);
    }
    if (a.stat_partial) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = st_s[j], s2 = st_s2[j];
            s += __shfl_xor(s, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s += __shfl_xor(s, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) { red[wave][j * 16 + li][0] = s; red[wave][j * 16 + li][1] = s2; }
        }
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

// First layer: Cin = 1.  A streaming kernel (53 GFLOP at B = 256 against 12 GB written): a thread owns 4 consecutive f positions of a
// row (16-byte stores of every output channel), walks the rows grid-stride, and keeps its per-channel sum / sum of squares in
// registers -- the batch-statistics partials are reduced ONCE per workgroup at the end (the first version reduced 40 values across the
// wave for every position and wrote 92 MB of partials).  C1_BLOCKS workgroups = that many partial rows.
#define C1_BLOCKS 2048
__global__ __launch_bounds__(256) void conv3x3_c1(ConvArgs a) {
    __shared__ float lw[20 * 9];
    __shared__ float red[4][20][2];
    for (int e = threadIdx.x; e < a.Cout * 9; e += 256) lw[e] = a.w[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qpr = (a.F + 3) / 4;                               // quads per row
    const long nquads = (long)a.B * a.T * qpr;
    const bool vec = (a.F % 4 == 0) && (((uintptr_t)a.y & 15) == 0);
    float s1[20], s2[20], m1[20];
#pragma unroll
    for (int co = 0; co < 20; ++co) { s1[co] = 0.f; s2[co] = 0.f; m1[co] = 0.f; }
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long)gridDim.x * 256) {
        const int f0 = (int)(q % qpr) * 4;
        const long row = q / qpr;                                // b * T + t
        const int t = (int)(row % a.T);
        float v[3][6];                                           // input window: rows t-1..t+1, columns f0-1..f0+4
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int tt = t + dt - 1;
            const float* xr = a.x + (row + dt - 1) * a.F;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int ff = f0 + c - 1;
                v[dt][c] = (tt >= 0 && tt < a.T && ff >= 0 && ff < a.F) ? xr[ff] : 0.f;
            }
        }
#pragma unroll
        for (int co = 0; co < 20; ++co) {
            if (co >= a.Cout) break;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    const float w = lw[co * 9 + dt * 3 + df];
#pragma unroll
                    for (int p = 0; p < 4; ++p) o[p] = fmaf(v[dt][p + df], w, o[p]);
                }
            float* dst = a.y + (row * a.Cout + co) * a.F + f0;
            if (vec) {
                *reinterpret_cast<f32x4*>(dst) = o;
#pragma unroll
                for (int p = 0; p < 4; ++p) { s1[co] += o[p]; s2[co] = fmaf(o[p], o[p], s2[co]); }
                m1[co] = fmaxf(fmaxf(m1[co], fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) if (f0 + p < a.F) { dst[p] = o[p]; s1[co] += o[p]; s2[co] = fmaf(o[p], o[p], s2[co]); m1[co] = fmaxf(m1[co], fabsf(o[p])); }
            }
        }
    }
    if (a.out_absmax) {          // per-channel max |y|: the next layer derives its operand scale from it (a2s_conv_rows.hip)
#pragma unroll
        for (int co = 0; co < 20; ++co) {
            if (co >= a.Cout) break;
            const float m = wave_max(m1[co]);
            if (lane == 0) {
                const unsigned bits = __float_as_uint(m);
                if (bits > __hip_atomic_load(reinterpret_cast<unsigned*>(a.out_absmax + co), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    atomicMax(reinterpret_cast<unsigned*>(a.out_absmax + co), bits);
            }
        }
    }
    if (a.stat_partial) {
#pragma unroll
        for (int co = 0; co < 20; ++co) {
            if (co >= a.Cout) break;
            const float w1 = wave_sum(s1[co]), w2 = wave_sum(s2[co]);
            if (lane == 0) { red[wave][co][0] = w1; red[wave][co][1] = w2; }
        }
        __syncthreads();
        if (threadIdx.x < a.Cout) {
            float s = 0.f, q2 = 0.f;
            for (int w = 0; w < 4; ++w) { s += red[w][threadIdx.x][0]; q2 += red[w][threadIdx.x][1]; }
            a.stat_partial[((long)blockIdx.x * a.Cout + threadIdx.x) * 2 + 0] = s;
            a.stat_partial[((long)blockIdx.x * a.Cout + threadIdx.x) * 2 + 1] = q2;
        }
    }
}

// The same kernel with the channel count fixed at compile time and F a multiple of 4 (the model's 20 x 480): with a run-time Cout the
// per-channel accumulators above are indexed dynamically (s_set_gpr_idx and ~50 register moves per channel: 3070 VALU instructions per
// quad, the kernel ran at the VALU's pace, 4.4 ms at B = 256 against 2.1 ms of HBM time).  Here everything unrolls: one aligned 16-byte
// load and two edge scalars per window row, packed fp32 FMAs, the same order of operations per output -- results are bit-identical.
template <int COUT>
__global__ __launch_bounds__(256, 4) void conv3x3_c1_fixed(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float lw[COUT * 12];             // 9 taps per channel, padded to 12 for 16-byte reads
    __shared__ float red[4][COUT][2];
    for (int e = threadIdx.x; e < COUT * 12; e += 256) lw[e] = (e % 12 < 9) ? a.w[(e / 12) * 9 + e % 12] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qpr = a.F / 4;
    const long nquads = (long)a.B * a.T * qpr;
    float s1[COUT], s2[COUT], m1[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) { s1[co] = 0.f; s2[co] = 0.f; m1[co] = 0.f; }
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long)gridDim.x * 256) {
        const int f0 = (int)(q % qpr) * 4;
        const long row = q / qpr;
        const int t = (int)(row % a.T);
        float v[3][6];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int tt = t + dt - 1;
            const bool in = tt >= 0 && tt < a.T;
            const float* xr = a.x + (row + dt - 1) * a.F + f0;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            float l = 0.f, r = 0.f;
            if (in) {
                c = *reinterpret_cast<const f32x4*>(xr);
                if (f0 > 0) l = xr[-1];
                if (f0 + 4 < a.F) r = xr[4];
            }
            v[dt][0] = l; v[dt][1] = c[0]; v[dt][2] = c[1]; v[dt][3] = c[2]; v[dt][4] = c[3]; v[dt][5] = r;
        }
        float* dst = a.y + row * COUT * a.F + f0;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            asm volatile("" ::: "memory");            // the taps are re-read from LDS per channel: hoisted out of the loop they cost 180 registers
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(&lw[co * 12]), w1 = *reinterpret_cast<const f32x4*>(&lw[co * 12 + 4]);
            const float w8 = lw[co * 12 + 8];
            const float w[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w8};
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int df = 0; df < 3; ++df)
#pragma unroll
                    for (int p = 0; p < 4; ++p) o[p] = fmaf(v[dt][p + df], w[dt * 3 + df], o[p]);
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dst + (long)co * a.F));
#pragma unroll
            for (int p = 0; p < 4; ++p) { s1[co] += o[p]; s2[co] = fmaf(o[p], o[p], s2[co]); }
            m1[co] = fmaxf(fmaxf(m1[co], fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        }
    }
    if (a.out_absmax) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float m = wave_max(m1[co]);
            if (lane == 0) {
                const unsigned bits = __float_as_uint(m);
                if (bits > __hip_atomic_load(reinterpret_cast<unsigned*>(a.out_absmax + co), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    atomicMax(reinterpret_cast<unsigned*>(a.out_absmax + co), bits);
            }
        }
    }
    if (a.stat_partial) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float w1 = wave_sum(s1[co]), w2 = wave_sum(s2[co]);
            if (lane == 0) { red[wave][co][0] = w1; red[wave][co][1] = w2; }
        }
        __syncthreads();
        if (threadIdx.x < COUT) {
            float s = 0.f, q2 = 0.f;
            for (int w = 0; w < 4; ++w) { s += red[w][threadIdx.x][0]; q2 += red[w][threadIdx.x][1]; }
            a.stat_partial[((long)blockIdx.x * COUT + threadIdx.x) * 2 + 0] = s;
            a.stat_partial[((long)blockIdx.x * COUT + threadIdx.x) * 2 + 1] = q2;
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

// out[0] = max_c (|scale_c| absmax_c + |shift_c|): a hard bound of relu(x scale_c + shift_c) over the tensor, given the per-channel
// max |x_c| its producer wrote (a2s_conv3x3_ranged) -- the operand range of the kernels that read the activated tensor on the fly
__global__ void act_bound_kernel(const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ absmax, int C, float* __restrict__ out) {
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) m = fmaxf(m, fabsf(scale[c]) * absmax[c] + fabsf(shift[c]));
    m = wave_max(m);
    if (threadIdx.x == 0) *out = m;
}
int a2s_act_bound_impl(hipStream_t st, const float* scale, const float* shift, const float* absmax, int C, float* out) {
    A2S_REQUIRE(scale && shift && absmax && out && C > 0, "act_bound: null tensor");
    hipLaunchKernelGGL(act_bound_kernel, dim3(1), dim3(64), 0, st, scale, shift, absmax, C, out);
    A2S_CHECK_LAUNCH("act_bound");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- launchers
static int g_conv_c1_fast = 1;        // the first layer's compile-time-shaped kernels (conv3x3_c1_fixed, conv3x3_wgrad_c1_stream); 0 = the generic ones (tests: bit-equal)
void a2s_conv_c1_fast_set(int on) { g_conv_c1_fast = on; }
int a2s_conv_c1_fast_enabled(void) { return g_conv_c1_fast; }
static int g_conv_bf16x3 = 3;         // convolutions on the bf16 matrix pipes with 3-term split operands (conv3x3_split<.., 3>): bit 0 forward, bit 1 data-gradient launches
void a2s_conv_bf16x3_set(int on) { g_conv_bf16x3 = on; }
int a2s_conv_bf16x3_enabled(void) { return g_conv_bf16x3; }
// ... and of those, which use the TWO-term fp16 split instead (conv3x3_split<.., 2>: three products instead of six); a data-gradient
// launch additionally needs the max |dy| scalar of its operand (a2s_conv3x3_dgrad_bnstats_scaled) and stays on three terms without it
static int g_conv_f16x2 = -1;
void a2s_conv_f16x2_set(int on) { g_conv_f16x2 = on; }
int a2s_conv_f16x2_enabled(void) {
    if (g_conv_f16x2 < 0) g_conv_f16x2 = 3;
    return g_conv_f16x2;
}

// the row-streaming kernel of a2s_conv_rows.hip (forward and data-gradient launches with F % 4 == 0, 20 / 40 channels)
bool a2s_conv_rows_eligible(int F, int Cin);
int a2s_conv_rows_blocks(int B, int T, int F);
size_t a2s_conv_rows_workspace_floats(int Cin);
int a2s_channel_absmax_impl(hipStream_t, const float*, long, int, int, float*);
int a2s_conv3x3_rows_impl(hipStream_t, const float*, const float*, float*, const float*, const float*, const float*, float*, float*, int, int, int, int, int, int,
                          float*, const float*, const float*, const float*, const float*, const float*, const float*);

size_t a2s_conv3x3_workspace_floats_impl(int Cin) {
    if (Cin == 1) return 0;
    const size_t f32_image = (size_t)(Cin / CV_CK) * C2_WCHUNK, split_image = (size_t)c4_chunks(Cin) * c4_chunk_bytes(40) / 4;
    const size_t tiled = (f32_image > split_image ? f32_image : split_image) + 4;          // + the max |w| scalar of the two-term path
    const size_t rows = a2s_conv_rows_workspace_floats(Cin);
    return tiled > rows ? tiled : rows;
}

int a2s_conv3x3_stat_blocks_impl(int B, int T, int F, int Cin);

int a2s_conv3x3_impl(hipStream_t st, const float* x, const float* w, float* y, const float* in_scale,
                     const float* in_shift, float* stat_partial, int B, int T, int F, int Cin, int Cout, int flip, float* ws,
                     const float* yl, const float* yl_mean, const float* yl_invstd, const float* yl_scale, const float* yl_shift,
                     const float* x_absmax, const float* in_absmax, float* out_absmax) {
    A2S_REQUIRE(x && w && y, "conv3x3: null tensor");
    A2S_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3x3: scale/shift must come together");
    A2S_REQUIRE(!yl || (yl_mean && yl_invstd && yl_scale && yl_shift && stat_partial && Cin != 1), "conv3x3: the fused BatchNorm-backward statistics need all of their tensors");
    ConvArgs a{x, w, y, in_scale, in_shift, stat_partial, B, T, F, Cin, Cout, flip, yl, yl_mean, yl_invstd, yl_scale, yl_shift, x_absmax, nullptr, out_absmax};
    if (Cin != 1 && a2s_conv_rows_eligible(F, Cin) && (Cout == 20 || Cout == 40)) {
        A2S_REQUIRE(ws, "conv3x3: needs a workspace of a2s_conv3x3_workspace_floats(Cin) floats for the packed weights");
        return a2s_conv3x3_rows_impl(st, x, w, y, in_scale, in_shift, in_absmax, stat_partial, out_absmax, B, T, F, Cin, Cout, flip, ws,
                                     yl, yl_mean, yl_invstd, yl_scale, yl_shift, x_absmax);
    }
    if (out_absmax && Cin == 1) {       // accumulated by atomic max in the first-layer kernel
        const hipError_t me = hipMemsetAsync(out_absmax, 0, sizeof(float) * Cout, st);
        A2S_REQUIRE(me == hipSuccess, "conv3x3: hipMemsetAsync(out_absmax): %s", hipGetErrorString(me));
    }
    if (out_absmax && Cin != 1) {       // the tiled kernels do not track it (A/B switch, odd shapes): one extra pass over y afterwards
        const int rc = a2s_conv3x3_impl(st, x, w, y, in_scale, in_shift, stat_partial, B, T, F, Cin, Cout, flip, ws, yl, yl_mean, yl_invstd, yl_scale, yl_shift,
                                        x_absmax, in_absmax, nullptr);
        return rc != A2S_OK ? rc : a2s_channel_absmax_impl(st, y, (long)B * T, Cout, F, out_absmax);
    }
    if (Cin == 1) {
        A2S_REQUIRE(Cout <= 20 && !flip && !in_scale, "conv3x3: Cin=1 path supports Cout<=20, no flip, no input affine");
        const bool fixed = g_conv_c1_fast && Cout == 20 && F % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0;
        if (fixed) hipLaunchKernelGGL(conv3x3_c1_fixed<20>, dim3(a2s_conv3x3_stat_blocks_impl(B, T, F, 1)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(conv3x3_c1, dim3(a2s_conv3x3_stat_blocks_impl(B, T, F, 1)), dim3(256), 0, st, a);
    } else {
        A2S_REQUIRE(Cin % 4 == 0 && Cin % CV_CK == 0, "conv3x3: Cin must be a multiple of %d", CV_CK);
        A2S_REQUIRE(ws, "conv3x3: needs a workspace of a2s_conv3x3_workspace_floats(Cin) floats for the packed weights");
        A2S_REQUIRE(Cout == 20 || Cout == 40, "conv3x3: Cout must be 20 or 40 (got %d)", Cout);
        const int nblk4 = B * a2s_cdiv(a2s_cdiv(T, CV_TR), C3_TPW) * a2s_cdiv(F, C2_FT);
        if (g_conv_bf16x3 & (flip ? 2 : 1)) {
            // two fp16 terms when enabled -- a gradient operand (flip) only with its max |x| scalar: fp16 has no exponent range to spare
            const bool two = (a2s_conv_f16x2_enabled() & (flip ? 2 : 1)) && (!flip || x_absmax);
            const int terms = two ? 2 : 3;
            const int n = c4_chunks(Cin) * C4_SLOTS * Cout * 8;
            const size_t image = c4_chunks(Cin) * c4_chunk_bytes(Cout, terms);
            const hipError_t me = hipMemsetAsync(ws, 0, image, st);
            A2S_REQUIRE(me == hipSuccess, "conv3x3: hipMemsetAsync(packed weights): %s", hipGetErrorString(me));
            float* wmax = ws + a2s_conv3x3_workspace_floats_impl(Cin) - 4;
            const unsigned char* wp = (const unsigned char*)ws;
            if (two) {
                hipLaunchKernelGGL(absmax_small, dim3(1), dim3(256), 0, st, w, Cout * Cin * 9, wmax);
                hipLaunchKernelGGL(conv_pack_weights_split<2>, dim3(a2s_cdiv(n, 256)), dim3(256), 0, st, w, (unsigned short*)ws, Cin, Cout, flip, (const float*)wmax);
                A2S_CHECK_LAUNCH("conv_pack_weights_split<2>");
                a.w_absmax = wmax;
                if (Cout == 20 && !yl) hipLaunchKernelGGL((conv3x3_split<20, false, 2>), dim3(nblk4), dim3(256), 0, st, a, wp);
                else if (Cout == 20) hipLaunchKernelGGL((conv3x3_split<20, true, 2>), dim3(nblk4), dim3(256), 0, st, a, wp);
                else if (!yl) hipLaunchKernelGGL((conv3x3_split<40, false, 2>), dim3(nblk4), dim3(256), 0, st, a, wp);
                else hipLaunchKernelGGL((conv3x3_split<40, true, 2>), dim3(nblk4), dim3(256), 0, st, a, wp);
                A2S_CHECK_LAUNCH("conv3x3_split<2>");
                return A2S_OK;
            }
            hipLaunchKernelGGL(conv_pack_weights_split<3>, dim3(a2s_cdiv(n, 256)), dim3(256), 0, st, w, (unsigned short*)ws, Cin, Cout, flip, (const float*)nullptr);
            A2S_CHECK_LAUNCH("conv_pack_weights_split<3>");
            if (Cout == 20 && !yl) hipLaunchKernelGGL((conv3x3_split<20, false, 3>), dim3(nblk4), dim3(256), 0, st, a, wp);
            else if (Cout == 20) hipLaunchKernelGGL((conv3x3_split<20, true, 3>), dim3(nblk4), dim3(256), 0, st, a, wp);
            else if (!yl) hipLaunchKernelGGL((conv3x3_split<40, false, 3>), dim3(nblk4), dim3(256), 0, st, a, wp);
            else hipLaunchKernelGGL((conv3x3_split<40, true, 3>), dim3(nblk4), dim3(256), 0, st, a, wp);
            A2S_CHECK_LAUNCH("conv3x3_split<3>");
            return A2S_OK;
        }
        const int chunks = Cin / CV_CK;
        hipLaunchKernelGGL(conv_pack_weights, dim3(a2s_cdiv(chunks * C2_WCHUNK, 256)), dim3(256), 0, st, w, ws, Cin, Cout, flip, chunks);
        A2S_CHECK_LAUNCH("conv_pack_weights");
        const int nblk = B * a2s_cdiv(a2s_cdiv(T, CV_TR), C3_TPW) * a2s_cdiv(F, C2_FT);
        if (Cout == 20 && !yl) hipLaunchKernelGGL((conv3x3_mfma<20, false>), dim3(nblk), dim3(256), 0, st, a, (const float*)ws);
        else if (Cout == 20) hipLaunchKernelGGL((conv3x3_mfma<20, true>), dim3(nblk), dim3(256), 0, st, a, (const float*)ws);
        else if (!yl) hipLaunchKernelGGL((conv3x3_mfma<40, false>), dim3(nblk), dim3(256), 0, st, a, (const float*)ws);
        else hipLaunchKernelGGL((conv3x3_mfma<40, true>), dim3(nblk), dim3(256), 0, st, a, (const float*)ws);
    }
    A2S_CHECK_LAUNCH("conv3x3");
    return A2S_OK;
}

int a2s_conv3x3_stat_blocks_impl(int B, int T, int F, int Cin) {
    if (Cin == 1) {       // the streaming first-layer kernel: at most C1_BLOCKS grid-stride workgroups
        const long want = a2s_cdiv((long)B * T * ((F + 3) / 4), 256);
        return (int)(want < C1_BLOCKS ? want : C1_BLOCKS);
    }
    if (a2s_conv_rows_eligible(F, Cin)) return a2s_conv_rows_blocks(B, T, F);
    return B * a2s_cdiv(a2s_cdiv(T, CV_TR), C3_TPW) * a2s_cdiv(F, C2_FT);
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

// =========================================================================================== backward
// BatchNorm (+ReLU, + optional dropout) backward in three steps, with y = x*scale + shift, xh = (x-mean)*invstd:
//   g' = g * [y > 0] (* keep*inv_keep);  s1 = sum g', s2 = sum g' xh  (per channel, fixed-order reduction);
//   dbeta += s1, dgamma += s2;  dx = gamma*invstd * (g' - s1/N - xh * s2/N).
// Plane layout (rows, C, F): block = one row, wave per channel plane.  Matrix layout (rows, C): see *_cols.
__global__ __launch_bounds__(256) void bn_bwd_reduce_planes(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ partial, int C, int F) {
    const long row = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < C; c += 4) {
        const float* gp = g + (row * C + c) * F;
        const float* xp = x + (row * C + c) * F;
        const float m = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
        float s1 = 0.f, s2 = 0.f;
        for (int f = lane; f < F; f += 64) {
            const float xv = xp[f];
            const float gv = (xv * sc + sh > 0.f) ? gp[f] : 0.f;
            s1 += gv; s2 += gv * (xv - m) * is;
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { partial[(row * C + c) * 2 + 0] = s1; partial[(row * C + c) * 2 + 1] = s2; }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_cols(const float* __restrict__ g, const float* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const uint8_t* __restrict__ mask, float inv_keep,
                                                          float* __restrict__ partial, long rows, int C, int rows_per_block) {
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
        float s1 = 0.f, s2 = 0.f;
        for (long r = r0; r < r1; ++r) {
            const float xv = x[r * C + c];
            float gv = (xv * sc + sh > 0.f) ? g[r * C + c] : 0.f;
            if (mask) gv = mask[r * C + c] ? gv * inv_keep : 0.f;
            s1 += gv; s2 += gv * (xv - m) * is;
        }
        partial[((long)blockIdx.x * C + c) * 2 + 0] = s1;
        partial[((long)blockIdx.x * C + c) * 2 + 1] = s2;
    }
}

// one block per channel: fixed-order double reduction of the partials -> dgamma/dbeta (+=) and c1 = s1/N, c2 = s2/N
__global__ __launch_bounds__(256) void bn_bwd_finalize(const float* __restrict__ partial, int nblocks, int C, double count,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ c12) {
    const int c = blockIdx.x;
    __shared__ double r1[256], r2[256];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        s1 += (double)partial[((long)i * C + c) * 2 + 0];
        s2 += (double)partial[((long)i * C + c) * 2 + 1];
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dbeta[c] += (float)r1[0];
        dgamma[c] += (float)r2[0];
        c12[2 * c + 0] = (float)(r1[0] / count);
        c12[2 * c + 1] = (float)(r2[0] / count);
    }
}

// dx = scale_c * (g' - c1 - xh*c2), element i has channel (i / F) % C; may run in place (dx == g)
__global__ void bn_bwd_apply(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ mean,
                             const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
                             const float* __restrict__ c12, const uint8_t* __restrict__ mask, float inv_keep,
                             float* __restrict__ dx, long n, int C, int F, float* __restrict__ absmax) {
    const long stride = (long)gridDim.x * blockDim.x;
    float am = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)((i / F) % C);
        const float xv = x[i];
        float gv = (xv * scale[c] + shift[c] > 0.f) ? g[i] : 0.f;
        if (mask) gv = mask[i] ? gv * inv_keep : 0.f;
        const float o = scale[c] * (gv - c12[2 * c] - (xv - mean[c]) * invstd[c] * c12[2 * c + 1]);
        dx[i] = o;
        am = fmaxf(am, fabsf(o));
    }
    if (absmax) {           // max |dx| for the consumer's power-of-two operand scale
        __shared__ float red[16];
        block_absmax_to(absmax, am, red);
    }
}

// The same for the (rows, C, F) activation layout with F % 4 == 0: one workgroup per (b,t) row, 16-byte accesses, the channel of a
// quad computed once (the generic kernel above spends two integer divisions per element and moves 4 bytes per access).
__global__ __launch_bounds__(256) void bn_bwd_apply_planes(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ c12,
                                                           float* __restrict__ dx, int C, int F, float* __restrict__ absmax) {
    __shared__ float k[64 * 6];
    float am = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        k[c * 6 + 0] = mean[c]; k[c * 6 + 1] = invstd[c]; k[c * 6 + 2] = scale[c]; k[c * 6 + 3] = shift[c];
        k[c * 6 + 4] = c12[2 * c]; k[c * 6 + 5] = c12[2 * c + 1];
    }
    __syncthreads();
    const int qpp = F / 4, nq = C * qpp;                       // quads per plane, per row
    const long base = (long)blockIdx.x * C * F;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + base);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + base);
    f32x4* d4 = reinterpret_cast<f32x4*>(dx + base);
    for (int q0 = threadIdx.x; q0 < nq; q0 += 4 * 256) {        // 4 quads in flight per thread
        f32x4 gv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * 256;
            if (q < nq) { gv[u] = g4[q]; xv[u] = x4[q]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * 256;
            if (q >= nq) continue;
            const float* kc = k + (q / qpp) * 6;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gm = (xv[u][e] * kc[2] + kc[3] > 0.f) ? gv[u][e] : 0.f;
                o[e] = kc[2] * (gm - kc[4] - (xv[u][e] - kc[0]) * kc[1] * kc[5]);
                am = fmaxf(am, fabsf(o[e]));
            }
            d4[q] = o;
        }
    }
    if (absmax) {
        __shared__ float red[16];
        block_absmax_to(absmax, am, red);
    }
}

// ------------------------------------------------------------------------------------------- conv weight gradient
// Optional fused BatchNorm backward of the layer's OUTPUT side: instead of a ready dy the kernel gets g (gradient wrt the activation
// relu(bn(y))) and y (the layer's pre-BN output) and forms  dy = scale_c * (g' - c1_c - (y - mean_c) * invstd_c * c2_c),
// g' = g where bn(y) > 0 else 0  (exactly bn_bwd_apply) while staging -- and writes dy out for the data-gradient convolution that
// follows.  The weight-gradient kernels are MFMA-bound with HBM bandwidth to spare, so the separate apply pass (read g, y, write dx:
// 71 GB per 40-channel layer at B = 256) disappears from the step.
struct BnBwdFuse {
    const float* y;          // (B, T, Cout, F) pre-BN output of this layer; null = dy is given ready-made
    const float* mean; const float* invstd; const float* scale; const float* shift;
    const float* c12;        // [Cout][2] = (sum g', sum g' xhat) / count  (bn_bwd_finalize)
    float* dy_out;           // (B, T, Cout, F) or null
};
__device__ __forceinline__ float bn_bwd_value(float g, float yv, float mean, float invstd, float scale, float shift, float c1, float c2) {
    const float gm = (yv * scale + shift > 0.f) ? g : 0.f;
    return scale * (gm - c1 - (yv - mean) * invstd * c2);
}

// dW[co][ci][tap] += sum_{b,t,f} dy[b,t,co,f] * in[b, t+dt-1, ci, f+df-1],  in = relu(x*scale+shift) (or x).
// GEMM view on v_mfma_f32_16x16x4_f32: M = co, N = (ci_local, tap) of one 20-channel chunk (blockIdx.y), K = positions
// (4 consecutive f per MFMA).  gridDim.x persistent workgroups walk the (b, 4-row strip, 32-column) tiles; each writes
// ONE partial slab [Cout][180]; wgrad_reduce sums the slabs in fixed order (deterministic).
template <int COUT, bool BN>
__global__ __launch_bounds__(256, 3) void conv3x3_wgrad(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                     float* __restrict__ partial, int B, int T, int F, int Cin, BnBwdFuse bn) {
    constexpr int MT = (COUT + 15) / 16;
    constexpr int NTW = 3;                               // n-tiles per wave: 4 waves x 3 x 16 = 192 >= 180
    // LDS layouts chosen by exhaustive search for conflict-free fragment reads (each half-wave = 16 M/N indices x 2 k-slots must
    // hit 32 distinct banks; PMC on the first layout: 77 % of the LDS cycles were bank-conflict cycles, 8-way on the dy image):
    //   dy   : [r][co][34]  -> lane stride 34 floats (== 2 mod 32) between output channels;
    //   input: [c][6 rows][36], plane 240; interior at column 2 (halos at 1 / 34).  Rows are 8-byte aligned -> float2 stores.
    constexpr int DRS = CV_FT + 2;                       // 34
    constexpr int WRS = 36;
    constexpr int WPL = 240;
    constexpr int WC0 = 2;                               // LDS column of f0
    __shared__ __attribute__((aligned(16))) float lin[CV_CK * WPL];
    __shared__ __attribute__((aligned(16))) float ldy[CV_TR * MT * 16 * DRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int c0 = blockIdx.y * CV_CK;
    const int tilesF = (F + CV_FT - 1) / CV_FT, tilesT = (T + CV_TR - 1) / CV_TR;
    const long ntiles = (long)B * tilesT * tilesF;
    const bool vec_ok = (F % 4 == 0) && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0);

    int boff[NTW];                                       // LDS offset of this lane's (ci, dt, df) column per n-tile
    bool bok[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = (wave * NTW + j) * 16 + li;
        bok[j] = n < CV_CK * 9 && (c0 + n / 9) < Cin;
        const int c = n / 9, tap = n % 9;
        boff[j] = bok[j] ? c * WPL + (tap / 3) * WRS + (tap % 3) + WC0 - 1 : 0;
    }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // rows of the dy image that belong to padded output channels (co >= COUT) stay zero for the whole kernel
    for (int e = tid; e < CV_TR * (MT * 16 - COUT) * DRS; e += 256) {
        const int r = e / ((MT * 16 - COUT) * DRS), rem = e % ((MT * 16 - COUT) * DRS);
        ldy[(r * MT * 16 + COUT) * DRS + rem] = 0.f;
    }

    // XCD-aware walk (see conv3x3_wgrad_split): each XCD's workgroups (id % 8) share a contiguous eighth of the tile space
    long t_begin = blockIdx.x, t_end = ntiles, t_step = gridDim.x;
    if (gridDim.x % 8 == 0) {
        const long chunk = (ntiles + 7) / 8;
        t_begin = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
        t_end = min(ntiles, (long)(blockIdx.x % 8 + 1) * chunk);
        t_step = gridDim.x / 8;
    }
    for (long tile = t_begin; tile < t_end; tile += t_step) {
        long bid = tile;
        const int ft = (int)(bid % tilesF); bid /= tilesF;
        const int tt = (int)(bid % tilesT); const int b = (int)(bid / tilesT);
        const int t0 = tt * CV_TR, f0 = ft * CV_FT;
        // ---- staging: ALL global loads of the tile are issued first (one round trip instead of ~10 dependent ones), then transformed
        // (producer's BN+ReLU) and stored to LDS
        constexpr int XIT = (CV_CK * (CV_TR + 2) * (CV_FT / 4) + 255) / 256;      // 4
        constexpr int DIT = (COUT * CV_TR * (CV_FT / 4) + 255) / 256;             // 5 (Cout 40) / 3 (Cout 20)
        f32x4 xreg[XIT], dreg[DIT], yreg[BN ? DIT : 1];
        float hreg = 0.f;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = tid + 256 * it;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < CV_CK * (CV_TR + 2) * (CV_FT / 4)) {
                const int j = e % (CV_FT / 4), row = e / (CV_FT / 4);
                const int r = row % (CV_TR + 2), c = row / (CV_TR + 2);
                const int t = t0 + r - 1, f = f0 + 4 * j, ci = c0 + c;
                if (t >= 0 && t < T && f < F && ci < Cin) {
                    const float* src = x + (((long)b * T + t) * Cin + ci) * F + f;
                    if (vec_ok && f + 3 < F) v = *reinterpret_cast<const f32x4*>(src);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (f + q < F) v[q] = src[q];
                    }
                }
            }
            xreg[it] = v;
        }
        if (tid < CV_CK * (CV_TR + 2) * 2) {
            const int side = tid & 1, row = tid >> 1;
            const int r = row % (CV_TR + 2), c = row / (CV_TR + 2);
            const int t = t0 + r - 1, f = side ? f0 + CV_FT : f0 - 1, ci = c0 + c;
            hreg = (t >= 0 && t < T && f >= 0 && f < F && ci < Cin) ? x[(((long)b * T + t) * Cin + ci) * F + f] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int e = tid + 256 * it;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < COUT * CV_TR * (CV_FT / 4)) {
                const int j = e % (CV_FT / 4), row = e / (CV_FT / 4);
                const int r = row % CV_TR, co = row / CV_TR;
                const int t = t0 + r, f = f0 + 4 * j;
                f32x4 yv = {0.f, 0.f, 0.f, 0.f};
                if (t < T && f < F) {
                    const long off = (((long)b * T + t) * COUT + co) * F + f;
                    if (vec_ok && f + 3 < F) {
                        v = *reinterpret_cast<const f32x4*>(dy + off);
                        if (BN) yv = *reinterpret_cast<const f32x4*>(bn.y + off);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (f + q < F) { v[q] = dy[off + q]; if (BN) yv[q] = bn.y[off + q]; }
                    }
                }
                if (BN) yreg[it] = yv;
            }
            dreg[it] = v;
        }
        __syncthreads();                   // previous tile fully consumed (the loads above are already in flight)
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = tid + 256 * it;
            if (e < CV_CK * (CV_TR + 2) * (CV_FT / 4)) {
                const int j = e % (CV_FT / 4), row = e / (CV_FT / 4);
                const int r = row % (CV_TR + 2), c = row / (CV_TR + 2);
                const int t = t0 + r - 1, f = f0 + 4 * j, ci = c0 + c;
                f32x4 v = xreg[it];
                if (in_scale && t >= 0 && t < T && f < F && ci < Cin) {
                    const float sc = in_scale[ci], sh = in_shift[ci];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (f + q < F) ? fmaxf(v[q] * sc + sh, 0.f) : 0.f;
                }
                float2* dst = reinterpret_cast<float2*>(lin + c * WPL + r * WRS + WC0 + 4 * j);
                dst[0] = make_float2(v[0], v[1]); dst[1] = make_float2(v[2], v[3]);
            }
        }
        if (tid < CV_CK * (CV_TR + 2) * 2) {
            const int side = tid & 1, row = tid >> 1;
            const int r = row % (CV_TR + 2), c = row / (CV_TR + 2);
            const int t = t0 + r - 1, f = side ? f0 + CV_FT : f0 - 1, ci = c0 + c;
            float v = hreg;
            if (in_scale && t >= 0 && t < T && f >= 0 && f < F && ci < Cin) v = fmaxf(v * in_scale[ci] + in_shift[ci], 0.f);
            lin[c * WPL + r * WRS + (side ? WC0 + CV_FT : WC0 - 1)] = v;
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int e = tid + 256 * it;
            if (e < COUT * CV_TR * (CV_FT / 4)) {
                const int j = e % (CV_FT / 4), row = e / (CV_FT / 4);
                const int r = row % CV_TR, co = row / CV_TR;
                f32x4 v = dreg[it];
                if (BN) {                   // fused BatchNorm backward (positions outside the image stay 0)
                    const int t = t0 + r, f = f0 + 4 * j;
                    const float mean = bn.mean[co], invstd = bn.invstd[co], sc = bn.scale[co], sh = bn.shift[co];
                    const float c1 = bn.c12[2 * co], c2 = bn.c12[2 * co + 1];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (t < T && f + q < F) ? bn_bwd_value(v[q], yreg[it][q], mean, invstd, sc, sh, c1, c2) : 0.f;
                    if (bn.dy_out && blockIdx.y == 0 && t < T && f < F) {
                        float* dst = bn.dy_out + (((long)b * T + t) * COUT + co) * F + f;
                        if (vec_ok && f + 3 < F) *reinterpret_cast<f32x4*>(dst) = v;
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (f + q < F) dst[q] = v[q];
                        }
                    }
                }
                float2* dst = reinterpret_cast<float2*>(ldy + (r * MT * 16 + co) * DRS + 4 * j);
                dst[0] = make_float2(v[0], v[1]); dst[1] = make_float2(v[2], v[3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CV_TR; ++r) {
#pragma unroll
            for (int ks = 0; ks < CV_FT / 4; ++ks) {
                float a[MT], bv[NTW];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = ldy[(r * MT * 16 + i * 16 + li) * DRS + ks * 4 + lk];
#pragma unroll
                for (int j = 0; j < NTW; ++j) bv[j] = bok[j] ? lin[boff[j] + r * WRS + ks * 4 + lk] : 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // D[row = co (lk*4+r)][col = n (li)]
    float* slab = partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * COUT * (CV_CK * 9);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = (wave * NTW + j) * 16 + li;
            if (n >= CV_CK * 9) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = i * 16 + lk * 4 + r;
                if (co < COUT) slab[(long)co * (CV_CK * 9) + n] = acc[i][j][r];
            }
        }
}

// dW[co][c0+ci][tap] += sum over slabs (fixed order)
__global__ void wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dW, int nslabs, int Cout, int Cin, int chunks) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = Cout * CV_CK * 9;
    if (idx >= per * chunks) return;
    const int chunk = idx / per, rem = idx % per;
    const int co = rem / (CV_CK * 9), n = rem % (CV_CK * 9);
    const int ci = chunk * CV_CK + n / 9, tap = n % 9;
    if (ci >= Cin) return;
    const float* p = partial + (long)chunk * nslabs * per + rem;
    float s = 0.f;
    for (int i = 0; i < nslabs; ++i) s += p[(long)i * per];
    dW[((long)co * Cin + ci) * 9 + tap] += s;
}

#define WGRAD_SLABS 768

// First layer (Cin = 1): dW[co][tap] = sum_{b,t,f} dy[b,t,co,f] * x[b, t+dt-1, f+df-1].  53 GFLOP at B = 256 against 12 GB of dy: a
// streaming kernel (the MFMA kernel above would spend a full 20-channel chunk on one real input channel).  Persistent workgroups walk
// the (b,t) rows; a row's dy block (Cout x F) and the 3 x (F+2) input window go through LDS; thread = (co, slice of F) keeps its 9
// tap sums in registers; one slab [Cout][9] per workgroup, reduced in fixed order by wgrad_reduce_c1.
#define C1W_PARTS 12
#define C1W_XIT 10                          // float4 loads per thread and row in the fused path (20 x 480 / 4 / 256 = 9.4)
__global__ __launch_bounds__(256) void conv3x3_wgrad_c1(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ partial,
                                                        int B, int T, int F, int Cout, BnBwdFuse bn) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ldy = sm;                            // Cout * F
    float* lx = sm + Cout * F;                  // 3 * (F + 2)
    __shared__ float lconst[20 * 6];            // per channel: mean, invstd, scale, shift, c1, c2 (fused BatchNorm backward)
    const int tid = threadIdx.x;
    if (bn.y && tid < Cout) {
        float* k = lconst + tid * 6;
        k[0] = bn.mean[tid]; k[1] = bn.invstd[tid]; k[2] = bn.scale[tid]; k[3] = bn.shift[tid]; k[4] = bn.c12[2 * tid]; k[5] = bn.c12[2 * tid + 1];
    }
    const int co = tid / C1W_PARTS, part = tid % C1W_PARTS;
    const int fper = (F + C1W_PARTS - 1) / C1W_PARTS;
    const int fa = part * fper, fb = min(F, fa + fper);
    const bool worker = co < Cout;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    const long rows = (long)B * T;
    const bool vec_ok = (F % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const int t = (int)(row % T);
        const long b = row / T;
        __syncthreads();
        const float* drow = dy + row * (long)Cout * F;
        if (bn.y && vec_ok && (((uintptr_t)bn.y & 15) == 0) && Cout * F <= 4 * 256 * C1W_XIT) {
            // fused BatchNorm backward of the dy operand (see BnBwdFuse): all loads of the row first (one round trip), constants from LDS
            const float* yrow = bn.y + row * (long)Cout * F;
            f32x4 g4[C1W_XIT], y4[C1W_XIT];
#pragma unroll
            for (int it = 0; it < C1W_XIT; ++it) {
                const int e = tid + 256 * it;
                if (e < Cout * F / 4) { g4[it] = reinterpret_cast<const f32x4*>(drow)[e]; y4[it] = reinterpret_cast<const f32x4*>(yrow)[e]; }
            }
#pragma unroll
            for (int it = 0; it < C1W_XIT; ++it) {
                const int e = tid + 256 * it;
                if (e < Cout * F / 4) {
                    const int c = (e * 4) / F;
                    const float* k = lconst + c * 6;
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = bn_bwd_value(g4[it][q], y4[it][q], k[0], k[1], k[2], k[3], k[4], k[5]);
                    reinterpret_cast<f32x4*>(ldy)[e] = v;
                    if (bn.dy_out) reinterpret_cast<f32x4*>(bn.dy_out + row * (long)Cout * F)[e] = v;
                }
            }
        } else if (bn.y) {
            const float* yrow = bn.y + row * (long)Cout * F;
            for (int e = tid; e < Cout * F; e += 256) {
                const int c = e / F;
                const float v = bn_bwd_value(drow[e], yrow[e], bn.mean[c], bn.invstd[c], bn.scale[c], bn.shift[c], bn.c12[2 * c], bn.c12[2 * c + 1]);
                ldy[e] = v;
                if (bn.dy_out) bn.dy_out[row * (long)Cout * F + e] = v;
            }
        } else if (vec_ok) {
            for (int e = tid; e < Cout * F / 4; e += 256) reinterpret_cast<f32x4*>(ldy)[e] = reinterpret_cast<const f32x4*>(drow)[e];
        } else {
            for (int e = tid; e < Cout * F; e += 256) ldy[e] = drow[e];
        }
        for (int e = tid; e < 3 * (F + 2); e += 256) {
            const int r = e / (F + 2), c = e % (F + 2);
            const int tt = t + r - 1, f = c - 1;
            lx[e] = (tt >= 0 && tt < T && f >= 0 && f < F) ? x[(b * T + tt) * F + f] : 0.f;
        }
        __syncthreads();
        if (worker) {
            const float* d = ldy + co * F;
            float w[3][3];                                  // input window, slid along f: 3 new values per position instead of 9
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) { w[dt][1] = lx[dt * (F + 2) + fa]; w[dt][2] = lx[dt * (F + 2) + fa + 1]; }
            for (int f = fa; f < fb; ++f) {
                const float g = d[f];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    w[dt][0] = w[dt][1]; w[dt][1] = w[dt][2]; w[dt][2] = lx[dt * (F + 2) + f + 2];
#pragma unroll
                    for (int df = 0; df < 3; ++df) acc[dt * 3 + df] = fmaf(g, w[dt][df], acc[dt * 3 + df]);
                }
            }
        }
    }
    __syncthreads();
    float* red = sm;                            // 256 * 9 floats (fits: Cout * F >= 2304 for the model's F)
#pragma unroll
    for (int k = 0; k < 9; ++k) red[tid * 9 + k] = worker ? acc[k] : 0.f;
    __syncthreads();
    if (tid < Cout * 9) {
        const int c = tid / 9, k = tid % 9;
        float s = 0.f;
        for (int p = 0; p < C1W_PARTS; ++p) s += red[(c * C1W_PARTS + p) * 9 + k];
        partial[(long)blockIdx.x * Cout * 9 + tid] = s;
    }
}

__global__ void wgrad_reduce_c1(const float* __restrict__ partial, float* __restrict__ dW, int nslabs, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float s = 0.f;
    for (int i = 0; i < nslabs; ++i) s += partial[(long)i * n + idx];
    dW[idx] += s;
}

// The model's case of conv3x3_wgrad_c1 (fused BatchNorm backward, F a multiple of 4 * C1W_PARTS) restructured for the memory system -- same
// rows per workgroup, same order of additions per tap sum, so the slabs are bit-identical:
//   * the NEXT row's dy / y / input loads are issued before the current row's tap sums are computed (the old kernel alternated a load phase
//     and a compute phase per workgroup, 4.1 TB/s);
//   * the tap sums read LDS 16 bytes at a time: one quad of dy and one quad per window row for 4 positions (the old loop: 4 scalar reads
//     per position, 2-4-way bank conflicts -- SQ_LDS_BANK_CONFLICT / SQ_LDS_ACTIVE = 0.67).  The window rows are stored at a stride of
//     F + 8 with x[f] at column 4 + f, so the quads are aligned and the zero borders sit at columns 3 and F + 4.
#define C1W_XN 2                            // input-window quads per thread and row: 3 * F / 4 <= 2 * 256
template <int COUT, int F>
__global__ __launch_bounds__(256, 3) void conv3x3_wgrad_c1_stream(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ partial,
                                                                  int B, int T, BnBwdFuse bn) {
    static_assert(F % (4 * C1W_PARTS) == 0 && 3 * F / 4 <= 256 * C1W_XN && COUT * F <= 4 * 256 * C1W_XIT && COUT * C1W_PARTS <= 256, "shape");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int LXS = F + 8;
    float* ldy = sm;                            // COUT * F
    float* lx = sm + COUT * F;                  // 3 * LXS
    __shared__ float lconst[COUT * 6];
    const int tid = threadIdx.x;
    if (tid < COUT) {
        float* k = lconst + tid * 6;
        k[0] = bn.mean[tid]; k[1] = bn.invstd[tid]; k[2] = bn.scale[tid]; k[3] = bn.shift[tid]; k[4] = bn.c12[2 * tid]; k[5] = bn.c12[2 * tid + 1];
    }
    const int co = tid / C1W_PARTS, part = tid % C1W_PARTS;
    constexpr int fper = F / C1W_PARTS, nq = COUT * F / 4, nx = 3 * F / 4;      // quads of a dy row, of the 3-row input window
    const int fa = part * fper;
    const bool worker = co < COUT;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    const long rows = (long)B * T;
    f32x4 g4[C1W_XIT], y4[C1W_XIT];
    f32x4 xv[C1W_XN];
    auto issue = [&](long row) {
        const f32x4* drow = reinterpret_cast<const f32x4*>(dy + row * (long)COUT * F);
        const f32x4* yrow = reinterpret_cast<const f32x4*>(bn.y + row * (long)COUT * F);
#pragma unroll
        for (int it = 0; it < C1W_XIT; ++it) {
            const int e = tid + 256 * it;
            if (e < nq) { g4[it] = drow[e]; y4[it] = yrow[e]; }
        }
        const int t = (int)(row % T);
#pragma unroll
        for (int it = 0; it < C1W_XN; ++it) {
            const int e = tid + 256 * it, r = e / (F / 4), tt = t + r - 1;
            xv[it] = (e < nx && tt >= 0 && tt < T) ? reinterpret_cast<const f32x4*>(x + (row + r - 1) * F)[e % (F / 4)] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (tid < 6) lx[(tid >> 1) * LXS + ((tid & 1) ? F + 4 : 3)] = 0.f;        // the zero borders left and right of the window rows
    if ((long)blockIdx.x < rows) issue(blockIdx.x);
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        __syncthreads();                        // the previous row's tap sums are done with LDS (and lconst is written)
#pragma unroll
        for (int it = 0; it < C1W_XIT; ++it) {
            const int e = tid + 256 * it;
            if (e < nq) {
                const float* k = lconst + ((e * 4) / F) * 6;
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = bn_bwd_value(g4[it][q], y4[it][q], k[0], k[1], k[2], k[3], k[4], k[5]);
                reinterpret_cast<f32x4*>(ldy)[e] = v;
                if (bn.dy_out) reinterpret_cast<f32x4*>(bn.dy_out + row * (long)COUT * F)[e] = v;
            }
        }
#pragma unroll
        for (int it = 0; it < C1W_XN; ++it) {
            const int e = tid + 256 * it;
            if (e < nx) *reinterpret_cast<f32x4*>(lx + (e / (F / 4)) * LXS + 4 + 4 * (e % (F / 4))) = xv[it];
        }
        if (row + gridDim.x < rows) issue(row + gridDim.x);      // in flight while this row's tap sums are computed
        __syncthreads();
        if (worker) {
            const float* d = ldy + co * F + fa;
            const float* xr = lx + 4 + fa;
            float left[3];
            f32x4 cur[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) { left[dt] = xr[dt * LXS - 1]; cur[dt] = *reinterpret_cast<const f32x4*>(xr + dt * LXS); }
#pragma unroll 2
            for (int j = 0; j < fper; j += 4) {                 // (fully unrolled, the 40 quad reads are all issued first: 100 spilled registers)
                const f32x4 g = *reinterpret_cast<const f32x4*>(d + j);
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    const f32x4 nxt = *reinterpret_cast<const f32x4*>(xr + dt * LXS + j + 4);      // only [0] is used past the row's end: the border
                    const float w[6] = {left[dt], cur[dt][0], cur[dt][1], cur[dt][2], cur[dt][3], nxt[0]};
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int df = 0; df < 3; ++df) acc[dt * 3 + df] = fmaf(g[p], w[p + df], acc[dt * 3 + df]);
                    left[dt] = cur[dt][3]; cur[dt] = nxt;
                }
            }
        }
    }
    __syncthreads();
    float* red = sm;                            // 256 * 9 floats
#pragma unroll
    for (int k = 0; k < 9; ++k) red[tid * 9 + k] = worker ? acc[k] : 0.f;
    __syncthreads();
    if (tid < COUT * 9) {
        const int c = tid / 9, k = tid % 9;
        float s = 0.f;
        for (int p = 0; p < C1W_PARTS; ++p) s += red[(c * C1W_PARTS + p) * 9 + k];
        partial[(long)blockIdx.x * COUT * 9 + tid] = s;
    }
}

// ---- weight gradient on the bf16 matrix pipes with 3-term split operands (see conv3x3_bf16x3).  K = positions: a fragment is 8
// consecutive f positions of ONE channel row, so dy (M = co) and the input (N = ci) both stay position-contiguous in LDS; an N-tile is
// 16 input channels at ONE tap, which makes the +-1 column shift of the tap uniform per MFMA: the lane reads the aligned 8 positions plus
// the dword before and after -- df = 1 is the aligned window, df = 0 / 2 are four v_alignbit_b32 each.  One 512-thread workgroup per CU
// walks (clip, 2-row strip, 64-column) tiles; the 9 x ceil(Cin/16) (tap, channel-tile) column tiles are dealt round robin to the 8 waves
// (<= 4 each, 12 accumulator tiles); every other tile accumulates the negated sum (dy negated while staging) against the pipe's
// truncation bias.  One partial slab [Cout][Cin][9] per workgroup, summed in fixed order by wgrad_reduce_c1.
#define W4_XROW 160                       // bytes per staged input row: 80 bf16, interior at element 8 (halo columns at 7 and 72)
#define W4_XCI (4 * W4_XROW + 16)         // 656: channel stride (pad -> conflict-free fragment reads across the 16 channels of a tile)
#define W4_DYCO (2 * 128 + 16)            // 272: output-channel stride of the dy image [co][2 rows][64]
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// TERMS = 3: bf16 three-term split (six products); TERMS = 2: fp16 two-term split (three products), dy scaled by the power of two that
// brings max |dy| (dy_absmax, device scalar; null = unscaled) to 2^12, the slab unscaled on the way out -- see conv3x3_split.
template <int COUT, int TERMS>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_split(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                              float* __restrict__ partial, int B, int T, int F, int Cin,
                                                              const float* __restrict__ dy_absmax, const float* __restrict__ act_absmax) {
    constexpr int MT = (COUT + 15) / 16;
    constexpr int XPL = 48 * W4_XCI;          // bytes per term plane of the input image (48 channel planes; beyond Cin they stay zero)
    constexpr int DPL = MT * 16 * W4_DYCO;
    __shared__ __attribute__((aligned(16))) unsigned char lx[TERMS * XPL];
    __shared__ __attribute__((aligned(16))) unsigned char ldy[TERMS * DPL];
    __shared__ __attribute__((aligned(16))) float lsc[48], lsh[48];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int nc16 = (Cin + 15) / 16, npair = 9 * nc16;
    for (int e = tid; e < TERMS * XPL / 16; e += 512) reinterpret_cast<uint4*>(lx)[e] = make_uint4(0u, 0u, 0u, 0u);
    for (int e = tid; e < TERMS * DPL / 16; e += 512) reinterpret_cast<uint4*>(ldy)[e] = make_uint4(0u, 0u, 0u, 0u);
    float dscale = 1.f, unscale = 1.f;
    if (TERMS == 2 && dy_absmax) {
        const int kd = pow2_scale_exp(*dy_absmax, 12);
        dscale = ldexpf(1.f, kd); unscale = ldexpf(1.f, -kd);
    }
    // act_absmax: device scalar bounding the activated operand relu(x scale_c + shift_c) (a2s_act_bound).  Its power-of-two scale is
    // folded into scale / shift (relu commutes with it) and undone with the slab: the operand's fp16 terms neither overflow nor sink
    // into the subnormal range whatever BatchNorm's gamma is.  Without it the operand is used as it is (and clamped at +-65000).
    int ka = 0;
    if (TERMS == 2 && act_absmax && in_scale) {
        ka = pow2_scale_exp(*act_absmax, 14);
        unscale *= ldexpf(1.f, -ka);
    }
    if (tid < 48) {
        lsc[tid] = (in_scale && tid < Cin) ? ldexpf(in_scale[tid], ka) : 0.f;
        lsh[tid] = (in_scale && tid < Cin) ? ldexpf(in_shift[tid], ka) : 0.f;
    }
    f32x4 acc[4][MT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tilesF = (F + 63) / 64, tilesT = (T + 1) / 2;
    const long ntiles = (long)B * tilesT * tilesF;
    const bool vec_ok = (F % 4 == 0) && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0);
    int pj_off[4], pj_df[4];                  // this wave's (tap, channel-tile) column tiles: LDS offset of the lane's channel row, column shift
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = min(wave + 8 * j, npair - 1), tap = p / nc16, c16 = p % nc16;
        pj_off[j] = (c16 * 16 + li) * W4_XCI + (tap / 3) * W4_XROW;
        pj_df[j] = tap % 3;
    }
    bool acc_neg = false;
    int round = 0;
    // register prefetch: the global loads of the NEXT tile are issued before the multiply of the current one and committed after it
    constexpr int DIT = (COUT * 32 + 511) / 512, XIT = 5;           // 16-byte loads per thread: dy (3 / 2), input (<= 40 ch x 4 rows x 16)
    f32x4 dreg[DIT], xreg[XIT];
    float hreg = 0.f;
    // one slice = one 16-byte load per thread (slices 0 .. DIT-1: dy, then XIT of the input; the halo columns ride on the last)
    auto issue_slice = [&](long tile, int slice) {
        long bid = tile;
        const int ft = (int)(bid % tilesF); bid /= tilesF;
        const int tt = (int)(bid % tilesT); const int b = (int)(bid / tilesT);
        const int t0 = tt * 2, f0 = ft * 64;
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            if (it != slice) continue;
            const int e = tid + 512 * it;
            const int co = e >> 5, r = (e >> 4) & 1, fq = e & 15;
            const int t = t0 + r, f = f0 + 4 * fq;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < COUT * 32 && t < T && f < F) {
                const float* src = dy + (((long)b * T + t) * COUT + co) * F + f;
                if (vec_ok && f + 3 < F) v = *reinterpret_cast<const f32x4*>(src);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (f + q < F) v[q] = src[q];
                }
            }
            dreg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            if (DIT + it != slice) continue;
            const int e = tid + 512 * it;
            const int ci = e >> 6, row = (e >> 4) & 3, fq = e & 15;
            const int t = t0 + row - 1, f = f0 + 4 * fq;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < Cin * 64 && t >= 0 && t < T && f < F) {
                const float* src = x + (((long)b * T + t) * Cin + ci) * F + f;
                if (vec_ok && f + 3 < F) v = *reinterpret_cast<const f32x4*>(src);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (f + q < F) v[q] = src[q];
                }
            }
            xreg[it] = v;
        }
        if (slice != DIT + XIT - 1) return;
        hreg = 0.f;
        if (tid < Cin * 8) {                 // halo columns f0 - 1 and f0 + 64
            const int ci = tid >> 3, row = (tid >> 1) & 3, side = tid & 1;
            const int t = t0 + row - 1, f = side ? f0 + 64 : f0 - 1;
            if (t >= 0 && t < T && f >= 0 && f < F) hreg = x[(((long)b * T + t) * Cin + ci) * F + f];
        }
    };
    auto commit = [&](long tile, unsigned sgn) {
        long bid = tile;
        const int ft = (int)(bid % tilesF); bid /= tilesF;
        const int tt = (int)(bid % tilesT);
        const int t0 = tt * 2, f0 = ft * 64;
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int e = tid + 512 * it;
            if (e >= COUT * 32) continue;
            const int co = e >> 5, r = (e >> 4) & 1, fq = e & 15;
            f32x4 v = dreg[it];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = __uint_as_float(__float_as_uint(v[q]) ^ sgn);
            uint2 o[TERMS];
            if (TERMS == 3) {
                split3_pair(v[0], v[1], o[0].x, o[1].x, o[TERMS - 1].x);
                split3_pair(v[2], v[3], o[0].y, o[1].y, o[TERMS - 1].y);
            } else {
                split2_pair_f16(v[0] * dscale, v[1] * dscale, o[0].x, o[1].x);
                split2_pair_f16(v[2] * dscale, v[3] * dscale, o[0].y, o[1].y);
            }
#pragma unroll
            for (int sp = 0; sp < TERMS; ++sp) *reinterpret_cast<uint2*>(ldy + sp * DPL + co * W4_DYCO + r * 128 + fq * 8) = o[sp];
        }
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = tid + 512 * it;
            if (e >= Cin * 64) continue;
            const int ci = e >> 6, row = (e >> 4) & 3, fq = e & 15;
            const int t = t0 + row - 1, f = f0 + 4 * fq;
            f32x4 v = xreg[it];
            if (in_scale && t >= 0 && t < T && f < F) {
                const float sc = lsc[ci], sh = lsh[ci];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (f + q < F) ? fmaxf(v[q] * sc + sh, 0.f) : 0.f;
            }
            uint2 o[TERMS];
            if (TERMS == 3) {
                split3_pair(v[0], v[1], o[0].x, o[1].x, o[TERMS - 1].x);
                split3_pair(v[2], v[3], o[0].y, o[1].y, o[TERMS - 1].y);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fminf(fmaxf(v[q], -65000.f), 65000.f);
                split2_pair_f16(v[0], v[1], o[0].x, o[1].x);
                split2_pair_f16(v[2], v[3], o[0].y, o[1].y);
            }
#pragma unroll
            for (int sp = 0; sp < TERMS; ++sp) *reinterpret_cast<uint2*>(lx + sp * XPL + ci * W4_XCI + row * W4_XROW + (8 + 4 * fq) * 2) = o[sp];
        }
        if (tid < Cin * 8) {
            const int ci = tid >> 3, row = (tid >> 1) & 3, side = tid & 1;
            const int t = t0 + row - 1, f = side ? f0 + 64 : f0 - 1;
            float v = hreg;
            if (in_scale && t >= 0 && t < T && f >= 0 && f < F) v = fmaxf(v * lsc[ci] + lsh[ci], 0.f);
            unsigned p0, p1, p2 = 0;
            if (TERMS == 3) split3_pair(v, 0.f, p0, p1, p2);
            else split2_pair_f16(fminf(fmaxf(v, -65000.f), 65000.f), 0.f, p0, p1);
            unsigned char* dst = lx + ci * W4_XCI + row * W4_XROW + (side ? 72 : 7) * 2;
            *reinterpret_cast<unsigned short*>(dst) = (unsigned short)p0;
            *reinterpret_cast<unsigned short*>(dst + XPL) = (unsigned short)p1;
            if (TERMS == 3) *reinterpret_cast<unsigned short*>(dst + (TERMS - 1) * XPL) = (unsigned short)p2;
        }
    };
    // XCD-aware walk: workgroup ids are dealt round-robin to the 8 XCDs (id % 8); every XCD walks its own contiguous eighth of the tile
    // space, its workgroups side by side on consecutive tiles -- the 8 column tiles of a row strip (the whole 1920-byte rows) and the
    // strips above / below (shared halo rows) then meet in ONE L2 instead of eight.
    long t_begin = blockIdx.x, t_end = ntiles, t_step = gridDim.x;
    if (gridDim.x % 8 == 0) {
        const long chunk = (ntiles + 7) / 8;
        t_begin = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
        t_end = min(ntiles, (long)(blockIdx.x % 8 + 1) * chunk);
        t_step = gridDim.x / 8;
    }
#ifdef C4_TRACE
    const int trace_wg = (blockIdx.x >= 64 && blockIdx.x < 64 + C4_TRACE_WGS) ? (int)blockIdx.x - 64 : -1;
#define W4_STAMP(k) do { if (trace_wg >= 0 && round >= 100 && round < 100 + C4_TRACE_STAGES && tid == 0) c4_trace[(trace_wg * C4_TRACE_STAGES + round - 100) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define W4_STAMP(k) do {} while (0)
#endif
    auto issue = [&](long tile) {
#pragma unroll
        for (int sl = 0; sl < DIT + XIT; ++sl) issue_slice(tile, sl);
    };
    if (t_begin < t_end) issue(t_begin);
    for (long tile = t_begin; tile < t_end; tile += t_step, ++round) {
        const bool neg = round & 1;
        W4_STAMP(0);
        __syncthreads();                     // previous tile consumed (first round: the zero fill is complete)
        W4_STAMP(1);
        commit(tile, neg ? 0x80000000u : 0u);
        W4_STAMP(2);
        __syncthreads();
        W4_STAMP(3);
        // in flight during the multiply below.  (Issuing them slice by slice BETWEEN the MFMAs of the multiply instead of as one burst was
        // measured: the burst's 2.3 k clocks of back-pressure disappear, but the multiply grows from 8.1 k to 12.8 k clocks per tile -- a
        // wave stalled on a full memory queue issues no MFMAs either.)
        if (tile + t_step < t_end) issue(tile + t_step);
        W4_STAMP(4);
        if (neg != acc_neg) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[j][i][r] = -acc[j][i][r];
            acc_neg = neg;
        }
        // A lane group owns 16 consecutive positions of a row: two k-steps (positions 16 lk + 8 h + [0, 8), h = 0 / 1 -- any k order is
        // fine as long as dy and the input agree) share ONE set of loads, so the two conflict-prone 4-byte neighbour reads are paid
        // once per 16 positions.
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            u32x4 a[2][TERMS][MT];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int sp = 0; sp < TERMS; ++sp)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        a[h][sp][i] = *reinterpret_cast<const u32x4*>(ldy + sp * DPL + (i * 16 + li) * W4_DYCO + r * 128 + (lk * 16 + h * 8) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (wave + 8 * j >= npair) continue;            // uniform per wave
                const int df = pj_df[j];
                const unsigned char* base = lx + pj_off[j] + r * W4_XROW + (lk * 16 + 8) * 2;
                u32x4 bfrag[2][TERMS];
#pragma unroll
                for (int sp = 0; sp < TERMS; ++sp) {
                    const u32x4 d0 = *reinterpret_cast<const u32x4*>(base + sp * XPL);
                    const u32x4 d1 = *reinterpret_cast<const u32x4*>(base + sp * XPL + 16);
                    u32x4 w0 = d0, w1 = d1;                      // df == 1: the aligned windows
                    if (df == 0) {
                        const unsigned dm1 = *reinterpret_cast<const unsigned*>(base + sp * XPL - 4);
                        w0[0] = __builtin_amdgcn_alignbit(d0[0], dm1, 16); w0[1] = __builtin_amdgcn_alignbit(d0[1], d0[0], 16);
                        w0[2] = __builtin_amdgcn_alignbit(d0[2], d0[1], 16); w0[3] = __builtin_amdgcn_alignbit(d0[3], d0[2], 16);
                        w1[0] = __builtin_amdgcn_alignbit(d1[0], d0[3], 16); w1[1] = __builtin_amdgcn_alignbit(d1[1], d1[0], 16);
                        w1[2] = __builtin_amdgcn_alignbit(d1[2], d1[1], 16); w1[3] = __builtin_amdgcn_alignbit(d1[3], d1[2], 16);
                    } else if (df == 2) {
                        const unsigned d8 = *reinterpret_cast<const unsigned*>(base + sp * XPL + 32);
                        w0[0] = __builtin_amdgcn_alignbit(d0[1], d0[0], 16); w0[1] = __builtin_amdgcn_alignbit(d0[2], d0[1], 16);
                        w0[2] = __builtin_amdgcn_alignbit(d0[3], d0[2], 16); w0[3] = __builtin_amdgcn_alignbit(d1[0], d0[3], 16);
                        w1[0] = __builtin_amdgcn_alignbit(d1[1], d1[0], 16); w1[1] = __builtin_amdgcn_alignbit(d1[2], d1[1], 16);
                        w1[2] = __builtin_amdgcn_alignbit(d1[3], d1[2], 16); w1[3] = __builtin_amdgcn_alignbit(d8, d1[3], 16);
                    }
                    bfrag[0][sp] = w0;
                    bfrag[1][sp] = w1;
                }
#define W4_PRODUCT(SA, SB)                                                                                                   \
                _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                \
                    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                           \
                        acc[j][i] = (TERMS == 3) ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[h][SA][i]), __builtin_bit_cast(bf16x8, bfrag[h][SB]), acc[j][i], 0, 0, 0) \
                                                 : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[h][SA][i]), __builtin_bit_cast(f16x8, bfrag[h][SB]), acc[j][i], 0, 0, 0);
                if (TERMS == 3) { W4_PRODUCT(TERMS - 1, 0) W4_PRODUCT(1, 1) W4_PRODUCT(0, TERMS - 1) }
                W4_PRODUCT(1, 0) W4_PRODUCT(0, 1) W4_PRODUCT(0, 0)
#undef W4_PRODUCT
            }
        }
        W4_STAMP(5);
    }
    // slab [Cout][Cin][9]; C/D map: lane holds column n = li (input channel of the tile), rows 4 lk + r (output channel)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = wave + 8 * j;
        if (p >= npair) continue;
        const int tap = p / nc16, c16 = p % nc16, ci = c16 * 16 + li;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = i * 16 + lk * 4 + r;
                if (co < COUT && ci < Cin) partial[(((long)blockIdx.x * COUT + co) * Cin + ci) * 9 + tap] = (acc_neg ? -acc[j][i][r] : acc[j][i][r]) * unscale;
            }
    }
}

static int g_wgrad_split = 1;          // conv3x3_wgrad_split for the plain weight-gradient launches: 1 = where it is faster, 2 = every eligible launch
void a2s_wgrad_split_set(int on) { g_wgrad_split = on; }
int a2s_wgrad_split_enabled(void) { return g_wgrad_split; }
static int g_wgrad_f16x2 = -1;         // ... with two fp16 terms (needs the max |dy| scalar) instead of three bf16 terms
void a2s_wgrad_f16x2_set(int on) { g_wgrad_f16x2 = on; }
int a2s_wgrad_f16x2_enabled(void) {
    if (g_wgrad_f16x2 < 0) g_wgrad_f16x2 = 1;
    return g_wgrad_f16x2;
}

bool a2s_wgrad_rows_eligible(int F, int Cin, int Cout);
int a2s_conv3x3_wgrad_rows_impl(hipStream_t st, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW, float* ws,
                                size_t ws_bytes, int B, int T, int F, int Cin, int Cout, const float* dy_absmax, const float* act_absmax);

size_t a2s_conv3x3_wgrad_workspace_bytes_impl(int Cin, int Cout) {
    const int chunks = (Cin + CV_CK - 1) / CV_CK;
    return (size_t)chunks * 1024 * Cout * CV_CK * 9 * sizeof(float);
}

int a2s_conv3x3_wgrad_impl(hipStream_t st, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW,
                           float* ws, size_t ws_bytes, int B, int T, int F, int Cin, int Cout, const float* bn_y, const float* bn_mean,
                           const float* bn_invstd, const float* bn_scale, const float* bn_shift, const float* bn_c12, float* dy_out,
                           const float* dy_absmax, const float* act_absmax) {
    A2S_REQUIRE(dy && x && dW && ws, "conv3x3_wgrad: null tensor");
    A2S_REQUIRE(!bn_y || (bn_mean && bn_invstd && bn_scale && bn_shift && bn_c12), "conv3x3_wgrad: the fused BatchNorm backward needs all of its tensors");
    A2S_REQUIRE(!dy_out || bn_y, "conv3x3_wgrad: dy_out is only written by the fused BatchNorm backward");
    const BnBwdFuse bn{bn_y, bn_mean, bn_invstd, bn_scale, bn_shift, bn_c12, dy_out};
    A2S_REQUIRE(ws_bytes >= a2s_conv3x3_wgrad_workspace_bytes_impl(Cin, Cout), "conv3x3_wgrad: workspace too small");
    if (Cin == 1 && !in_scale && Cout * C1W_PARTS <= 256 && (size_t)Cout * F >= 256 * 9) {
        const size_t shm = ((size_t)Cout * F + 3 * (F + 8)) * sizeof(float);
        if (shm <= 64 * 1024) {
            const bool stream = g_conv_c1_fast && bn_y && Cout == 20 && F == 480 && (((uintptr_t)dy | (uintptr_t)bn_y | (uintptr_t)dy_out | (uintptr_t)x) & 15) == 0;
            if (stream) hipLaunchKernelGGL((conv3x3_wgrad_c1_stream<20, 480>), dim3(WGRAD_SLABS), dim3(256), shm, st, dy, x, ws, B, T, bn);
            else hipLaunchKernelGGL(conv3x3_wgrad_c1, dim3(WGRAD_SLABS), dim3(256), shm, st, dy, x, ws, B, T, F, Cout, bn);
            A2S_CHECK_LAUNCH("conv3x3_wgrad_c1");
            hipLaunchKernelGGL(wgrad_reduce_c1, dim3(a2s_cdiv(Cout * 9, 256)), dim3(256), 0, st, ws, dW, WGRAD_SLABS, Cout * 9);
            A2S_CHECK_LAUNCH("wgrad_reduce_c1");
            return A2S_OK;
        }
    }
    const bool two = a2s_wgrad_f16x2_enabled() && dy_absmax;          // two fp16 terms: with the operand's max |dy| only
    // round 3: the row-streaming kernel (a2s_conv_wrows.hip) wherever its operand ranges are known
    if (two && !bn_y && a2s_wgrad_rows_eligible(F, Cin, Cout) && (!in_scale || act_absmax))
        return a2s_conv3x3_wgrad_rows_impl(st, dy, x, in_scale, in_shift, dW, ws, ws_bytes, B, T, F, Cin, Cout, dy_absmax, act_absmax);
    // measured at B = 64 (tools/conv_f16x2_check.py; fp32-input kernel / three bf16 terms / two fp16 terms): 40 -> 40: 12.3 / 9.9 / 7.1 ms,
    // 20 -> 40: 6.3 / 7.4 / 5.3 ms, 20 -> 20: 4.6 / 6.2 / 4.6 ms -- the three-term kernel only pays off at 40 -> 40 channels, the two-term
    // one for 40 output channels (wgrad_bf16x3 = 2 / wgrad_f16x2 = 2: every eligible launch regardless)
    const bool split_here = g_wgrad_split > 1 || (Cin == 40 && Cout == 40) || (two && (Cout == 40 || a2s_wgrad_f16x2_enabled() > 1));
    if (g_wgrad_split && !bn_y && Cin > 1 && Cin <= 40 && (Cout == 20 || Cout == 40) && split_here) {
        const int nslabs = 256;               // one 512-thread workgroup per CU
        A2S_REQUIRE(ws_bytes >= (size_t)nslabs * Cout * Cin * 9 * sizeof(float), "conv3x3_wgrad: workspace too small for the split-operand kernel");
        if (two && Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_split<20, 2>), dim3(nslabs), dim3(512), 0, st, dy, x, in_scale, in_shift, ws, B, T, F, Cin, dy_absmax, act_absmax);
        else if (two) hipLaunchKernelGGL((conv3x3_wgrad_split<40, 2>), dim3(nslabs), dim3(512), 0, st, dy, x, in_scale, in_shift, ws, B, T, F, Cin, dy_absmax, act_absmax);
        else if (Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_split<20, 3>), dim3(nslabs), dim3(512), 0, st, dy, x, in_scale, in_shift, ws, B, T, F, Cin, (const float*)nullptr, (const float*)nullptr);
        else hipLaunchKernelGGL((conv3x3_wgrad_split<40, 3>), dim3(nslabs), dim3(512), 0, st, dy, x, in_scale, in_shift, ws, B, T, F, Cin, (const float*)nullptr, (const float*)nullptr);
        A2S_CHECK_LAUNCH("conv3x3_wgrad_split");
        hipLaunchKernelGGL(wgrad_reduce_c1, dim3(a2s_cdiv(Cout * Cin * 9, 256)), dim3(256), 0, st, ws, dW, nslabs, Cout * Cin * 9);
        A2S_CHECK_LAUNCH("wgrad_reduce_c1");
        return A2S_OK;
    }
    const int chunks = (Cin + CV_CK - 1) / CV_CK;
    // persistent workgroups: one full round of the occupancy the kernel reaches (__launch_bounds__(256, 3): 3 per CU for Cout 40,
    // 4 per CU for Cout 20 -- without the bound the compiler spent 200 registers and dropped Cout 40 to 2 per CU)
    const int slabs = Cout == 20 ? 1024 : WGRAD_SLABS;
    dim3 grid(slabs, chunks);
    if (Cout == 20 && !bn_y) hipLaunchKernelGGL((conv3x3_wgrad<20, false>), grid, dim3(256), 0, st, dy, x, in_scale, in_shift, ws, B, T, F, Cin, bn);
    else if (Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad<20, true>), grid, dim3(256), 0, st, dy, x, in_scale, in_shift, ws, B, T, F, Cin, bn);
    else if (Cout == 40 && !bn_y) hipLaunchKernelGGL((conv3x3_wgrad<40, false>), grid, dim3(256), 0, st, dy, x, in_scale, in_shift, ws, B, T, F, Cin, bn);
    else if (Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad<40, true>), grid, dim3(256), 0, st, dy, x, in_scale, in_shift, ws, B, T, F, Cin, bn);
    else A2S_FAIL(A2S_ERR_ARG, "conv3x3_wgrad: Cout must be 20 or 40 (got %d)", Cout);
    A2S_CHECK_LAUNCH("conv3x3_wgrad");
    const int n = Cout * CV_CK * 9 * chunks;
    hipLaunchKernelGGL(wgrad_reduce, dim3(a2s_cdiv(n, 256)), dim3(256), 0, st, ws, dW, slabs, Cout, Cin, chunks);
    A2S_CHECK_LAUNCH("wgrad_reduce");
    return A2S_OK;
}

int a2s_bn_bwd_impl(hipStream_t st, const float* g, const float* x, const float* mean, const float* invstd, const float* scale,
                    const float* shift, const uint8_t* mask, float inv_keep, float* dgamma, float* dbeta, float* dx, float* partial,
                    float* c12, long rows, int C, int F, float* dx_absmax) {
    A2S_REQUIRE(g && x && mean && invstd && scale && shift && dgamma && dbeta && partial && c12, "bn_bwd: null tensor");
    if (dx_absmax && dx) {
        const hipError_t me = hipMemsetAsync(dx_absmax, 0, sizeof(float), st);
        A2S_REQUIRE(me == hipSuccess, "bn_bwd: hipMemsetAsync: %s", hipGetErrorString(me));
    }
    int nblocks;
    if (F > 1) {
        A2S_REQUIRE(!mask, "bn_bwd: dropout mask only supported on the (rows, C) layout");
        nblocks = (int)rows;
        hipLaunchKernelGGL(bn_bwd_reduce_planes, dim3(nblocks), dim3(256), 0, st, g, x, mean, invstd, scale, shift, partial, C, F);
    } else {
        const int rpb = 64;
        nblocks = a2s_cdiv(rows, rpb);
        hipLaunchKernelGGL(bn_bwd_reduce_cols, dim3(nblocks), dim3(256), 0, st, g, x, mean, invstd, scale, shift, mask, inv_keep, partial, rows, C, rpb);
    }
    A2S_CHECK_LAUNCH("bn_bwd_reduce");
    hipLaunchKernelGGL(bn_bwd_finalize, dim3(C), dim3(256), 0, st, partial, nblocks, C, (double)rows * F, dgamma, dbeta, c12);
    A2S_CHECK_LAUNCH("bn_bwd_finalize");
    if (!dx) return A2S_OK;            // statistics only: the input gradient is formed by the consumer (a2s_conv3x3_wgrad_bn)
    const long n = rows * C * F;
    if (F > 1 && F % 4 == 0 && C <= 64 && !mask && ((((uintptr_t)g | (uintptr_t)x | (uintptr_t)dx) & 15) == 0)) {
        hipLaunchKernelGGL(bn_bwd_apply_planes, dim3((unsigned)rows), dim3(256), 0, st, g, x, mean, invstd, scale, shift, c12, dx, C, F, dx_absmax);
        A2S_CHECK_LAUNCH("bn_bwd_apply_planes");
        return A2S_OK;
    }
    hipLaunchKernelGGL(bn_bwd_apply, dim3(min((long)4096, (n + 255) / 256)), dim3(256), 0, st, g, x, mean, invstd, scale, shift, c12, mask,
                       inv_keep, dx, n, C, F, dx_absmax);
    A2S_CHECK_LAUNCH("bn_bwd_apply");
    return A2S_OK;
}

// BatchNorm backward whose statistics partials were produced elsewhere (the data-gradient convolution's epilogue,
// a2s_conv3x3_dgrad_bnstats): finalize (dgamma, dbeta, c12) + apply.  (rows, C, F) layout only.
int a2s_bn_bwd_from_partial_impl(hipStream_t st, const float* g, const float* x, const float* mean, const float* invstd, const float* scale,
                                 const float* shift, float* dgamma, float* dbeta, float* dx, const float* partial, int nblocks, float* c12,
                                 long rows, int C, int F, float* dx_absmax) {
    A2S_REQUIRE(g && x && mean && invstd && scale && shift && dgamma && dbeta && partial && c12 && nblocks > 0, "bn_bwd_from_partial: null tensor");
    if (dx_absmax && dx) {
        const hipError_t me = hipMemsetAsync(dx_absmax, 0, sizeof(float), st);
        A2S_REQUIRE(me == hipSuccess, "bn_bwd_from_partial: hipMemsetAsync: %s", hipGetErrorString(me));
    }
    hipLaunchKernelGGL(bn_bwd_finalize, dim3(C), dim3(256), 0, st, partial, nblocks, C, (double)rows * F, dgamma, dbeta, c12);
    A2S_CHECK_LAUNCH("bn_bwd_finalize");
    if (!dx) return A2S_OK;
    const long n = rows * C * F;
    if (F > 1 && F % 4 == 0 && C <= 64 && ((((uintptr_t)g | (uintptr_t)x | (uintptr_t)dx) & 15) == 0)) {
        hipLaunchKernelGGL(bn_bwd_apply_planes, dim3((unsigned)rows), dim3(256), 0, st, g, x, mean, invstd, scale, shift, c12, dx, C, F, dx_absmax);
        A2S_CHECK_LAUNCH("bn_bwd_apply_planes");
        return A2S_OK;
    }
    hipLaunchKernelGGL(bn_bwd_apply, dim3(min((long)4096, (n + 255) / 256)), dim3(256), 0, st, g, x, mean, invstd, scale, shift, c12,
                       (const uint8_t*)nullptr, 1.f, dx, n, C, F, dx_absmax);
    A2S_CHECK_LAUNCH("bn_bwd_apply");
    return A2S_OK;
}

// ---- split form for synchronised BatchNorm (statistics exchanged between ranks by the host between the two calls)
// sums[c] = {s1, s2} of THIS rank (fixed-order double reduction of the partials)
__global__ __launch_bounds__(256) void bn_bwd_finalize_sums(const float* __restrict__ partial, int nblocks, int C, float* __restrict__ sums) {
    const int c = blockIdx.x;
    __shared__ double r1[256], r2[256];
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        s1 += (double)partial[((long)i * C + c) * 2 + 0];
        s2 += (double)partial[((long)i * C + c) * 2 + 1];
    }
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[2 * c] = (float)r1[0]; sums[2 * c + 1] = (float)r2[0]; }
}
// dgamma/dbeta += LOCAL sums (what torch.nn.SyncBatchNorm does: parameter gradients are rank-local, DDP averages them);
// c12 = GLOBAL sums / global count (the input gradient sees the statistics of the whole global batch)
__global__ void bn_bwd_c12_from_sums(const float* __restrict__ local, const float* __restrict__ global, double count, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, float* __restrict__ c12, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] += local[2 * c]; dgamma[c] += local[2 * c + 1];
    c12[2 * c] = (float)((double)global[2 * c] / count); c12[2 * c + 1] = (float)((double)global[2 * c + 1] / count);
}

int a2s_bn_bwd_stats_impl(hipStream_t st, const float* g, const float* x, const float* mean, const float* invstd, const float* scale,
                          const float* shift, const uint8_t* mask, float inv_keep, float* partial, float* sums, long rows, int C, int F) {
    A2S_REQUIRE(g && x && mean && invstd && scale && shift && partial && sums, "bn_bwd_stats: null tensor");
    int nblocks;
    if (F > 1) {
        A2S_REQUIRE(!mask, "bn_bwd_stats: dropout mask only supported on the (rows, C) layout");
        nblocks = (int)rows;
        hipLaunchKernelGGL(bn_bwd_reduce_planes, dim3(nblocks), dim3(256), 0, st, g, x, mean, invstd, scale, shift, partial, C, F);
    } else {
        const int rpb = 64;
        nblocks = a2s_cdiv(rows, rpb);
        hipLaunchKernelGGL(bn_bwd_reduce_cols, dim3(nblocks), dim3(256), 0, st, g, x, mean, invstd, scale, shift, mask, inv_keep, partial, rows, C, rpb);
    }
    A2S_CHECK_LAUNCH("bn_bwd_reduce");
    hipLaunchKernelGGL(bn_bwd_finalize_sums, dim3(C), dim3(256), 0, st, partial, nblocks, C, sums);
    A2S_CHECK_LAUNCH("bn_bwd_finalize_sums");
    return A2S_OK;
}

int a2s_bn_bwd_apply_impl(hipStream_t st, const float* g, const float* x, const float* mean, const float* invstd, const float* scale,
                          const float* shift, const uint8_t* mask, float inv_keep, const float* sums_local, const float* sums_global,
                          double count_global, float* dgamma, float* dbeta, float* dx, float* c12, long rows, int C, int F) {
    A2S_REQUIRE(g && x && sums_local && sums_global && dgamma && dbeta && dx && c12, "bn_bwd_apply: null tensor");
    hipLaunchKernelGGL(bn_bwd_c12_from_sums, dim3(a2s_cdiv(C, 256)), dim3(256), 0, st, sums_local, sums_global, count_global, dgamma, dbeta, c12, C);
    A2S_CHECK_LAUNCH("bn_bwd_c12_from_sums");
    const long n = rows * C * F;
    hipLaunchKernelGGL(bn_bwd_apply, dim3(min((long)4096, (n + 255) / 256)), dim3(256), 0, st, g, x, mean, invstd, scale, shift, c12, mask,
                       inv_keep, dx, n, C, F, (float*)nullptr);
    A2S_CHECK_LAUNCH("bn_bwd_apply");
    return A2S_OK;
}

// The two halves again for statistics that were already reduced into per-block partials by the kernel that produced g (the data-gradient
// convolutions' and the Linear data gradient's epilogues): (1) partials -> this rank's sums; [host: all-reduce]; (2) dgamma / dbeta += local sums,
// c12 from the global sums -- no pass over (g, x) at all, the input gradient is then formed by the fused consumer (a2s_conv3x3_wgrad_bn_ranged).
int a2s_bn_bwd_sums_from_partial_impl(hipStream_t st, const float* partial, int nblocks, int C, float* sums) {
    A2S_REQUIRE(partial && sums && nblocks > 0 && C > 0, "bn_bwd_sums_from_partial: null tensor");
    hipLaunchKernelGGL(bn_bwd_finalize_sums, dim3(C), dim3(256), 0, st, partial, nblocks, C, sums);
    A2S_CHECK_LAUNCH("bn_bwd_finalize_sums");
    return A2S_OK;
}
int a2s_bn_bwd_c12_from_sums_impl(hipStream_t st, const float* sums_local, const float* sums_global, double count_global, float* dgamma, float* dbeta,
                                  float* c12, int C) {
    A2S_REQUIRE(sums_local && sums_global && dgamma && dbeta && c12 && count_global > 0, "bn_bwd_c12_from_sums: null tensor");
    hipLaunchKernelGGL(bn_bwd_c12_from_sums, dim3(a2s_cdiv(C, 256)), dim3(256), 0, st, sums_local, sums_global, count_global, dgamma, dbeta, c12, C);
    A2S_CHECK_LAUNCH("bn_bwd_c12_from_sums");
    return A2S_OK;
}

size_t a2s_bn_bwd_partial_floats_impl(long rows, int C, int F) {
    const long nblocks = F > 1 ? rows : (rows + 63) / 64;
    return (size_t)nblocks * C * 2;
}
