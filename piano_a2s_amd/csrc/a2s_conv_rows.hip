// Row-streaming 3x3 convolution for gfx950 (reference models.py:475-502,525-534): the forward and data-gradient
// convolutions of the ConvStack since round 3 (a2s_debug_set("conv_rows", 0) selects the tiled kernels of a2s_conv.hip).
//
// What round 2's tiled kernel (conv3x3_split) paid for: a 4 x 64 tile re-staged 6/4 x 66/64 = 1.55x its input per 16-channel chunk
// through a register transpose (32 scalar loads per thread and stage: the MFMA wants 8 CHANNELS per lane, memory has F contiguous),
// K = (tap, channel) padded 360 -> 416, and four barrier-separated phases per stage that did not overlap (profiles/r02_conv_analysis.txt).
//
// This kernel:
//   * N = (df, co), K = (dt, ci).  D[p][(df, co)] = sum_{dt, ci} x[ci][t + dt - 1][p] w[co][ci][dt][df] is a plain GEMM over INPUT
//     positions p (no column shift inside the operand), and out[f][co] = D[f-1][0,co] + D[f][1,co] + D[f+1][2,co] is formed in the
//     epilogue.  Padding: K = 3 Cin = 120 -> 128, N = 3 Cout = 120 -> 128 (5 output channels x 3 df + 1 idle column per 16-wide
//     n-tile): 82 % of the issued MFMA work is algorithmic (72 % before).
//   * No transpose on the way in.  The fp16 operand image in LDS is [channel][position] -- what a 16-byte global load of 4 positions
//     of one channel gives -- and the A fragments are read with ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group
//     reads a [4 channels][16 positions] block and every lane receives the 4 channels of ITS position).  The k index of a lane is
//     (read r, element e) of 4-channel block kb = 8 s + 4 r + (lane >> 4); consecutive blocks sit on disjoint banks (row stride 288 B).
//   * Every input row is loaded and converted ONCE.  A workgroup owns 120 output columns of one clip and walks down T; the rows live
//     in a ring of 5 LDS slots (fp32 -> BatchNorm + ReLU of the producer -> two fp16 terms under per-channel power-of-two scales).
//     Halo: 8 of 128 staged columns (was 55 %), rows: 2 per strip of ~600.
//   * The weights never touch LDS in the main loop: a wave owns 2 n-tiles (1 for Cout = 20) and keeps their B fragments of all
//     k-steps in registers (64 VGPRs) for the whole kernel.
//   * 8 waves = 2 row parities x 4 n-groups; the two waves of a SIMD are one of each parity.  An iteration produces two output rows
//     in two barrier-separated phases: while the even-row waves multiply, the odd-row waves run the epilogue of their previous row
//     (df-combination with DPP row shifts, scale, store, batch statistics), then the roles swap -- matrix pipe beside VALU / memory
//     on every SIMD, by construction instead of by luck.  The odd-row waves hold the NEGATED weights (sign undone in the epilogue):
//     the matrix pipe's internal truncation bias (round 1: -3e-9 sum|a||b|, always negative) cancels between neighbouring rows.
//
// Operand representation (two fp16 terms, three products, a2s_common.h split2_pair_f16) and its scales -- all exact powers of two:
//   * activation operand (forward): channel c is scaled by 2^k_c with k_c = 14 - exponent(|scale_c| max|x_c| + |shift_c|), a HARD bound
//     of the activation relu(scale_c x + shift_c) given the producer's per-channel max|x_c| (`in_absmax`, written by the kernel that
//     produced x): nothing is ever clamped, whatever BatchNorm's gamma is; 2^-k_c is folded into the weights of input channel c;
//   * gradient operand (data gradient): 2^(14 - exponent(max|dy|)) from the device scalar the producer of dy reduced;
//   * weights: output channel co by 2^(13 - exponent(max_{ci,tap} |w 2^-k_ci|)), undone per lane in the epilogue.
//   An element x of an operand whose scaled magnitude is below 2^-3 has an absolute error of 2^-25 (second term subnormal): relative to
//   the channel's (tensor's) largest magnitude that is 2^-39; everything above carries 22 significand bits.
#include "a2s_common.h"
#include <type_traits>

#define RW_P 120             // output columns per workgroup
#define RW_SROW 288          // bytes per channel row of the fp16 image: 128 positions + 32 B (rows 8 banks apart)
#define RW_SLOTS 5
#ifndef RW_EB
#define RW_EB 4              // row quads per epilogue batch (lane permutes in flight together)
#endif
#define RW_HDR 160           // header floats behind the packed weight image: asc[40], ash[40], unsc[40], xscale, ...

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define RW_LDS(p) ((__attribute__((address_space(3))) s16x4*)(p))

template <int CIN>
struct RwGeom {
    static constexpr int NB = CIN / 4;                 // 4-channel blocks per input row
    static constexpr int KS = (3 * NB + 7) / 8;        // k-steps of 32 (8 blocks)
    static constexpr int TS = CIN * RW_SROW;           // bytes per term plane of a slot
    static constexpr int SLOT = 2 * TS + (CIN % 8 ? 128 : 0);   // CIN = 20: consecutive rows' slots offset by 32 banks (block cb = 4 -> cb = 0 across dt)
    static constexpr int ITEMS = CIN * 32;             // 16-byte items per staged row
    static constexpr int XIT = (ITEMS + 255) / 256;   // per thread of the 4 waves (256 threads) that stage a row
};

struct RowsArgs {
    const float* x; float* y;
    const unsigned char* wimg; const float* hdr;
    float* stat_partial; float* out_absmax;
    const float* yl; const float* yl_mean; const float* yl_invstd; const float* yl_scale; const float* yl_shift;
    int B, T, F, tilesF, nstrips, strip_len, nwork;
};

// block (dt, cb) of k-block index kb; the padding blocks alias real ones of the right bank parity (their weights are zero)
template <int CIN>
__host__ __device__ inline void rw_block(int kb, int& dt, int& cb) {
    constexpr int NB = CIN / 4;
    if (kb >= 3 * NB) kb = (CIN % 8) ? 3 * NB - 2 : kb - 2;      // CIN = 20: kb 15 -> (2, 3); CIN = 40: kb 30, 31 -> (2, 8), (2, 9)
    dt = kb / NB; cb = kb % NB;
}

// ------------------------------------------------------------------------------------------- weight packing
// One block per n-tile (5 output channels x 3 df + 1 idle column).  Writes the B fragments in register order:
// wimg[((jt * KS + s) * 2 + term) * 1024 + lane * 16 + (4 r + e) * 2], and (block 0) the header.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void rows_pack(const float* __restrict__ w, int flip, const float* __restrict__ in_scale,
                                                 const float* __restrict__ in_shift, const float* __restrict__ in_absmax,
                                                 const float* __restrict__ x_absmax, unsigned char* __restrict__ wimg, float* __restrict__ hdr,
                                                 float* __restrict__ out_absmax) {
    using G = RwGeom<CIN>;
    __shared__ int kci[CIN];
    __shared__ float red[16];
    __shared__ int kw[5];
    const int jt = blockIdx.x, tid = threadIdx.x;
    if (tid < CIN) {
        int k = 0;
        if (in_scale) {
            const float bound = fabsf(in_scale[tid]) * (in_absmax ? in_absmax[tid] : 1.f) + fabsf(in_shift[tid]);
            k = pow2_scale_exp(bound, 14);
        }
        kci[tid] = k;
        if (jt == 0) {
            hdr[tid] = in_scale ? ldexpf(in_scale[tid], k) : 1.f;
            hdr[40 + tid] = in_scale ? ldexpf(in_shift[tid], k) : 0.f;
        }
    }
    if (jt == 0 && tid == 0) hdr[120] = ldexpf(1.f, (!in_scale && x_absmax) ? pow2_scale_exp(*x_absmax, 14) : 0);
    if (jt == 0 && out_absmax && tid < COUT) out_absmax[tid] = 0.f;
    __syncthreads();
    auto weight = [&](int co, int ci, int tap) -> float {
        const float wv = flip ? w[((long)ci * COUT + co) * 9 + (8 - tap)] : w[((long)co * CIN + ci) * 9 + tap];
        return ldexpf(wv, -kci[ci]);
    };
    for (int c5 = 0; c5 < 5; ++c5) {
        const int co = jt * 5 + c5;
        float m = 0.f;
        for (int e = tid; e < CIN * 9; e += 256) m = fmaxf(m, fabsf(weight(co, e / 9, e % 9)));
        m = block_max(m, red);
        if (tid == 0) {
            kw[c5] = pow2_scale_exp(m, 13);
            hdr[80 + co] = ldexpf(1.f, -kw[c5]);
        }
        __syncthreads();
    }
    for (int idx = tid; idx < G::KS * 512; idx += 256) {          // (s, lane, e8): one fp16 pair (both terms) each
        const int s = idx / 512, lane = (idx >> 3) & 63, e8 = idx & 7;
        const int li = lane & 15, g = lane >> 4, r = e8 >> 2, e = e8 & 3;
        const int kb = 8 * s + 4 * r + g;
        float v = 0.f;
        if (li < 15 && kb < 3 * G::NB) {
            const int c5 = li / 3, df = li % 3, dt = kb / G::NB, ci = 4 * (kb % G::NB) + e;
            v = ldexpf(weight(jt * 5 + c5, ci, dt * 3 + df), kw[c5]);
        }
        unsigned t0, t1;
        split2_pair_f16(v, 0.f, t0, t1);
        unsigned short* dst = reinterpret_cast<unsigned short*>(wimg + ((size_t)(jt * G::KS + s) * 2) * 1024 + lane * 16 + e8 * 2);
        dst[0] = (unsigned short)t0;
        dst[512] = (unsigned short)t1;
    }
}

// per-channel max |x| of a (rows, C, F) tensor: the fallback producer of `in_absmax` for callers that do not have it (tests, old API)
__global__ __launch_bounds__(256) void rows_channel_absmax(const float* __restrict__ x, long rows, int C, int F, float* __restrict__ out) {
    __shared__ float red[16];
    const int c = blockIdx.x % C;
    float m = 0.f;
    for (long r = blockIdx.x / C; r < rows; r += gridDim.x / C) {
        const float* p = x + (r * C + c) * F;
        for (int f = threadIdx.x; f < F; f += 256) m = fmaxf(m, fabsf(p[f]));
    }
    block_absmax_to(out + c, m, red);
}
__global__ void rows_zero(float* p, int n) { if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f; }
// out[c] = max |x[:, c, :]| (slow path: callers that did not get the ranges from the producing kernel)
int a2s_channel_absmax_impl(hipStream_t st, const float* x, long rows, int C, int F, float* out) {
    A2S_REQUIRE(C <= 64, "channel_absmax: at most 64 channels");
    hipLaunchKernelGGL(rows_zero, dim3(1), dim3(64), 0, st, out, C);
    const long per = (rows + 63) / 64;
    hipLaunchKernelGGL(rows_channel_absmax, dim3(C * (int)(per < 2048 ? per : 2048)), dim3(256), 0, st, x, rows, C, F, out);
    A2S_CHECK_LAUNCH("rows_channel_absmax");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- main kernel
#ifdef RW_TRACE
// timing instrumentation (tools/conv_rows_trace.py; never in the product build): shader-clock stamps of lane 0 of one even-row and one
// odd-row wave of a few mid-grid workgroups, 8 stamps per iteration
#define RW_TRACE_WGS 8
#define RW_TRACE_ITERS 32
__device__ unsigned long long rw_trace[RW_TRACE_WGS * 2 * RW_TRACE_ITERS * 8];
extern "C" int a2s_rows_trace_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rw_trace), sizeof(rw_trace)); }
#define RW_STAMP(k)                                                                                                           \
    do {                                                                                                                      \
        if (trace_wg >= 0 && trace_it >= 0 && trace_it < RW_TRACE_ITERS && ng == 0 && lane == 0)                              \
            rw_trace[((trace_wg * 2 + rp) * RW_TRACE_ITERS + trace_it) * 8 + (k)] = __builtin_readcyclecounter();             \
    } while (0)
#else
#define RW_STAMP(k) do {} while (0)
#endif
__device__ __forceinline__ float rw_shr1(float x) {       // lane li of a 16-lane row receives lane li - 1's value
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float rw_shl1(float x) {       // ... lane li + 1's
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x101, 0xf, 0xf, true));
}

template <int CIN, int COUT, bool AFFINE, bool BNRED>
__global__ __launch_bounds__(512, 2) void conv3x3_rows(RowsArgs a) {
    using G = RwGeom<CIN>;
    constexpr int NJ = COUT / 20;             // n-tiles per wave
    constexpr int KS = G::KS, XIT = G::XIT;
    constexpr int MG = (NJ == 1) ? 4 : 2;     // m-tiles per fragment group
    // Where the second fp16 term of the weights and (data gradient) the yl rows of the statistics epilogue live -- a register / LDS trade:
    //   YLDS (data gradient with two n-tiles per wave): yl is fetched by LDS-DMA into a private slice while the wave multiplies (the 64
    //        registers it would need in the epilogue do not exist) and both weight terms stay in registers;
    //   otherwise the second weight term is re-read from LDS per k-step (32 registers saved) and yl, if needed, goes through registers.
    constexpr bool YLDS = false;          // (measured: both forms fit the 256 registers of the 40 -> 40 data gradient only just; registers: 2 spills, LDS-DMA: 11)
    constexpr bool B1L = !YLDS;
    __shared__ __attribute__((aligned(16))) unsigned char ring[RW_SLOTS * G::SLOT];
    // second fp16 term of the weights, in fragment order [n-tile][k-step][lane x 16 B] (the first term lives in registers; this one is
    // needed by one product in three and re-read per row: 8 ds_read_b128 against 128 transposing reads of the activations)
    __shared__ __attribute__((aligned(16))) unsigned char b1img[B1L ? (COUT / 5) * G::KS * 1024 : 16];
    __shared__ __attribute__((aligned(16))) unsigned char ylbuf[YLDS ? 8 * NJ * 5 * 32 * 16 : 16];      // [wave][channel 5 NJ][32 x 4 positions]
    __shared__ float red[8][NJ * 5][3];
    __shared__ float tab[(AFFINE ? 2 * CIN : 0) + (BNRED ? 4 * COUT : 0) + 4];      // [asc | ash] or [mean | invstd | scale | shift]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, g = lane >> 4;
    const int rp = __builtin_amdgcn_readfirstlane(wave >> 2), ng = __builtin_amdgcn_readfirstlane(wave & 3);      // wave-uniform roles

    int bid = blockIdx.x;
    {   // XCD-aware order (workgroup ids go round-robin to the 8 XCDs): every XCD gets a contiguous run of logical tiles, so the
        // column tiles of a row strip -- which share cache lines at their edges -- run on one L2 at the same time
        const int per = (int)gridDim.x / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    const int ft = bid % a.tilesF; bid /= a.tilesF;
    const int strip = bid % a.nstrips, b = bid / a.nstrips;
    const int f_base = ft * RW_P;
    const int t_lo = strip * a.strip_len, t_hi = min(a.T, t_lo + a.strip_len);

    // ---- B fragments: resident in registers
    const unsigned sgn = rp ? 0x80008000u : 0u;          // odd-row waves multiply by the negated weights
    s16x8 bw[NJ][KS];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 v = *reinterpret_cast<const uint4*>(a.wimg + ((size_t)((ng * NJ + j) * KS + s) * 2) * 1024 + lane * 16);
            v.x ^= sgn; v.y ^= sgn; v.z ^= sgn; v.w ^= sgn;
            bw[j][s] = __builtin_bit_cast(s16x8, v);
        }
    s16x8 bw1[B1L ? 1 : NJ][B1L ? 1 : KS];
    if (B1L) {
        for (int e = tid; e < (COUT / 5) * KS * 64; e += 512)
            *reinterpret_cast<uint4*>(b1img + (size_t)e * 16) = *reinterpret_cast<const uint4*>(a.wimg + ((size_t)(e >> 6) * 2 + 1) * 1024 + (e & 63) * 16);
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                uint4 v = *reinterpret_cast<const uint4*>(a.wimg + ((size_t)((ng * NJ + j) * KS + s) * 2 + 1) * 1024 + lane * 16);
                v.x ^= sgn; v.y ^= sgn; v.z ^= sgn; v.w ^= sgn;
                bw1[B1L ? 0 : j][B1L ? 0 : s] = __builtin_bit_cast(s16x8, v);
            }
    }
    // ---- per-lane epilogue constants: column li = 3 c5 + df of n-tile j; lanes with df == 1 finish the output channel
    const bool useful = (q % 3 == 1) && q < 15;
    float unsc[NJ];
    const float xscale = a.hdr[120];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int co = (ng * NJ + j) * 5 + q / 3;
        unsc[j] = useful ? a.hdr[80 + co] / xscale * (rp ? -1.f : 1.f) : 0.f;
    }
    if (AFFINE && tid < 2 * CIN) tab[tid] = a.hdr[tid < CIN ? tid : 40 + tid - CIN];
    if (BNRED && tid < 4 * COUT) {
        const float* src = tid < COUT ? a.yl_mean : tid < 2 * COUT ? a.yl_invstd : tid < 3 * COUT ? a.yl_scale : a.yl_shift;
        tab[tid] = src[tid % COUT];
    }
    __syncthreads();
    f32x2 st_s[NJ], st_s2[NJ];            // two partial sums each (packed adds / FMAs)
    float st_m[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { st_s[j] = st_s2[j] = (f32x2){0.f, 0.f}; st_m[j] = 0.f; }

    // ---- A fragment addressing: running byte address (inside the ring) of this lane's 4-channel block for (s, r), for the wave's
    // first input row (dt = 0 of its output row); advanced by 2 slots per iteration modulo the ring
    unsigned aaddr[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int dt, cb;
            rw_block<CIN>(8 * s + 4 * r + g, dt, cb);
            aaddr[s][r] = (unsigned)((rp + dt) * G::SLOT + (4 * cb + q / 4) * RW_SROW + (q % 4) * 8);
        }
    // ---- staging items of this thread: (channel, 4 positions)
    const int clip_rows = a.T;
    const float* __restrict__ xclip = a.x + (long)b * clip_rows * CIN * a.F;
    // A row is staged by the 4 waves that MULTIPLY in the phase it is converted in (256 threads, lt = their thread index): they issued its
    // loads at the end of their preceding epilogue phase -- a full phase of latency cover -- and the waves that run an epilogue never have
    // input-row loads in flight when they fetch yl (vector loads return in order).  Item `it` of a thread: e = lt + 256 it -> channel e >> 5,
    // positions 4 (e & 31) ..: the column group is the same for all items; offsets are re-derived from a laundered lt where they are used
    // (the compiler otherwise keeps several loop-invariant registers per item alive across the multiply, and spills).
    const int lt = ng * 64 + lane;
    const int fcol = f_base - 4 + 4 * (lt & 31);
    const bool colok = fcol >= 0 && fcol < a.F;
    // Input rows through a raw buffer over this clip.  The whole element offset sits in the VGPR offset: a row above / below the clip
    // wraps out of the buffer's range and reads 0; columns outside the row are masked when the row is converted.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xclip), 0, (unsigned)((long)a.T * CIN * a.F * 4), 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto issue = [&](int row, f32x4 (&xr)[XIT]) {
        const int rbase = row * CIN * a.F + f_base - 4;
        int tl = lt;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = tl + 256 * it;
            const int goff = (e >> 5) * a.F + 4 * (e & 31) + rbase;
            xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (colok && e < G::ITEMS) ? goff * 4 : -4, 0, 0));
        }
    };
    // one staged item: BatchNorm + ReLU of the producer (or the gradient's power-of-two scale), two fp16 terms, two 8-byte LDS stores
    auto commit_item = [&](int it, bool ok, unsigned char* base, const f32x4& x) {
        int tl = lt;
        asm volatile("" : "+v"(tl));
        const int e = tl + 256 * it;
        if (e >= G::ITEMS) return;
        const int ch = e >> 5, loff = ch * RW_SROW + (e & 31) * 8;
        f32x4 v = x;
        if (AFFINE) {
            const float sc = tab[ch], sh = tab[CIN + ch];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ok ? fmaxf(fmaf(v[k], sc, sh), 0.f) : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ok ? v[k] * xscale : 0.f;
        }
        uint2 t0, t1;
        split2_pair_f16(v[0], v[1], t0.x, t1.x);
        split2_pair_f16(v[2], v[3], t0.y, t1.y);
        *reinterpret_cast<uint2*>(base + loff) = t0;
        *reinterpret_cast<uint2*>(base + G::TS + loff) = t1;
    };
    auto commit = [&](int row, int slot, const f32x4 (&xr)[XIT]) {
        const bool ok = row >= 0 && row < a.T && colok;
#pragma unroll
        for (int it = 0; it < XIT; ++it) commit_item(it, ok, ring + slot * G::SLOT, xr[it]);
    };

    f32x4 acc[8][NJ];
    // ---- multiply: output row of this wave from the three input rows at the running addresses
    // ... and, between its k-steps, converts the input row `crow` this wave holds in xr into ring slot `cslot` (the VALU / LDS-store work
    // rides in the shadow of the MFMAs: ~3 issue slots per 16-clock MFMA are free)
    auto multiply = [&](bool converts, int crow, int cslot, const f32x4 (&xr)[XIT]) {
        const bool cok = crow >= 0 && crow < a.T && colok;
        unsigned char* const cbase = ring + cslot * G::SLOT;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            s16x8 b1[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (B1L) {
                    uint4 v = *reinterpret_cast<const uint4*>(b1img + ((ng * NJ + j) * KS + s) * 1024 + lane * 16);
                    v.x ^= sgn; v.y ^= sgn; v.z ^= sgn; v.w ^= sgn;
                    b1[j] = __builtin_bit_cast(s16x8, v);
                } else b1[j] = bw1[B1L ? 0 : j][B1L ? 0 : s];
            }
#pragma unroll
            for (int ig = 0; ig < 8 / MG; ++ig) {
                s16x8 av[2][MG];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int ii = 0; ii < MG; ++ii) {
                        const int off = tm * G::TS + (ig * MG + ii) * 32;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(RW_LDS(ring + aaddr[s][0] + off));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(RW_LDS(ring + aaddr[s][1] + off));
                        av[tm][ii] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#define RW_PRODUCT(TA, TB)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                           \
        _Pragma("unroll") for (int ii = 0; ii < MG; ++ii)                                                                    \
            acc[ig * MG + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[TA][ii]), __builtin_bit_cast(f16x8, TB), acc[ig * MG + ii][j], 0, 0, 0);
                RW_PRODUCT(1, bw[j][s])
                RW_PRODUCT(0, b1[j])
                RW_PRODUCT(0, bw[j][s])
#undef RW_PRODUCT
            }
#pragma unroll
            for (int it = 0; it < XIT; ++it)
                if (converts && it * KS / XIT == s) commit_item(it, cok, cbase, xr[it]);
        }
    };
    auto advance = [&]() {      // two rows further down the ring
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned n = aaddr[s][r] + 2 * G::SLOT;
                aaddr[s][r] = min(n, n - RW_SLOTS * G::SLOT);
            }
    };
    // ---- epilogue of output row t (the wave's own accumulators): df-combination, scale, store, statistics
    const int pv_addr = (((g + 3) & 3) * 16 + ((q + 15) & 15)) * 4;      // lane (li - 1, g - 1 mod 4)
    const int nx_addr = (((g + 1) & 3) * 16 + ((q + 1) & 15)) * 4;       // lane (li + 1, g + 1 mod 4)
    // which of its 8 row quads (m-tile i, rows 16 i + 4 g ..) a lane stores: finishing lane, inside the 120 output columns, inside F
    const bool full_tile = f_base + RW_P <= a.F;           // uniform
    unsigned okbits = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = 16 * i + 4 * g;
        if (useful && p >= 4 && p < 4 + RW_P && f_base - 4 + p < a.F) okbits |= 1u << i;
    }
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + (long)b * a.T * COUT * a.F, 0, (unsigned)((long)a.T * COUT * a.F * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ylrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNRED ? a.yl + (long)b * a.T * COUT * a.F : a.x), 0,
                                                                            BNRED ? (unsigned)((long)a.T * COUT * a.F * 4) : 0u, 0x00020000);
    // data-gradient launches: yl at the positions this lane stores (same offsets as its stores), ALL quads of the wave's row, issued at the
    // start of the epilogue phase BEFORE the phase's input-row loads (loads return in order: the epilogue must not queue behind those)
    unsigned char* const ylw = ylbuf + (YLDS ? wave * (NJ * 5 * 32 * 16) : 0);          // this wave's private slice
    // YLDS: yl of output row t for this wave's channels by LDS-DMA -- NJ * 5 channels x 32 items of 16 B, lane-linear in item order
    auto yl_fetch = [&](int t) {
        if (!YLDS || t >= t_hi) return;
        const float* __restrict__ ylrow = a.yl + (((long)b * a.T + t) * COUT + (ng * NJ) * 5) * a.F;
#pragma unroll
        for (int it = 0; it < (NJ * 5 * 32 + 63) / 64; ++it) {
            const int e = it * 64 + lane, c = e >> 5, p4 = e & 31;
            const int f = f_base - 4 + 4 * p4;
            if (e < NJ * 5 * 32 && f >= 0 && f < a.F)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ylrow + (long)c * a.F + f),
                                                 (__attribute__((address_space(3))) void*)(ylw + it * 1024), 16, 0, 0);
        }
    };
    f32x4 ylv[(BNRED && !YLDS) ? NJ : 1][(BNRED && !YLDS) ? 8 : 1];
    auto yl_issue = [&](int t) {
        if (!BNRED) return;
        if (YLDS) {      // fetched a phase ago by this very wave; nothing else of its vector memory traffic is outstanding by now
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        const int yrow_off = t * COUT * a.F * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int yvo = (((ng * NJ + j) * 5 + q / 3) * a.F + f_base - 4 + 4 * g) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                ylv[(BNRED && !YLDS) ? j : 0][(BNRED && !YLDS) ? i : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ylrsrc, yvo + 64 * i, yrow_off, 0));      // (lanes / quads that store nothing read a harmless in-range or range-checked address)
        }
    };
    auto epilogue = [&](int t) {
        if (t >= t_hi) return;
        const int yrow_off = t * COUT * a.F * 4;                         // uniform: the s-offset of the stores
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = (ng * NJ + j) * 5 + q / 3;
            const int yvo = (co * a.F + f_base - 4 + 4 * g) * 4;         // + 64 i: the instruction's immediate offset
            float bm = 0.f, bi = 0.f, bsc = 0.f, bsh = 0.f;
            if (BNRED && useful) { bm = tab[co]; bi = tab[COUT + co]; bsc = tab[2 * COUT + co]; bsh = tab[3 * COUT + co]; }
            unsigned ob = okbits;                 // laundered: the per-quad store masks / scales are formed here, not kept in 16 registers
            asm volatile("" : "+v"(ob));
            // a workgroup whose 120 columns lie inside F (every one but the last column tile of a ragged F): only the first quad of row
            // group 0 and the last of row group 3 fall outside the output columns
            const float us_first = g > 0 ? unsc[j] : 0.f, us_last = g < 3 ? unsc[j] : 0.f;
            // (1) the cross-group neighbours of all 8 quads first: 16 lane permutes in flight together (rows p - 1 / p + 1 of the
            // neighbouring columns live in another 16-lane group, and in the neighbouring m-tile for g = 0 / 3)
            // (in batches of RW_EB quads, fenced for the instruction scheduler: left alone it hoists every permute of both n-tiles and spills)
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += RW_EB) {
            float X[RW_EB], Y[RW_EB];
#pragma unroll
            for (int ii = 0; ii < RW_EB; ++ii) {
                const int i = i0 + ii;
                const float m3 = (g == 3 && i > 0) ? acc[i > 0 ? i - 1 : 0][j][3] : acc[i][j][3];
                const float m0 = (g == 0 && i < 7) ? acc[i < 7 ? i + 1 : 7][j][0] : acc[i][j][0];
#ifdef RW_X_NOPERM
                X[ii] = m3; Y[ii] = m0;
#else
                X[ii] = __int_as_float(__builtin_amdgcn_ds_bpermute(pv_addr, __float_as_int(m3)));
                Y[ii] = __int_as_float(__builtin_amdgcn_ds_bpermute(nx_addr, __float_as_int(m0)));
#endif
            }
            // (2) per quad, every lane (the DPP row shifts read all lanes): out = D0[p - 1] + D1[p] + D2[p + 1], scaled; lanes that do
            // not finish a channel carry unsc = 0
#pragma unroll
            for (int ii = 0; ii < RW_EB; ++ii) {
                const int i = i0 + ii;
                const f32x4 v = acc[i][j];
                // quads / lanes that store nothing come out as exact zeros: the statistics need no mask
                const bool ok = full_tile ? (useful && (i == 0 ? g > 0 : i == 7 ? g < 3 : true)) : bool(ob >> i & 1);
                const float us = full_tile ? (i == 0 ? us_first : i == 7 ? us_last : unsc[j]) : ((ob >> i & 1) ? unsc[j] : 0.f);
                f32x4 o;
                o[0] = ((X[ii] + v[0]) + rw_shl1(v[1])) * us;
                o[1] = ((rw_shr1(v[0]) + v[1]) + rw_shl1(v[2])) * us;
                o[2] = ((rw_shr1(v[1]) + v[2]) + rw_shl1(v[3])) * us;
                o[3] = ((rw_shr1(v[2]) + v[3]) + Y[ii]) * us;
#ifndef RW_X_NOSTORE
                if (ok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, yvo + 64 * i, yrow_off, 0);
#else
                if (o[0] == 123.f && o[1] == 7.f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, yvo + 64 * i, yrow_off, 0);
#endif
                if (BNRED) {
                    const f32x4 xv = YLDS ? *reinterpret_cast<const f32x4*>(ylw + ((j * 5 + q / 3) * 32 + 4 * i + g) * 16) : ylv[(BNRED && !YLDS) ? j : 0][(BNRED && !YLDS) ? i : 0];
                    f32x4 gm, xh;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gm[k] = (fmaf(xv[k], bsc, bsh) > 0.f) ? o[k] : 0.f;
                        xh[k] = fmaf(xv[k], bi, -bm * bi);
                    }
                    st_s[j] += (f32x2){gm[0], gm[1]}; st_s[j] += (f32x2){gm[2], gm[3]};
                    st_s2[j] += (f32x2){gm[0], gm[1]} * (f32x2){xh[0], xh[1]}; st_s2[j] += (f32x2){gm[2], gm[3]} * (f32x2){xh[2], xh[3]};
                    // max |g| (round 4: the range of the BatchNorm backward fused into the next weight gradient).  ONE running maximum per lane, kept in
                    // st_m[0] whatever the n-tile: its consumer takes the maximum over the channels anyway, and a register per n-tile costs the
                    // 20 -> 20 instance its fourth wave per SIMD (130 registers: 9.0 -> 11.8 ms)
                    // (the 20 -> 20 data gradient -- conv2's, whose consumer below is conv1's own kernel -- does not track it at all: see the launcher)
                    if (!(CIN == 20 && COUT == 20) && a.out_absmax) st_m[0] = fmaxf(fmaxf(st_m[0], fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                } else {
#ifndef RW_X_NOSTATS
                    st_s[j] += (f32x2){o[0], o[1]}; st_s[j] += (f32x2){o[2], o[3]};
                    st_s2[j] += (f32x2){o[0], o[1]} * (f32x2){o[0], o[1]}; st_s2[j] += (f32x2){o[2], o[3]} * (f32x2){o[2], o[3]};
                    st_m[j] = fmaxf(fmaxf(st_m[j], fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
#endif
                }
                // pin the accumulation to this quad (left alone, the adds sink below all 16 quads and keep 64 output registers alive)
                if (BNRED && CIN == 20 && COUT == 20) asm volatile("" : "+v"(st_s[j]), "+v"(st_s2[j]));
                else asm volatile("" : "+v"(st_s[j]), "+v"(st_s2[j]), "+v"(st_m[BNRED ? 0 : j]));
            }
            __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue: rows t_lo - 1 .. t_lo + 2 into slots 0 .. 3 (even-row waves: slots 0, 2; odd-row waves: 1, 3)
    f32x4 xa[XIT];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        issue(t_lo - 1 + 2 * r + rp, xa);
        commit(t_lo - 1 + 2 * r + rp, 2 * r + rp, xa);
    }
    issue(t_lo + 3 + rp, xa);                      // the row this wave converts during the first iteration
    __syncthreads();

    // Per iteration (two output rows, t by the even-row waves and t + 1 by the odd-row waves):
    //   M: every wave multiplies its row (two waves per SIMD keep the matrix pipe fed and cover each other's LDS latency); the even-row
    //      waves convert input row t + 3 into the one free ring slot between their k-steps;
    //   E: the odd-row waves convert row t + 4 into the slot row t - 1 occupied (free once every wave has left M), every wave runs its
    //      epilogue (df-combination, scale, store, statistics) and issues the loads of the row it converts in the next iteration.
    // (Measured first, then dropped: even-row waves multiplying WHILE the odd-row waves run their epilogue and vice versa.  A wave issuing
    // back-to-back MFMAs leaves its SIMD partner ~2 issue slots per MFMA: the epilogue took 6.5 k clocks beside a multiplying partner and
    // 1.5 k alone, while a lone multiplying wave ran at 27 clocks per MFMA -- profiles/r03_conv_rows_ablation.txt.)
    int slot_a = 4, slot_b = 0;
#ifdef RW_TRACE
    const int trace_first = (int)gridDim.x / 2;
    const int trace_wg = ((int)blockIdx.x >= trace_first && (int)blockIdx.x < trace_first + RW_TRACE_WGS) ? (int)blockIdx.x - trace_first : -1;
#endif
    for (int t = t_lo; t < t_hi; t += 2) {
#ifdef RW_TRACE
        const int trace_it = (t - t_lo) / 2 - 20;
#endif
        RW_STAMP(0);
#ifndef RW_X_NOMUL
        multiply(rp == 0, t + 3, slot_a, xa);
#else
        if (rp == 0) commit(t + 3, slot_a, xa);
#endif
        RW_STAMP(1);
        __syncthreads();
        RW_STAMP(2);
        if (rp == 1) commit(t + 4, slot_b, xa);
        RW_STAMP(3);
#ifndef RW_X_NOEPI
        if (t + rp < t_hi) { yl_issue(t + rp); epilogue(t + rp); }
#endif
        RW_STAMP(4);
        issue(t + 5 + rp, xa);
        slot_a = slot_a >= RW_SLOTS - 2 ? slot_a + 2 - RW_SLOTS : slot_a + 2;
        slot_b = slot_b >= RW_SLOTS - 2 ? slot_b + 2 - RW_SLOTS : slot_b + 2;
        advance();
        RW_STAMP(5);
        __syncthreads();
        RW_STAMP(6);
    }

    // ---- statistics: lanes of a column over the 4 row groups, then the two row-parity waves of the n-group
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float s = st_s[j][0] + st_s[j][1], s2 = st_s2[j][0] + st_s2[j][1], m = st_m[j];
        s += __shfl_xor(s, 16, 64); s2 += __shfl_xor(s2, 16, 64); m = fmaxf(m, __shfl_xor(m, 16, 64));
        s += __shfl_xor(s, 32, 64); s2 += __shfl_xor(s2, 32, 64); m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (g == 0 && useful) { red[wave][j * 5 + q / 3][0] = s; red[wave][j * 5 + q / 3][1] = s2; red[wave][j * 5 + q / 3][2] = m; }
    }
    __syncthreads();
    if (tid < COUT) {
        const int grp = tid / (NJ * 5), c = tid % (NJ * 5);       // n-group, channel inside it
        const float s = red[grp][c][0] + red[grp + 4][c][0], s2 = red[grp][c][1] + red[grp + 4][c][1];
        const float m = fmaxf(red[grp][c][2], red[grp + 4][c][2]);
        if (a.stat_partial) {
            a.stat_partial[((long)blockIdx.x * COUT + tid) * 2 + 0] = s;
            a.stat_partial[((long)blockIdx.x * COUT + tid) * 2 + 1] = s2;
        }
        if (a.out_absmax) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __hip_atomic_load(reinterpret_cast<unsigned*>(a.out_absmax + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(reinterpret_cast<unsigned*>(a.out_absmax + tid), bits);
        }
    }
}

// =========================================================================================== rows16
// Second generation of the row-streaming kernel (Cout = 40 launches): the n-tiles are [df][16 output channels] -- tile (cg, df) holds
// D_df of the 16 channels of channel group cg, one channel per lane -- so the df-combination out[p][co] = D0[p-1] + D1[p] + D2[p+1] adds
// three ACCUMULATOR REGISTERS OF THE SAME LANE (the row shift p +- 1 is a register rename inside a lane's 4 rows, one lane permute per
// quad for the rows that cross a 16-lane group), every lane finishes an output channel, and a store writes 64 lanes x 16 B.  The first
// generation (5 channels x 3 df per tile, above) spent 2.9 VALU instructions per MFMA on the combination with 5 of 16 lanes useful.
//   * 40 channels = two groups of 16 (3 tiles each) + one of 8, laid out as 2 tiles ([df0 x 8 | df1 x 8], [df2 x 8 | idle]: D1 comes
//     from lane + 8 by a DPP row rotate): 8 n-tiles, as many as the first generation needs;
//   * 8 waves: waves 0-3 own (group 0 / 1, column half: 3 tiles x 4 m-tiles), waves 4-7 a column QUARTER of group 2 (2 tiles x 2 m-tiles):
//     every SIMD carries 3 x 4 + 2 x 2 = 16 tile products per k-step and term product (192 MFMAs per row);
//   * neighbouring column ranges of a group exchange the one accumulator row that crosses their boundary through a few floats of LDS;
//   * one output row per iteration, ring of 4 slots, ONE barrier per row: all waves multiply (row t + 2 is converted into the free slot
//     between the k-steps), barrier, epilogue + next row's loads;
//   * both weight terms live in registers (96).
// One block per n-tile: 0-5 = (group cg, df) of groups 0 / 1, 6 = [df0 | df1] and 7 = [df2 | idle] of group 2.
// wimg[(tile * KS + s) * 2 + term][lane * 16 + (4 r + e) * 2]; header as rows_pack.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void rows16_pack(const float* __restrict__ w, int flip, const float* __restrict__ in_scale,
                                                   const float* __restrict__ in_shift, const float* __restrict__ in_absmax,
                                                   const float* __restrict__ x_absmax, unsigned char* __restrict__ wimg, float* __restrict__ hdr,
                                                   float* __restrict__ out_absmax) {
    static_assert(COUT == 40, "rows16: 40 output channels");
    using G = RwGeom<CIN>;
    __shared__ int kci[CIN];
    __shared__ float red[16];
    __shared__ int kw[16];
    const int tile = blockIdx.x, tid = threadIdx.x;
    // column li of this tile -> (output channel, df); co < 0: idle column
    auto column = [&](int li, int& co, int& df) {
        if (tile < 6) { co = (tile / 3) * 16 + li; df = tile % 3; }
        else if (tile == 6) { co = 32 + (li & 7); df = li >> 3; }
        else { co = li < 8 ? 32 + li : -1; df = 2; }
    };
    if (tid < CIN) {
        int k = 0;
        if (in_scale) k = pow2_scale_exp(fabsf(in_scale[tid]) * (in_absmax ? in_absmax[tid] : 1.f) + fabsf(in_shift[tid]), 14);
        kci[tid] = k;
        if (tile == 0) {
            hdr[tid] = in_scale ? ldexpf(in_scale[tid], k) : 1.f;
            hdr[40 + tid] = in_scale ? ldexpf(in_shift[tid], k) : 0.f;
        }
    }
    if (tile == 0 && tid == 0) hdr[120] = ldexpf(1.f, (!in_scale && x_absmax) ? pow2_scale_exp(*x_absmax, 14) : 0);
    if (tile == 0 && out_absmax && tid < COUT) out_absmax[tid] = 0.f;
    __syncthreads();
    auto weight = [&](int co, int ci, int tap) -> float {
        const float wv = flip ? w[((long)ci * COUT + co) * 9 + (8 - tap)] : w[((long)co * CIN + ci) * 9 + tap];
        return ldexpf(wv, -kci[ci]);
    };
    for (int c = 0; c < 16; ++c) {          // per-channel weight scale (over all taps: the same in every tile that holds the channel)
        int co, df;
        column(c, co, df);
        float m = 0.f;
        if (co >= 0)
            for (int e = tid; e < CIN * 9; e += 256) m = fmaxf(m, fabsf(weight(co, e / 9, e % 9)));
        m = block_max(m, red);
        if (tid == 0) {
            kw[c] = pow2_scale_exp(m, 13);
            if (co >= 0 && df == 0) hdr[80 + co] = ldexpf(1.f, -kw[c]);
        }
        __syncthreads();
    }
    for (int idx = tid; idx < G::KS * 512; idx += 256) {
        const int s = idx / 512, lane = (idx >> 3) & 63, e8 = idx & 7;
        const int li = lane & 15, g = lane >> 4, r = e8 >> 2, e = e8 & 3;
        const int kb = 8 * s + 4 * r + g;
        int co, df;
        column(li, co, df);
        float v = 0.f;
        if (co >= 0 && kb < 3 * G::NB) v = ldexpf(weight(co, 4 * (kb % G::NB) + e, (kb / G::NB) * 3 + df), kw[li]);
        unsigned t0, t1;
        split2_pair_f16(v, 0.f, t0, t1);
        unsigned short* dst = reinterpret_cast<unsigned short*>(wimg + ((size_t)(tile * G::KS + s) * 2) * 1024 + lane * 16 + e8 * 2);
        dst[0] = (unsigned short)t0;
        dst[512] = (unsigned short)t1;
    }
}

__device__ __forceinline__ float rw_ror8(float x) {        // lane li of a 16-lane row receives lane (li + 8) % 16's value
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));
}

#define R16_SLOTS 4
// PIPE (round 6): the epilogue of output row t - 1 runs UNDER the multiply of row t.  The kernel above spends 3.1 k of its 7.0 k clocks per row
// outside the multiply (epilogue 1.45 k, barrier, next row's load issue: HISTORY.md Part II 3.1) with the matrix pipes idle, and a second
// accumulator set did not fit beside both weight terms in registers (214 / 250 of 256).  Here the second weight term lives in LDS (32 KB in
// fragment order, 3 ds_read_b128 per k-step: it feeds one product in three), which pays for a second accumulator set: the multiply of row t
// fills set t & 1 while the df-combination, scaling, stores and statistics of row t - 1 -- set (t - 1) & 1, complete since the last barrier --
// are issued between its k-steps, one m-tile per k-step.  A wave issues one MFMA per ~32 clocks when two waves share a SIMD: the ~350 vector
// instructions of an epilogue ride in the issue slots in between.  The barrier per row stays (ring slot hand-off, boundary rows).
template <int CIN, int COUT, bool AFFINE, bool BNRED, bool PIPE = false>
__global__ __launch_bounds__(512, 2) void conv3x3_rows16(RowsArgs a) {
    static_assert(COUT == 40, "rows16: 40 output channels");
    using G = RwGeom<CIN>;
    constexpr int NTHR = 512, KS = G::KS;
    constexpr int XIT = (G::ITEMS + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) unsigned char ring[R16_SLOTS * G::SLOT];
    __shared__ __attribute__((aligned(16))) unsigned char b1img[PIPE ? 8 * G::KS * 1024 : 16];      // second fp16 term of the weights: [tile][k-step][lane x 16 B]
    // accumulator rows that cross the boundary between neighbouring column ranges of a channel group, as TRUE values (sign undone):
    // [iteration parity][boundary: 0 / 1 = halves of group 0 / 1, 2..4 = quarters of group 2][0: left range's last row of D0 | 1: right
    // range's first row of D2][channel]
    __shared__ float xch[2][5][2][16];
    __shared__ float red[8][16][3];
    __shared__ float tab[(AFFINE ? 2 * CIN : 0) + (BNRED ? 4 * COUT : 0) + 4];
    // where the threads without an item in the last staging round put their (meaningless) 8 + 8 bytes: the conversion then has no
    // branch, and the scheduler can weave it between the MFMAs
    __shared__ __attribute__((aligned(16))) unsigned char dump[(G::ITEMS % NTHR) ? NTHR * 16 : 16];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, g = lane >> 4;
    const bool wide = wave < 4;                           // waves 0-3: (group wave >> 1, half wave & 1); waves 4-7: quarter wave - 4 of group 2
    const int m0 = wide ? 4 * (wave & 1) : 2 * (wave - 4); // first m-tile of the wave's column range
    const int nm = wide ? 4 : 2;
    const int tile0 = wide ? 3 * (wave >> 1) : 6;
    const bool neg = wave & 1;                            // odd column ranges multiply by the negated weights (the matrix pipe's truncation
    const float sg = neg ? -1.f : 1.f;                    // bias then cancels between neighbouring ranges in every per-channel sum)
    const int bnd_l = wide ? (wave & 1 ? (wave >> 1) : -1) : (wave > 4 ? wave - 3 : -1);      // boundary index to the left / right (-1: tile edge)
    const int bnd_r = wide ? (wave & 1 ? -1 : (wave >> 1)) : (wave < 7 ? wave - 2 : -1);

    int bid = blockIdx.x;
    {
        const int per = (int)gridDim.x / 8;
        if (bid < per * 8) bid = (bid % 8) * per + bid / 8;
    }
    const int ft = bid % a.tilesF; bid /= a.tilesF;
    const int strip = bid % a.nstrips, b = bid / a.nstrips;
    const int f_base = ft * RW_P;
    const int t_lo = strip * a.strip_len, t_hi = min(a.T, t_lo + a.strip_len);

    // ---- weights: both terms of the wave's tiles (3 or 2), all k-steps
    const unsigned sgn = neg ? 0x80008000u : 0u;
    constexpr int BWT = PIPE ? 1 : 2;                       // weight terms kept in registers
    s16x8 bw[3][KS][BWT];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int tm = 0; tm < BWT; ++tm) {
                uint4 v = *reinterpret_cast<const uint4*>(a.wimg + ((size_t)((tile0 + ((wide || d < 2) ? d : 0)) * KS + s) * 2 + tm) * 1024 + lane * 16);
                v.x ^= sgn; v.y ^= sgn; v.z ^= sgn; v.w ^= sgn;
                bw[d][s][tm] = __builtin_bit_cast(s16x8, v);
            }
    if (PIPE) {
        for (int e = tid; e < 8 * KS * 64; e += NTHR)
            *reinterpret_cast<uint4*>(b1img + (size_t)e * 16) = *reinterpret_cast<const uint4*>(a.wimg + ((size_t)(e >> 6) * 2 + 1) * 1024 + (e & 63) * 16);
    }
    // second weight term of tile d, k-step s (this wave's sign)
    auto bterm1 = [&](int d, int s) -> s16x8 {
        if (PIPE) {
            uint4 v = *reinterpret_cast<const uint4*>(b1img + ((tile0 + ((wide || d < 2) ? d : 0)) * KS + s) * 1024 + lane * 16);
            v.x ^= sgn; v.y ^= sgn; v.z ^= sgn; v.w ^= sgn;
            return __builtin_bit_cast(s16x8, v);
        }
        return bw[d][s][BWT - 1];
    };
    const int co = wide ? (wave >> 1) * 16 + q : 32 + (q & 7);      // this lane's output channel
    const bool useful = wide || q < 8;
    const float xscale = a.hdr[120];
    const float unsc = useful ? a.hdr[80 + co] / xscale * sg : 0.f;
    if (AFFINE && tid < 2 * CIN) tab[tid] = a.hdr[tid < CIN ? tid : 40 + tid - CIN];
    if (BNRED && tid < 4 * COUT) {
        const float* src = tid < COUT ? a.yl_mean : tid < 2 * COUT ? a.yl_invstd : tid < 3 * COUT ? a.yl_scale : a.yl_shift;
        tab[tid] = src[tid % COUT];
    }
    __syncthreads();
    float bm = 0.f, bi = 0.f, bsc = 0.f, bsh = 0.f;
    if (BNRED && useful) { bm = tab[co]; bi = tab[COUT + co]; bsc = tab[2 * COUT + co]; bsh = tab[3 * COUT + co]; }
    f32x2 st_s = {0.f, 0.f}, st_s2 = {0.f, 0.f};
    float st_m = 0.f;

    // ---- A fragment addressing: running byte address of this lane's 4-channel block for (s, r), first input row of the output row
    unsigned aaddr[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int dt, cb;
            rw_block<CIN>(8 * s + 4 * r + g, dt, cb);
            aaddr[s][r] = (unsigned)(dt * G::SLOT + (4 * cb + q / 4) * RW_SROW + (q % 4) * 8 + m0 * 32);
        }
    // ---- staging: every thread, items e = tid + NTHR it (channel e >> 5, positions 4 (e & 31) ..)
    const float* __restrict__ xclip = a.x + (long)b * a.T * CIN * a.F;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xclip), 0, (unsigned)((long)a.T * CIN * a.F * 4), 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int fcol = f_base - 4 + 4 * (tid & 31);
    const bool colok = fcol >= 0 && fcol < a.F;          // (NTHR is a multiple of 32: the column group is the same for all items of a thread)
    auto issue = [&](int row, f32x4 (&xr)[XIT]) {
        const int rbase = row * CIN * a.F + f_base - 4;
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = tl + NTHR * it;
            const int goff = (e >> 5) * a.F + 4 * (e & 31) + rbase;
            xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (colok && e < G::ITEMS) ? goff * 4 : -4, 0, 0));
        }
    };
    auto commit_item = [&](int it, bool ok, unsigned char* base, const f32x4& x) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const int e = tl + NTHR * it;
        const bool has = (it + 1) * NTHR <= G::ITEMS || e < G::ITEMS;
        const int ch = has ? e >> 5 : 0;
        unsigned char* const p0 = has ? base + ch * RW_SROW + (e & 31) * 8 : dump + tl * 16;
        unsigned char* const p1 = has ? p0 + G::TS : p0 + 8;
        f32x4 v = x;
        if (AFFINE) {
            const float sc = tab[ch], sh = tab[CIN + ch];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ok ? fmaxf(fmaf(v[k], sc, sh), 0.f) : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ok ? v[k] * xscale : 0.f;
        }
        uint2 t0, t1;
        split2_pair_f16(v[0], v[1], t0.x, t1.x);
        split2_pair_f16(v[2], v[3], t0.y, t1.y);
        *reinterpret_cast<uint2*>(p0) = t0;
        *reinterpret_cast<uint2*>(p1) = t1;
    };

    f32x4 acc[4][3];                                     // [m-tile][tile of the wave]
    // NM m-tiles x NT tiles; the conversion of input row crow (held in xr) rides between the k-steps
    auto multiply = [&](auto NMc, auto NTc, int crow, int cslot, const f32x4 (&xr)[XIT]) {
        constexpr int NM = decltype(NMc)::value, NT = decltype(NTc)::value;
        const bool cok = crow >= 0 && crow < a.T && colok;
        unsigned char* const cbase = ring + cslot * G::SLOT;
#pragma unroll
        for (int i = 0; i < NM; ++i)
#pragma unroll
            for (int d = 0; d < NT; ++d) acc[i][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int ig = 0; ig < NM / 2; ++ig) {
                s16x8 av[2][2];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int off = tm * G::TS + (ig * 2 + ii) * 32;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(RW_LDS(ring + aaddr[s][0] + off));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(RW_LDS(ring + aaddr[s][1] + off));
                        av[tm][ii] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#define R16_PRODUCT(TA, TB)                                                                                                  \
    _Pragma("unroll") for (int d = 0; d < NT; ++d)                                                                           \
        _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                     \
            acc[ig * 2 + ii][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[TA][ii]), __builtin_bit_cast(f16x8, bw[d][s][TB < BWT ? TB : 0]), acc[ig * 2 + ii][d], 0, 0, 0);
                R16_PRODUCT(1, 0)
                R16_PRODUCT(0, 1)
                R16_PRODUCT(0, 0)
#undef R16_PRODUCT
            }
            // (the conversion has no branch: the compiler's scheduler weaves it and the next k-step's fragment reads between the MFMAs.
            // Measured without gain: explicit sched_group_barrier interleaving, hand double-buffered fragments -- 244 registers in the
            // data-gradient instance, slower.)
#pragma unroll
            for (int it = 0; it < XIT; ++it)
                if (it * KS / XIT == s) commit_item(it, cok, cbase, xr[it]);
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const unsigned n = aaddr[s][r] + G::SLOT;
                aaddr[s][r] = min(n, n - R16_SLOTS * G::SLOT);
            }
    };

    // ---- epilogue.  Lane (q, g) holds, per m-tile i, rows p = 16 (m0 + i) + 4 g + r of its column.
    const int up_addr = (((g + 3) & 3) * 16 + q) * 4;     // same column, previous 4-row group (wraps to the previous m-tile's last group)
    const int dn_addr = (((g + 1) & 3) * 16 + q) * 4;     // same column, next 4-row group
    unsigned okbits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = 16 * (m0 + i) + 4 * g;
        if (useful && i < nm && p >= 4 && p < 4 + RW_P && f_base - 4 + p < a.F) okbits |= 1u << i;
    }
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + (long)b * a.T * COUT * a.F, 0, (unsigned)((long)a.T * COUT * a.F * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ylrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNRED ? a.yl + (long)b * a.T * COUT * a.F : a.x), 0,
                                                                            BNRED ? (unsigned)((long)a.T * COUT * a.F * 4) : 0u, 0x00020000);
    const int yvo = (co * a.F + f_base - 4 + 16 * m0 + 4 * g) * 4;       // + 64 i
    // WIDE: D0 / D1 / D2 = tiles 0 / 1 / 2, same lane.  Narrow (group 2): D0 = tile 0 lanes 0-7, D1 = tile 0 lanes 8-15 (row rotate by 8),
    // D2 = tile 1 lanes 0-7.
    // data gradient: yl at this lane's output positions, fetched BEFORE the multiply of the row (in flight under the MFMAs)
    f32x4 ylv[BNRED ? 4 : 1];
    auto yl_fetch = [&](int t) {
        if (!BNRED) return;
        const int yrow_off = t * COUT * a.F * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nm) ylv[BNRED ? i : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ylrsrc, yvo + 64 * i, yrow_off, 0));
    };
    auto epilogue = [&](auto WIDEc, int t, int par) {
        constexpr bool WIDE = decltype(WIDEc)::value;
        constexpr int NM = WIDE ? 4 : 2, D2 = WIDE ? 2 : 1;
        const int yrow_off = t * COUT * a.F * 4;
        // rows across the boundaries to the neighbouring column ranges (true values -> this wave's sign)
        const float xl = bnd_l >= 0 ? xch[par][bnd_l >= 0 ? bnd_l : 0][0][q] * sg : 0.f;
        const float xr_ = bnd_r >= 0 ? xch[par][bnd_r >= 0 ? bnd_r : 0][1][q] * sg : 0.f;
        unsigned ob = okbits;
        asm volatile("" : "+v"(ob));
        float X[NM], Y[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            // row p - 1 of D0 lives in the previous 4-row group (lane - 16; for g = 0: the previous m-tile's group 3, for the first
            // m-tile of the range: the left neighbour's value), row p + 1 of D2 in the next one
            const float up_src = (g == 3) ? (i > 0 ? acc[i > 0 ? i - 1 : 0][0][3] : xl) : acc[i][0][3];
            const float dn_src = (g == 0) ? (i < NM - 1 ? acc[i < NM - 1 ? i + 1 : NM - 1][D2][0] : xr_) : acc[i][D2][0];
            X[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(up_addr, __float_as_int(up_src)));
            Y[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(dn_addr, __float_as_int(dn_src)));
        }
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const bool ok = ob >> i & 1;
            const float us = ok ? unsc : 0.f;
            f32x4 d1;
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[r] = WIDE ? acc[i][1][r] : rw_ror8(acc[i][0][r]);
            f32x4 o;
            o[0] = ((X[i] + d1[0]) + acc[i][D2][1]) * us;
            o[1] = ((acc[i][0][0] + d1[1]) + acc[i][D2][2]) * us;
            o[2] = ((acc[i][0][1] + d1[2]) + acc[i][D2][3]) * us;
            o[3] = ((acc[i][0][2] + d1[3]) + Y[i]) * us;
            if (ok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, yvo + 64 * i, yrow_off, 0);
            if (BNRED) {
                const f32x4 xv = ylv[BNRED ? i : 0];
                f32x4 gm, xh;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gm[k] = (fmaf(xv[k], bsc, bsh) > 0.f) ? o[k] : 0.f;
                    xh[k] = fmaf(xv[k], bi, -bm * bi);
                }
                st_s += (f32x2){gm[0], gm[1]}; st_s += (f32x2){gm[2], gm[3]};
                st_s2 += (f32x2){gm[0], gm[1]} * (f32x2){xh[0], xh[1]}; st_s2 += (f32x2){gm[2], gm[3]} * (f32x2){xh[2], xh[3]};
                if (a.out_absmax) st_m = fmaxf(fmaxf(st_m, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));                  // max |g|: round 4
            } else {
                st_s += (f32x2){o[0], o[1]}; st_s += (f32x2){o[2], o[3]};
                st_s2 += (f32x2){o[0], o[1]} * (f32x2){o[0], o[1]}; st_s2 += (f32x2){o[2], o[3]} * (f32x2){o[2], o[3]};
                st_m = fmaxf(fmaxf(st_m, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            }
        }
    };

    // ---- prologue: rows t_lo - 1, t_lo, t_lo + 1 into slots 0, 1, 2; row t_lo + 2 in flight
    f32x4 xa[XIT];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        issue(t_lo - 1 + r, xa);
#pragma unroll
        for (int it = 0; it < XIT; ++it) commit_item(it, t_lo - 1 + r >= 0 && t_lo - 1 + r < a.T && colok, ring + r * G::SLOT, xa[it]);
    }
    issue(t_lo + 2, xa);
    __syncthreads();

    if constexpr (PIPE) {
        // ======================================================================== two accumulator sets: epilogue of row t - 1 under the multiply of row t
        f32x4 accp[2][4][3];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int d = 0; d < 3; ++d) accp[e][i][d] = (f32x4){0.f, 0.f, 0.f, 0.f};      // (the first row's "previous row" is masked out, but must be finite)
        // one m-tile of the epilogue of row t from accumulator set SET (the arithmetic of `epilogue` above)
        // (data gradient) yl of one m-tile of row t: requested at the START of the k-step whose end consumes it -- ~36 MFMAs of latency cover,
        // 4 registers live instead of the 16 of a whole row (the instance has none to spare)
        auto yl_tile = [&](int i, int t) -> f32x4 {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ylrsrc, yvo + 64 * i, t * COUT * a.F * 4, 0));
        };
        auto epi_tile = [&](auto WIDEc, auto SETc, int i, int t, float xl, float xr_, unsigned ob, const f32x4& ylt) {
            constexpr bool WIDE = decltype(WIDEc)::value;
            constexpr int SET = decltype(SETc)::value, NM = WIDE ? 4 : 2, D2 = WIDE ? 2 : 1;
            const int yrow_off = t * COUT * a.F * 4;
            const float up_src = (g == 3) ? (i > 0 ? accp[SET][i > 0 ? i - 1 : 0][0][3] : xl) : accp[SET][i][0][3];
            const float dn_src = (g == 0) ? (i < NM - 1 ? accp[SET][i < NM - 1 ? i + 1 : NM - 1][D2][0] : xr_) : accp[SET][i][D2][0];
            const float X = __int_as_float(__builtin_amdgcn_ds_bpermute(up_addr, __float_as_int(up_src)));
            const float Y = __int_as_float(__builtin_amdgcn_ds_bpermute(dn_addr, __float_as_int(dn_src)));
            const bool ok = ob >> i & 1;
            const float us = ok ? unsc : 0.f;
            f32x4 d1;
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[r] = WIDE ? accp[SET][i][1][r] : rw_ror8(accp[SET][i][0][r]);
            f32x4 o;
            o[0] = ((X + d1[0]) + accp[SET][i][D2][1]) * us;
            o[1] = ((accp[SET][i][0][0] + d1[1]) + accp[SET][i][D2][2]) * us;
            o[2] = ((accp[SET][i][0][1] + d1[2]) + accp[SET][i][D2][3]) * us;
            o[3] = ((accp[SET][i][0][2] + d1[3]) + Y) * us;
            if (ok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrsrc, yvo + 64 * i, yrow_off, 0);
            if (BNRED) {
                const f32x4 xv = ylt;
                f32x4 gm, xh;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gm[k] = (fmaf(xv[k], bsc, bsh) > 0.f) ? o[k] : 0.f;
                    xh[k] = fmaf(xv[k], bi, -bm * bi);
                }
                st_s += (f32x2){gm[0], gm[1]}; st_s += (f32x2){gm[2], gm[3]};
                st_s2 += (f32x2){gm[0], gm[1]} * (f32x2){xh[0], xh[1]}; st_s2 += (f32x2){gm[2], gm[3]} * (f32x2){xh[2], xh[3]};
                if (a.out_absmax) st_m = fmaxf(fmaxf(st_m, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            } else {
                st_s += (f32x2){o[0], o[1]}; st_s += (f32x2){o[2], o[3]};
                st_s2 += (f32x2){o[0], o[1]} * (f32x2){o[0], o[1]}; st_s2 += (f32x2){o[2], o[3]} * (f32x2){o[2], o[3]};
                st_m = fmaxf(fmaxf(st_m, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            }
        };
        // boundary rows of row t_e (accumulator set 1 - CUR, exchanged before the last barrier) and the store mask: nothing is stored for a row
        // in front of the strip (the first iteration has no previous row: its accumulators are the zeros above)
        auto multiply_p = [&](auto WIDEc, auto CURc, int t_e, int crow, int cslot, const f32x4 (&xr)[XIT]) {
            constexpr bool WIDE = decltype(WIDEc)::value;
            constexpr int CUR = decltype(CURc)::value, NM = WIDE ? 4 : 2, NT = WIDE ? 3 : 2;
            const bool cok = crow >= 0 && crow < a.T && colok;
            unsigned char* const cbase = ring + cslot * G::SLOT;
            const bool ev = t_e >= t_lo;
            const float xl = (ev && bnd_l >= 0) ? xch[1 - CUR][bnd_l >= 0 ? bnd_l : 0][0][q] * sg : 0.f;
            const float xr_ = (ev && bnd_r >= 0) ? xch[1 - CUR][bnd_r >= 0 ? bnd_r : 0][1][q] * sg : 0.f;
            unsigned ob = ev ? okbits : 0u;
            asm volatile("" : "+v"(ob));
#pragma unroll
            for (int i = 0; i < NM; ++i)
#pragma unroll
                for (int d = 0; d < NT; ++d) accp[CUR][i][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                f32x4 ylt[(BNRED && NM > KS) ? NM / KS : 1];          // yl of the m-tiles whose epilogue ends this k-step
                if (BNRED) {
#pragma unroll
                    for (int i = 0; i < NM; ++i)
                        if (i * KS / NM == s) ylt[(NM > KS) ? i - s * NM / KS : 0] = (ev || true) ? yl_tile(i, t_e) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                s16x8 b1[NT];
#pragma unroll
                for (int d = 0; d < NT; ++d) b1[d] = bterm1(d, s);
#pragma unroll
                for (int ig = 0; ig < NM / 2; ++ig) {
                    s16x8 av[2][2];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const int off = tm * G::TS + (ig * 2 + ii) * 32;
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(RW_LDS(ring + aaddr[s][0] + off));
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(RW_LDS(ring + aaddr[s][1] + off));
                            av[tm][ii] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
#define R16P_PRODUCT(TA, BV)                                                                                                 \
    _Pragma("unroll") for (int d = 0; d < NT; ++d)                                                                           \
        _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                     \
            accp[CUR][ig * 2 + ii][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[TA][ii]), __builtin_bit_cast(f16x8, BV), accp[CUR][ig * 2 + ii][d], 0, 0, 0);
                    R16P_PRODUCT(1, bw[d][s][0])
                    R16P_PRODUCT(0, b1[d])
                    R16P_PRODUCT(0, bw[d][s][0])
#undef R16P_PRODUCT
                }
#pragma unroll
                for (int it = 0; it < XIT; ++it)
                    if (it * KS / XIT == s) commit_item(it, cok, cbase, xr[it]);
                // the previous row's epilogue, spread over the k-steps
#pragma unroll
                for (int i = 0; i < NM; ++i)
                    if (i * KS / NM == s) epi_tile(WIDEc, std::integral_constant<int, 1 - CUR>{}, i, t_e, xl, xr_, ob, ylt[(BNRED && NM > KS) ? i - s * NM / KS : 0]);
            }
        };
        int slot_w = 3;
        auto row = [&](auto CURc, int t) {
            constexpr int CUR = decltype(CURc)::value;
            if (wide) multiply_p(std::true_type{}, CURc, t - 1, t + 2, slot_w, xa);
            else multiply_p(std::false_type{}, CURc, t - 1, t + 2, slot_w, xa);
            if (bnd_r >= 0 && g == 3) xch[CUR][bnd_r >= 0 ? bnd_r : 0][0][q] = (wide ? accp[CUR][3][0][3] : accp[CUR][1][0][3]) * sg;
            if (bnd_l >= 0 && g == 0) xch[CUR][bnd_l >= 0 ? bnd_l : 0][1][q] = (wide ? accp[CUR][0][2][0] : accp[CUR][0][1][0]) * sg;
            __syncthreads();
            issue(t + 3, xa);
            slot_w = slot_w == R16_SLOTS - 1 ? 0 : slot_w + 1;
            advance();
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        int t = t_lo, last = 0;
        for (;;) {
            row(C0{}, t);
            if (++t >= t_hi) { last = 0; break; }
            row(C1{}, t);
            if (++t >= t_hi) { last = 1; break; }
        }
        // the last row's epilogue
        {
            const float xl = bnd_l >= 0 ? xch[last][bnd_l >= 0 ? bnd_l : 0][0][q] * sg : 0.f;
            const float xr_ = bnd_r >= 0 ? xch[last][bnd_r >= 0 ? bnd_r : 0][1][q] * sg : 0.f;
            unsigned ob = okbits;
            asm volatile("" : "+v"(ob));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < nm) {
                    const f32x4 ylt = BNRED ? yl_tile(i, t_hi - 1) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (wide) { if (last) epi_tile(std::true_type{}, C1{}, i, t_hi - 1, xl, xr_, ob, ylt); else epi_tile(std::true_type{}, C0{}, i, t_hi - 1, xl, xr_, ob, ylt); }
                    else if (i < 2) { if (last) epi_tile(std::false_type{}, C1{}, i < 2 ? i : 0, t_hi - 1, xl, xr_, ob, ylt); else epi_tile(std::false_type{}, C0{}, i < 2 ? i : 0, t_hi - 1, xl, xr_, ob, ylt); }
                }
            }
        }
    } else {
    int slot_w = 3, par = 0;
#ifdef RW_TRACE
    const int rp = wave & 1, ng = wave >> 1;  // (stamp slots: [workgroup][column half][iteration], waves 0 / 1 = channel group 0)
    const int trace_first = (int)gridDim.x / 2;
    const int trace_wg = ((int)blockIdx.x >= trace_first && (int)blockIdx.x < trace_first + RW_TRACE_WGS) ? (int)blockIdx.x - trace_first : -1;
#endif
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
    for (int t = t_lo; t < t_hi; ++t) {
#ifdef RW_TRACE
        const int trace_it = t - t_lo - 40;
#endif
        RW_STAMP(0);
        yl_fetch(t);
        if (wide) multiply(I4{}, I3{}, t + 2, slot_w, xa);
        else multiply(I2{}, I2{}, t + 2, slot_w, xa);
        RW_STAMP(1);
        // boundary rows for the neighbouring column ranges (lanes of 4-row group 3 / 0 hold them), as true values
        if (bnd_r >= 0 && g == 3) xch[par][bnd_r >= 0 ? bnd_r : 0][0][q] = (wide ? acc[3][0][3] : acc[1][0][3]) * sg;
        if (bnd_l >= 0 && g == 0) xch[par][bnd_l >= 0 ? bnd_l : 0][1][q] = (wide ? acc[0][2][0] : acc[0][1][0]) * sg;
        __syncthreads();
        RW_STAMP(2);
        if (wide) epilogue(std::true_type{}, t, par);
        else epilogue(std::false_type{}, t, par);
        RW_STAMP(3);
        issue(t + 3, xa);
        RW_STAMP(4);
        slot_w = slot_w == R16_SLOTS - 1 ? 0 : slot_w + 1;
        par ^= 1;
        advance();
    }
    }

    // ---- statistics: a column's four 4-row groups, then the column ranges of the group
    {
        float s = st_s[0] + st_s[1], s2 = st_s2[0] + st_s2[1], m = st_m;
        s += __shfl_xor(s, 16, 64); s2 += __shfl_xor(s2, 16, 64); m = fmaxf(m, __shfl_xor(m, 16, 64));
        s += __shfl_xor(s, 32, 64); s2 += __shfl_xor(s2, 32, 64); m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (g == 0) { red[wave][q][0] = s; red[wave][q][1] = s2; red[wave][q][2] = m; }
    }
    __syncthreads();
    if (tid < COUT) {
        float s = 0.f, s2 = 0.f, m = 0.f;
        const int w0 = tid < 32 ? 2 * (tid >> 4) : 4, nw = tid < 32 ? 2 : 4, c = tid < 32 ? tid & 15 : tid - 32;
        for (int k = 0; k < nw; ++k) { s += red[w0 + k][c][0]; s2 += red[w0 + k][c][1]; m = fmaxf(m, red[w0 + k][c][2]); }
        if (a.stat_partial) {
            a.stat_partial[((long)blockIdx.x * COUT + tid) * 2 + 0] = s;
            a.stat_partial[((long)blockIdx.x * COUT + tid) * 2 + 1] = s2;
        }
        if (a.out_absmax) {
            const unsigned bits = __float_as_uint(m);
            if (bits > __hip_atomic_load(reinterpret_cast<unsigned*>(a.out_absmax + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(reinterpret_cast<unsigned*>(a.out_absmax + tid), bits);
        }
    }
}

// ------------------------------------------------------------------------------------------- launcher
static int g_conv_rows = -1;
void a2s_conv_rows_set(int on) { g_conv_rows = on; }
int a2s_conv_rows_enabled(void) {
    // bit 0: the row-streaming kernels; bit 1: their second generation (conv3x3_rows16) where it exists (Cout = 40); bits 2 / 3: its form with two
    // accumulator sets (epilogue under the next row's multiply) for the forward / data-gradient launches
    if (g_conv_rows < 0) { const char* e = getenv("A2S_CONV_ROWS"); g_conv_rows = e ? atoi(e) : 7; }
    return g_conv_rows;
}
bool a2s_conv_rows_eligible(int F, int Cin) { return a2s_conv_rows_enabled() && F % 4 == 0 && (Cin == 20 || Cin == 40); }

static void rows_geometry(int B, int T, int F, int* tilesF, int* nstrips, int* strip_len) {
    *tilesF = a2s_cdiv(F, RW_P);
    int ns = a2s_cdiv(2048, (long)B * *tilesF);                 // ~8 workgroups per CU over the launch
    ns = ns < 1 ? 1 : ns;
    const int max_ns = T / 16 > 1 ? T / 16 : 1;
    ns = ns > max_ns ? max_ns : ns;
    int len = a2s_cdiv(T, ns);
    len += len & 1;
    *strip_len = len;
    *nstrips = a2s_cdiv(T, len);
}
int a2s_conv_rows_blocks(int B, int T, int F) {
    int tf, ns, len;
    rows_geometry(B, T, F, &tf, &ns, &len);
    return B * tf * ns;
}
size_t a2s_conv_rows_workspace_floats(int Cin) {
    // packed image for Cout = 40 (8 n-tiles) + header + per-channel max|x| scratch of the fallback path
    return (size_t)9 * RwGeom<40>::KS * 2 * 1024 / 4 + RW_HDR + 64;     // (rows16: 9 tiles)
}

template <int CIN, int COUT>
static int rows16_launch(hipStream_t st, const RowsArgs& a, bool affine, bool bnred, int nwork) {
    constexpr int NT = 512;
    // the two-accumulator-set form (round 6): bit 2 the forward instances, bit 3 the data-gradient instance (24 spilled registers at 40 -> 40)
    const int en = a2s_conv_rows_enabled();
    if (affine && (en & 4)) hipLaunchKernelGGL((conv3x3_rows16<CIN, COUT, true, false, true>), dim3(nwork), dim3(NT), 0, st, a);
    else if (affine) hipLaunchKernelGGL((conv3x3_rows16<CIN, COUT, true, false>), dim3(nwork), dim3(NT), 0, st, a);
    else if (bnred && (en & 8)) hipLaunchKernelGGL((conv3x3_rows16<CIN, COUT, false, true, true>), dim3(nwork), dim3(NT), 0, st, a);
    else if (bnred) hipLaunchKernelGGL((conv3x3_rows16<CIN, COUT, false, true>), dim3(nwork), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_rows16<CIN, COUT, false, false>), dim3(nwork), dim3(NT), 0, st, a);
    A2S_CHECK_LAUNCH("conv3x3_rows16");
    return A2S_OK;
}
template <int CIN, int COUT>
static int rows_launch(hipStream_t st, const RowsArgs& a, bool affine, bool bnred, int nwork) {
    if (affine) hipLaunchKernelGGL((conv3x3_rows<CIN, COUT, true, false>), dim3(nwork), dim3(512), 0, st, a);
    else if (bnred) hipLaunchKernelGGL((conv3x3_rows<CIN, COUT, false, true>), dim3(nwork), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_rows<CIN, COUT, false, false>), dim3(nwork), dim3(512), 0, st, a);
    A2S_CHECK_LAUNCH("conv3x3_rows");
    return A2S_OK;
}
template <int CIN, int COUT>
static void rows_pack_launch(hipStream_t st, const float* w, int flip, const float* in_scale, const float* in_shift, const float* in_absmax,
                             const float* x_absmax, unsigned char* wimg, float* hdr, float* out_absmax) {
    hipLaunchKernelGGL((rows_pack<CIN, COUT>), dim3(COUT / 5), dim3(256), 0, st, w, flip, in_scale, in_shift, in_absmax, x_absmax, wimg, hdr, out_absmax);
}

int a2s_absmax_impl(hipStream_t, const float*, long, float*);

int a2s_conv3x3_rows_impl(hipStream_t st, const float* x, const float* w, float* y, const float* in_scale, const float* in_shift,
                          const float* in_absmax, float* stat_partial, float* out_absmax, int B, int T, int F, int Cin, int Cout, int flip,
                          float* ws, const float* yl, const float* yl_mean, const float* yl_invstd, const float* yl_scale, const float* yl_shift,
                          const float* x_absmax) {
    A2S_REQUIRE(F % 4 == 0 && (Cin == 20 || Cin == 40) && (Cout == 20 || Cout == 40), "conv3x3_rows: unsupported shape F=%d Cin=%d Cout=%d", F, Cin, Cout);
    A2S_REQUIRE(!(in_scale && yl), "conv3x3_rows: input affine and BatchNorm-backward statistics are exclusive");
    if (yl && Cin == 20 && Cout == 20 && out_absmax) {       // the one data-gradient instance that does not track max |g| in its epilogue: an extra pass
        const int rc = a2s_conv3x3_rows_impl(st, x, w, y, in_scale, in_shift, in_absmax, stat_partial, nullptr, B, T, F, Cin, Cout, flip, ws, yl, yl_mean, yl_invstd,
                                             yl_scale, yl_shift, x_absmax);
        return rc != A2S_OK ? rc : a2s_channel_absmax_impl(st, y, (long)B * T, Cout, F, out_absmax);
    }
    unsigned char* wimg = reinterpret_cast<unsigned char*>(ws);
    float* hdr = ws + (size_t)9 * RwGeom<40>::KS * 2 * 1024 / 4;
    float* scratch = hdr + RW_HDR;
    // operand ranges the caller did not supply are measured here (one extra pass over x: the engine always supplies them)
    if (in_scale && !in_absmax) {
        const int rc = a2s_channel_absmax_impl(st, x, (long)B * T, Cin, F, scratch);
        if (rc != A2S_OK) return rc;
        in_absmax = scratch;
    }
    if (!in_scale && !x_absmax) {
        const int rc = a2s_absmax_impl(st, x, (long)B * T * Cin * F, scratch);
        if (rc != A2S_OK) return rc;
        x_absmax = scratch;
    }
    RowsArgs a{x, y, wimg, hdr, stat_partial, out_absmax, yl, yl_mean, yl_invstd, yl_scale, yl_shift, B, T, F, 0, 0, 0, 0};
    rows_geometry(B, T, F, &a.tilesF, &a.nstrips, &a.strip_len);
    a.nwork = B * a.tilesF * a.nstrips;
    const bool affine = in_scale != nullptr, bnred = yl != nullptr;
#define R16_CASE(CI, CO)                                                                                                     \
    if (Cin == CI && Cout == CO && (a2s_conv_rows_enabled() & 2)) {                                                          \
        hipLaunchKernelGGL((rows16_pack<CI, CO>), dim3(8), dim3(256), 0, st, w, flip, in_scale, in_shift, in_absmax, x_absmax, wimg, hdr, out_absmax); \
        A2S_CHECK_LAUNCH("rows16_pack");                                                                                     \
        return rows16_launch<CI, CO>(st, a, affine, bnred, a.nwork);                                                         \
    }
    R16_CASE(20, 40) R16_CASE(40, 40)
#undef R16_CASE
#define RW_CASE(CI, CO)                                                                                                      \
    if (Cin == CI && Cout == CO) {                                                                                           \
        rows_pack_launch<CI, CO>(st, w, flip, in_scale, in_shift, in_absmax, x_absmax, wimg, hdr, out_absmax);               \
        A2S_CHECK_LAUNCH("rows_pack");                                                                                       \
        return rows_launch<CI, CO>(st, a, affine, bnred, a.nwork);                                                           \
    }
    RW_CASE(20, 20) RW_CASE(20, 40) RW_CASE(40, 40) RW_CASE(40, 20)
#undef RW_CASE
    A2S_FAIL(A2S_ERR_ARG, "conv3x3_rows: no instance for %d -> %d", Cin, Cout);
}
