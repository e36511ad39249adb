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

#define RW_P 120             // output columns per workgroup
#define RW_SROW 288          // bytes per channel row of the fp16 image: 128 positions + 32 B (rows 8 banks apart)
#define RW_SLOTS 5
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
    static constexpr int XIT = (ITEMS + 511) / 512;
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
    __shared__ __attribute__((aligned(16))) unsigned char ring[RW_SLOTS * G::SLOT];
    // data-gradient launches: the rows of yl (BatchNorm-backward statistics) a wave's epilogue needs, fetched by LDS-DMA while the wave
    // multiplies -- [row parity][n-group][channel 5 NJ][32 x 4 positions]; private to the wave that fetched them
    __shared__ __attribute__((aligned(16))) unsigned char ylbuf[BNRED ? 2 * COUT * 32 * 16 : 16];
    __shared__ float red[8][NJ * 5][3];
    __shared__ float tab[(AFFINE ? 2 * CIN : 0) + (BNRED ? 4 * COUT : 0) + 4];      // [asc | ash] or [mean | invstd | scale | shift]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, g = lane >> 4;
    const int rp = wave >> 2, ng = wave & 3;

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
    s16x8 bw[NJ][KS][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                uint4 v = *reinterpret_cast<const uint4*>(a.wimg + ((size_t)((ng * NJ + j) * KS + s) * 2 + tm) * 1024 + lane * 16);
                if (rp) { v.x ^= 0x80008000u; v.y ^= 0x80008000u; v.z ^= 0x80008000u; v.w ^= 0x80008000u; }
                bw[j][s][tm] = __builtin_bit_cast(s16x8, v);
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
    float st_s[NJ], st_s2[NJ], st_m[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) st_s[j] = st_s2[j] = st_m[j] = 0.f;

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
    // item `it` of a thread: e = tid + 512 it -> channel e >> 5, positions 4 (e & 31) ...; everything is re-derived from tid where it is
    // used (a laundered copy, so that the compiler does not keep 4 loop-invariant registers per item alive across the multiply)
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int e = tid + 512 * it;
        const int f = f_base - 4 + 4 * (e & 31);
        if (e < G::ITEMS && f >= 0 && f < a.F) okmask |= 1u << it;
    }
    // always XIT loads per thread (the counted s_waitcnt of the data-gradient epilogue relies on it): rows / columns outside the clip
    // read a clamped address and are zeroed when the row is converted
    auto issue = [&](int row, f32x4 (&xr)[XIT]) {
        const long rbase = (long)min(max(row, 0), a.T - 1) * CIN * a.F;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            int e = tid + 512 * it;
            asm volatile("" : "+v"(e));
            const int goff = (e >> 5) * a.F + f_base - 4 + 4 * (e & 31);
            xr[it] = *reinterpret_cast<const f32x4*>(xclip + rbase + ((okmask >> it & 1) ? goff : 0));
        }
    };
    auto commit = [&](int row, int slot, const f32x4 (&xr)[XIT]) {
        const bool rowok = row >= 0 && row < a.T;
        unsigned char* base = ring + slot * G::SLOT;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            int e = tid + 512 * it;
            asm volatile("" : "+v"(e));
            if (e >= G::ITEMS) continue;
            const int ch = e >> 5, loff = ch * RW_SROW + (e & 31) * 8;
            f32x4 v = xr[it];
            const bool ok = rowok && (okmask >> it & 1);
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
        }
    };

    f32x4 acc[8][NJ];
    unsigned char* const ylw = ylbuf + (rp * 4 + ng) * (NJ * 5 * 32 * 16);          // this wave's private slice
    // yl of output row t for this wave's channels: NJ * 5 channels x 32 items of 16 B = 2.5 (1.25) KB-instructions per lane
    auto yl_fetch = [&](int t) {
        if (!BNRED || t >= t_hi) return;
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
    // ---- multiply: output row of this wave from the three input rows at the running addresses
    auto multiply = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
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
            acc[ig * MG + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[TA][ii]), __builtin_bit_cast(f16x8, bw[j][s][TB]), acc[ig * MG + ii][j], 0, 0, 0);
                RW_PRODUCT(1, 0)
                RW_PRODUCT(0, 1)
                RW_PRODUCT(0, 0)
#undef RW_PRODUCT
            }
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
    auto epilogue = [&](int t) {
        if (t >= t_hi) return;
        float* __restrict__ yrow = a.y + ((long)b * a.T + t) * COUT * a.F;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = (ng * NJ + j) * 5 + q / 3;
            float bm = 0.f, bi = 0.f, bsc = 0.f, bsh = 0.f;
            if (BNRED && useful) { bm = tab[co]; bi = tab[COUT + co]; bsc = tab[2 * COUT + co]; bsh = tab[3 * COUT + co]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 v = acc[i][j];
                // rows p - 1 / p + 1 of the neighbouring columns across the 4-row register groups (and across m-tiles for g = 0 / 3)
                const float m3 = (g == 3 && i > 0) ? acc[i > 0 ? i - 1 : 0][j][3] : v[3];
                const float m0 = (g == 0 && i < 7) ? acc[i < 7 ? i + 1 : 7][j][0] : v[0];
                const float X = __int_as_float(__builtin_amdgcn_ds_bpermute(pv_addr, __float_as_int(m3)));
                const float Y = __int_as_float(__builtin_amdgcn_ds_bpermute(nx_addr, __float_as_int(m0)));
                f32x4 o;
                o[0] = (X + v[0]) + rw_shl1(v[1]);
                o[1] = (rw_shr1(v[0]) + v[1]) + rw_shl1(v[2]);
                o[2] = (rw_shr1(v[1]) + v[2]) + rw_shl1(v[3]);
                o[3] = (rw_shr1(v[2]) + v[3]) + Y;
                const int p = 16 * i + 4 * g, f = f_base - 4 + p;
                if (useful && p >= 4 && p < 4 + RW_P && f < a.F) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] *= unsc[j];
                    *reinterpret_cast<f32x4*>(yrow + (long)co * a.F + f) = o;
                    if (BNRED) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(ylw + ((j * 5 + q / 3) * 32 + 4 * i + g) * 16);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float gm = (fmaf(xv[k], bsc, bsh) > 0.f) ? o[k] : 0.f;
                            st_s[j] += gm; st_s2[j] = fmaf(gm * (xv[k] - bm), bi, st_s2[j]);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { st_s[j] += o[k]; st_s2[j] = fmaf(o[k], o[k], st_s2[j]); }
                        st_m[j] = fmaxf(fmaxf(st_m[j], fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    }
                }
            }
        }
    };

    // ---- prologue: rows t_lo - 1 .. t_lo + 2 into slots 0 .. 3, row t_lo + 3 in flight
    f32x4 xa[XIT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        issue(t_lo - 1 + r, xa);
        commit(t_lo - 1 + r, r, xa);
    }
    __syncthreads();

    // Per phase: [barrier] -> next input row's loads issued -> role (multiply | epilogue) -> that row converted into its slot -> [barrier].
    // Even-row waves: multiply row t (+ fetch its yl), then its epilogue; odd-row waves: epilogue of row t - 1, then multiply row t + 1.
    int slot_w = 4;
    for (int t = t_lo; t < t_hi; t += 2) {
        issue(t + 3, xa);
        if (rp == 0) { yl_fetch(t); multiply(); }
        else if (t > t_lo) { if (BNRED) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XIT) : "memory"); epilogue(t - 1); }
        commit(t + 3, slot_w, xa);
        slot_w = slot_w == RW_SLOTS - 1 ? 0 : slot_w + 1;
        __syncthreads();
        issue(t + 4, xa);
        if (rp == 0) { if (BNRED) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XIT) : "memory"); epilogue(t); }
        else { yl_fetch(t + 1); multiply(); }
        commit(t + 4, slot_w, xa);
        slot_w = slot_w == RW_SLOTS - 1 ? 0 : slot_w + 1;
        advance();
        __syncthreads();
    }
    if (rp == 1 && t_hi > t_lo) {
        if (BNRED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        epilogue(t_lo + ((t_hi - t_lo + 1) / 2) * 2 - 1);
    }

    // ---- statistics: lanes of a column over the 4 row groups, then the two row-parity waves of the n-group
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float s = st_s[j], s2 = st_s2[j], m = st_m[j];
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
        if (!BNRED && a.out_absmax) {
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
    if (g_conv_rows < 0) { const char* e = getenv("A2S_CONV_ROWS"); g_conv_rows = e ? atoi(e) : 1; }
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
    return (size_t)8 * RwGeom<40>::KS * 2 * 1024 / 4 + RW_HDR + 64;
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
    unsigned char* wimg = reinterpret_cast<unsigned char*>(ws);
    float* hdr = ws + (size_t)8 * RwGeom<40>::KS * 2 * 1024 / 4;
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
