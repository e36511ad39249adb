// Data gradient of the ConvStack's 19200 -> 256 Linear with the layer-4 BatchNorm-backward statistics in its epilogue
// (reference models.py:68 `self.out = nn.Linear(...)`, backward of y = relu(bn4(y4)) W^T):
//
//   da[m][n] = sum_k dz[m][k] Wt[n][k]        M = B*T rows (307 456 at B = 256), N = 40 * F = 19200 columns, K = 256
//   s1[c] += g' , s2[c] += g' xhat            g' = da where bn4(y4) > 0, c = n / F
//
// The generic two-term tile of a2s_gemm.hip (256 x 256, one workgroup per tile) spends this launch outside the matrix pipe: K is only 8
// k-tiles, every workgroup re-splits its 256 x 256 slice of W into fp16 terms (the whole of W 1201 times per launch), and its epilogue -- a
// 256 KB store and a 256 KB re-read of y4 per tile, 47 GB per launch -- overlaps with nothing (one workgroup per CU, 256 registers):
// 23.8 ms at B = 256 against ~9.5 ms of HBM time and 3.6 ms of matrix time.  Here:
//   * W is split ONCE per launch into fp16 term planes laid out in MFMA-fragment order (lin_pack_planes_k: 19.7 MB, 1 KB contiguous per
//     fragment); the sweep below reads its B fragments straight from those planes (L2-resident: all workgroups of an XCD walk the
//     column tiles together) -- no conversion, no LDS, no barrier for B;
//   * a workgroup owns 128 ROWS for ALL column tiles: its slice of dz is split once into LDS (147 KB: [128-k block][term][row][256 B + 32])
//     and is read-only from then on, so the sweep over the 75 column tiles of 256 has NO barrier at all: the 8 waves (32 columns each)
//     drift apart and one wave's epilogue (stores, the y4 re-read, the statistics) runs under the other waves' MFMAs on the same SIMD;
//   * the statistics of a wave's 32-column slices (one channel each: F % 32 == 0) stay in registers until the channel changes.
// Measured at B = 256 (tools/linear_bench.py 256; profiles/r04_linear_dgrad.txt): 15.1 ms (generic tile: 23.8).  Ablations (-DLIN_X): multiply
// alone 7.2 ms (553 M MFMAs = 3.6 ms at the nominal clock; without the B loads 6.4), + stores 9.6, + the y4 re-read alone 10.9, both 15.1-15.6:
// the two streams together cost more than their sum -- the launch is bound by the mixed read / write HBM traffic of 64-byte pieces of 16 rows
// per instruction (PMC: 25.6 GB fetched = y4 + dz, the B fragments are L2 hits: 722 M hits against 392 M misses; non-temporal stores wrote
// 32.7 GB for 23.6 GB of output, plain stores the exact bytes: hence plain).  Wave priorities or a start-up stagger between the two waves of
// a SIMD change nothing.
#include "a2s_common.h"
#include "../../include/a2s.h"

#define LIN_BM 128
#define LIN_K 256
#define LIN_RS 288                       // bytes per LDS row of one 128-k block: 256 of fp16 + 32.  ds_read_b128 of the fragment pattern (16 rows x 64 B):
                                         // 4.0 clocks per wave-level read with a row stride that is an ODD multiple of 32 bytes (96, 160, 224, 288), 6.7
                                         // with 80 / 144 / 272 (tools/ubench/b128_read.hip: profiles/r04_lds_patterns.txt)
#define LIN_NTH 512
#define LIN_LDS (2 * 2 * LIN_BM * LIN_RS)

typedef unsigned lu32x4 __attribute__((ext_vector_type(4)));

struct LinDgradArgs {
    const float* A; long lda;            // dz (M x 256)
    const unsigned char* planes;         // packed fp16 term planes of Wt (lin_pack_planes_k)
    float* C; long ldc;                  // da (M x N)
    const float* ep_y;                   // y4 (M x N, leading dimension ldc)
    const float* mean; const float* invstd; const float* scale; const float* shift;
    float* partial;                      // [M / 128 rounded up][8 waves][channels][2]
    const float* a_absmax; const float* b_absmax;
    int M, N, period, channels;
    float* c_absmax_out;                 // max |da| folded in by atomic max (may be null): the range of the BatchNorm backward that follows
};

// W (N x K, k contiguous, K % 32 == 0) -> [n-tile][k-step][term][lane][8 halves]: the 16 bytes of lane (lr = n % 16, lk) of the fragment of
// k-step ks hold k = 32 ks + 8 lk .. + 7 of row n, scaled by the power of two that brings max |W| to 2^12.  One thread per (n, 8 k).
__global__ __launch_bounds__(256) void lin_pack_planes_k(const float* __restrict__ W, long ld, int N, int K, const float* __restrict__ absmax,
                                                         unsigned char* __restrict__ out) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int kgroups = K >> 3, nks = K >> 5;
    const int n = (int)(id / kgroups), kg = (int)(id % kgroups);
    if (n >= N) return;
    const float ps = ldexpf(1.f, pow2_scale_exp(*absmax, 12));
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(W + (long)n * ld + kg * 8);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(W + (long)n * ld + kg * 8 + 4);
    float x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[i] = __builtin_amdgcn_fmed3f(v0[i] * ps, -65000.f, 65000.f); x[4 + i] = __builtin_amdgcn_fmed3f(v1[i] * ps, -65000.f, 65000.f); }
    lu32x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned h, l; split2_pair_f16(x[2 * i], x[2 * i + 1], h, l); hi[i] = h; lo[i] = l; }
    const int nt = n >> 4, lr = n & 15, ks = kg >> 2, lk = kg & 3;
    unsigned char* o = out + ((((long)nt * nks + ks) * 2) * 64 + (lk * 16 + lr)) * 16;
    *reinterpret_cast<lu32x4*>(o) = hi;
    *reinterpret_cast<lu32x4*>(o + 64 * 16) = lo;
}

__global__ __launch_bounds__(LIN_NTH) void lin_dgrad_bnstats(LinDgradArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LIN_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int m0 = blockIdx.x * LIN_BM;
    const int ka = pow2_scale_exp(*a.a_absmax, 12), kb = pow2_scale_exp(*a.b_absmax, 12);
    const float psa = ldexpf(1.f, ka), unscale = ldexpf(1.f, -(ka + kb));

    // ---- this workgroup's 128 rows of dz -> two fp16 term images in LDS (rows beyond M: zeros)
    {
        f32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int slot = tid + LIN_NTH * i, row = slot >> 6, k = (slot & 63) * 4;
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (m0 + row < a.M) v[i] = *reinterpret_cast<const f32x4*>(a.A + (long)(m0 + row) * a.lda + k);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int slot = tid + LIN_NTH * i, row = slot >> 6, k = (slot & 63) * 4;
            float x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = __builtin_amdgcn_fmed3f(v[i][j] * psa, -65000.f, 65000.f);
            uint2 hi, lo;
            split2_pair_f16(x[0], x[1], hi.x, lo.x);
            split2_pair_f16(x[2], x[3], hi.y, lo.y);
            const int blk = k >> 7, off = (k & 127) * 2;
            *reinterpret_cast<uint2*>(lds + ((blk * 2 + 0) * LIN_BM + row) * LIN_RS + off) = hi;
            *reinterpret_cast<uint2*>(lds + ((blk * 2 + 1) * LIN_BM + row) * LIN_RS + off) = lo;
        }
    }
    __syncthreads();                     // the only barrier: the images are read-only from here on

    const int ntiles = (a.N + 255) >> 8;
    // addresses as (workgroup-uniform base) + (32-bit lane offset that never changes): the bases live in scalar registers
    const unsigned lane_off = (unsigned)(lr * a.ldc + lk * 4) * 4u;
    float s1 = 0.f, s2 = 0.f, vmax = 0.f;
    int cur_c = -1;
    auto flush = [&]() {
        if (cur_c < 0) return;
        const float w1 = wave_sum(s1), w2 = wave_sum(s2);
        if (lane == 0) {
            float* p = a.partial + (((long)blockIdx.x * 8 + wave) * a.channels + cur_c) * 2;
            p[0] = w1; p[1] = w2;
        }
    };
    // B fragments of k-step ks of the 32-column slice at n0: planes + ((((n0 / 16 + nt) * 8 + ks) * 2 + term) * 64 + lane) * 16.  They come straight
    // from L2 (~1 us under load) while a k-step is 0.3 us of MFMA per wave: a ring of three k-steps keeps two in flight, and the first two of the
    // NEXT column tile are issued before the epilogue of this one.
    lu32x4 bf[3][2][2];
    auto load_b = [&](int n0, int ks, lu32x4 (&dst)[2][2]) {
        const unsigned char* bp = a.planes + (long)(n0 >> 4) * 8 * 2 * 64 * 16;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) dst[nt][sp] = *reinterpret_cast<const lu32x4*>(bp + ((long)(nt * 8 + ks) * 2 + sp) * 1024 + (unsigned)lane * 16u);
    };
    if (wave * 32 < a.N) { load_b(wave * 32, 0, bf[0]); load_b(wave * 32, 1, bf[1]); }
    for (int j = 0; j < ntiles; ++j) {
        const int n0 = j * 256 + wave * 32;
        if (n0 >= a.N) break;             // (wave-uniform; N % 32 == 0)
        // the epilogue's re-read of y4 (16 bytes per lane and tile: 64 registers) is issued BEFORE the multiply: it costs 6.7 of 17 ms when the
        // loads are issued where they are used (eight dependent round trips per column tile)
        f32x4 yq[8][2];
        auto load_y = [&](int mt) {
            const int m = m0 + mt * 16 + lr;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#if defined(LIN_X) && LIN_X >= 1 && LIN_X != 4      /* timing ablations (-DLIN_X=: 1 no re-read of y4, 2 nor stores, 3 nor B loads in the k loop, 4 no stores only) */
                yq[mt][nt] = (f32x4){1.f, 1.f, 1.f, 1.f};
#else
                yq[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (m < a.M)
                    yq[mt][nt] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.ep_y + (long)(m0 + mt * 16) * a.ldc + n0 + nt * 16) + lane_off);
#endif
            }
        };
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) load_y(mt);                      // (the rest once the multiply has released its registers)
        f32x4 acc[8][2];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#if defined(LIN_X) && LIN_X >= 3
            if (ks + 2 < 8 && a.M < 0) load_b(n0, ks + 2, bf[(ks + 2) % 3]);      // (timing experiment: no B traffic inside the k loop)
#else
            if (ks + 2 < 8) load_b(n0, ks + 2, bf[(ks + 2) % 3]);
#endif
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                lu32x4 af[2][4];
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        af[sp][q] = *reinterpret_cast<const lu32x4*>(lds + (((ks >> 2) * 2 + sp) * LIN_BM + (half * 4 + q) * 16 + lr) * LIN_RS + (ks & 3) * 64 + lk * 16);
#define LIN_PRODUCT(SA, SB)                                                                                                   \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                              \
                    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                             \
                        acc[half * 4 + q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[ks % 3][nt][SB]), \
                                                                                       __builtin_bit_cast(f16x8, af[SA][q]), acc[half * 4 + q][nt], 0, 0, 0);
                LIN_PRODUCT(1, 0) LIN_PRODUCT(0, 1) LIN_PRODUCT(0, 0)
#undef LIN_PRODUCT
            }
        }
        if (n0 + 256 < a.N) { load_b(n0 + 256, 0, bf[0]); load_b(n0 + 256, 1, bf[1]); }
#pragma unroll
        for (int mt = 3; mt < 8; ++mt) load_y(mt);
        // ---- epilogue: lane (lr, lk) of tile (mt, nt) owns da[m0 + 16 mt + lr][n0 + 16 nt + 4 lk .. + 3] (transposed accumulators)
        const int c = n0 / a.period;
        if (c != cur_c) { flush(); s1 = 0.f; s2 = 0.f; cur_c = c; }
        const float mean = a.mean[c], invstd = a.invstd[c], sc = a.scale[c], sh = a.shift[c];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int m = m0 + mt * 16 + lr;
            if (m >= a.M) continue;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 v = acc[mt][nt];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] *= unscale; vmax = fmaxf(vmax, fabsf(v[r])); }
#if defined(LIN_X) && LIN_X >= 2
                if (v[0] == 123.456f)
#endif
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.C + (long)(m0 + mt * 16) * a.ldc + n0 + nt * 16) + lane_off) = v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gm = (fmaf(yq[mt][nt][r], sc, sh) > 0.f) ? v[r] : 0.f;
                    s1 += gm; s2 = fmaf(gm * (yq[mt][nt][r] - mean), invstd, s2);
                }
            }
        }
    }
    flush();
    if (a.c_absmax_out) {                // non-negative floats order like their bit patterns
        const float m = wave_max(vmax);
        if (lane == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.c_absmax_out), __float_as_uint(m));
    }
}

size_t a2s_linear_dgrad_ws_bytes_impl(int N, int K) { return (K > 0 && K % 32 == 0 && N > 0) ? (size_t)N * K * 2 * 2 : 0; }      // the weight as two fp16 planes
int a2s_linear_dgrad_blocks_impl(int M) { return M > 0 ? ((M + LIN_BM - 1) / LIN_BM) * 8 : 0; }

bool a2s_linear_dgrad_ok(int M, int N, int K, long lda, long sBk, long sBn, long ldc, int period, const void* A, const void* B, const void* C, const void* y) {
    return K == LIN_K && M >= LIN_BM && N >= 256 && N % 32 == 0 && period > 0 && period % 32 == 0 && N % period == 0 && N / period <= 4096 && sBk == 1 && sBn == K &&
           lda % 4 == 0 && ldc % 4 == 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)y) % 16 == 0);
}

int a2s_linear_dgrad_bnstats_impl(hipStream_t st, int M, int N, int K, const float* A, long lda, const float* Wt, long sBk, long sBn, float* C, long ldc,
                                  const float* ep_y, const float* mean, const float* invstd, const float* scale, const float* shift, int period,
                                  float* partial, const float* a_absmax, const float* b_absmax, float* ws, size_t ws_bytes, float* c_absmax_out) {
    A2S_REQUIRE(A && Wt && C && ep_y && mean && invstd && scale && shift && partial && a_absmax && b_absmax && ws, "linear_dgrad_bnstats: null argument");
    A2S_REQUIRE(a2s_linear_dgrad_ok(M, N, K, lda, sBk, sBn, ldc, period, A, Wt, C, ep_y),
                "linear_dgrad_bnstats: needs K = 256, M >= 128, N >= 256, N %% 32 == 0, period %% 32 == 0, a k-contiguous weight (N x K) and 16-byte aligned rows");
    A2S_REQUIRE(((uintptr_t)ws % 16 == 0) && ws_bytes >= a2s_linear_dgrad_ws_bytes_impl(N, K), "linear_dgrad_bnstats: workspace too small (%zu bytes needed)",
                a2s_linear_dgrad_ws_bytes_impl(N, K));
    const int channels = N / period, nblk = a2s_linear_dgrad_blocks_impl(M);
    hipError_t e = hipMemsetAsync(partial, 0, sizeof(float) * 2 * (size_t)nblk * channels, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "linear_dgrad_bnstats memset: %s", hipGetErrorString(e));
    unsigned char* planes = reinterpret_cast<unsigned char*>(ws);
    hipLaunchKernelGGL(lin_pack_planes_k, dim3((unsigned)(((long)N * (K / 8) + 255) / 256)), dim3(256), 0, st, Wt, (long)K, N, K, b_absmax, planes);
    A2S_CHECK_LAUNCH("lin_pack_planes_k");
    LinDgradArgs a{A, lda, planes, C, ldc, ep_y, mean, invstd, scale, shift, partial, a_absmax, b_absmax, M, N, period, channels, c_absmax_out};
    hipLaunchKernelGGL(lin_dgrad_bnstats, dim3((M + LIN_BM - 1) / LIN_BM), dim3(LIN_NTH), 0, st, a);
    A2S_CHECK_LAUNCH("lin_dgrad_bnstats");
    return A2S_OK;
}

// =========================================================================================== forward
// z[m][n] = sum_k relu(y4[m][k] * scale[c] + shift[c]) * W[n][k]      (reference models.py:68,537-539: ConvStack.out applied to relu(bn4(y4)));
// M = B*T rows, N = 256, K = 40 * F = 19200, c = k / F.
// The generic tile (256 x 256, a2s_gemm.hip) re-splits its 256 x 32 slice of W in every k-tile of every workgroup (as much conversion work as
// the activations themselves), stages both operands through ONE LDS buffer with two barriers per k-tile and gives the next tile's global loads
// one multiply phase to arrive: 12.9 ms at B = 256 against 4.7 ms of HBM time (y4 once) and 3.6 ms of matrix time.  Here:
//   * W as fp16 term planes in fragment order (lin_pack_planes_k, as for the data gradient): B fragments straight from L2, ring of three k-steps;
//   * a workgroup owns 128 rows x all 256 columns (8 waves x 32 columns, 64 accumulators): y4 is read exactly once;
//   * the activations go global -> registers (two 64-k blocks in flight) -> BatchNorm + ReLU + split -> a ring of THREE LDS stages (40 KB each):
//     one barrier per 64 k, the stage written in iteration b is read in iteration b + 1 (32-k stages, a barrier per k-step: 11.0 ms at B = 256;
//     the barrier and the conversion block stand between the k-steps' MFMAs);
//   * F % 32 == 0: a k-step lies inside one channel, so the BatchNorm constants of a k-step are two scalars.
// Measured at B = 256 (profiles/r04_linear_fwd.txt): 10.4 ms (generic tile 13.2).  Ablations (-DLF_X, additive almost to the millisecond): matrix work
// + fragment reads alone 4.9 ms, + the barrier 5.6, + the B fragment loads 7.4 (1.8), + conversion and LDS writes 9.5 (2.0), + the y4 loads 11.5
// (3.5): all eight waves leave the barrier in the same phase, so whatever is not an MFMA is time the matrix pipe idles.  Reading the next quarter's
// fragments under this quarter's MFMAs, spreading the conversion over the quarters (sched_group_barrier) and issuing the L2-hit B loads ahead of the
// HBM loads (in-order completion) moved it from 11.5 to 11.0, running the two waves of a SIMD in opposite order (convert-then-multiply /
// multiply-then-convert) to 10.4.  The y4 loads still cost ~3 ms although they are in flight for two blocks (7 us) and although one contiguous
// 32 KB run per block instead of 128 x 256 B changes nothing (10.34): with the HBM stream beside the matrix pipes the shader clock drops
// (a2s_conv_wrows.hip: 2.3 -> 1.7 GHz), i.e. the "cost of the loads" is largely the multiply running slower.
// Two independent workgroups per CU instead (64 rows x 256 columns, 4 waves x 64 columns, so that one's staging runs under the other's MFMAs):
// 15.1 ms -- every B fragment is then fetched from L2 by twice as many workgroups.
#define LF_RS 160                        // bytes per LDS row of a stage: 64 k of fp16 + 32 (odd multiple of 32 B: conflict-free fragment reads, see LIN_RS)
#define LF_STAGE (2 * LIN_BM * LF_RS)
#define LF_NS 3

struct LinFwdArgs {
    const float* A; long lda;            // y4 (M x K)
    const unsigned char* planes;         // packed fp16 term planes of W (N x K): lin_pack_planes_k
    float* C; long ldc;                  // z (M x 256)
    const float* a_scale; const float* a_shift;     // per channel (may be null: the operand is used as is, no ReLU)
    const float* a_absmax; const float* b_absmax;
    int M, K, period;
};

// (The symmetric form -- all eight waves stage and multiply, 10.4 ms against 8.2 -- is in the history: round 4, HISTORY.md Part II 3.2.)
// ---- the forward with the two waves of a SIMD in FIXED ROLES (the split of a2s_conv_wrows.hip): waves 0-3 only read fragments and
// multiply (128 rows x 64 columns each: 128 accumulators), waves 4-7 only load, convert and write the next stages -- the conversion, the
// global-load issue and their waits never stand in front of an MFMA.
__global__ __launch_bounds__(LIN_NTH) void lin_fwd_roles(LinFwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LF_NS * LF_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * LIN_BM;
    const int ka = a.a_absmax ? pow2_scale_exp(*a.a_absmax, 12) : 0, kb = pow2_scale_exp(*a.b_absmax, 12);
    const float psa = ldexpf(1.f, ka), unscale = ldexpf(1.f, -(ka + kb));
    const int nks = a.K >> 5, nblk = a.K >> 6, kpc = a.period >> 5;
    const bool affine = a.a_scale != nullptr;
    if (wave >= 4) {
        // ================================================================ staging role: 256 threads, 8 items (row, 4 k) per 64-k block each
        const int st_ = tid - 256;
        const int sk = (st_ & 15) * 4;
        const bool second = sk >= 32;
        f32x4 ar[2][8];
        auto issue_a = [&](int blk, f32x4 (&r)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = (st_ + 256 * i) >> 4;
                r[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (m0 + row < a.M && blk < nblk) r[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.A + (long)(m0 + row) * a.lda + (long)blk * 64 + sk));
            }
        };
        auto commit_a = [&](int blk, const f32x4 (&r)[8]) {
            if (blk >= nblk) return;
            const int c0 = (2 * blk) / kpc, c1 = (2 * blk + 1) / kpc;
            const float sc = affine ? (second ? a.a_scale[c1] : a.a_scale[c0]) * psa : psa, sh = affine ? (second ? a.a_shift[c1] : a.a_shift[c0]) * psa : 0.f;
            const float floor_ = affine ? 0.f : -INFINITY;
            unsigned char* st = lds + (blk % LF_NS) * LF_STAGE;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = (st_ + 256 * i) >> 4;
                float x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = __builtin_amdgcn_fmed3f(fmaxf(fmaf(r[i][j], sc, sh), floor_), -65000.f, 65000.f);
                uint2 hi, lo;
                split2_pair_f16(x[0], x[1], hi.x, lo.x);
                split2_pair_f16(x[2], x[3], hi.y, lo.y);
                *reinterpret_cast<uint2*>(st + (0 * LIN_BM + row) * LF_RS + sk * 2) = hi;
                *reinterpret_cast<uint2*>(st + (1 * LIN_BM + row) * LF_RS + sk * 2) = lo;
            }
        };
        issue_a(0, ar[0]);
        commit_a(0, ar[0]);
        issue_a(1, ar[1]); issue_a(2, ar[0]);
#pragma unroll 1
        for (int b0 = 0; b0 < nblk; b0 += 2) {
            // iteration blk: (barrier) stage blk is complete and stage blk + 1 free; write stage blk + 1, fetch block blk + 3
            __syncthreads();
            commit_a(b0 + 1, ar[1]); issue_a(b0 + 3, ar[1]);
            if (b0 + 1 < nblk) {
                __syncthreads();
                commit_a(b0 + 2, ar[0]); issue_a(b0 + 4, ar[0]);
            }
        }
        return;
    }
    // ==================================================================== multiply role: wave w = columns 64 w .. + 63, all 128 rows
    lu32x4 bf[2][4][2];                                // [ring of two k-steps][nt][term]
    auto load_b = [&](int ks, lu32x4 (&dst)[4][2]) {
        if (ks >= nks) return;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
                dst[nt][sp] = *reinterpret_cast<const lu32x4*>(a.planes + (((long)(wave * 4 + nt) * nks + ks) * 2 + sp) * 1024 + (unsigned)lane * 16u);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_b(0, bf[0]);
    // two row tiles at a time (16 fragment registers), the next pair read while this one multiplies
    auto load_af = [&](const unsigned char* st, int h, int qt, lu32x4 (&af)[2][2]) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int q = 0; q < 2; ++q) af[sp][q] = *reinterpret_cast<const lu32x4*>(st + (sp * LIN_BM + (qt * 2 + q) * 16 + lr) * LF_RS + h * 64 + lk * 16);
    };
    auto multiply = [&](int qt, const lu32x4 (&af)[2][2], const lu32x4 (&b)[4][2]) {
#define LR_PRODUCT(SA, SB)                                                                                                        \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                          \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                         \
                acc[qt * 2 + q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b[nt][SB]),                \
                                                                             __builtin_bit_cast(f16x8, af[SA][q]), acc[qt * 2 + q][nt], 0, 0, 0);
        LR_PRODUCT(1, 0) LR_PRODUCT(0, 1) LR_PRODUCT(0, 0)
#undef LR_PRODUCT
    };
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
        __syncthreads();
        const unsigned char* st = lds + (blk % LF_NS) * LF_STAGE;
        lu32x4 fa[2][2], fb[2][2];
        load_af(st, 0, 0, fa);
        load_b(2 * blk + 1, bf[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_af(st, 0, 1, fb); multiply(0, fa, bf[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_af(st, 0, 2, fa); multiply(1, fb, bf[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_af(st, 0, 3, fb); multiply(2, fa, bf[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_af(st, 1, 0, fa); multiply(3, fb, bf[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_b(2 * blk + 2, bf[0]);
        load_af(st, 1, 1, fb); multiply(0, fa, bf[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_af(st, 1, 2, fa); multiply(1, fb, bf[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_af(st, 1, 3, fb); multiply(2, fa, bf[1]);
        __builtin_amdgcn_sched_barrier(0);
        multiply(3, fb, bf[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = m0 + mt * 16 + lr;
        if (m >= a.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[mt][nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= unscale;
            *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + wave * 64 + nt * 16 + lk * 4) = v;
        }
    }
}

bool a2s_linear_fwd_ok(int M, int N, int K, long lda, long ldc, int period, const void* A, const void* W, const void* C) {
    return N == 256 && M >= 1 && K >= 128 && K % 64 == 0 && (period == 0 || (period % 32 == 0 && K % period == 0)) && lda % 4 == 0 && ldc % 4 == 0 &&
           (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) % 16 == 0);
}

int a2s_linear_fwd_impl(hipStream_t st, int M, int N, int K, const float* A, long lda, const float* W, float* C, long ldc, const float* a_scale,
                        const float* a_shift, int period, const float* a_absmax, const float* w_absmax, float* ws, size_t ws_bytes) {
    A2S_REQUIRE(A && W && C && w_absmax && ws && ((a_scale == nullptr) == (a_shift == nullptr)), "linear_fwd: null argument");
    A2S_REQUIRE(a2s_linear_fwd_ok(M, N, K, lda, ldc, a_scale ? period : 0, A, W, C),
                "linear_fwd: needs N = 256, K %% 64 == 0, period %% 32 == 0 and 16-byte aligned rows");
    A2S_REQUIRE(((uintptr_t)ws % 16 == 0) && ws_bytes >= (size_t)N * K * 4, "linear_fwd: workspace too small (%zu bytes needed)", (size_t)N * K * 4);
    unsigned char* planes = reinterpret_cast<unsigned char*>(ws);
    hipLaunchKernelGGL(lin_pack_planes_k, dim3((unsigned)(((long)N * (K / 8) + 255) / 256)), dim3(256), 0, st, W, (long)K, N, K, w_absmax, planes);
    A2S_CHECK_LAUNCH("lin_pack_planes_k");
    LinFwdArgs a{A, lda, planes, C, ldc, a_scale, a_shift, a_absmax, w_absmax, M, K, a_scale ? period : K};
    hipLaunchKernelGGL(lin_fwd_roles, dim3((M + LIN_BM - 1) / LIN_BM), dim3(LIN_NTH), 0, st, a);
    A2S_CHECK_LAUNCH("lin_fwd");
    return A2S_OK;
}

// =========================================================================================== weight gradient
// G[n][k] += sum_m dz[m][n] * relu(y4[m][k] * scale[c] + shift[c])      (dW of ConvStack.out, reference models.py:68; N = 256, K = 19200, M = B*T)
// The reduction index is the ROW m, the slow dimension of both tensors.  Same skeleton as lin_fwd with the roles turned:
//   * dz (M x 256, small) is transposed, scaled and split ONCE per launch into fragment-order planes [32-row step][n-tile][term][lane][8 m]
//     (lin_pack_dz_planes; every workgroup reads them from L2 -- they play the part the weight planes play in the forward);
//   * a workgroup owns 128 columns k and a range of 64-row blocks (split-K over the rows: 150 column tiles x 5 ranges = three full rounds of the
//     chip); the activations go global -> registers (a 4 rows x 4 columns block per thread) -> BatchNorm + ReLU + split -> the LDS stage
//     TRANSPOSED, [column][64 m]: for each of its 4 columns a thread writes 4 consecutive m as one 8-byte store, and a fragment (8 consecutive m
//     of one column) is one ds_read_b128 exactly as in the forward;
//   * the matrix pipe's internal sum truncates toward -infinity: every other group of 8 row steps is accumulated negated (dz negated in the
//     planes, accumulators flipped), as the generic tiles do (a2s_gemm.hip);
//   * partial slabs [range][256][K], summed in fixed order into G by lin_wgrad_reduce.
struct LinWgradArgs {
    const float* A; long lda;            // y4 (M x K)
    const unsigned char* planes;         // dz planes (lin_pack_dz_planes)
    float* partial;                      // [splits][256][K]
    const float* a_scale; const float* a_shift;
    const float* a_absmax; const float* d_absmax;
    int M, K, period, nblk, blk_per_split;
};

// dz (M x 256) -> [step s = m / 32][n-tile][term][lane][8 halves]: lane (lr = n % 16, lk) holds dz[32 s + 8 lk .. + 7][n], scaled by the power of
// two that brings max |dz| to 2^12, negated where (s >> 3) is odd; rows >= M: zeros.  One thread per (s, lk, n).
__global__ __launch_bounds__(256) void lin_pack_dz_planes(const float* __restrict__ dz, long ld, int M, int nsteps, const float* __restrict__ absmax,
                                                          unsigned char* __restrict__ out) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int n = (int)(id & 255), lk = (int)((id >> 8) & 3), s = (int)(id >> 10);
    if (s >= nsteps) return;
    float ps = ldexpf(1.f, pow2_scale_exp(*absmax, 12));
    if ((s >> 3) & 1) ps = -ps;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = s * 32 + lk * 8 + j;
        x[j] = m < M ? __builtin_amdgcn_fmed3f(dz[(long)m * ld + n] * ps, -65000.f, 65000.f) : 0.f;
    }
    lu32x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned h, l; split2_pair_f16(x[2 * i], x[2 * i + 1], h, l); hi[i] = h; lo[i] = l; }
    unsigned char* o = out + ((((long)s * 16 + (n >> 4)) * 2) * 64 + (lk * 16 + (n & 15))) * 16;
    *reinterpret_cast<lu32x4*>(o) = hi;
    *reinterpret_cast<lu32x4*>(o + 64 * 16) = lo;
}

// ---- fixed wave roles (see lin_fwd_roles; the symmetric form, 11.05 ms against 10.1, is in the history): waves 0-3 multiply (128 columns x 64 n each: 128 accumulators), waves 4-7 stage
__global__ __launch_bounds__(LIN_NTH) void lin_wgrad_roles(LinWgradArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LF_NS * LF_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k0 = blockIdx.x * LIN_BM;
    const int b_lo = blockIdx.y * a.blk_per_split, b_hi = min(a.nblk, b_lo + a.blk_per_split);
    const int ka = a.a_absmax ? pow2_scale_exp(*a.a_absmax, 12) : 0, kd = pow2_scale_exp(*a.d_absmax, 12);
    const float psa = ldexpf(1.f, ka), unscale = ldexpf(1.f, -(ka + kd));
    const bool affine = a.a_scale != nullptr;
    if (wave >= 4) {
        // ================================================================ staging role: two 4 rows x 4 columns blocks per thread and 64-row block
        const int st_ = tid - 256;
        const int rq = st_ & 15, cq0 = st_ >> 4;             // row quad rq; column quads cq0 and cq0 + 16 (16 lanes fill one stage row: see lin_wgrad)
        float sc[2], sh[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = affine ? (k0 + 4 * (cq0 + 16 * u)) / a.period : 0;
            sc[u] = affine ? a.a_scale[ch] * psa : psa; sh[u] = affine ? a.a_shift[ch] * psa : 0.f;
        }
        const float floor_ = affine ? 0.f : -INFINITY;
        f32x4 ar[2][8];
        auto issue_a = [&](int blk, f32x4 (&r)[8]) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = blk * 64 + 4 * rq + j;
                    r[4 * u + j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (m < a.M && blk < b_hi) r[4 * u + j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.A + (long)m * a.lda + k0 + 4 * (cq0 + 16 * u)));
                }
        };
        auto commit_a = [&](int blk, const f32x4 (&r)[8]) {
            if (blk >= b_hi) return;
            unsigned char* st = lds + (blk % LF_NS) * LF_STAGE;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = __builtin_amdgcn_fmed3f(fmaxf(fmaf(r[4 * u + j][c], sc[u], sh[u]), floor_), -65000.f, 65000.f);
                    uint2 hi, lo;
                    split2_pair_f16(x[0], x[1], hi.x, lo.x);
                    split2_pair_f16(x[2], x[3], hi.y, lo.y);
                    *reinterpret_cast<uint2*>(st + (0 * LIN_BM + 4 * (cq0 + 16 * u) + c) * LF_RS + rq * 8) = hi;
                    *reinterpret_cast<uint2*>(st + (1 * LIN_BM + 4 * (cq0 + 16 * u) + c) * LF_RS + rq * 8) = lo;
                }
        };
        if (b_lo < b_hi) {
            issue_a(b_lo, ar[0]);
            commit_a(b_lo, ar[0]);
            issue_a(b_lo + 1, ar[1]); issue_a(b_lo + 2, ar[0]);
        }
#pragma unroll 1
        for (int b = b_lo; b < b_hi; b += 2) {
            __syncthreads();
            commit_a(b + 1, ar[1]); issue_a(b + 3, ar[1]);
            if (b + 1 < b_hi) {
                __syncthreads();
                commit_a(b + 2, ar[0]); issue_a(b + 4, ar[0]);
            }
        }
        return;
    }
    // ==================================================================== multiply role: wave w = n 64 w .. + 63 (dz planes), all 128 columns (LDS)
    lu32x4 df[2][4][2];                                // [ring of two steps][nt][term]
    auto load_d = [&](int step, lu32x4 (&dst)[4][2]) {
        if (step >= 2 * b_hi) return;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
                dst[nt][sp] = *reinterpret_cast<const lu32x4*>(a.planes + ((((long)step) * 16 + wave * 4 + nt) * 2 + sp) * 1024 + (unsigned)lane * 16u);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool neg = false;
    auto flip = [&]() {
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[kt][nt][r] = -acc[kt][nt][r];
    };
    auto load_bf = [&](const unsigned char* st, int h, int qt, lu32x4 (&f)[2][2]) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int q = 0; q < 2; ++q) f[sp][q] = *reinterpret_cast<const lu32x4*>(st + (sp * LIN_BM + (qt * 2 + q) * 16 + lr) * LF_RS + h * 64 + lk * 16);
    };
    auto multiply = [&](int qt, const lu32x4 (&f)[2][2], const lu32x4 (&d)[4][2]) {
#define LW_PRODUCT(SX, SD)                                                                                                        \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                                          \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                         \
                acc[qt * 2 + q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f[SX][q]),                 \
                                                                             __builtin_bit_cast(f16x8, d[nt][SD]), acc[qt * 2 + q][nt], 0, 0, 0);
        LW_PRODUCT(1, 0) LW_PRODUCT(0, 1) LW_PRODUCT(0, 0)
#undef LW_PRODUCT
    };
    load_d(2 * b_lo, df[0]);
#pragma unroll 1
    for (int blk = b_lo; blk < b_hi; ++blk) {
        const bool want = ((2 * blk) >> 3) & 1;
        if (want != neg) { flip(); neg = want; }
        __syncthreads();
        const unsigned char* st = lds + (blk % LF_NS) * LF_STAGE;
        lu32x4 fa[2][2], fb[2][2];
        load_bf(st, 0, 0, fa);
        load_d(2 * blk + 1, df[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_bf(st, 0, 1, fb); multiply(0, fa, df[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_bf(st, 0, 2, fa); multiply(1, fb, df[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_bf(st, 0, 3, fb); multiply(2, fa, df[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_bf(st, 1, 0, fa); multiply(3, fb, df[0]);
        __builtin_amdgcn_sched_barrier(0);
        load_d(2 * blk + 2, df[0]);
        load_bf(st, 1, 1, fb); multiply(0, fa, df[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_bf(st, 1, 2, fa); multiply(1, fb, df[1]);
        __builtin_amdgcn_sched_barrier(0);
        load_bf(st, 1, 3, fb); multiply(2, fa, df[1]);
        __builtin_amdgcn_sched_barrier(0);
        multiply(3, fb, df[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (neg) flip();
    float* slab = a.partial + (long)blockIdx.y * 256 * a.K;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[kt][nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= unscale;
            *reinterpret_cast<f32x4*>(slab + (long)(wave * 64 + nt * 16 + lr) * a.K + k0 + kt * 16 + lk * 4) = v;
        }
}

__global__ __launch_bounds__(256) void lin_wgrad_reduce(const float* __restrict__ partial, int splits, long n, float* __restrict__ G, long ldg, int K) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(partial + i);
    for (int p = 1; p < splits; ++p) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(partial + (long)p * n + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += v[r];
    }
    float* g = G + (i / K) * ldg + (i % K);
    const f32x4 o = *reinterpret_cast<const f32x4*>(g);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] += o[r];
    *reinterpret_cast<f32x4*>(g) = s;
}

static int lin_wgrad_splits(int M, int K) {
    const int ncol = K / LIN_BM, nblk = (M + 63) / 64;
    int s = (768 + ncol - 1) / ncol;
    if (s > 16) s = 16;
    if (s > nblk) s = nblk;
    return s < 1 ? 1 : s;
}
size_t a2s_linear_wgrad_ws_bytes_impl(int M, int K) {
    if (M < 1 || K < LIN_BM || K % LIN_BM) return 0;
    const size_t nsteps = 2 * (size_t)((M + 63) / 64);
    return nsteps * 16 * 2 * 1024 + (size_t)lin_wgrad_splits(M, K) * 256 * K * sizeof(float);
}
bool a2s_linear_wgrad_ok(int M, int N, int K, long ldz, long lda, long ldg, int period, const void* dz, const void* A, const void* G) {
    return N == 256 && M >= 64 && K >= LIN_BM && K % LIN_BM == 0 && (period == 0 || (period % 4 == 0 && K % period == 0)) && lda % 4 == 0 && ldg % 4 == 0 && ldz >= 256 &&
           (((uintptr_t)dz | (uintptr_t)A | (uintptr_t)G) % 16 == 0);
}
int a2s_linear_wgrad_impl(hipStream_t st, int M, int N, int K, const float* dz, long ldz, const float* A, long lda, float* G, long ldg, const float* a_scale,
                          const float* a_shift, int period, const float* dz_absmax, const float* a_absmax, float* ws, size_t ws_bytes) {
    A2S_REQUIRE(dz && A && G && dz_absmax && ws && ((a_scale == nullptr) == (a_shift == nullptr)), "linear_wgrad: null argument");
    A2S_REQUIRE(a2s_linear_wgrad_ok(M, N, K, ldz, lda, ldg, a_scale ? period : 0, dz, A, G), "linear_wgrad: needs N = 256, K %% 128 == 0, period %% 4 == 0, 16-byte aligned rows");
    A2S_REQUIRE(((uintptr_t)ws % 16 == 0) && ws_bytes >= a2s_linear_wgrad_ws_bytes_impl(M, K), "linear_wgrad: workspace too small (%zu bytes needed)",
                a2s_linear_wgrad_ws_bytes_impl(M, K));
    const int nblk = (M + 63) / 64, nsteps = 2 * nblk, splits = lin_wgrad_splits(M, K);
    unsigned char* planes = reinterpret_cast<unsigned char*>(ws);
    float* partial = reinterpret_cast<float*>(planes + (size_t)nsteps * 16 * 2 * 1024);
    hipLaunchKernelGGL(lin_pack_dz_planes, dim3((unsigned)(((long)nsteps * 1024 + 255) / 256)), dim3(256), 0, st, dz, ldz, M, nsteps, dz_absmax, planes);
    A2S_CHECK_LAUNCH("lin_pack_dz_planes");
    LinWgradArgs a{A, lda, planes, partial, a_scale, a_shift, a_absmax, dz_absmax, M, K, a_scale ? period : K, nblk, (nblk + splits - 1) / splits};
    hipLaunchKernelGGL(lin_wgrad_roles, dim3(K / LIN_BM, splits), dim3(LIN_NTH), 0, st, a);
    A2S_CHECK_LAUNCH("lin_wgrad");
    const long n = 256L * K;
    hipLaunchKernelGGL(lin_wgrad_reduce, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, partial, splits, n, G, ldg, K);
    A2S_CHECK_LAUNCH("lin_wgrad_reduce");
    return A2S_OK;
}
