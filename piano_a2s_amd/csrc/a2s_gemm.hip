// fp32 GEMM on the CDNA4 matrix cores: v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate -- bit-for-bit an
// fmaf chain, so results stay inside the 1e-4 parity budget against the reference's fp32 CPU path).
//
//   C[m,n] = act( alpha * sum_k A(m,k) * B(k,n) + beta * C[m,n] + bias[n] )
//   A(m,k) = A[m*sAm + k*sAk],  B(k,n) = B[k*sBk + n*sBn]   (any strides; 16-byte vector loads when the
//   unit-stride dimension allows it), optional batch dimension, optional deterministic split-K.
//
// Used for every dense contraction of the hot path: Linear 19200->256 (a-2), GRU input
// projections (a-3), attention key projection (a-7), per-step decoder GEMMs (a-9), heads (a-10) and all
// their dgrad / wgrad forms (which are the same contraction with other strides).
//
// Structure: 256 threads = 4 waves; block tile BM x BN x 32; register-prefetched global->LDS staging
// (global loads for tile t+1 are in flight while tile t is multiplied); LDS images padded so the
// fragment reads (lane = (m&15, k>>... ) one dword each) are at most 2-way conflicted.
#include "a2s_common.h"
int a2s_attn_bulk_cap_enabled(void);

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long sAm, sAk, sBk, sBn, ldc;
    float alpha, beta;
    int act;                // 0 none, 1 relu, 2 tanh, 3 exp(2x) (attention key image)
    int batch; long bsA, bsB, bsC;
    int splitk; int kchunk; // kchunk: K range per split (multiple of 32)
    float* partial;         // [batch*splitk][M][N] when splitk > 1
    int vecA, vecB;         // 16-byte loads allowed along the operand's unit-stride dimension
    // optional BatchNorm+ReLU of an operand applied while it is staged: element -> max(0, e * scale[c] + shift[c]) with
    // c = (index along the operand's unit-stride dimension) / period.  Lets the 19200->256 Linear read the last convolution's raw
    // output (forward: A, k-contiguous; weight gradient: B, n-contiguous) -- the activated copy is never written.
    const float* a_scale; const float* a_shift; int a_period;
    const float* b_scale; const float* b_shift; int b_period;
    // optional epilogue: the output C (M rows x N = channels * period columns) is the gradient wrt relu(bn(Y)) of a ConvStack layer; its
    // BatchNorm-backward statistics (sum g', sum g' xhat per channel; g' = g where bn(y) > 0) are accumulated here, one partial row per
    // (row tile, column-tile slot of the channel) in the [blocks][channels][2] layout bn_bwd_finalize reads.  128x128 tiles only.
    const float* ep_y; const float* ep_mean; const float* ep_invstd; const float* ep_scale; const float* ep_shift;
    float* ep_partial; int ep_period, ep_channels, ep_slots;
    // two-term fp16 split path (128x128 tiles): device scalars max|A| / max|B| for the power-of-two operand scales (null = the operand is
    // used unscaled: O(1) activations)
    int two_term; const float* a_absmax; const float* b_absmax;
};

#define GEMM_BK 32
#ifndef GEMM_DEPTH
#define GEMM_DEPTH 1              // k-tiles of global loads in flight in the small fp32-input tiles (-DGEMM_DEPTH=4: measured, see the main loop)
#endif

template <int ROWS, bool KC> struct LdsTile {
    // KC  : image [ROWS][BK+4]   (k contiguous, rows 16-byte aligned -> ds_write_b128)
    // !KC : image [BK][ROWS+16]  (m contiguous)
    static constexpr int STRIDE = KC ? (GEMM_BK + 4) : (ROWS + 16);
    static constexpr int SIZE = KC ? ROWS * STRIDE : GEMM_BK * STRIDE;
    __device__ static __forceinline__ int at(int r, int k) { return KC ? r * STRIDE + k : k * STRIDE + r; }
};

// One operand tile: ROWS (m or n) x 32 (k).  elem(r,k) = P[r*sR + k*sK].
template <int ROWS, bool KC, int NLD, int NTH = 256>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, long sR, long sK, int r0, int k0,
                                          int rmax, int kmax, bool vec, f32x4 (&reg)[NLD], float (&aff)[NLD][2],
                                          const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr, int period = 1) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int slot = tid + i * NTH;
        int r, k;
        if (KC) { r = slot >> 3; k = (slot & 7) * 4; }                 // 8 float4 per row of 32 k
        else    { constexpr int PER = ROWS / 4; k = slot / PER; r = (slot % PER) * 4; }
        if (scale) {      // channel constants of this float4 (aligned quads share a channel when period % 4 == 0): fetched with the operand,
            const int u0 = KC ? k0 + k : r0 + r;                      // so that they too are in flight during the multiply
            const int c = min((int)(((float)u0 + 0.5f) * (1.f / (float)period)), ((KC ? kmax : rmax) - 1) / period);
            aff[i][0] = scale[max(c, 0)]; aff[i][1] = shift[max(c, 0)];
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const bool in_tile = KC ? (r < ROWS) : (k < GEMM_BK);
        if (in_tile) {
            const int gr = r0 + r, gk = k0 + k;
            if (KC) {
                if (gr < rmax) {
                    const float* p = P + (long)gr * sR + (long)gk * sK;
                    if (vec && gk + 3 < kmax) v = *reinterpret_cast<const f32x4*>(p);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gk + j < kmax) v[j] = p[(long)j * sK];
                    }
                }
            } else {
                if (gk < kmax) {
                    const float* p = P + (long)gr * sR + (long)gk * sK;
                    if (vec && gr + 3 < rmax) v = *reinterpret_cast<const f32x4*>(p);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (gr + j < rmax) v[j] = p[(long)j * sR];
                    }
                }
            }
        }
        reg[i] = v;
    }
}

// registers -> LDS.  The optional operand BatchNorm+ReLU is applied HERE, not at load time: the global loads of the next tile stay in
// flight during the multiply of the current one, and the transform only touches them once they have landed.
template <int ROWS, bool KC, int NLD>
__device__ __forceinline__ void tile_store(float* __restrict__ lds, const f32x4 (&reg)[NLD], const float (&aff)[NLD][2], int r0 = 0, int k0 = 0,
                                           int rmax = 0, int kmax = 0, const float* __restrict__ scale = nullptr,
                                           const float* __restrict__ shift = nullptr, int period = 1) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int slot = tid + i * 256;
        int r, k;
        if (KC) { r = slot >> 3; k = (slot & 7) * 4; }
        else    { constexpr int PER = ROWS / 4; k = slot / PER; r = (slot % PER) * 4; }
        const bool in_tile = KC ? (r < ROWS) : (k < GEMM_BK);
        f32x4 v = reg[i];
        if (scale && in_tile) {      // operand affine + ReLU; elements outside the matrix stay exactly 0
            const int gr = r0 + r, gk = k0 + k;
            const int u0 = KC ? gk : gr;                            // index along the unit-stride dimension of the first of the 4
            const bool row_ok = KC ? (gr < rmax) : (gk < kmax);
            const int umax = KC ? kmax : rmax;
            const float inv = 1.f / (float)period;                  // exact channel for indices < 2^22: (u + 0.5) / period truncated
            if ((period & 3) == 0 && (u0 & 3) == 0) {               // the 4 elements share one channel
                const float sc = aff[i][0], sh = aff[i][1];     // prefetched by tile_load
#pragma unroll
                for (int j = 0; j < 4; ++j) if (row_ok && u0 + j < umax) v[j] = fmaxf(fmaf(v[j], sc, sh), 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (row_ok && u0 + j < umax) { const int c = (int)(((float)(u0 + j) + 0.5f) * inv); v[j] = fmaxf(fmaf(v[j], scale[c], shift[c]), 0.f); }
            }
        }
        if (in_tile) *reinterpret_cast<f32x4*>(lds + LdsTile<ROWS, KC>::at(r, k)) = v;
    }
}

// Split path (both operands k-contiguous): registers -> LDS as three bf16 term planes [term][ROWS][32 k (+16 pad)], 96-byte rows so
// that a lane's fragment (8 consecutive k of one row) is one ds_read_b128.  Same optional operand BatchNorm+ReLU as tile_store.
#ifndef GEMM_SPLIT_RS
#define GEMM_SPLIT_RS 96           // (64 bytes of k + 32: an odd multiple of 32 bytes -- ds_read_b128 fragment reads at 4.0 instead of 6.7 clocks: profiles/r04_lds_patterns.txt)
#endif
template <int ROWS, int NLD, int TERMS = 3, int NTH = 256>
__device__ __forceinline__ void tile_store_split(unsigned char* __restrict__ lds, const f32x4 (&reg)[NLD], const float (&aff)[NLD][2], int r0, int k0,
                                                 int rmax, int kmax, const float* __restrict__ scale, const float* __restrict__ shift, int period,
                                                 bool negate, float pscale = 1.f) {
    const int tid = threadIdx.x;
    const unsigned sgn = negate ? 0x80000000u : 0u;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int slot = tid + i * NTH;
        const int r = slot >> 3, k = (slot & 7) * 4;
        if (r >= ROWS) continue;
        f32x4 v = reg[i];
        if (scale) {
            const int gr = r0 + r, gk = k0 + k;
            const bool row_ok = gr < rmax;
            if ((period & 3) == 0 && (gk & 3) == 0) {
                const float sc = aff[i][0], sh = aff[i][1];
#pragma unroll
                for (int j = 0; j < 4; ++j) if (row_ok && gk + j < kmax) v[j] = fmaxf(fmaf(v[j], sc, sh), 0.f);
            } else {
                const float inv = 1.f / (float)period;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (row_ok && gk + j < kmax) { const int c = (int)(((float)(gk + j) + 0.5f) * inv); v[j] = fmaxf(fmaf(v[j], scale[c], shift[c]), 0.f); }
            }
        }
        uint2 o[TERMS];
        if (TERMS == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(__float_as_uint(v[j]) ^ sgn);
            split3_pair(v[0], v[1], o[0].x, o[1].x, o[TERMS - 1].x);
            split3_pair(v[2], v[3], o[0].y, o[1].y, o[TERMS - 1].y);
        } else {
            const float ps = negate ? -pscale : pscale;           // exact power-of-two operand scale (sign: the negated k-tile blocks)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j] * ps, -65000.f, 65000.f);
            split2_pair_f16(v[0], v[1], o[0].x, o[1].x);
            split2_pair_f16(v[2], v[3], o[0].y, o[1].y);
        }
#pragma unroll
        for (int sp = 0; sp < TERMS; ++sp) *reinterpret_cast<uint2*>(lds + (sp * ROWS + r) * GEMM_SPLIT_RS + k * 2) = o[sp];
    }
}

// Split path, operand with the ROW dimension contiguous in memory (weight-gradient forms: K = the long row count).  A thread owns a
// 4 rows x 4 consecutive k block (four 16-byte loads along the rows), so that it can emit k-contiguous bf16 terms for each of its rows;
// thread -> (row quad = tid / 8, k chunk = tid % 8): global segments of 128 bytes, LDS stores conflict-free (16 lanes = 2 row quads x 8
// chunks -> 32 banks).  128-row tiles only.
template <int ROWS>
__device__ __forceinline__ void tile_load_T(const float* __restrict__ P, long sK, int r0, int k0, int rmax, int kmax, bool vec, f32x4 (&reg)[4],
                                            float (&aff)[2], const float* __restrict__ scale, const float* __restrict__ shift, int period) {
    // ROWS = threads / 2 (128-row tiles with 256 threads, 256-row tiles with 512): enforced by the kernel
    const int rq = threadIdx.x >> 3, kc = threadIdx.x & 7;
    const int gr = r0 + rq * 4;
    if (scale) {
        const int c = min((int)(((float)gr + 0.5f) * (1.f / (float)period)), (rmax - 1) / period);
        aff[0] = scale[max(c, 0)]; aff[1] = shift[max(c, 0)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gk = k0 + kc * 4 + i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gk < kmax && gr < rmax) {
            const float* p = P + (long)gk * sK + gr;
            if (vec && gr + 3 < rmax) v = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (gr + j < rmax) v[j] = p[j];
            }
        }
        reg[i] = v;
    }
}

template <int ROWS, int TERMS = 3>
__device__ __forceinline__ void tile_store_split_T(unsigned char* __restrict__ lds, const f32x4 (&reg)[4], const float (&aff)[2], int r0, int k0,
                                                   int rmax, int kmax, const float* __restrict__ scale, const float* __restrict__ shift, int period,
                                                   bool negate, float pscale = 1.f) {
    const int rq = threadIdx.x >> 3, kc = threadIdx.x & 7;
    const unsigned sgn = negate ? 0x80000000u : 0u;
    const int gr = r0 + rq * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                      // row gr + j, k = k0 + 4 kc .. + 3
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = reg[i][j];
        if (scale) {
            float sc = aff[0], sh = aff[1];
            if ((period & 3) != 0) { const int c = (int)(((float)(gr + j) + 0.5f) * (1.f / (float)period)); sc = scale[min(c, (rmax - 1) / period)]; sh = shift[min(c, (rmax - 1) / period)]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) if (gr + j < rmax && k0 + kc * 4 + i < kmax) v[i] = fmaxf(fmaf(v[i], sc, sh), 0.f);
        }
        uint2 o[TERMS];
        if (TERMS == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(__float_as_uint(v[i]) ^ sgn);
            split3_pair(v[0], v[1], o[0].x, o[1].x, o[TERMS - 1].x);
            split3_pair(v[2], v[3], o[0].y, o[1].y, o[TERMS - 1].y);
        } else {
            const float ps = negate ? -pscale : pscale;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i] * ps, -65000.f, 65000.f);
            split2_pair_f16(v[0], v[1], o[0].x, o[1].x);
            split2_pair_f16(v[2], v[3], o[0].y, o[1].y);
        }
#pragma unroll
        for (int sp = 0; sp < TERMS; ++sp) *reinterpret_cast<uint2*>(lds + (sp * ROWS + rq * 4 + j) * GEMM_SPLIT_RS + kc * 8) = o[sp];
    }
}

#ifdef GEMM_TRACE
// timing instrumentation (tools/linear_bench.py --trace): shader-clock stamps of thread 0 of 8 mid-grid workgroups, k-tiles 100..123
__device__ unsigned long long gemm_trace[8 * 24 * 8];
extern "C" int a2s_gemm_trace_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gemm_trace), sizeof(gemm_trace)); }
#define G_STAMP(k) do { if (trace_wg >= 0 && t >= 100 && t < 124 && threadIdx.x == 0) gemm_trace[(trace_wg * 24 + t - 100) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define G_STAMP(k) do {} while (0)
#endif
// SPLIT: 0 = fp32-input MFMA; 3 = three bf16 terms, six products; 2 = two fp16 terms, three products (operands scaled by exact powers of two
// from their max-magnitude scalars, the accumulators unscaled before the epilogue -- see conv3x3_split in a2s_conv.hip)
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int SPLIT = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f32_kernel(GemmArgs g) {
    constexpr int NTH = WM * WN * 64;                   // 256 threads; 512 for the 256x256 two-term tile
    constexpr int TM = BM / (WM * 16), TN = BN / (WN * 16);
    constexpr int NLA = (BM * 8 + NTH - 1) / NTH, NLB = (BN * 8 + NTH - 1) / NTH;
    static_assert(NTH == 256 || SPLIT == 2, "only the two-term split kernel is written for 8 waves");
    using LA = LdsTile<BM, A_KC>;
    using LB = LdsTile<BN, B_KC>;
    // the split paths stage ONE tile as term planes (2 terms, 128x128: 40 KB -- three workgroups per CU where the registers allow);
    // the fp32 path double-buffers fp32 tiles
    constexpr int LDS_FLOATS = SPLIT != 0 ? SPLIT * (BM + BN) * GEMM_SPLIT_RS / 4 : 2 * (LA::SIZE + LB::SIZE);
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int zb = blockIdx.z / g.splitk, zs = blockIdx.z % g.splitk;
    const float* A = g.A + (long)zb * g.bsA;
    const float* B = g.B + (long)zb * g.bsB;
    // XCD-aware tile order: the hardware deals consecutive workgroup ids (x fastest) round-robin to the 8 XCDs, each with its own L2.
    // Logical tile L = (id % 8) * (n / 8) + id / 8 gives every XCD a contiguous run of tiles: the column tiles of one row block (which
    // read the same A rows) and neighbouring row blocks (the same B k-slices at the same time) then share ONE L2.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int n = gridDim.x * gridDim.y, id = by * gridDim.x + bx, per = n / 8;
        if (n >= 64 && id < per * 8) {
            const int l = (id % 8) * per + id / 8;
            bx = l % gridDim.x; by = l / gridDim.x;
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = zs * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave / WN) * (TM * 16), wn = (wave % WN) * (TN * 16);
    const int lr = lane & 15, lk = lane >> 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[NLA], rb[NLB];
    float fa[NLA][2], fb[NLB][2];          // per-float4 operand BatchNorm constants (only touched when an operand affine is given)
    const int ntiles = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
    if (ntiles > 0) {
        tile_load<BM, A_KC, NLA, NTH>(A, g.sAm, g.sAk, m0, kbeg, g.M, kend, g.vecA, ra, fa, g.a_scale, g.a_shift, g.a_period);
        tile_load<BN, B_KC, NLB, NTH>(B, g.sBn, g.sBk, n0, kbeg, g.N, kend, g.vecB, rb, fb, g.b_scale, g.b_shift, g.b_period);
    }
    if constexpr (SPLIT != 0) {
        // fp32 operands as three exact bf16 terms on the bf16 matrix pipes (six term products per k-step of 32, see conv3x3_bf16x3 in
        // a2s_conv.hip); single LDS buffer (2 x 30 KB), the next tile's global loads stay in flight during the multiply.  The pipe truncates
        // its internal sum toward -infinity: every other block of 8 k-tiles accumulates the negated sum (A negated while staging).
        static_assert(SPLIT * (BM + BN) * GEMM_SPLIT_RS <= (int)sizeof(lds), "split path: the term planes must fit the fp32 path's LDS");
        static_assert((A_KC || BM == NTH / 2) && (B_KC || BN == NTH / 2), "split path: row-contiguous operands are staged 4 rows x 4 k per thread");
        unsigned char* la = reinterpret_cast<unsigned char*>(lds);
        unsigned char* lb = la + SPLIT * BM * GEMM_SPLIT_RS;
        bool neg = false;
        float psa = 1.f, psb = 1.f, unscale = 1.f;
        if (SPLIT == 2) {
            const int ka = g.a_absmax ? pow2_scale_exp(*g.a_absmax, 12) : 0, kb = g.b_absmax ? pow2_scale_exp(*g.b_absmax, 12) : 0;
            psa = ldexpf(1.f, ka); psb = ldexpf(1.f, kb); unscale = ldexpf(1.f, -(ka + kb));
        }
        auto flip = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = -acc[i][j][r];
        };
        f32x4 ta[4], tb[4];                   // row-contiguous operands: 4 rows x 4 k per thread (tile_load_T)
        float tfa[2], tfb[2];
        auto load_tile = [&](int k0) {
            if constexpr (A_KC) tile_load<BM, true, NLA, NTH>(A, g.sAm, g.sAk, m0, k0, g.M, kend, g.vecA, ra, fa, g.a_scale, g.a_shift, g.a_period);
            else tile_load_T<BM>(A, g.sAk, m0, k0, g.M, kend, g.vecA, ta, tfa, g.a_scale, g.a_shift, g.a_period);
            if constexpr (B_KC) tile_load<BN, true, NLB, NTH>(B, g.sBn, g.sBk, n0, k0, g.N, kend, g.vecB, rb, fb, g.b_scale, g.b_shift, g.b_period);
            else tile_load_T<BN>(B, g.sBk, n0, k0, g.N, kend, g.vecB, tb, tfb, g.b_scale, g.b_shift, g.b_period);
        };
        if (ntiles > 0 && !(A_KC && B_KC)) load_tile(kbeg);      // (the k-contiguous prologue loads above are dead code for a transposed operand)
#ifdef GEMM_TRACE
        const int trace_id = (int)(blockIdx.y * gridDim.x + blockIdx.x), trace_first = (int)(gridDim.x * gridDim.y) / 2;
        const int trace_wg = (trace_id >= trace_first && trace_id < trace_first + 8) ? trace_id - trace_first : -1;
#endif
        for (int t = 0; t < ntiles; ++t) {
            const bool want = (t >> 3) & 1;
            G_STAMP(0);
            __syncthreads();                  // the previous tile's fragments are consumed
            G_STAMP(1);
            if constexpr (A_KC) tile_store_split<BM, NLA, SPLIT, NTH>(la, ra, fa, m0, kbeg + t * GEMM_BK, g.M, kend, g.a_scale, g.a_shift, g.a_period, want, psa);
            else tile_store_split_T<BM, SPLIT>(la, ta, tfa, m0, kbeg + t * GEMM_BK, g.M, kend, g.a_scale, g.a_shift, g.a_period, want, psa);
            if constexpr (B_KC) tile_store_split<BN, NLB, SPLIT, NTH>(lb, rb, fb, n0, kbeg + t * GEMM_BK, g.N, kend, g.b_scale, g.b_shift, g.b_period, false, psb);
            else tile_store_split_T<BN, SPLIT>(lb, tb, tfb, n0, kbeg + t * GEMM_BK, g.N, kend, g.b_scale, g.b_shift, g.b_period, false, psb);
            G_STAMP(2);
            __syncthreads();
            G_STAMP(3);
            if (t + 1 < ntiles) load_tile(kbeg + (t + 1) * GEMM_BK);
            G_STAMP(4);
            if (want != neg) { flip(); neg = want; }
            typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
            gu32x4 af[SPLIT][TM];
#pragma unroll
            for (int sp = 0; sp < SPLIT; ++sp)
#pragma unroll
                for (int i = 0; i < TM; ++i) af[sp][i] = *reinterpret_cast<const gu32x4*>(la + (sp * BM + wm + i * 16 + lr) * GEMM_SPLIT_RS + lk * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                gu32x4 bf[SPLIT];
#pragma unroll
                for (int sp = 0; sp < SPLIT; ++sp) bf[sp] = *reinterpret_cast<const gu32x4*>(lb + (sp * BN + wn + j * 16 + lr) * GEMM_SPLIT_RS + lk * 16);
#define GEMM_PRODUCT(SA, SB)                                                                                                    \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                  \
                    acc[i][j] = (SPLIT == 3) ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[SB]), __builtin_bit_cast(bf16x8, af[SA][i]), acc[i][j], 0, 0, 0) \
                                             : __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[SB]), __builtin_bit_cast(f16x8, af[SA][i]), acc[i][j], 0, 0, 0);
                if (SPLIT == 3) { GEMM_PRODUCT(SPLIT - 1, 0) GEMM_PRODUCT(1, 1) GEMM_PRODUCT(0, SPLIT - 1) }
                GEMM_PRODUCT(1, 0) GEMM_PRODUCT(0, 1) GEMM_PRODUCT(0, 0)
#undef GEMM_PRODUCT
            }
            G_STAMP(5);
        }
        if (neg) flip();
        if (SPLIT == 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] *= unscale;
        }
    } else {
        // fp32-input path.  GEMM_DEPTH register sets of global loads in a ring (small tiles only).  Round 5 measured depth 4 against 1 on the
        // decoder's per-step products (M = 512 / 1280 rows, N = 1536, K = 528: profiles/r05_gemm_depth.txt): 23.9 -> 23.0 us alone -- these
        // launches are bound by the L2 traffic of their 64x32 tiles (78 MB in 23 us), not by a memory round trip per k-tile -- and the training
        // step 450 -> 460 ms (164-208 VGPRs instead of ~100: fewer workgroups per CU beside the attention sweeps).  Depth 1 stays.
        constexpr int D = (BM * BN <= 64 * 64) ? GEMM_DEPTH : 1;
        f32x4 qa[D][NLA], qb[D][NLB];
        float ga[D][NLA][2], gb[D][NLB][2];
#pragma unroll
        for (int i = 0; i < NLA; ++i) { qa[0][i] = ra[i]; ga[0][i][0] = fa[i][0]; ga[0][i][1] = fa[i][1]; }
#pragma unroll
        for (int i = 0; i < NLB; ++i) { qb[0][i] = rb[i]; gb[0][i][0] = fb[i][0]; gb[0][i][1] = fb[i][1]; }
#pragma unroll
        for (int d = 1; d < D; ++d) {
            if (d < ntiles) {
                const int k0 = kbeg + d * GEMM_BK;
                tile_load<BM, A_KC, NLA>(A, g.sAm, g.sAk, m0, k0, g.M, kend, g.vecA, qa[d], ga[d], g.a_scale, g.a_shift, g.a_period);
                tile_load<BN, B_KC, NLB>(B, g.sBn, g.sBk, n0, k0, g.N, kend, g.vecB, qb[d], gb[d], g.b_scale, g.b_shift, g.b_period);
            }
        }
        for (int t0 = 0; t0 < ntiles; t0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int t = t0 + d;
                if (t >= ntiles) break;                   // (uniform over the workgroup)
                float* la = lds + (t & 1) * (LA::SIZE + LB::SIZE);
                float* lb = la + LA::SIZE;
                tile_store<BM, A_KC, NLA>(la, qa[d], ga[d], m0, kbeg + t * GEMM_BK, g.M, kend, g.a_scale, g.a_shift, g.a_period);
                tile_store<BN, B_KC, NLB>(lb, qb[d], gb[d], n0, kbeg + t * GEMM_BK, g.N, kend, g.b_scale, g.b_shift, g.b_period);
                __syncthreads();
                if (t + D < ntiles) {   // refill this register set: its tile is multiplied D iterations from now
                    const int k0 = kbeg + (t + D) * GEMM_BK;
                    tile_load<BM, A_KC, NLA>(A, g.sAm, g.sAk, m0, k0, g.M, kend, g.vecA, qa[d], ga[d], g.a_scale, g.a_shift, g.a_period);
                    tile_load<BN, B_KC, NLB>(B, g.sBn, g.sBk, n0, k0, g.N, kend, g.vecB, qb[d], gb[d], g.b_scale, g.b_shift, g.b_period);
                }
#pragma unroll
                for (int ks = 0; ks < GEMM_BK / 4; ++ks) {
                    float a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = la[LA::at(wm + i * 16 + lr, ks * 4 + lk)];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = lb[LB::at(wn + j * 16 + lr, ks * 4 + lk)];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);     // transposed tile: see the epilogue
                }
                // the other LDS buffer is written next iteration; its readers finished before the barrier above
            }
        }
    }


    // epilogue.  The MFMAs were issued with the operands swapped (B fragment first), so every accumulator holds the TRANSPOSED 16x16 tile:
    // lane = (m = lane & 15, n-quad = lane >> 4) owns C[m][4q .. 4q+3] -- four CONSECUTIVE columns of one output row, i.e. one 16-byte
    // store per tile and lane instead of four scattered 4-byte stores (the 19200-wide data-gradient GEMM writes 23.6 GB this way).
    const bool partial = g.splitk > 1;
    float* C = partial ? g.partial + (long)blockIdx.z * g.M * g.N : g.C + (long)zb * g.bsC;
    const long ldc = partial ? g.N : g.ldc;
    const bool vec_c = (ldc % 4 == 0) && (((uintptr_t)C & 15) == 0);
    const int ep_c0 = g.ep_y ? n0 / g.ep_period : 0;             // first channel this column tile touches (it touches at most two)
    float ep_s[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm + i * 16 + lr;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + j * 16 + lk * 4;
            if (n >= g.N) continue;
            f32x4 v = acc[i][j];
            float* dst = C + (long)m * ldc + n;
            const bool full = vec_c && n + 3 < g.N;
            if (!partial) {
                f32x4 old = {0.f, 0.f, 0.f, 0.f};
                if (g.beta != 0.f) {
                    if (full) old = *reinterpret_cast<const f32x4*>(dst);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < g.N) old[r] = dst[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = g.alpha * v[r] + ((g.bias && n + r < g.N) ? g.bias[n + r] : 0.f);
                    if (g.beta != 0.f) x += g.beta * old[r];
                    if (g.act == 1) x = fmaxf(x, 0.f);
                    else if (g.act == 2) x = fast_tanh(x);
                    else if (g.act == 3) x = exp2x_clamped(x);
                    v[r] = x;
                }
            }
            if (full) *reinterpret_cast<f32x4*>(dst) = v;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < g.N) dst[r] = v[r];
            }
            if (g.ep_y && full) {      // (host guarantees N % 4 == 0, period % 4 == 0: a quad never straddles a channel)
                const int c = n / g.ep_period;
                const f32x4 yq = *reinterpret_cast<const f32x4*>(g.ep_y + (long)m * ldc + n);
                const float mean = g.ep_mean[c], invstd = g.ep_invstd[c], sc = g.ep_scale[c], sh = g.ep_shift[c];
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gm = (yq[r] * sc + sh > 0.f) ? v[r] : 0.f;
                    a1 += gm; a2 += gm * (yq[r] - mean) * invstd;
                }
                if (c == ep_c0) { ep_s[0][0] += a1; ep_s[0][1] += a2; } else { ep_s[1][0] += a1; ep_s[1][1] += a2; }
            }
        }
    }
    if (g.ep_y) {                      // workgroup totals of the (at most) two channels -> their (row tile, slot) partial rows
        float* red = lds;              // the operand tiles are dead by now
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float w = wave_sum(ep_s[sl][k]);
                if (lane == 0) red[wave * 4 + sl * 2 + k] = w;
            }
        __syncthreads();
        if (threadIdx.x < 4) {
            const int sl = threadIdx.x >> 1, k = threadIdx.x & 1;
            const int c = ep_c0 + sl;
            if (c < g.ep_channels && (long)c * g.ep_period < (long)n0 + BN) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < WM * WN; ++w) tot += red[w * 4 + threadIdx.x];
                const int slot = bx - (int)(((long)c * g.ep_period) / BN);        // which of the channel's column tiles this is
                g.ep_partial[(((long)by * g.ep_slots + slot) * g.ep_channels + c) * 2 + k] = tot;
            }
        }
    }
}

// Fixed-order reduction of the split-K partial slabs (deterministic), with the epilogue applied once.
__global__ void gemm_splitk_reduce(GemmArgs g) {
    const long mn = (long)g.M * g.N;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int zb = blockIdx.y;
    if (idx >= mn) return;
    const float* p = g.partial + (long)zb * g.splitk * mn + idx;
    float s = 0.f;
    for (int i = 0; i < g.splitk; ++i) s += p[(long)i * mn];
    const int m = (int)(idx / g.N), n = (int)(idx % g.N);
    float* c = g.C + (long)zb * g.bsC + (long)m * g.ldc + n;
    float v = g.alpha * s + (g.bias ? g.bias[n] : 0.f);
    if (g.beta != 0.f) v += g.beta * *c;
    if (g.act == 1) v = fmaxf(v, 0.f);
    else if (g.act == 2) v = fast_tanh(v);
    else if (g.act == 3) v = exp2x_clamped(v);
    *c = v;
}

// 1: 128x128 launches with two k-contiguous operands run on the bf16 matrix pipes with 3-term split operands (a2s_debug_set "gemm_bf16x3")
static int g_gemm_split = 1;
void a2s_gemm_split_set(int on) { g_gemm_split = on; }
int a2s_gemm_split_enabled(void) { return g_gemm_split; }
// ... two fp16 terms instead of three bf16 terms where the caller vouches for the operands' ranges (a2s_gemm_f32_desc: two_term)
static int g_gemm_f16x2 = -1;
void a2s_gemm_f16x2_set(int on) { g_gemm_f16x2 = on; }
int a2s_gemm_f16x2_enabled(void) {
    if (g_gemm_f16x2 < 0) g_gemm_f16x2 = 1;          // (a2s_debug_set("gemm_f16x2", 0) / A2S_ARITH=bf16x3: the three-term bf16 split)
    return g_gemm_f16x2;
}

// max |x| of a tensor into a device scalar (atomicMax on the float bits; *out must be zero before): the power-of-two operand scales of
// the two-term fp16 kernels
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    __shared__ float red[16];
    block_absmax_to(out, m, red);
}
int a2s_absmax_impl(hipStream_t st, const float* x, long n, float* out) {
    A2S_REQUIRE(x && out && n >= 0, "absmax: null tensor");
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "absmax memset: %s", hipGetErrorString(e));
    if (n == 0) return A2S_OK;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)min((long)1024, (n + 255) / 256)), dim3(256), 0, st, x, n, out);
    A2S_CHECK_LAUNCH("absmax_kernel");
    return A2S_OK;
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(const GemmArgs& g, bool akc, bool bkc, hipStream_t st) {
    dim3 grid(a2s_cdiv(g.N, BN), a2s_cdiv(g.M, BM), g.batch * g.splitk);
    if constexpr (BM == 128 && BN == 128) {
        // row-contiguous operands need unit row stride and 16-byte loads for the transposed staging (else the fp32-input path)
        const bool a_ok = akc || (g.sAm == 1 && g.vecA), b_ok = bkc || (g.sBn == 1 && g.vecB);
        if (g_gemm_split && g.K >= (g.two_term ? 128 : 256) && a_ok && b_ok) {      // (a short K amortises the two-term staging, not the three-term one)
            if (g.two_term && a2s_gemm_f16x2_enabled()) {
                if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true, 2>), grid, dim3(256), 0, st, g);
                else if (akc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false, 2>), grid, dim3(256), 0, st, g);
                else if (bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true, 2>), grid, dim3(256), 0, st, g);
                else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false, 2>), grid, dim3(256), 0, st, g);
                return;
            }
            if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true, 3>), grid, dim3(256), 0, st, g);
            else if (akc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false, 3>), grid, dim3(256), 0, st, g);
            else if (bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true, 3>), grid, dim3(256), 0, st, g);
            else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false, 3>), grid, dim3(256), 0, st, g);
            return;
        }
    }
    // Occupancy cap of the bulk clip group's per-step products (M >= 256 rows, small tiles) while another clip group decodes beside it
    // (a2s_attn_bulk_cap_enabled(): the same condition as the attention sweeps' cap): 16 KB of unused dynamic LDS per workgroup.
    // The long-clip chain's kernels wait for a place beside whole grids of these workgroups (profiles/r05_trace_overlap.txt).  Default 16 KB (3 instead of
    // 4 workgroups of the 64x32 tile per CU): 447.3 -> 443.5 ms per step, 32 KB 445.2 (profiles/r05_prefix_percent.txt).
    const size_t pad = (BM * BN <= 64 * 64 && g.M >= 256 && a2s_attn_bulk_cap_enabled()) ? 16384 : 0;
    if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), pad, st, g);
    else if (akc && !bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), pad, st, g);
    else if (!akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), pad, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), pad, st, g);
}

// 256x256 tiles, 8 waves, two-term fp16 split: the big two-term products (19200 -> 256 Linear forward and weight gradient).  The 128x128
// kernel spends its k-tile period feeding operands into the CU (tools/linear_bench.py trace: 1.3 k clocks issuing the next tile's loads
// + 2.8 k waiting for them against 1.2 k of multiply): the time goes with the bytes staged per MFMA, and a 256x256 tile stages half.
static bool big_two_term_ok(const GemmArgs& g, bool akc, bool bkc) {
    const bool a_ok = akc || (g.sAm == 1 && g.vecA), b_ok = bkc || (g.sBn == 1 && g.vecB);
    // (the BatchNorm-statistics epilogue keeps its partial layout: a 256-column tile touches at most two channels of period >= 256 and
    // at most as many tiles per channel as the 128-column layout has slots; unused partial rows stay zero)
    return g.two_term && a2s_gemm_f16x2_enabled() && g_gemm_split && (!g.ep_y || g.ep_period >= 256) && g.K >= 128 && a_ok && b_ok && g.M >= 256 && g.N >= 256 &&
           (long)a2s_cdiv(g.M, 256) * a2s_cdiv(g.N, 256) * g.batch * g.splitk >= 192;
}
static void launch_big_two_term(const GemmArgs& g, bool akc, bool bkc, hipStream_t st) {
    dim3 grid(a2s_cdiv(g.N, 256), a2s_cdiv(g.M, 256), g.batch * g.splitk);
    if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<256, 256, 4, 2, true, true, 2>), grid, dim3(512), 0, st, g);
    else if (akc) hipLaunchKernelGGL((gemm_f32_kernel<256, 256, 4, 2, true, false, 2>), grid, dim3(512), 0, st, g);
    else if (bkc) hipLaunchKernelGGL((gemm_f32_kernel<256, 256, 4, 2, false, true, 2>), grid, dim3(512), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<256, 256, 4, 2, false, false, 2>), grid, dim3(512), 0, st, g);
}

size_t a2s_gemm_workspace_bytes_impl(int M, int N, int batch, int splitk) {
    return splitk > 1 ? (size_t)batch * splitk * M * N * sizeof(float) : 0;
}

// Heuristic split count for contractions with a small output and a huge K (wgrad forms).
int a2s_gemm_pick_splitk_impl(int M, int N, int K, int batch) {
    const long tiles = (long)a2s_cdiv(M, 64) * a2s_cdiv(N, 64) * batch;
    if (K < 4096) return 1;
    if (tiles >= 256) {
        // enough 64x64 tiles, but the launch uses 128x128 tiles, two workgroups per CU = 512 slots: a tile count that is not a
        // multiple of the slots leaves CUs idle for the whole (long) K loop (Linear 19200->256 wgrad: 300 tiles = 59 % of the slots).
        // Pick the split that fills whole rounds best.
        const long t128 = (long)a2s_cdiv(M, 128) * a2s_cdiv(N, 128) * batch;
        if (t128 >= 2048 || K < 32768) return 1;
        // (judged for both tilings the launch may use: 128x128 at two workgroups per CU, 256x256 two-term at one)
        const long t256 = (long)a2s_cdiv(M, 256) * a2s_cdiv(N, 256) * batch;
        int best = 1; double best_eff = 0.0;
        for (int s = 1; s <= 12; ++s) {
            const long wg = t128 * s, wg2 = t256 * s;
            double eff = (double)wg / (double)(a2s_cdiv(wg, 512) * 512);
            if (M >= 256 && N >= 256) eff = fmin(eff, (double)wg2 / (double)(a2s_cdiv(wg2, 256) * 256));
            if (eff > best_eff + 0.02) { best_eff = eff; best = s; }
        }
        return best;
    }
    // Mid-size outputs with a huge K (the decoder's deferred weight gradients: 1536 x 528 / 1536 x 512 / 173 x 1024 outputs, K = steps x
    // rows = tens of thousands): with the split below they would run as ~430 workgroups of 64x64 fp32-input tiles (~55 TFLOP/s, 59 ms
    // per step in profiles/r02_kernel_stats.txt).  Split K far enough that the 128x128 tiles -- the split-operand path, ~2x the rate --
    // give one workgroup per CU instead.
    const long t128 = (long)a2s_cdiv(M, 128) * a2s_cdiv(N, 128) * batch;
    if (K >= 8192 && t128 >= 12) {
        long s = a2s_cdiv(256, t128);
        if (s > K / 2048) s = K / 2048;
        if (s > 16) s = 16;
        if (s >= 1 && t128 * s >= 192) return (int)s;
    }
    long s = 512 / tiles;
    const long maxs = K / 512;
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

// column tiles (of 128) one channel of `period` columns can touch: the partial rows per row tile of the BatchNorm-statistics epilogue
int a2s_gemm_bnstats_slots(int period) { return period > 0 ? (period + 127) / 128 + 1 : 0; }

// tuning aid (tools/gemm_sweep.py): force a tile configuration for M > 64; 0 = the heuristic below
static int g_force_tile = 0;
void a2s_gemm_debug_tile_impl(int cfg) { g_force_tile = cfg; }

int a2s_gemm_affine_impl(hipStream_t st, int M, int N, int K, float alpha, const float* A, long sAm, long sAk,
                  const float* B, long sBk, long sBn, float beta, float* C, long ldc, const float* bias, int act,
                  int batch, long bsA, long bsB, long bsC, int splitk, float* ws, size_t ws_bytes,
                  const float* a_scale, const float* a_shift, int a_period, const float* b_scale, const float* b_shift, int b_period,
                  const float* ep_y, const float* ep_mean, const float* ep_invstd, const float* ep_scale, const float* ep_shift,
                  float* ep_partial, int ep_period, int two_term, const float* a_absmax, const float* b_absmax) {
    if (M <= 0 || N <= 0 || batch <= 0) return A2S_OK;
    A2S_REQUIRE(K >= 0 && A && B && C, "gemm: null operand or negative K");
    A2S_REQUIRE(splitk >= 0, "gemm: splitk must be >= 0 (0 = choose automatically when a workspace is given)");
    // Mid-size row counts (the per-step products of a fused-bars decoder call: M = bars x batch rows, N <= 1792, K <= 1792):
    // tools/gemm_sweep.py on MI355X -- 64x32 tiles beat 64x64 up to M = 1024 (more workgroups for the same work), and K is split
    // 4 ways only while the tile count stays under one workgroup per CU.
    const bool mid = M > 64 && M <= 2048 && batch == 1 && N <= 2048;
    int mid_tile = 0;
    if (mid) mid_tile = M <= 1024 ? 2 : 3;                 // 2: 64x32, 3: 64x64
    if (splitk == 0) {
        // Skinny per-step products (M = batch rows, K up to 1792): a handful of output tiles would leave most of the 256 CUs
        // idle and serialise 50+ K-tiles behind one barrier each; spread K over workgroups instead (deterministic reduce).
        splitk = 1;
        if (mid) {
            const long tiles = (long)a2s_cdiv(M, 64) * a2s_cdiv(N, mid_tile == 2 ? 32 : 64);
            if (ws && tiles < 256 && K >= 512 && a2s_gemm_workspace_bytes_impl(M, N, batch, 4) <= ws_bytes) splitk = 4;
        } else {
            const long bm = M <= 16 ? 16 : (M <= 32 ? 32 : 64), bn = M <= 64 ? (M <= 16 ? 64 : (M <= 32 ? 64 : 32)) : 64;
            const long tiles = (long)a2s_cdiv(M, bm) * a2s_cdiv(N, bn) * batch;
            if (ws && tiles < 192 && K >= 256) {
                long s = (256 + tiles - 1) / tiles;
                if (s > K / 64) s = K / 64;
                if (s > 16) s = 16;
                while (s > 1 && a2s_gemm_workspace_bytes_impl(M, N, batch, (int)s) > ws_bytes) --s;
                if (s > 1) splitk = (int)s;
            }
        }
    } else {
        mid_tile = 0;                                       // explicit split count: the caller's (wgrad-type) shape rules apply
    }
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.sAm = sAm; g.sAk = sAk; g.sBk = sBk; g.sBn = sBn; g.ldc = ldc; g.alpha = alpha; g.beta = beta; g.act = act;
    g.batch = batch; g.bsA = bsA; g.bsB = bsB; g.bsC = bsC;
    g.splitk = splitk; g.partial = ws;
    g.two_term = two_term; g.a_absmax = a_absmax; g.b_absmax = b_absmax;
    A2S_REQUIRE((a_scale == nullptr) == (a_shift == nullptr) && (b_scale == nullptr) == (b_shift == nullptr), "gemm: operand scale/shift must come together");
    A2S_REQUIRE((!a_scale || a_period > 0) && (!b_scale || b_period > 0), "gemm: operand affine needs a positive period");
    A2S_REQUIRE(!a_scale || sAk == 1 || sAm == 1, "gemm: operand affine needs a unit-stride dimension on A");
    A2S_REQUIRE(!b_scale || sBk == 1 || sBn == 1, "gemm: operand affine needs a unit-stride dimension on B");
    g.a_scale = a_scale; g.a_shift = a_shift; g.a_period = a_period > 0 ? a_period : 1;
    g.b_scale = b_scale; g.b_shift = b_shift; g.b_period = b_period > 0 ? b_period : 1;
    g.ep_y = ep_y; g.ep_mean = ep_mean; g.ep_invstd = ep_invstd; g.ep_scale = ep_scale; g.ep_shift = ep_shift; g.ep_partial = ep_partial;
    g.ep_period = ep_period > 0 ? ep_period : 1; g.ep_channels = ep_y ? N / g.ep_period : 0; g.ep_slots = ep_y ? a2s_gemm_bnstats_slots(ep_period) : 0;
    if (ep_y) {
        A2S_REQUIRE(ep_mean && ep_invstd && ep_scale && ep_shift && ep_partial && ep_period >= 128 && ep_period % 4 == 0 && N % ep_period == 0,
                    "gemm: the BatchNorm-statistics epilogue needs its tensors, period >= 128, period %% 4 == 0, N a multiple of the period");
        A2S_REQUIRE(splitk == 1 && batch == 1 && M > 64 && ldc % 4 == 0 && ((uintptr_t)C % 16 == 0) && ((uintptr_t)ep_y % 16 == 0) && beta == 0.f,
                    "gemm: the BatchNorm-statistics epilogue needs one split, one batch, M > 64, beta 0 and 16-byte aligned rows");
        hipError_t e = hipMemsetAsync(ep_partial, 0, sizeof(float) * 2 * (size_t)a2s_cdiv(M, 128) * g.ep_slots * g.ep_channels, st);
        if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "gemm bn-stats memset: %s", hipGetErrorString(e));
    }
    g.kchunk = a2s_cdiv(a2s_cdiv(K, splitk), GEMM_BK) * GEMM_BK;
    if (g.kchunk == 0) g.kchunk = GEMM_BK;
    if (splitk > 1)
        A2S_REQUIRE(ws && ws_bytes >= a2s_gemm_workspace_bytes_impl(M, N, batch, splitk), "gemm: split-K workspace too small");
    // operand orientation: which dimension is unit-stride decides the staging path
    const bool akc = (sAk == 1) || (sAm != 1);     // generic strides go through the (scalar) k-major path
    const bool bkc = (sBk == 1) || (sBn != 1);
    auto aligned = [](const void* p, long s0, long s1) { return ((uintptr_t)p % 16 == 0) && (s0 % 4 == 0) && (s1 % 4 == 0); };
    g.vecA = (akc ? (sAk == 1 && aligned(A, sAm, bsA)) : (sAm == 1 && aligned(A, sAk, bsA))) ? 1 : 0;
    g.vecB = (bkc ? (sBk == 1 && aligned(B, sBn, bsB)) : (sBn == 1 && aligned(B, sBk, bsB))) ? 1 : 0;

    if (!g_force_tile && big_two_term_ok(g, akc, bkc)) launch_big_two_term(g, akc, bkc, st);
    else if (ep_y) launch_cfg<128, 128, 2, 2>(g, akc, bkc, st);
    else if (!g_force_tile && mid_tile == 2 && g.splitk >= 1 && M * (long)N < (1L << 22)) launch_cfg<64, 32, 4, 1>(g, akc, bkc, st);
    else if (!g_force_tile && mid_tile == 3 && M * (long)N < (1L << 22)) launch_cfg<64, 64, 2, 2>(g, akc, bkc, st);
    else if (g_force_tile && M > 64) {
        switch (g_force_tile) {
            case 1: launch_cfg<32, 64, 2, 2>(g, akc, bkc, st); break;
            case 2: launch_cfg<64, 32, 4, 1>(g, akc, bkc, st); break;
            case 3: launch_cfg<64, 64, 2, 2>(g, akc, bkc, st); break;
            default: launch_cfg<128, 128, 2, 2>(g, akc, bkc, st); break;
        }
    }
    else if (M <= 16) launch_cfg<16, 64, 1, 4>(g, akc, bkc, st);
    else if (M <= 32) launch_cfg<32, 64, 2, 2>(g, akc, bkc, st);
    else if (M <= 64) launch_cfg<64, 32, 4, 1>(g, akc, bkc, st);
    else if ((long)a2s_cdiv(M, 128) * a2s_cdiv(N, 128) * batch * splitk >= 192) launch_cfg<128, 128, 2, 2>(g, akc, bkc, st);
    else launch_cfg<64, 64, 2, 2>(g, akc, bkc, st);
    A2S_CHECK_LAUNCH("gemm_f32_kernel");
    if (splitk > 1) {
        dim3 grid(a2s_cdiv((long)M * N, 256), batch);
        hipLaunchKernelGGL(gemm_splitk_reduce, grid, dim3(256), 0, st, g);
        A2S_CHECK_LAUNCH("gemm_splitk_reduce");
    }
    return A2S_OK;
}

int a2s_gemm_impl(hipStream_t st, int M, int N, int K, float alpha, const float* A, long sAm, long sAk,
                  const float* B, long sBk, long sBn, float beta, float* C, long ldc, const float* bias, int act,
                  int batch, long bsA, long bsB, long bsC, int splitk, float* ws, size_t ws_bytes) {
    return a2s_gemm_affine_impl(st, M, N, K, alpha, A, sAm, sAk, B, sBk, sBn, beta, C, ldc, bias, act, batch, bsA, bsB, bsC, splitk, ws, ws_bytes,
                                nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr);
}
