// Row-streaming weight gradient of the 3x3 convolutions (autograd of nn.Conv2d, reference models.py:525-534), round 3.
//
//   dW[co][ci][dt][df] = sum_{b,t,p} dz[b][t][co][p] * a[b][t+dt-1][ci][p+df-1]        a = relu(x * scale[ci] + shift[ci])  (or x itself)
//
// i.e. a GEMM whose K index is the POSITION: M = 16-channel tiles of dz, N = (dt, ci) x df, both operands position-contiguous exactly as
// they lie in memory -- a 16-byte global load of 4 positions of one channel becomes 8 + 8 bytes of the two fp16 term images
// [channel][position] in LDS, and a fragment is one ds_read_b128 of 8 consecutive positions.  The only shift is df on the input side: a
// lane reads its aligned 8 positions plus the 8 bytes before and after; df = 1 is the aligned window, df = 0 / 2 are four v_alignbit_b32
// each (one window serves the three df fragments of a (dt, ci) column tile).  What round 2's kernel (conv3x3_wgrad_split, a2s_conv.hip:
// 25 ms at B = 256 for 40 -> 40, 3x its HBM time) paid for and this one does not: it tiled (clip, 2 rows, 64 columns) and re-staged the
// input rows for every tile (3/2 x 66/64), shifted the operand per (tap, channel tile) with 4-byte neighbour reads (4-way bank conflicts),
// and ran its staging and multiply phases one after the other.
//
//   * a workgroup walks DOWN a (clip, 128-column) strip over all T rows: every input row is loaded and converted once into a ring of 4
//     LDS slots (rows t-1, t, t+1 feed dt = 0, 1, 2), every dz row once into a double buffer; one barrier per row;
//   * TWO waves per SIMD in fixed roles: waves 0-3 ("multiply", one per SIMD) only read fragments and issue MFMAs, waves 4-7 ("staging",
//     one per SIMD) only load, convert and write the next rows.  Measured on the way here (40 -> 40, B = 64, profiles/r03_wgrad_rows.txt):
//     8 waves that each mix MFMA and conversion work 5.07 ms (two such waves on a SIMD starve each other's vector issue); one wave per
//     SIMD with 512 registers doing both 4.1 ms (a single in-order wave hides ~2 issue slots per 16-cycle MFMA and this path has ~2
//     non-MFMA instructions per MFMA: every fragment wait stalls the conversions too); the role split 3.7 ms;
//   * multiply wave = two 16-wide (dt, ci) column tiles (one for 20 input channels) x all channel tiles of dz x df: 18 accumulators
//     (9 / 6); the fragments of k-step ks + 1 are read between the MFMAs of k-step ks (sched_group_barrier: 3 MFMA : 1 read : 2 VALU);
//   * the staging waves keep the global loads of two rows in flight (two register sets);
//   * persistent workgroups (one per CU), one partial slab [Cout][Cin][9] each, summed in fixed order by wgrad_rows_reduce;
//   * operands as two exact fp16 terms / three MFMA products, scaled by exact powers of two from their ranges (dz: max|dz| scalar written by
//     the kernel that produced dz; activated input: the a2s_act_bound scalar); odd work items accumulate the negated
//     sum (dz negated while staging, accumulators flipped) against the matrix pipe's truncation bias.
// What bounds it now (tools/wgrad_rows_check.py --trace, -DWR_TRACE): the multiply waves wait 2-5 % of their time in the row barrier, the
// staging waves 40-57 %: the multiply stream is the critical one at ~5200 cycles per row against 3672 of bare MFMA issue (216 x 17), and
// the shader clock falls from ~2.3 GHz to ~1.7 GHz once the HBM stream runs beside the matrix pipes (same cycles per row with the loads
// redirected to one resident row: 2.83 ms instead of 3.72 ms) -- power, not latency: 3 or 4 rows of loads in flight change nothing.
#include "a2s_common.h"

#define WR_TP 128              // positions per strip
#ifndef WR_SROW
#define WR_SROW 288            // bytes per channel row of an fp16 image (8 margin + 128 positions + 8); 272 / 304 measured the same
#endif
#define WR_SLABS 256
#ifndef WR_X
// ablation bits (timing only, results are wrong): 1 no MFMA, 4 no df shifts, 8 no conversion arithmetic, 16 no global loads,
// 32 all loads from row 0 (cache-resident), 64 staging waves idle, 128 no neighbour reads (the shifts use the lane's own window)
#define WR_X 0
#endif
#ifndef WR_KD
#define WR_KD 14              // exponent the largest |dz| is scaled to
#endif
#ifndef WR_SCHED2
#define WR_SCHED2 1            // multiply waves: ask the scheduler for 3 MFMA : 1 LDS read : 2 VALU groups
#endif
#ifndef WR_NS
#define WR_NS 2                // rows of global loads in flight per workgroup (register sets of the staging waves); 3 / 4 measured the same
#endif
#ifndef WR_PRIO
#define WR_PRIO 1              // 1: staging waves at priority 1, 2: multiply waves at priority 1, 0: neither (all three measured the same)
#endif
// WR_TRACE: cycles each role spends waiting in the per-row barrier (which role is the critical one): per workgroup {wait, total} of multiply
// wave 0 and staging wave 4, written behind the slabs (tools/wgrad_rows_check.py --trace)
#ifdef WR_TRACE
#define WR_BAR_INIT() long long wr_wait = 0; const long long wr_t0 = __builtin_readcyclecounter()
#define WR_BAR() do { const long long b0 = __builtin_readcyclecounter(); __syncthreads(); wr_wait += __builtin_readcyclecounter() - b0; } while (0)
#define WR_BAR_OUT(role) do { if ((tid & 255) == 0) { long long* o = reinterpret_cast<long long*>(a.partial + (long)gridDim.x * COUT * CIN * 9) + (blockIdx.x * 2 + role) * 2; \
                              o[0] = wr_wait; o[1] = __builtin_readcyclecounter() - wr_t0; } } while (0)
#else
#define WR_BAR_INIT()
#define WR_BAR() __syncthreads()
#define WR_BAR_OUT(role)
#endif

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
struct WgRowsArgs {
    const float* dy; const float* x; const float* in_scale; const float* in_shift;
    const float* dy_absmax; const float* act_absmax;
    float* partial;
    int B, T, F, tilesF, nwork;
    // fused BatchNorm backward (BNF instances): `dy` is then g = the gradient wrt relu(bn(y)); the staging waves form dz = bn_bwd(g, y) per
    // element (constants bn_k[Cout][6] = mean, invstd, scale, shift, c1, c2), write it to dz_out for the data-gradient convolution and fold
    // max |dz| into dz_absmax_out; dy_absmax is then an upper BOUND of |dz| (a2s_bn_bwd_bound)
    const float* bn_y; const float* bn_k; float* dz_out; float* dz_absmax_out;
};

template <int CIN, int COUT, bool BNF = false>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_rows(WgRowsArgs a) {
    constexpr int NT = (3 * CIN + 15) / 16;          // (dt, ci) column tiles: 8 / 4
    constexpr int NTW = NT / 4;                      // per multiply wave: 2 / 1
    constexpr int CT = (COUT + 15) / 16;             // channel tiles of dz: 3 / 2
    constexpr int TSX = CIN * WR_SROW, XSLOT = 2 * TSX;
    constexpr int TSD = CT * 16 * WR_SROW, DBUF = 2 * TSD;
    constexpr int XITEMS = CIN * 32, DITEMS = COUT * 32;
    constexpr int XIT = (XITEMS + 255) / 256, DIT = (DITEMS + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char xring[4 * XSLOT];
    __shared__ __attribute__((aligned(16))) unsigned char dzbuf[2 * DBUF];
    __shared__ __attribute__((aligned(16))) unsigned char zrow[WR_SROW];
    __shared__ float tab[2 * CIN];
    __shared__ __attribute__((aligned(16))) float tabk[BNF ? 8 * COUT : 4];      // per channel: mean, invstd, scale, shift, c1, c2, -, -

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WR_BAR_INIT();
    for (int e = tid; e < (int)sizeof(dzbuf) / 16; e += 512) reinterpret_cast<uint4*>(dzbuf)[e] = make_uint4(0u, 0u, 0u, 0u);
    for (int e = tid; e < (int)sizeof(xring) / 16; e += 512) reinterpret_cast<uint4*>(xring)[e] = make_uint4(0u, 0u, 0u, 0u);
    for (int e = tid; e < WR_SROW / 16; e += 512) reinterpret_cast<uint4*>(zrow)[e] = make_uint4(0u, 0u, 0u, 0u);
    const bool affine = a.in_scale != nullptr;                      // without: scale 1, shift 0 and no relu -- the same arithmetic, exact
    const float relu_floor = affine ? 0.f : -INFINITY;
    const int ka = (affine && a.act_absmax) ? pow2_scale_exp(*a.act_absmax, 14) : 0;
    const int kd = a.dy_absmax ? pow2_scale_exp(*a.dy_absmax, WR_KD) : 0;
    const float dscale = ldexpf(1.f, kd), unscale = ldexpf(1.f, -(ka + kd));
    if (tid < CIN) {
        tab[tid] = affine ? ldexpf(a.in_scale[tid], ka) : 1.f;
        tab[CIN + tid] = affine ? ldexpf(a.in_shift[tid], ka) : 0.f;
    }
    if (BNF) for (int e = tid; e < 8 * COUT; e += 512) tabk[e] = (e & 7) < 6 ? a.bn_k[(e >> 3) * 6 + (e & 7)] : 0.f;
    __syncthreads();

    if (wave < 4) {
        // ================================================================================ multiply role
        if (WR_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        const int li = lane & 15, g = lane >> 4;
        int ndt[NTW], nci[NTW]; bool ncol[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = 16 * (NTW * wave + j) + li;
            ncol[j] = n < 3 * CIN;
            ndt[j] = ncol[j] ? n / CIN : 0; nci[j] = ncol[j] ? n % CIN : 0;
        }
        f32x4 acc[NTW][CT][3];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) acc[j][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bool acc_neg = false;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        for (int work = blockIdx.x; work < a.nwork; work += gridDim.x) {
            const bool item_neg = work & 1;
            if (item_neg != acc_neg) {                // switch the sign convention of the accumulators (exact)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
#pragma unroll
                        for (int d = 0; d < 3; ++d)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[j][c][d][r] = -acc[j][c][d][r];
                acc_neg = item_neg;
            }
            __syncthreads();                          // (A) this role is done with the previous strip
            __syncthreads();                          // (B) rows 0, 1 of x and row 0 of dz are staged
#pragma unroll 1
            for (int t = 0; t < a.T; ++t) {
                const unsigned char* bsrc[NTW]; int bts[NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    const int row = t + ndt[j] - 1;
                    const bool live = ncol[j] && row >= 0 && row < a.T;       // rows outside the clip: the zero row
                    bsrc[j] = live ? xring + ((t + ndt[j]) & 3) * XSLOT + nci[j] * WR_SROW + 16 + 16 * g : zrow + 16;
                    bts[j] = live ? TSX : 0;
                }
                const unsigned char* asrc = dzbuf + (t & 1) * DBUF + li * WR_SROW + 16 * g;
                // The fragments of k-step ks + 1 are read while k-step ks multiplies (two register sets, the order pinned with
                // sched_barrier: left to itself the compiler issues most reads right before their first use and the wave sits out the
                // LDS latency a dozen times per row).
                struct Raw { f16x8 af[CT][2]; u32x4 v[NTW][2]; unsigned wb[NTW][2], wa[NTW][2]; };
                auto load = [&](int ks, Raw& r) {
#pragma unroll
                    for (int c = 0; c < CT; ++c)
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm) r.af[c][tm] = *reinterpret_cast<const f16x8*>(asrc + tm * TSD + c * 16 * WR_SROW + ks * 64);
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm) {
                            const unsigned char* p = bsrc[j] + tm * bts[j] + ks * 64;
                            r.v[j][tm] = *reinterpret_cast<const u32x4*>(p);
                            r.wb[j][tm] = (WR_X & 128) ? r.v[j][tm][0] : *reinterpret_cast<const unsigned*>(p - 4);          // positions p - 2, p - 1
                            r.wa[j][tm] = (WR_X & 128) ? r.v[j][tm][3] : *reinterpret_cast<const unsigned*>(p + 16);         // positions p + 8, p + 9
                        }
                };
                auto compute = [&](const Raw& r) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        f16x8 bf[3][2];
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm) {
                            const u32x4 v = r.v[j][tm];
                            const u32x4 f0 = {__builtin_amdgcn_alignbit(v[0], r.wb[j][tm], 16), __builtin_amdgcn_alignbit(v[1], v[0], 16),
                                              __builtin_amdgcn_alignbit(v[2], v[1], 16), __builtin_amdgcn_alignbit(v[3], v[2], 16)};          // a[p - 1]
                            const u32x4 f2 = {__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                                              __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(r.wa[j][tm], v[3], 16)};   // a[p + 1]
                            bf[0][tm] = __builtin_bit_cast(f16x8, (WR_X & 4) ? v : f0);
                            bf[1][tm] = __builtin_bit_cast(f16x8, v);
                            bf[2][tm] = __builtin_bit_cast(f16x8, (WR_X & 4) ? v : f2);
                        }
                        if (WR_X & 1) {
#pragma unroll
                            for (int c = 0; c < CT; ++c)
#pragma unroll
                                for (int d = 0; d < 3; ++d) asm volatile("" :: "v"(r.af[c][0]), "v"(r.af[c][1]), "v"(bf[d][0]), "v"(bf[d][1]));
                            continue;
                        }
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.af[c][1], bf[d][0], acc[j][c][d], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.af[c][0], bf[d][1], acc[j][c][d], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.af[c][0], bf[d][0], acc[j][c][d], 0, 0, 0);
                    }
                };
                Raw r0, r1;
                // one k-step: the reads of the next k-step are spread between this one's MFMAs (the four multiply waves leave the row
                // barrier together: a block of 15 reads per wave drains all four matrix pipes while the one LDS pipe serves 60 reads)
                auto step = [&](const Raw& cur, Raw& nxt, int ks_next) {
                    if (ks_next < 4) load(ks_next, nxt);
                    compute(cur);
                    if (WR_SCHED2) {
                        constexpr int NMF = NTW * CT * 9, NRD = CT * 2 + NTW * 6, NVA = NTW * 16;
                        __builtin_amdgcn_sched_group_barrier(0x002, NVA / NTW, 0);         // the first column tile's fragments
                        constexpr int G = NMF / 3;
#pragma unroll
                        for (int q = 0; q < G; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                            if (ks_next < 4 && q < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            if (q < (NVA - NVA / NTW + 1) / 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                load(0, r0);
                __builtin_amdgcn_sched_barrier(0);
                step(r0, r1, 1);
                step(r1, r0, 2);
                step(r0, r1, 3);
                step(r1, r0, 4);
                WR_BAR();
            }
        }
        WR_BAR_OUT(0);
        // ---- the slab: accumulator (j, c, df) holds rows co = 16 c + 4 g + r, column n_j = (ndt, nci)
        const float us = acc_neg ? -unscale : unscale;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
            if (ncol[j]) {
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = 16 * c + 4 * g + r;
                            if (co < COUT) a.partial[(((long)blockIdx.x * COUT + co) * CIN + nci[j]) * 9 + ndt[j] * 3 + d] = acc[j][c][d][r] * us;
                        }
            }
    } else {
        // ================================================================================ staging role
        if (WR_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        const int st = tid - 256;                     // item it: channel (st >> 5) + 8 it, positions 4 (st & 31) ..+3
        float xsc[XIT], xsh[XIT];
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int ch = (st >> 5) + 8 * it;
            xsc[it] = ch < CIN ? tab[ch] : 0.f; xsh[it] = ch < CIN ? tab[CIN + ch] : 0.f;
        }
        const int hch = st >> 1, hside = st & 1;      // the halo of the input rows: thread h < 2 CIN carries channel h >> 1, position -1 or 128
        const bool hthread = st < 2 * CIN;
        const float hsc = hthread ? tab[hch] : 0.f, hsh = hthread ? tab[CIN + hch] : 0.f;
        float dzmax = 0.f;
        for (int work = blockIdx.x; work < a.nwork; work += gridDim.x) {
            const int ft = work % a.tilesF, b = work / a.tilesF;
            const int f_base = ft * WR_TP;
            const float dsc = (work & 1) ? -dscale : dscale;
            const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * a.T * CIN * a.F), 0, (unsigned)((long)a.T * CIN * a.F * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + (long)b * a.T * COUT * a.F), 0, (unsigned)((long)a.T * COUT * a.F * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((BNF ? a.bn_y : a.dy) + (long)b * a.T * COUT * a.F), 0, (unsigned)((long)a.T * COUT * a.F * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((BNF ? a.dz_out : a.partial) + (BNF ? (long)b * a.T * COUT * a.F : 0), 0, BNF ? (unsigned)((long)a.T * COUT * a.F * 4) : 0u, 0x00020000);
            const int fcol = f_base + 4 * (st & 31);
            const bool colok = fcol < a.F;            // (F % 4 == 0: a 4-position item is inside or outside as a whole)
            const int hf = hside ? f_base + WR_TP : f_base - 1;
            const bool hok = hthread && hf >= 0 && hf < a.F;
            const int xoff = ((st >> 5) * a.F + fcol) * 4;
            // Columns beyond F: the buffer loads return zeros and the affine of those items is zeroed for the strip (relu(0 * 0 + 0) = 0).
            // Rows outside the clip are never staged as zeros: the fragments of their (dt, ci) columns read the zero row instead.
            float csc[XIT], csh[XIT];
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                csc[it] = colok ? xsc[it] : 0.f; csh[it] = colok ? xsh[it] : 0.f;
                asm volatile("" : "+v"(csc[it]), "+v"(csh[it]));
            }
            float chsc = hok ? hsc : 0.f, chsh = hok ? hsh : 0.f;
            asm volatile("" : "+v"(chsc), "+v"(chsh));
            auto issue_x = [&](int row, f32x4 (&xr)[XIT], float& xh) {
                const int rbase = (WR_X & 32) ? 0 : row * CIN * a.F * 4;
#pragma unroll
                for (int it = 0; it < XIT; ++it) {
                    const bool has = (it + 1) * 256 <= XITEMS || st + 256 * it < XITEMS;
                    xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (colok && has && !(WR_X & 16)) ? rbase + xoff + 32 * it * a.F : -4, 0, 0));
                }
                xh = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, hok ? rbase + (hch * a.F + hf) * 4 : -4, 0, 0));
            };
            auto issue_d = [&](int row, f32x4 (&dr)[DIT], f32x4 (&yr)[BNF ? DIT : 1]) {
                const int rbase = (WR_X & 32) ? 0 : row * COUT * a.F * 4;
#pragma unroll
                for (int it = 0; it < DIT; ++it) {
                    const bool has = (it + 1) * 256 <= DITEMS || st + 256 * it < DITEMS;
                    dr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drsrc, (colok && has && !(WR_X & 16)) ? rbase + xoff + 32 * it * a.F : -4, 0, 0));
                    if (BNF) yr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, (colok && has && !(WR_X & 16)) ? rbase + xoff + 32 * it * a.F : -4, 0, 0));
                }
            };

            // threads without an item in the last round store into bytes 0..7 / 256..263 of an image row: positions no fragment uses
            auto commit_x = [&](int slot, const f32x4 (&xr)[XIT], float xh) {
#pragma unroll
                for (int it = 0; it < XIT; ++it) {
                    const bool has = (it + 1) * 256 <= XITEMS || st + 256 * it < XITEMS;
                    unsigned char* const p0 = xring + slot * XSLOT + (has ? ((st >> 5) + 8 * it) * WR_SROW + 16 + (st & 31) * 8 : (st >> 5) * WR_SROW);
                    f32x4 v = xr[it];
                    uint2 t0, t1;
                    if (WR_X & 8) { t0 = make_uint2(__float_as_uint(v[0]), __float_as_uint(v[1])); t1 = make_uint2(__float_as_uint(v[2]), __float_as_uint(v[3])); } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], csc[it], csh[it]), relu_floor);
                    split2_pair_f16(v[0], v[1], t0.x, t1.x);
                    split2_pair_f16(v[2], v[3], t0.y, t1.y);
                    }
                    *reinterpret_cast<uint2*>(p0) = t0;
                    *reinterpret_cast<uint2*>(p0 + TSX) = t1;
                }
                unsigned p0, p1;
                split2_pair_f16(fmaxf(fmaf(xh, chsc, chsh), relu_floor), 0.f, p0, p1);
                unsigned char* dst = xring + slot * XSLOT + (hthread ? hch * WR_SROW + (hside ? 8 + WR_TP : 7) * 2 : (st >> 5) * WR_SROW + 8);
                *reinterpret_cast<unsigned short*>(dst) = (unsigned short)p0;
                *reinterpret_cast<unsigned short*>(dst + TSX) = (unsigned short)p1;
            };
            auto commit_d = [&](int buf, int row, f32x4 (&dr)[DIT], const f32x4 (&yr)[BNF ? DIT : 1]) {
#pragma unroll
                for (int it = 0; it < DIT; ++it) {
                    const bool has = (it + 1) * 256 <= DITEMS || st + 256 * it < DITEMS;
                    unsigned char* const p0 = dzbuf + buf * DBUF + (has ? ((st >> 5) + 8 * it) * WR_SROW + (st & 31) * 8 : (st >> 5) * WR_SROW + 256);
                    if (BNF) {
                        // dz = scale (g' - c1 - xhat c2), g' = g where bn(y) > 0 (bn_bwd_value of a2s_conv.hip): formed here, written once for the
                        // data-gradient convolution (every dz element lies in exactly one strip), then split like a loaded dz row
                        const int ch = min((st >> 5) + 8 * it, COUT - 1);
                        const f32x4 k0 = *reinterpret_cast<const f32x4*>(tabk + 8 * ch);
                        const float c1 = tabk[8 * ch + 4], c2 = tabk[8 * ch + 5];
                        const bool live = colok && has && row < a.T;
                        f32x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float gm = (fmaf(yr[it][k], k0[2], k0[3]) > 0.f) ? dr[it][k] : 0.f;
                            v[k] = live ? k0[2] * (gm - c1 - (yr[it][k] - k0[0]) * k0[1] * c2) : 0.f;
                            dzmax = fmaxf(dzmax, fabsf(v[k]));
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), zrsrc, live ? row * COUT * a.F * 4 + xoff + 32 * it * a.F : -4, 0, 0);
                        dr[it] = v;
                    }
                    uint2 t0, t1;
                    if (WR_X & 8) { t0 = make_uint2(__float_as_uint(dr[it][0]), __float_as_uint(dr[it][1])); t1 = make_uint2(__float_as_uint(dr[it][2]), __float_as_uint(dr[it][3])); } else {
                    split2_pair_f16(dr[it][0] * dsc, dr[it][1] * dsc, t0.x, t1.x);
                    split2_pair_f16(dr[it][2] * dsc, dr[it][3] * dsc, t0.y, t1.y);
                    }
                    *reinterpret_cast<uint2*>(p0) = t0;
                    *reinterpret_cast<uint2*>(p0 + TSD) = t1;
                }
            };
            // WR_NS register sets keep the global loads of WR_NS rows in flight (one row per CU = 40 KB x 256 CUs does not cover HBM's
            // latency x bandwidth product).
            f32x4 xr[WR_NS][XIT], dr[WR_NS][DIT], yr[WR_NS][BNF ? DIT : 1];
            float xh[WR_NS];
            __syncthreads();                          // (A) the previous strip's last multiply is over
            issue_x(0, xr[0], xh[0]); issue_x(1, xr[1], xh[1]); issue_d(0, dr[0], yr[0]);
            commit_x(1, xr[0], xh[0]); commit_x(2, xr[1], xh[1]); commit_d(0, 0, dr[0], yr[0]);
#pragma unroll
            for (int u = 0; u < WR_NS; ++u) { issue_x(2 + u, xr[u], xh[u]); issue_d(1 + u, dr[u], yr[u]); }
            __syncthreads();                          // (B)
#pragma unroll 1
            for (int t0 = 0; t0 < a.T; t0 += WR_NS) {
#pragma unroll
                for (int u = 0; u < WR_NS; ++u) {
                    const int t = t0 + u;
                    if (t < a.T) {
                        if (!(WR_X & 64)) {
                        commit_x((t + 3) & 3, xr[u], xh[u]); commit_d((t + 1) & 1, t + 1, dr[u], yr[u]);            // input row t + 2, dz row t + 1
                        issue_x(t + 2 + WR_NS, xr[u], xh[u]); issue_d(t + 1 + WR_NS, dr[u], yr[u]);
                        }
                        WR_BAR();
                    }
                }
            }
        }
        WR_BAR_OUT(1);
        if (BNF && a.dz_absmax_out) {              // non-negative floats order like their bit patterns
            const float m = wave_max(dzmax);
            if (lane == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.dz_absmax_out), __float_as_uint(m));
        }
    }
}

// ------------------------------------------------------------------------------------------- launcher
static int g_wgrad_rows = -1;
void a2s_wgrad_rows_set(int on) { g_wgrad_rows = on; }
int a2s_wgrad_rows_enabled(void) {
    if (g_wgrad_rows < 0) { const char* e = getenv("A2S_WGRAD_ROWS"); g_wgrad_rows = e ? atoi(e) : 1; }
    return g_wgrad_rows;
}
bool a2s_wgrad_rows_eligible(int F, int Cin, int Cout) {
    return a2s_wgrad_rows_enabled() && F % 4 == 0 && (Cin == 20 || Cin == 40) && (Cout == 20 || Cout == 40);
}
__global__ void wgrad_rows_reduce(const float* __restrict__ partial, float* __restrict__ dW, int nslabs, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float s = 0.f;
    for (int i = 0; i < nslabs; ++i) s += partial[(long)i * n + idx];
    dW[idx] += s;
}

// Upper bound of |dz| of a BatchNorm backward dz = scale (g' - c1 - xhat c2) from ranges: max |g| (one scalar, reduced by the kernel that wrote
// g) and max |y_c| per channel (written by the forward convolution): out[0] = max_c |scale_c| (gmax + |c1_c| + xhatmax_c |c2_c|).  Also packs
// the six per-channel constants of the fused weight gradient: k[c] = mean, invstd, scale, shift, c1, c2.
__global__ void bn_bwd_bound_kernel(const float* __restrict__ g_absmax, int g_n, const float* __restrict__ y_absmax, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ c12, int C, float* __restrict__ k, float* __restrict__ out) {
    __shared__ float red[16];
    const int c = threadIdx.x;
    float gmax = 0.f;
    for (int i = 0; i < g_n; ++i) gmax = fmaxf(gmax, g_absmax[i]);
    float b = 0.f;
    if (c < C) {
        const float xh = (y_absmax[c] + fabsf(mean[c])) * fabsf(invstd[c]);
        b = fabsf(scale[c]) * (gmax + fabsf(c12[2 * c]) + xh * fabsf(c12[2 * c + 1]));
        k[6 * c + 0] = mean[c]; k[6 * c + 1] = invstd[c]; k[6 * c + 2] = scale[c]; k[6 * c + 3] = shift[c]; k[6 * c + 4] = c12[2 * c]; k[6 * c + 5] = c12[2 * c + 1];
    }
    b = block_max(b, red);
    if (threadIdx.x == 0) out[0] = b;
}

int a2s_conv3x3_wgrad_rows_bn_impl(hipStream_t st, const float* g, const float* y, const float* mean, const float* invstd, const float* scale,
                                   const float* shift, const float* c12, const float* g_absmax, int g_absmax_n, const float* y_absmax, float* dz_out,
                                   float* dz_absmax_out, const float* x, const float* in_scale, const float* in_shift, float* dW, float* ws, size_t ws_bytes,
                                   int B, int T, int F, int Cin, int Cout, const float* act_absmax) {
    const int tilesF = a2s_cdiv(F, WR_TP), nwork = B * tilesF;
    const int nslabs = nwork < WR_SLABS ? nwork : WR_SLABS;
    const size_t slab_bytes = (size_t)nslabs * Cout * Cin * 9 * sizeof(float);
    // workspace: [slabs][k: Cout x 6][bound: 1 float (+3 pad)]
    A2S_REQUIRE(ws_bytes >= slab_bytes + sizeof(float) * (6 * (size_t)Cout + 4), "conv3x3_wgrad_rows_bn: workspace too small");
    A2S_REQUIRE(g && y && mean && invstd && scale && shift && c12 && g_absmax && g_absmax_n >= 1 && y_absmax && dz_out && dz_absmax_out, "conv3x3_wgrad_rows_bn: null argument");
    A2S_REQUIRE((long)T * Cout * F * 4 < (1L << 31), "conv3x3_wgrad_rows_bn: a clip's dz must stay below 2 GiB");
    float* k = ws + slab_bytes / sizeof(float);
    float* bound = k + 6 * Cout;
    hipLaunchKernelGGL(bn_bwd_bound_kernel, dim3(1), dim3(64), 0, st, g_absmax, g_absmax_n, y_absmax, mean, invstd, scale, shift, c12, Cout, k, bound);
    A2S_CHECK_LAUNCH("bn_bwd_bound");
    hipError_t e = hipMemsetAsync(dz_absmax_out, 0, sizeof(float), st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "conv3x3_wgrad_rows_bn memset: %s", hipGetErrorString(e));
    WgRowsArgs a{g, x, in_scale, in_shift, bound, act_absmax, ws, B, T, F, tilesF, nwork, y, k, dz_out, dz_absmax_out};
    if (Cin == 40 && Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad_rows<40, 40, true>), dim3(nslabs), dim3(512), 0, st, a);
    else if (Cin == 20 && Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad_rows<20, 40, true>), dim3(nslabs), dim3(512), 0, st, a);
    else if (Cin == 20 && Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_rows<20, 20, true>), dim3(nslabs), dim3(512), 0, st, a);
    else A2S_FAIL(A2S_ERR_ARG, "conv3x3_wgrad_rows_bn: no instance for %d -> %d", Cin, Cout);
    A2S_CHECK_LAUNCH("conv3x3_wgrad_rows_bn");
    hipLaunchKernelGGL(wgrad_rows_reduce, dim3(a2s_cdiv(Cout * Cin * 9, 256)), dim3(256), 0, st, ws, dW, nslabs, Cout * Cin * 9);
    A2S_CHECK_LAUNCH("wgrad_rows_reduce");
    return A2S_OK;
}

int a2s_conv3x3_wgrad_rows_impl(hipStream_t st, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW, float* ws,
                                size_t ws_bytes, int B, int T, int F, int Cin, int Cout, const float* dy_absmax, const float* act_absmax) {
    const int tilesF = a2s_cdiv(F, WR_TP), nwork = B * tilesF;
    const int nslabs = nwork < WR_SLABS ? nwork : WR_SLABS;
    A2S_REQUIRE(ws_bytes >= (size_t)nslabs * Cout * Cin * 9 * sizeof(float), "conv3x3_wgrad_rows: workspace too small");
    WgRowsArgs a{dy, x, in_scale, in_shift, dy_absmax, act_absmax, ws, B, T, F, tilesF, nwork, nullptr, nullptr, nullptr, nullptr};
    if (Cin == 40 && Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad_rows<40, 40>), dim3(nslabs), dim3(512), 0, st, a);
    else if (Cin == 20 && Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad_rows<20, 40>), dim3(nslabs), dim3(512), 0, st, a);
    else if (Cin == 20 && Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_rows<20, 20>), dim3(nslabs), dim3(512), 0, st, a);
    else if (Cin == 40 && Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_rows<40, 20>), dim3(nslabs), dim3(512), 0, st, a);
    else A2S_FAIL(A2S_ERR_ARG, "conv3x3_wgrad_rows: no instance for %d -> %d", Cin, Cout);
    A2S_CHECK_LAUNCH("conv3x3_wgrad_rows");
    hipLaunchKernelGGL(wgrad_rows_reduce, dim3(a2s_cdiv(Cout * Cin * 9, 256)), dim3(256), 0, st, ws, dW, nslabs, Cout * Cin * 9);
    A2S_CHECK_LAUNCH("wgrad_rows_reduce");
    return A2S_OK;
}
