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
//   * ONE wave per SIMD (256 threads, up to 512 registers): the conversion of the next rows is interleaved by the compiler into the
//     shadow of the wave's own MFMAs.  Two waves per SIMD that mix MFMA and vector work starve each other (a wave waiting for the matrix
//     pipe blocks the SIMD's vector issue port: profiles/r03_trace_overlap.txt) -- the first version of this kernel (8 waves, one column
//     tile each) ran at 2533 clk per k-step against 864 clk of MFMA issue;
//   * wave = two 16-wide (dt, ci) column tiles (one for 20 input channels) x all channel tiles of dz x df: 18 accumulators (9 / 6);
//   * persistent workgroups (one per CU), one partial slab [Cout][Cin][9] each, summed in fixed order by wgrad_rows_reduce;
//   * operands as two exact fp16 terms / three MFMA products, scaled by exact powers of two from their ranges (dz: max|dz| scalar written by
//     the kernel that produced dz; activated input: the a2s_act_bound scalar); odd work items accumulate the negated
//     sum (dz negated while staging, accumulators flipped) against the matrix pipe's truncation bias.
#include "a2s_common.h"

#define WR_TP 128              // positions per strip
#ifndef WR_SROW
#define WR_SROW 288            // bytes per channel row of an fp16 image (8 margin + 128 positions + 8)
#endif
#define WR_SLABS 256
#ifndef WR_SCHED
#define WR_SCHED 0             // instruction-group pattern asked of the scheduler for a row's multiply (0 = none)
#endif
#ifndef WR_X
#define WR_X 0                 // ablation bits (tools/wgrad_rows_check.py timing only; results are wrong): 1 no MFMA, 2 no commits, 4 no neighbour reads
#endif

struct WgRowsArgs {
    const float* dy; const float* x; const float* in_scale; const float* in_shift;
    const float* dy_absmax; const float* act_absmax;
    float* partial;
    int B, T, F, tilesF, nwork;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_rows(WgRowsArgs a) {
    constexpr int NT = (3 * CIN + 15) / 16;          // (dt, ci) column tiles: 8 / 4
    constexpr int NTW = NT / 4;                      // per wave: 2 / 1
    constexpr int CT = (COUT + 15) / 16;             // channel tiles of dz: 3 / 2
    constexpr int TSX = CIN * WR_SROW;               // term plane of an x slot
    constexpr int XSLOT = 2 * TSX;
    constexpr int TSD = CT * 16 * WR_SROW;           // term plane of a dz buffer
    constexpr int DBUF = 2 * TSD;
    constexpr int XITEMS = CIN * 32, DITEMS = COUT * 32;
    constexpr int XIT = (XITEMS + 255) / 256, DIT = (DITEMS + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char xring[4 * XSLOT];
    __shared__ __attribute__((aligned(16))) unsigned char dzbuf[2 * DBUF];
    __shared__ __attribute__((aligned(16))) unsigned char zrow[WR_SROW];      // fragment source of the idle (dt, ci) columns
    __shared__ float tab[2 * CIN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;

    for (int e = tid; e < (int)sizeof(dzbuf) / 16; e += 256) reinterpret_cast<uint4*>(dzbuf)[e] = make_uint4(0u, 0u, 0u, 0u);      // rows >= COUT stay zero
    for (int e = tid; e < (int)sizeof(xring) / 16; e += 256) reinterpret_cast<uint4*>(xring)[e] = make_uint4(0u, 0u, 0u, 0u);      // margins beyond the halo
    for (int e = tid; e < WR_SROW / 16; e += 256) reinterpret_cast<uint4*>(zrow)[e] = make_uint4(0u, 0u, 0u, 0u);
    // operand scales (exact powers of two)
    const bool affine = a.in_scale != nullptr;                      // without: scale 1, shift 0 and no relu -- the same arithmetic, exact
    const float relu_floor = affine ? 0.f : -INFINITY;
    const int ka = (affine && a.act_absmax) ? pow2_scale_exp(*a.act_absmax, 14) : 0;
    const int kd = a.dy_absmax ? pow2_scale_exp(*a.dy_absmax, 14) : 0;
    const float dscale = ldexpf(1.f, kd), unscale = ldexpf(1.f, -(ka + kd));
    if (tid < CIN) {
        tab[tid] = affine ? ldexpf(a.in_scale[tid], ka) : 1.f;
        tab[CIN + tid] = affine ? ldexpf(a.in_shift[tid], ka) : 0.f;
    }
    __syncthreads();

    // ---- this lane's columns n = (dt, ci): B fragments from row ci of the slot of input row t + dt - 1
    int ndt[NTW], nci[NTW]; bool ncol[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = 16 * (NTW * wave + j) + li;
        ncol[j] = n < 3 * CIN;
        ndt[j] = ncol[j] ? n / CIN : 0; nci[j] = ncol[j] ? n % CIN : 0;
    }
    // ---- this thread's staging items: channel (tid >> 5) + 8 it, positions 4 (tid & 31) ..+3; the affine of its input channels
    float xsc[XIT], xsh[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int ch = (tid >> 5) + 8 * it;
        xsc[it] = ch < CIN ? tab[ch] : 0.f; xsh[it] = ch < CIN ? tab[CIN + ch] : 0.f;
    }
    // the halo of the input rows: thread h < 2 CIN carries channel h >> 1, position -1 (left) or 128 (right)
    const int hch = tid >> 1, hside = tid & 1;
    const bool hthread = tid < 2 * CIN;
    const float hsc = hthread ? tab[hch] : 0.f, hsh = hthread ? tab[CIN + hch] : 0.f;

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
        const int ft = work % a.tilesF, b = work / a.tilesF;
        const int f_base = ft * WR_TP;
        const bool item_neg = work & 1;
        if (item_neg != acc_neg) {                    // switch the sign convention of the accumulators (exact)
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
        const float dsc = item_neg ? -dscale : dscale;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * a.T * CIN * a.F), 0, (unsigned)((long)a.T * CIN * a.F * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + (long)b * a.T * COUT * a.F), 0, (unsigned)((long)a.T * COUT * a.F * 4), 0x00020000);
        const int fcol = f_base + 4 * (tid & 31);
        const bool colok = fcol < a.F;                // (F % 4 == 0: a 4-position item is inside or outside as a whole)
        const int hf = hside ? f_base + WR_TP : f_base - 1;
        const bool hok = hthread && hf >= 0 && hf < a.F;
        const int xoff = ((tid >> 5) * a.F + fcol) * 4;       // byte offset of item 0 inside a row; item it: + 8 it F 4
        // Columns beyond F: the buffer loads return zeros and the affine of those items is zeroed for the strip (relu(0 * 0 + 0) = 0).
        // Rows outside the clip are never staged as zeros: the fragments of their (dt, ci) columns read the zero row instead (multiply).
        float csc[XIT], csh[XIT];
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            csc[it] = colok ? xsc[it] : 0.f; csh[it] = colok ? xsh[it] : 0.f;
            asm volatile("" : "+v"(csc[it]), "+v"(csh[it]));
        }
        float chsc = hok ? hsc : 0.f, chsh = hok ? hsh : 0.f;
        asm volatile("" : "+v"(chsc), "+v"(chsh));

        auto issue_x = [&](int row, f32x4 (&xr)[XIT], float& xh) {
            const int rbase = row * CIN * a.F * 4;
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                const bool has = (it + 1) * 256 <= XITEMS || tid + 256 * it < XITEMS;
                xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (colok && has) ? rbase + xoff + 32 * it * a.F : -4, 0, 0));
            }
            xh = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, hok ? rbase + (hch * a.F + hf) * 4 : -4, 0, 0));
        };
        auto issue_d = [&](int row, f32x4 (&dr)[DIT]) {
            const int rbase = row * COUT * a.F * 4;
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const bool has = (it + 1) * 256 <= DITEMS || tid + 256 * it < DITEMS;
                dr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drsrc, (colok && has) ? rbase + xoff + 32 * it * a.F : -4, 0, 0));
            }
        };
        // threads without an item in the last round store into bytes 0..7 of an image row: margin positions no fragment uses
        auto commit_x_item = [&](int it, int slot, const f32x4& x) {
            if (WR_X & 2) return;
            const bool has = (it + 1) * 256 <= XITEMS || tid + 256 * it < XITEMS;
            unsigned char* const p0 = xring + slot * XSLOT + (has ? ((tid >> 5) + 8 * it) * WR_SROW + 16 + (tid & 31) * 8 : (tid >> 5) * WR_SROW);
            f32x4 v = x;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], csc[it], csh[it]), relu_floor);
            uint2 t0, t1;
            split2_pair_f16(v[0], v[1], t0.x, t1.x);
            split2_pair_f16(v[2], v[3], t0.y, t1.y);
            *reinterpret_cast<uint2*>(p0) = t0;
            *reinterpret_cast<uint2*>(p0 + TSX) = t1;
        };
        auto commit_x_halo = [&](int slot, float xh) {
            if (WR_X & 2) return;
            unsigned p0, p1;
            split2_pair_f16(fmaxf(fmaf(xh, chsc, chsh), relu_floor), 0.f, p0, p1);
            unsigned char* dst = xring + slot * XSLOT + (hthread ? hch * WR_SROW + (hside ? 8 + WR_TP : 7) * 2 : (tid >> 5) * WR_SROW + 8);
            *reinterpret_cast<unsigned short*>(dst) = (unsigned short)p0;
            *reinterpret_cast<unsigned short*>(dst + TSX) = (unsigned short)p1;
        };
        auto commit_d_item = [&](int it, int buf, const f32x4& d) {
            if (WR_X & 2) return;
            const bool has = (it + 1) * 256 <= DITEMS || tid + 256 * it < DITEMS;
            unsigned char* const p0 = dzbuf + buf * DBUF + (has ? ((tid >> 5) + 8 * it) * WR_SROW + (tid & 31) * 8 : (tid >> 5) * WR_SROW + 256);
            uint2 t0, t1;
            split2_pair_f16(d[0] * dsc, d[1] * dsc, t0.x, t1.x);
            split2_pair_f16(d[2] * dsc, d[3] * dsc, t0.y, t1.y);
            *reinterpret_cast<uint2*>(p0) = t0;
            *reinterpret_cast<uint2*>(p0 + TSD) = t1;
        };

        // ---- multiply of row t: x slots of rows t-1, t, t+1 (slot of row r = (r + 1) & 3), dz buffer t & 1; spread over the k-steps the
        // conversions of input row t + 2 and dz row t + 1
        auto multiply = [&](int t, const f32x4 (&xr)[XIT], float xh, const f32x4 (&dr)[DIT]) {
            const unsigned char* bsrc[NTW]; int bts[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int row = t + ndt[j] - 1;
                const bool live = ncol[j] && row >= 0 && row < a.T;
                bsrc[j] = live ? xring + ((t + ndt[j]) & 3) * XSLOT + nci[j] * WR_SROW + 16 + 16 * g : zrow + 16;
                bts[j] = live ? TSX : 0;
            }
            const unsigned char* asrc = dzbuf + (t & 1) * DBUF + li * WR_SROW + 16 * g;
            const int xslot = (t + 3) & 3, dbuf = (t + 1) & 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f16x8 af[CT][2];
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) af[c][tm] = *reinterpret_cast<const f16x8*>(asrc + tm * TSD + c * 16 * WR_SROW + ks * 64);
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    f16x8 bf[3][2];
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) {
                        const unsigned char* p = bsrc[j] + tm * bts[j] + ks * 64;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(p);
                        const uint2 wb = (WR_X & 4) ? make_uint2(v[0], v[1]) : *reinterpret_cast<const uint2*>(p - 8);
                        const uint2 wa = (WR_X & 4) ? make_uint2(v[2], v[3]) : *reinterpret_cast<const uint2*>(p + 16);
                        const u32x4 f0 = {__builtin_amdgcn_alignbit(v[0], wb.y, 16), __builtin_amdgcn_alignbit(v[1], v[0], 16),
                                          __builtin_amdgcn_alignbit(v[2], v[1], 16), __builtin_amdgcn_alignbit(v[3], v[2], 16)};          // a[p - 1]
                        const u32x4 f2 = {__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                                          __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(wa.x, v[3], 16)};       // a[p + 1]
                        bf[0][tm] = __builtin_bit_cast(f16x8, f0);
                        bf[1][tm] = __builtin_bit_cast(f16x8, v);
                        bf[2][tm] = __builtin_bit_cast(f16x8, f2);
                    }
                    if (WR_X & 1) {
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) asm volatile("" :: "v"(af[c][0]), "v"(af[c][1]), "v"(bf[d][0]), "v"(bf[d][1]));
                    } else {
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[c][1], bf[d][0], acc[j][c][d], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[c][0], bf[d][1], acc[j][c][d], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[c][0], bf[d][0], acc[j][c][d], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int it = 0; it < XIT; ++it)
                    if (it * 4 / XIT == ks) commit_x_item(it, xslot, xr[it]);
#pragma unroll
                for (int it = 0; it < DIT; ++it)
                    if (it * 4 / DIT == ks) commit_d_item(it, dbuf, dr[it]);
                if (ks == 3) commit_x_halo(xslot, xh);
            }
            if (WR_SCHED == 1) {
#pragma unroll
                for (int m = 0; m < 4 * NTW * CT * 9; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
            }
        };

        // ---- prologue of the strip: rows 0, 1 of x into slots 1, 2; dz row 0 into buffer 0.  Two register sets keep the global loads
        // of TWO rows in flight (one row per CU = 40 KB x 256 CUs does not cover HBM's latency x bandwidth product).
        f32x4 xr0[XIT], dr0[DIT], xr1[XIT], dr1[DIT];
        float xh0, xh1;
        __syncthreads();                              // the previous strip's last multiply is over
#pragma unroll 1
        for (int r = 0; r <= 1; ++r) {
            issue_x(r, xr0, xh0);
#pragma unroll
            for (int it = 0; it < XIT; ++it) commit_x_item(it, (r + 1) & 3, xr0[it]);
            commit_x_halo((r + 1) & 3, xh0);
        }
        issue_d(0, dr0);
#pragma unroll
        for (int it = 0; it < DIT; ++it) commit_d_item(it, 0, dr0[it]);
        issue_x(2, xr0, xh0);
        issue_d(1, dr0);
        issue_x(3, xr1, xh1);
        issue_d(2, dr1);
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < a.T; t += 2) {
            multiply(t, xr0, xh0, dr0);
            __syncthreads();
            issue_x(t + 4, xr0, xh0);
            issue_d(t + 3, dr0);
            if (t + 1 < a.T) {
                multiply(t + 1, xr1, xh1, dr1);
                __syncthreads();
                issue_x(t + 5, xr1, xh1);
                issue_d(t + 4, dr1);
            }
        }
    }

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
}

// ------------------------------------------------------------------------------------------- role-split form
// The same strip walk with TWO waves per SIMD in fixed roles: waves 0-3 ("multiply", one per SIMD) only read fragments and issue MFMAs,
// waves 4-7 ("staging", one per SIMD) only load, convert and write the next rows.  A single in-order wave hides at most ~2 issue slots
// per 16-cycle MFMA (MI355X_MICROARCH.md, one wave per SIMD), and this path needs ~2 non-MFMA instructions per MFMA -- every fragment
// wait of the one-wave form above stalls its conversions too.  Two streams stall independently; the staging stream is sparse (~190
// instructions per 3.6k-cycle row) and takes its issue slots with s_setprio.
#ifndef WR_PRIO
#define WR_PRIO 1              // 1: staging waves at priority 1, 2: multiply waves at priority 1, 0: neither
#endif
template <int CIN, int COUT>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_rows2(WgRowsArgs a) {
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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < (int)sizeof(dzbuf) / 16; e += 512) reinterpret_cast<uint4*>(dzbuf)[e] = make_uint4(0u, 0u, 0u, 0u);
    for (int e = tid; e < (int)sizeof(xring) / 16; e += 512) reinterpret_cast<uint4*>(xring)[e] = make_uint4(0u, 0u, 0u, 0u);
    for (int e = tid; e < WR_SROW / 16; e += 512) reinterpret_cast<uint4*>(zrow)[e] = make_uint4(0u, 0u, 0u, 0u);
    const bool affine = a.in_scale != nullptr;                      // without: scale 1, shift 0 and no relu -- the same arithmetic, exact
    const float relu_floor = affine ? 0.f : -INFINITY;
    const int ka = (affine && a.act_absmax) ? pow2_scale_exp(*a.act_absmax, 14) : 0;
    const int kd = a.dy_absmax ? pow2_scale_exp(*a.dy_absmax, 14) : 0;
    const float dscale = ldexpf(1.f, kd), unscale = ldexpf(1.f, -(ka + kd));
    if (tid < CIN) {
        tab[tid] = affine ? ldexpf(a.in_scale[tid], ka) : 1.f;
        tab[CIN + tid] = affine ? ldexpf(a.in_shift[tid], ka) : 0.f;
    }
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
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    f16x8 af[CT][2];
#pragma unroll
                    for (int c = 0; c < CT; ++c)
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm) af[c][tm] = *reinterpret_cast<const f16x8*>(asrc + tm * TSD + c * 16 * WR_SROW + ks * 64);
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        f16x8 bf[3][2];
#pragma unroll
                        for (int tm = 0; tm < 2; ++tm) {
                            const unsigned char* p = bsrc[j] + tm * bts[j] + ks * 64;
                            const u32x4 v = *reinterpret_cast<const u32x4*>(p);
                            const uint2 wb = *reinterpret_cast<const uint2*>(p - 8);
                            const uint2 wa = *reinterpret_cast<const uint2*>(p + 16);
                            const u32x4 f0 = {__builtin_amdgcn_alignbit(v[0], wb.y, 16), __builtin_amdgcn_alignbit(v[1], v[0], 16),
                                              __builtin_amdgcn_alignbit(v[2], v[1], 16), __builtin_amdgcn_alignbit(v[3], v[2], 16)};          // a[p - 1]
                            const u32x4 f2 = {__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                                              __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(wa.x, v[3], 16)};       // a[p + 1]
                            bf[0][tm] = __builtin_bit_cast(f16x8, f0);
                            bf[1][tm] = __builtin_bit_cast(f16x8, v);
                            bf[2][tm] = __builtin_bit_cast(f16x8, f2);
                        }
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[c][1], bf[d][0], acc[j][c][d], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[c][0], bf[d][1], acc[j][c][d], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int d = 0; d < 3; ++d) acc[j][c][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[c][0], bf[d][0], acc[j][c][d], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        }
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
        for (int work = blockIdx.x; work < a.nwork; work += gridDim.x) {
            const int ft = work % a.tilesF, b = work / a.tilesF;
            const int f_base = ft * WR_TP;
            const float dsc = (work & 1) ? -dscale : dscale;
            const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (long)b * a.T * CIN * a.F), 0, (unsigned)((long)a.T * CIN * a.F * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + (long)b * a.T * COUT * a.F), 0, (unsigned)((long)a.T * COUT * a.F * 4), 0x00020000);
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
                const int rbase = row * CIN * a.F * 4;
#pragma unroll
                for (int it = 0; it < XIT; ++it) {
                    const bool has = (it + 1) * 256 <= XITEMS || st + 256 * it < XITEMS;
                    xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (colok && has) ? rbase + xoff + 32 * it * a.F : -4, 0, 0));
                }
                xh = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, hok ? rbase + (hch * a.F + hf) * 4 : -4, 0, 0));
            };
            auto issue_d = [&](int row, f32x4 (&dr)[DIT]) {
                const int rbase = row * COUT * a.F * 4;
#pragma unroll
                for (int it = 0; it < DIT; ++it) {
                    const bool has = (it + 1) * 256 <= DITEMS || st + 256 * it < DITEMS;
                    dr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drsrc, (colok && has) ? rbase + xoff + 32 * it * a.F : -4, 0, 0));
                }
            };
            // threads without an item in the last round store into bytes 0..7 / 256..263 of an image row: positions no fragment uses
            auto commit_x = [&](int slot, const f32x4 (&xr)[XIT], float xh) {
#pragma unroll
                for (int it = 0; it < XIT; ++it) {
                    const bool has = (it + 1) * 256 <= XITEMS || st + 256 * it < XITEMS;
                    unsigned char* const p0 = xring + slot * XSLOT + (has ? ((st >> 5) + 8 * it) * WR_SROW + 16 + (st & 31) * 8 : (st >> 5) * WR_SROW);
                    f32x4 v = xr[it];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaf(v[k], csc[it], csh[it]), relu_floor);
                    uint2 t0, t1;
                    split2_pair_f16(v[0], v[1], t0.x, t1.x);
                    split2_pair_f16(v[2], v[3], t0.y, t1.y);
                    *reinterpret_cast<uint2*>(p0) = t0;
                    *reinterpret_cast<uint2*>(p0 + TSX) = t1;
                }
                unsigned p0, p1;
                split2_pair_f16(fmaxf(fmaf(xh, chsc, chsh), relu_floor), 0.f, p0, p1);
                unsigned char* dst = xring + slot * XSLOT + (hthread ? hch * WR_SROW + (hside ? 8 + WR_TP : 7) * 2 : (st >> 5) * WR_SROW + 8);
                *reinterpret_cast<unsigned short*>(dst) = (unsigned short)p0;
                *reinterpret_cast<unsigned short*>(dst + TSX) = (unsigned short)p1;
            };
            auto commit_d = [&](int buf, const f32x4 (&dr)[DIT]) {
#pragma unroll
                for (int it = 0; it < DIT; ++it) {
                    const bool has = (it + 1) * 256 <= DITEMS || st + 256 * it < DITEMS;
                    unsigned char* const p0 = dzbuf + buf * DBUF + (has ? ((st >> 5) + 8 * it) * WR_SROW + (st & 31) * 8 : (st >> 5) * WR_SROW + 256);
                    uint2 t0, t1;
                    split2_pair_f16(dr[it][0] * dsc, dr[it][1] * dsc, t0.x, t1.x);
                    split2_pair_f16(dr[it][2] * dsc, dr[it][3] * dsc, t0.y, t1.y);
                    *reinterpret_cast<uint2*>(p0) = t0;
                    *reinterpret_cast<uint2*>(p0 + TSD) = t1;
                }
            };
            // Two register sets keep the global loads of TWO rows in flight (one row per CU = 40 KB x 256 CUs does not cover HBM's
            // latency x bandwidth product).
            f32x4 xr0[XIT], dr0[DIT], xr1[XIT], dr1[DIT];
            float xh0, xh1;
            __syncthreads();                          // (A) the previous strip's last multiply is over
            issue_x(0, xr0, xh0); issue_x(1, xr1, xh1); issue_d(0, dr0);
            commit_x(1, xr0, xh0); commit_x(2, xr1, xh1); commit_d(0, dr0);
            issue_x(2, xr0, xh0); issue_d(1, dr0);
            issue_x(3, xr1, xh1); issue_d(2, dr1);
            __syncthreads();                          // (B)
#pragma unroll 1
            for (int t = 0; t < a.T; t += 2) {
                commit_x((t + 3) & 3, xr0, xh0); commit_d((t + 1) & 1, dr0);            // input row t + 2, dz row t + 1
                issue_x(t + 4, xr0, xh0); issue_d(t + 3, dr0);
                __syncthreads();
                if (t + 1 < a.T) {
                    commit_x((t + 4) & 3, xr1, xh1); commit_d((t + 2) & 1, dr1);        // input row t + 3, dz row t + 2
                    issue_x(t + 5, xr1, xh1); issue_d(t + 4, dr1);
                    __syncthreads();
                }
            }
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

int a2s_conv3x3_wgrad_rows_impl(hipStream_t st, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW, float* ws,
                                size_t ws_bytes, int B, int T, int F, int Cin, int Cout, const float* dy_absmax, const float* act_absmax) {
    const int tilesF = a2s_cdiv(F, WR_TP), nwork = B * tilesF;
    const int nslabs = nwork < WR_SLABS ? nwork : WR_SLABS;
    A2S_REQUIRE(ws_bytes >= (size_t)nslabs * Cout * Cin * 9 * sizeof(float), "conv3x3_wgrad_rows: workspace too small");
    WgRowsArgs a{dy, x, in_scale, in_shift, dy_absmax, act_absmax, ws, B, T, F, tilesF, nwork};
    if (a2s_wgrad_rows_enabled() == 2) {              // the role-split form
        if (Cin == 40 && Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad_rows2<40, 40>), dim3(nslabs), dim3(512), 0, st, a);
        else if (Cin == 20 && Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad_rows2<20, 40>), dim3(nslabs), dim3(512), 0, st, a);
        else if (Cin == 20 && Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_rows2<20, 20>), dim3(nslabs), dim3(512), 0, st, a);
        else if (Cin == 40 && Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_rows2<40, 20>), dim3(nslabs), dim3(512), 0, st, a);
        else A2S_FAIL(A2S_ERR_ARG, "conv3x3_wgrad_rows: no instance for %d -> %d", Cin, Cout);
    } else
    if (Cin == 40 && Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad_rows<40, 40>), dim3(nslabs), dim3(256), 0, st, a);
    else if (Cin == 20 && Cout == 40) hipLaunchKernelGGL((conv3x3_wgrad_rows<20, 40>), dim3(nslabs), dim3(256), 0, st, a);
    else if (Cin == 20 && Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_rows<20, 20>), dim3(nslabs), dim3(256), 0, st, a);
    else if (Cin == 40 && Cout == 20) hipLaunchKernelGGL((conv3x3_wgrad_rows<40, 20>), dim3(nslabs), dim3(256), 0, st, a);
    else A2S_FAIL(A2S_ERR_ARG, "conv3x3_wgrad_rows: no instance for %d -> %d", Cin, Cout);
    A2S_CHECK_LAUNCH("conv3x3_wgrad_rows");
    hipLaunchKernelGGL(wgrad_rows_reduce, dim3(a2s_cdiv(Cout * Cin * 9, 256)), dim3(256), 0, st, ws, dW, nslabs, Cout * Cin * 9);
    A2S_CHECK_LAUNCH("wgrad_rows_reduce");
    return A2S_OK;
}
