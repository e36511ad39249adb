// Fused per-step kernels of the note decoder for FEW rows (reference NoteDecoder.decode_notes step, models.py:388-419, and its reverse).
//
// A decode step is a chain of dependent products on a handful of rows: as library-style launches (4 GEMMs + their split-K reduces +
// gates + epilogue) it is 9-13 kernels of ~10 us each whatever the row count -- which is what bounds (i) the long-clip group of the
// fused training step (engine.Engine.forward, clip groups: ~400 dependent steps per segment on 20-40 rows), (ii) small-batch greedy
// decoding.  Here a step is FOUR launches forward (attention split, attention combine, `dec_gru_step`, `dec_out_step`) and FIVE
// backward (gate backward, `dec_bwd_products`, attention split, attention combine, `dec_bwd_query`):
//
//   dec_gru_step   gi = x W_ih^T and gh = h W_hh^T for a 16-row x 16-unit tile (x 3 gates) on the fp32 matrix cores, gate math on
//                  the accumulators -> h' (state slot, [h' | ctx] output row, saved gates).  Replaces 2 GEMMs + 2 reduces + gates.
//   dec_out_step   logits = [h' | ctx] W_out^T + b for 16 rows x all 173 symbols in ONE workgroup, log-softmax, argmax (lowest index on
//                  ties, torch's CPU rule), teacher-forced / fed-back token choice, its embedding (+ dropout) into the next input row,
//                  <eos> bookkeeping -- and, in other workgroups of the same launch, the NEXT step's attention query
//                  q = h' W_h^T + b.  Replaces 2 GEMMs + 2 reduces + the epilogue kernel.
//   dec_bwd_products   dx = dgi W_ih and dh_prev += dgh W_hh (both on transposed weight copies made once per call).
//   dec_bwd_query      dh_prev += dq W_h after the attention backward.
//
// All operands are K-contiguous rows fetched straight from L2 with 16-byte loads (no LDS staging, as gru_step_fwd_fused); the 8 waves
// of a workgroup take the 16-wide k-steps round robin and their partial tiles meet in a fixed-order tree through LDS (deterministic).
// Used when the call has at most a2s_debug_set("dec_fused_max_rows") rows (default 192): with many rows every workgroup re-reads the weights from
// L2 and the tiled GEMMs win again (they only run under the other staff's attention there anyway).
#include "a2s_common.h"
#include <type_traits>
#include "../../include/a2s.h"

int a2s_gemm_impl(hipStream_t st, int M, int N, int K, float alpha, const float* A, long sAm, long sAk,
                  const float* B, long sBk, long sBn, float beta, float* C, long ldc, const float* bias, int act,
                  int batch, long bsA, long bsB, long bsC, int splitk, float* ws, size_t ws_bytes);
int a2s_attn_step_fwd_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                           float* ctx, long ldctx, float* ctx2, long ldctx2, float* attw, int B, int T, int H, const int* n_done, int n_rows,
                           float* ws, const a2s_attn_rows* rows, a2s_attn_deferred* defer = nullptr);
int a2s_attn_step_bwd_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                           const float* attw, const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b, long lddb,
                           float* dctx_out, long lddo, float* dq, long lddq, float* ds_out, int B, int T, int H, float* ws, const a2s_attn_rows* rows);
int a2s_gru_gates_bwd_impl(hipStream_t st, const float* dh_a, long lda, const float* dh_b, long ldb, const float* save,
                           const float* hprev, long ldhp, float* dgi, long ldgi, float* dgh, long ldgh, float* dgh2, long ldgh2,
                           float* dhprev, long lddp, int R, int H);

#define NW 8                      // waves per workgroup
// k-steps of operand loads a wave has in flight at a time in the few-row kernels (the CH argument of mfma_rows8): a wave's share of the k-steps is
// fetched in batches of CH, each batch waited for before its MFMAs.  Round 5 measured the whole range in the training step (tools/lib_ab.sh-style
// alternating processes, 8 pairs each, profiles/r05_step_kernel_batches.txt), same bits everywhere:
//     one batch per operand (CH = 5 / 8 / 12)   437.9 / 441.8 ms
//     CH = 4 (rounds 3-5)                        435.4 / 437.0
//     CH = 2                                     432.1 / 428.5
//     CH = 1                                     429.2 / 429.0
// FEWER loads in flight per wave win: these launches run beside the attention sweeps, which keep the memory system's queues full -- a wave that
// asks for 12-36 float4 at once waits for the last of them, a wave that asks for 3-4 gets them back in the time of one, and the compiler
// overlaps the next k-step's loads with the MFMAs anyway (the loop is not unrolled across the wait).  The registers (46-70 instead of 119-177 for one batch per operand)
// also let the dispatcher place the workgroups sooner.
#ifndef DEC_CH_GRU
#define DEC_CH_GRU 1
#define DEC_CH_OUT 1
#define DEC_CH_PROD 1
#define DEC_CH_QUERY 1
#endif
#ifndef DEC_NW_PROD
#define DEC_NW_PROD 8             // waves of dec_bwd_products / dec_out_step (16 halves a wave's chain of k-steps: measured, no gain -- profiles/r05_step_kernel_batches.txt)
#define DEC_NW_OUT 8
#endif

// acc[g] += A-row-fragments x B-row-fragments over this wave's share of the k-steps (k-step u covers k in [16u, 16u+16); this lane
// reads 4 floats at 16u + 4*lk of its A row and of its NT B rows)
template <int NT, int CH, int NWV = NW>
__device__ __forceinline__ void mfma_rows8(const float* __restrict__ arow, const float* const (&brow)[NT], int ksteps, int wave, int lk,
                                           f32x4 (&acc)[NT]) {
    for (int u0 = wave; u0 < ksteps; u0 += NWV * CH) {
        f32x4 a[CH], b[NT][CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int u = u0 + NWV * c;
            const bool ok = u < ksteps;
            a[c] = ok ? *reinterpret_cast<const f32x4*>(arow + 16 * u + 4 * lk) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < NT; ++g)
                b[g][c] = ok ? *reinterpret_cast<const f32x4*>(brow[g] + 16 * u + 4 * lk) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (u0 + NWV * c >= ksteps) break;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < NT; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[g][c][j], acc[g], 0, 0, 0);
        }
    }
}

// Sum of the waves' partial tiles in wave 0, fixed order: (w, w+4) -> (w, w+2) -> (0, 1) for 8 waves.  part: (NWV / 2) * NT * 64 f32x4 of LDS.
template <int NT, int NWV = NW>
__device__ __forceinline__ void reduce_waves(f32x4 (&acc)[NT], f32x4* part, int wave, int lane) {
#pragma unroll
    for (int half = NWV / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int g = 0; g < NT; ++g) part[((wave - half) * NT + g) * 64 + lane] = acc[g];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                const f32x4 o = part[(wave * NT + g) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[g][r] += o[r];
            }
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------- forward: GRU cell
// GRU cell of one (row, hidden unit) from the four product sums (reference models.py NoteDecoder's nn.GRU): ONE definition with floating-point
// contraction off, so that every kernel that ends in it (dec_gru_step, its row-tiled and combine-folding forms) rounds alike.
struct GruCellOut { float rg, zg, ng, ghn, hn; };
__device__ __forceinline__ GruCellOut gru_cell(float ar, float az, float an, float ahn, float br, float bz, float bin, float bhn, float hp) {
#pragma clang fp contract(off)
    GruCellOut o;
    o.ghn = ahn + bhn;
    o.rg = fast_sigmoid(ar + br);
    o.zg = fast_sigmoid(az + bz);
    const float pre = an + bin;
    const float gate = o.rg * o.ghn;
    o.ng = fast_tanh(pre + gate);
    const float keep = (1.f - o.zg) * o.ng;
    const float carry = o.zg * hp;
    o.hn = keep + carry;
    return o;
}

struct DecGruArgs {
    const float* x; long ldx; int kx;                   // (R, ldx) rows [token | ctx]; kx = E + 2H
    const float* h;                                     // (R, H2) previous state
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;
    float* hout; float* o; long ldo; float* save;       // h' -> hout (R, H2) and o[:, :H2]; save (R, 4 H2) [r|z|n|gh_n] or null
    const int* n_done; int* skip;                       // greedy: the step is a no-op once *n_done >= R; *skip tells the step's later kernels
    const int* rowmap;                                  // optional: the launch covers rows rowmap[0 .. R) of the call (rows still running), not 0 .. R
    int R, H2;
};
// physical row of the v-th row a launch covers
__device__ __forceinline__ int dec_row(const int* __restrict__ rowmap, int v) { return rowmap ? rowmap[v] : v; }

__global__ __launch_bounds__(64 * NW) void dec_gru_step(DecGruArgs a) {
    __shared__ f32x4 part[4 * 4 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.n_done) {
        const bool done = *a.n_done >= a.R;               // nobody writes n_done while this kernel runs (the epilogue of the previous step is over)
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.skip = done ? 1 : 0;
        if (done) return;
    }
    const int H2 = a.H2, R = a.R;
    const int j0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int j = j0 + li;
    float hp[4], br = 0.f, bz = 0.f, bin = 0.f, bhn = 0.f;
    if (wave == 0) {                                      // epilogue operands: in flight while the products run
        br = a.b_ih[j] + a.b_hh[j]; bz = a.b_ih[H2 + j] + a.b_hh[H2 + j]; bin = a.b_ih[2 * H2 + j]; bhn = a.b_hh[2 * H2 + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = a.h[(long)dec_row(a.rowmap, min(row0 + lk * 4 + r, R - 1)) * H2 + j];
    }
    const int arow_i = dec_row(a.rowmap, min(row0 + li, R - 1));
    const float* bi[3];
    const float* bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { bi[g] = a.w_ih + ((long)g * H2 + j0 + li) * a.kx; bh[g] = a.w_hh + ((long)g * H2 + j0 + li) * H2; }
    f32x4 t[3], u[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) t[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_rows8<3, DEC_CH_GRU>(a.x + (long)arow_i * a.ldx, bi, a.kx / 16, wave, lk, t);          // gi: r, z, n
    u[0] = t[0]; u[1] = t[1]; u[2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_rows8<3, DEC_CH_GRU>(a.h + (long)arow_i * H2, bh, H2 / 16, wave, lk, u);               // + gh on r, z; gh_n apart
    f32x4 acc[4] = {u[0], u[1], t[2], u[2]};
    reduce_waves<4>(acc, part, wave, lane);
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (row0 + lk * 4 + r >= R) continue;
        const int row = dec_row(a.rowmap, row0 + lk * 4 + r);
        const GruCellOut cell = gru_cell(acc[0][r], acc[1][r], acc[2][r], acc[3][r], br, bz, bin, bhn, hp[r]);
        const float rg = cell.rg, zg = cell.zg, ng = cell.ng, ghn = cell.ghn, hn = cell.hn;
        a.hout[(long)row * H2 + j] = hn;
        a.o[(long)row * a.ldo + j] = hn;
        if (a.save) {
            float* sv = a.save + (long)row * 4 * H2;
            sv[j] = rg; sv[H2 + j] = zg; sv[2 * H2 + j] = ng; sv[3 * H2 + j] = ghn;
        }
    }
}


// ---- dec_gru_step with the attention combine of its rows folded into the prologue (round 5; few-clip training launches: a2s_attn_deferred).
// On the long-clip group's chain a decode step was  sweep -> combine -> GRU step -> output step,  and under the other clip group's traffic a
// launch costs ~20 us of fixed time plus ~18 us of gap however little it does (profiles/r05_trace_overlap.txt: the combine 23 us median for a few
// hundred KB).  Here every workgroup of the GRU step (32 hidden-unit tiles x row blocks) merges the G partials of ITS <= 16 rows into LDS --
// redundantly: rows x G x 2 KB from L2 per workgroup, and late in a bar segment only 1-3 rows still run -- and reads the context k-steps of
// its A operand from there; workgroup x < 16 of a row block also writes row x's context where the combine kernel wrote it (the GRU input row:
// the deferred weight gradients read it; the output step's operand row), workgroup 16 + x normalises row x's saved scores, and every
// workgroup zero-fills its share of the rows the attention skipped.  Arithmetic and summation order are the combine kernel's
// (attn_combine_row) and dec_gru_step's: either path gives the same bits.  Requires H2 == 512 (one context column per thread).
struct DecCmbArgs {
    const float* part; float* attw; float* xw;          // partials; raw scores (Rall, T) or null; the GRU input rows, writable (row stride a.ldx)
    const int* clip_rank; const int* row_until;
    int G, groups, n_clips, n_active, step, T, Rall, E;
};
#define CMB_LD 516                 // floats per context row in LDS (rows land 4 banks apart: conflict-free 16-byte fragment reads)
#define CMB_PS 516                 // floats per partial: [m, l, pad, pad, ctx(512)]

template <int NT, int CH>
__device__ __forceinline__ void mfma_rows8_split(const float* __restrict__ arow, int ke, const float* lrow, const float* const (&brow)[NT], int ksteps,
                                                 int wave, int lk, f32x4 (&acc)[NT]) {
    // as mfma_rows8; k-steps u < ke come from the global row, the others from the LDS row (columns 16 (u - ke) ..)
    for (int u0 = wave; u0 < ksteps; u0 += NW * CH) {
        f32x4 a[CH], b[NT][CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int u = u0 + NW * c;
            const bool ok = u < ksteps;
            a[c] = ok ? (u < ke ? *reinterpret_cast<const f32x4*>(arow + 16 * u + 4 * lk) : *reinterpret_cast<const f32x4*>(lrow + 16 * (u - ke) + 4 * lk))
                      : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < NT; ++g)
                b[g][c] = ok ? *reinterpret_cast<const f32x4*>(brow[g] + 16 * u + 4 * lk) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (u0 + NW * c >= ksteps) break;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < NT; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[g][c][j], acc[g], 0, 0, 0);
        }
    }
}

// LDS (dynamic, sized by the rows a block can hold -- min(16, rows of the launch): beside the bulk clip group's occupancy-capped sweeps a CU has
// ~32 KB of LDS free, and a kernel that asks for more waits for bulk workgroups to leave): [cross-wave reduction 16 KB][weights][bases][contexts]
static size_t dec_cmb_lds_bytes(int rows_blk) { return 4 * 4 * 64 * sizeof(f32x4) + 16 * 20 * sizeof(float) + 16 * sizeof(int) + (size_t)rows_blk * CMB_LD * sizeof(float); }
__global__ __launch_bounds__(64 * NW) void dec_gru_step_cmb(DecGruArgs a, DecCmbArgs c) {
    extern __shared__ __attribute__((aligned(16))) float dyn_cmb[];
    f32x4* part = reinterpret_cast<f32x4*>(dyn_cmb);      // 4 * 4 * 64
    float* cw = dyn_cmb + 4 * 4 * 64 * 4;                 // [16][20] per row: weights of the G partials [0, 16), max [16], 1 / sum [17], row computed [18]
    int* cbase = reinterpret_cast<int*>(cw + 16 * 20);    // per row: index of its first partial
    float* ctxs = cw + 16 * 20 + 16;                      // [rows of a block][CMB_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H2 = a.H2, R = a.R;
    const int j0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
    const int nb = min(16, R - row0);
    const int li = lane & 15, lk = lane >> 4;
    const int j = j0 + li;
    // ---- rows of the call that the attention skipped this step: zeros where the combine kernel wrote zeros
    {
        const int nwg = gridDim.x * gridDim.y, w = blockIdx.y * gridDim.x + blockIdx.x;
        for (int b = w; b < c.Rall; b += nwg) {
            const int clip = b % c.n_clips;
            const int slot = c.clip_rank ? c.clip_rank[clip] : clip;
            if (slot >= c.n_active || (c.row_until && c.step >= c.row_until[b])) {
                c.xw[(long)b * a.ldx + c.E + tid] = 0.f;
                a.o[(long)b * a.ldo + H2 + tid] = 0.f;
                if (c.attw) for (int t = tid; t < c.T; t += 64 * NW) c.attw[(long)b * c.T + t] = 0.f;
            }
        }
    }
    float hp[4], br = 0.f, bz = 0.f, bin = 0.f, bhn = 0.f;
    if (wave == 0) {                                      // epilogue operands: in flight while everything else runs
        br = a.b_ih[j] + a.b_hh[j]; bz = a.b_ih[H2 + j] + a.b_hh[H2 + j]; bin = a.b_ih[2 * H2 + j]; bhn = a.b_hh[2 * H2 + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[r] = a.h[(long)dec_row(a.rowmap, min(row0 + lk * 4 + r, R - 1)) * H2 + j];
    }
    // ---- softmax statistics of the block's rows: wave w takes rows 2 w, 2 w + 1 (lanes, reductions and expressions of attn_combine_row)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int i = 2 * wave + rr;
        if (i >= nb) break;
        const int b = dec_row(a.rowmap, row0 + i);
        const int clip = b % c.n_clips, grp = b / c.n_clips;
        const int slot = c.clip_rank ? c.clip_rank[clip] : clip;
        const bool on = slot < c.n_active && !(c.row_until && c.step >= c.row_until[b]);
        if (on) {
            const int base = slot * c.groups + grp;
            const float* pb = c.part + (long)base * c.G * CMB_PS;
            const bool have = lane < c.G;
            const float mg = have ? pb[(long)lane * CMB_PS] : -INFINITY, lg = have ? pb[(long)lane * CMB_PS + 1] : 0.f;
            float m, inv_l;
            const float wg = attn_merge_weight(mg, lg, have, m, inv_l);
            if (lane < 16) cw[i * 20 + lane] = wg;
            if (lane == 0) { cw[i * 20 + 16] = m; cw[i * 20 + 17] = inv_l; cw[i * 20 + 18] = 1.f; cbase[i] = base; }
        } else if (lane == 0) { cw[i * 20 + 18] = 0.f; cbase[i] = 0; }
    }
    __syncthreads();
    // ---- contexts: thread = column, two rows' partials in flight at a time; acc = fma chain over the partials in order, as the combine kernel
    for (int i0 = 0; i0 < nb; i0 += 2) {
        float p[2][16];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int i = min(i0 + rr, nb - 1);
            const bool on = cw[i * 20 + 18] != 0.f;
            const float* pb = c.part + (long)cbase[i] * c.G * CMB_PS + 4 + tid;
#pragma unroll
            for (int u = 0; u < 16; ++u) p[rr][u] = (on && u < c.G) ? pb[(long)u * CMB_PS] : 0.f;
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int i = i0 + rr;
            if (i >= nb) break;
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = fmaf(p[rr][u], cw[i * 20 + u], acc);
            ctxs[i * CMB_LD + tid] = acc;
        }
    }
    __syncthreads();
    // ---- what the combine kernel left in memory: row x's context (workgroup x), row x's normalised weights (workgroup 16 + x)
    if ((int)blockIdx.x < nb) {
        const int i = blockIdx.x;
        const long row = dec_row(a.rowmap, row0 + i);
        const float v = ctxs[i * CMB_LD + tid];
        c.xw[row * a.ldx + c.E + tid] = v;
        a.o[row * a.ldo + H2 + tid] = v;
    } else if ((int)blockIdx.x >= 16 && (int)blockIdx.x - 16 < nb && c.attw) {
        const int i = blockIdx.x - 16;
        float* aw = c.attw + (long)dec_row(a.rowmap, row0 + i) * c.T;
        const bool on = cw[i * 20 + 18] != 0.f;
        const float m = cw[i * 20 + 16], inv = cw[i * 20 + 17];
        float sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sv[u] = (on && tid + 64 * NW * u < c.T) ? aw[tid + 64 * NW * u] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) if (tid + 64 * NW * u < c.T) aw[tid + 64 * NW * u] = on ? __expf(sv[u] - m) * inv : 0.f;
        for (int t = tid + 4 * 64 * NW; t < c.T; t += 64 * NW) aw[t] = on ? __expf(aw[t] - m) * inv : 0.f;       // (T > 2048 only)
    }
    // ---- the GRU products, as dec_gru_step; the context k-steps of the A operand come from LDS
    const int arow_i = dec_row(a.rowmap, min(row0 + li, R - 1));
    const float* lrow = ctxs + min(li, nb - 1) * CMB_LD;
    const float* bi[3];
    const float* bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { bi[g] = a.w_ih + ((long)g * H2 + j0 + li) * a.kx; bh[g] = a.w_hh + ((long)g * H2 + j0 + li) * H2; }
    f32x4 t[3], u[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) t[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_rows8_split<3, DEC_CH_GRU>(a.x + (long)arow_i * a.ldx, c.E / 16, lrow, bi, a.kx / 16, wave, lk, t);          // gi: r, z, n
    u[0] = t[0]; u[1] = t[1]; u[2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_rows8<3, DEC_CH_GRU>(a.h + (long)arow_i * H2, bh, H2 / 16, wave, lk, u);               // + gh on r, z; gh_n apart
    f32x4 acc[4] = {u[0], u[1], t[2], u[2]};
    reduce_waves<4>(acc, part, wave, lane);
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (row0 + lk * 4 + r >= R) continue;
        const int row = dec_row(a.rowmap, row0 + lk * 4 + r);
        const GruCellOut cell = gru_cell(acc[0][r], acc[1][r], acc[2][r], acc[3][r], br, bz, bin, bhn, hp[r]);
        const float rg = cell.rg, zg = cell.zg, ng = cell.ng, ghn = cell.ghn, hn = cell.hn;
        a.hout[(long)row * H2 + j] = hn;
        a.o[(long)row * a.ldo + j] = hn;
        if (a.save) {
            float* sv = a.save + (long)row * 4 * H2;
            sv[j] = rg; sv[H2 + j] = zg; sv[2 * H2 + j] = ng; sv[3 * H2 + j] = ghn;
        }
    }
}

// ------------------------------------------------------------------------------------------- forward: GRU cell for HUNDREDS of rows (round 6)
// The bulk clip group's decode step -- ~500 rows -- ran its two gate products as library-style launches (gh before the attention sweep, gi
// behind it: 64 x 32 fp32 tiles that walk K as 17 barrier-separated k-tiles fetched through L2, 46-54 us EACH in the step,
// profiles/r05_step_trace_b256.txt) plus the elementwise gate kernel: ~150 of the ~200 us a step spends outside its sweep, during which the
// HBM idles.  The 16-row kernels above do not scale to these row counts (every workgroup pulls its weights from L2 again, one k-step of loads
// in flight per wave: 77 us at 496 rows).  This kernel is the same fused cell -- gi and gh for a tile of hidden units, gate math on the
// accumulators, no gi / gh round trip through memory -- organised for a few hundred rows:
//   * a workgroup owns 64 rows x 16 hidden units (their r, z, n gate columns); wave w owns rows 16 w .. 16 w + 15 for ALL of K = kx + H2:
//     no cross-wave reduction;
//   * the weights of a 64-wide k-chunk (48 rows x 64 floats) are staged ONCE per workgroup in LDS (double-buffered, 288-byte rows: the
//     conflict-free stride of the fragment pattern, profiles/r04_lds_patterns.txt) and read as ds_read_b128 fragments by all four waves;
//   * a wave's own rows never touch LDS: lane (li, lk) fetches 16 bytes at k + 4 lk of row li straight from memory, one chunk ahead; the
//     weight chunk after the next is in flight in registers meanwhile.  One barrier per chunk.
// Grid (H2 / 16, rows / 64): 256 workgroups at 496 rows; unit tile x lands on XCD x % 8 in every launch, so an XCD's L2 keeps ITS eighth of
// the weights (0.8 MB per decoder) across the steps.  The k order inside a 16-wide k-step is the 16-row kernels' (lane group lk supplies
// k = 4 lk + j for the j-th MFMA); the sum runs over the chunks in order, x part first.
#define MID_KC 64
#define MID_LDB 72                // floats per staged weight row: 64 + 8
#ifndef MID_WAVES
// waves per SIMD the mid-size kernels' register budget is sized for (launch bounds) and chunks the wave's own rows are requested ahead (1 or 2).
// Measured in the step (tools/lib_ab2.sh, three rounds of 12 steps, profiles/r06_lib_ab_variants.txt): (1 wave, 2 ahead: 156 / 108 / 112 / 104
// registers) 441.7 ms, (4 waves, 1 ahead: 122 / 88 / 90 / 84) 442.1 -- equal; the smaller footprint is the default.
#define MID_WAVES 4
#define MID_AHEAD 1
#endif
__global__ __launch_bounds__(256, MID_WAVES) void dec_gru_mid(DecGruArgs a) {
    __shared__ __attribute__((aligned(16))) float bs[2][48 * MID_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.n_done) {
        const bool done = *a.n_done >= a.R;
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.skip = done ? 1 : 0;
        if (done) return;
    }
    const int H2 = a.H2, R = a.R, kx = a.kx;
    const int j0 = blockIdx.x * 16, row0 = blockIdx.y * 64 + wave * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int j = j0 + li;
    const int ncx = (kx + MID_KC - 1) / MID_KC, nc = ncx + (H2 + MID_KC - 1) / MID_KC;
    // epilogue operands: in flight while the products run
    const float br = a.b_ih[j] + a.b_hh[j], bz = a.b_ih[H2 + j] + a.b_hh[H2 + j], bin = a.b_ih[2 * H2 + j], bhn = a.b_hh[2 * H2 + j];
    float hp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hp[r] = a.h[(long)dec_row(a.rowmap, min(row0 + lk * 4 + r, R - 1)) * H2 + j];
    const int arow = dec_row(a.rowmap, min(row0 + li, R - 1));
    const float* const xa = a.x + (long)arow * a.ldx + 4 * lk;
    const float* const ha = a.h + (long)arow * H2 + 4 * lk;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // weight chunk c -> registers: item e = tid + 256 i is (staged row e >> 4 = gate * 16 + unit, 16-byte column e & 15)
    auto bload = [&](int c, f32x4 (&v)[3]) {
        const bool px = c < ncx;
        const int k0 = (px ? c : c - ncx) * MID_KC, K = px ? kx : H2;
        const float* const W = px ? a.w_ih : a.w_hh;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + 256 * i, r = e >> 4, k = k0 + 4 * (e & 15);
            v[i] = k < K ? *reinterpret_cast<const f32x4*>(W + ((long)(r >> 4) * H2 + j0 + (r & 15)) * K + k) : zero4;
        }
    };
    auto bstore = [&](int buf, const f32x4 (&v)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + 256 * i;
            *reinterpret_cast<f32x4*>(&bs[buf][(e >> 4) * MID_LDB + 4 * (e & 15)]) = v[i];
        }
    };
    auto aload = [&](int c, f32x4 (&v)[4]) {
        const bool px = c < ncx;
        const int k0 = (px ? c : c - ncx) * MID_KC, K = px ? kx : H2;
        const float* const p = px ? xa : ha;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (k0 + 16 * u < K) ? *reinterpret_cast<const f32x4*>(p + k0 + 16 * u) : zero4;     // (K % 16 == 0: a k-step is inside the row or not at all)
    };
    f32x4 acc[4] = {zero4, zero4, zero4, zero4};          // r, z, n (input part), n (state part)
    auto compute = [&](auto PX, int c, int buf, const f32x4 (&av)[4]) {
        constexpr bool px = decltype(PX)::value;
        const int k0 = (px ? c : c - ncx) * MID_KC, K = px ? kx : H2;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + 16 * u >= K) break;                   // (uniform)
            f32x4 b[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) b[g] = *reinterpret_cast<const f32x4*>(&bs[buf][(g * 16 + li) * MID_LDB + 16 * u + 4 * lk]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][jj], b[0][jj], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][jj], b[1][jj], acc[1], 0, 0, 0);
                acc[px ? 2 : 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][jj], b[2][jj], acc[px ? 2 : 3], 0, 0, 0);
            }
        }
    };
    // the wave's own rows are MID_AHEAD chunks ahead (they were written by the previous launch, on any XCD: the longest round trip of the kernel -- in the
    // step, beside the other staff's sweep, 3-5 us against 0.7 us of multiply per chunk), the weight chunk two ahead in registers, one in LDS
    f32x4 bv[3], a_cur[4], a_n1[4], a_n2[4];
    bload(0, bv);
    aload(0, a_cur);
    if (MID_AHEAD > 1 && nc > 1) aload(1, a_n1);
    bstore(0, bv);
    if (nc > 1) bload(1, bv);
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
#if MID_AHEAD > 1
        if (c + 2 < nc) aload(c + 2, a_n2);
#else
        if (c + 1 < nc) aload(c + 1, a_n1);
#endif
        if (c + 1 < nc) {
            bstore((c + 1) & 1, bv);                       // (that buffer was last read in iteration c - 1, which ended with a barrier)
            if (c + 2 < nc) bload(c + 2, bv);
        }
        if (c < ncx) compute(std::true_type{}, c, c & 1, a_cur);
        else compute(std::false_type{}, c, c & 1, a_cur);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) { a_cur[u] = a_n1[u]; if (MID_AHEAD > 1) a_n1[u] = a_n2[u]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (row0 + lk * 4 + r >= R) continue;
        const int row = dec_row(a.rowmap, row0 + lk * 4 + r);
        const GruCellOut cell = gru_cell(acc[0][r], acc[1][r], acc[2][r], acc[3][r], br, bz, bin, bhn, hp[r]);
        a.hout[(long)row * H2 + j] = cell.hn;
        a.o[(long)row * a.ldo + j] = cell.hn;
        if (a.save) {
            float* sv = a.save + (long)row * 4 * H2;
            sv[j] = cell.rg; sv[H2 + j] = cell.zg; sv[2 * H2 + j] = cell.ng; sv[3 * H2 + j] = cell.ghn;
        }
    }
}

// ------------------------------------------------------------------------------------------- forward: output projection + epilogue + next query
// One launch, three kinds of workgroups (blockIdx.y), all 16 rows x 32 columns x full K on 8 waves:
//   [0, NVW)        logits tile = [h' | ctx] W_out^T + b  -> a.logits; the workgroup that finishes LAST for its 16 rows (ticket counter
//                   per row block, release/acquire through L2) runs the step epilogue for them: log-softmax, argmax, token choice,
//                   embedding, <eos> bookkeeping -- one wave per row, as note_step_finalize;
//   [NVW, NVW+NQW)  the NEXT step's attention query  q = h' W_h^T + b.
// (A single workgroup per 16 rows doing all 173 columns took 40 us at 1-4 workgroups per launch: one CU pulling the 708 KB of W_out
// alone; spread over 6 + 8 workgroups per row block the launch takes ~10.)
#define NTV 11                    // 16-column tiles of the vocabulary: 161..176 symbols (LabelsMultiple(extended=True): 173)
#define NVW 6                     // vocabulary workgroups per row block (2 tiles each; the last one has one)
struct DecOutArgs {
    const float* o; long ldo; int ko;                   // (R, 2 H2) rows [h' | ctx]; ko = 2 H2
    const float* out_w; const float* out_b;             // (V, ko), (V)
    float* logits; long ldl;                            // (R, ldl) scratch, ldl >= 16 * NTV
    int* tickets;                                       // one counter per row block, zero between launches
    float* probs; long probs_bstride;                   // log-probabilities: row b, step t at probs + b*probs_bstride + t*V
    const long long* gt; long gt_bstride;
    const float* emb; float* xnext; long ldx;           // token embedding of the next step -> xnext[:, :E]
    const uint8_t* drop; float inv_keep;
    int* argmax_out; long am_bstride;
    int* eos_seen; long long* lengths; int* n_done; int* steps_exec;
    const int* t_base; const int* row_until; const int* skip;
    // next step's attention query
    const float* hnew; const float* attn_w; long ld_aw; const float* attn_b; float* q_next; int H;
    const int* rowmap;                                  // as in DecGruArgs; the logits scratch and the tickets are indexed by the launch's own row numbers
    int n_clips, R, V, E, t, teacher_force, eos_id, max_t, H2;
};

// the step epilogue for one row, executed by one wave (reference models.py:401-419); lg: the row's logits, read past the L1 (they
// were written by other workgroups of this launch)
__device__ __forceinline__ void dec_row_epilogue(const DecOutArgs& a, int vrow, int t, int lane) {
    const volatile float* lg = a.logits + (long)vrow * a.ldl;
    const int row = dec_row(a.rowmap, vrow);
    const bool finished = a.row_until && t >= a.row_until[row];    // its bar's loop has ended in the reference or only <pad> targets remain
    float v[3];
    float m = -INFINITY; int mi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < a.V ? lg[j] : -INFINITY;
        if (v[k] > m) { m = v[k]; mi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (!finished) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) if (lane + 64 * k < a.V) s += expf(v[k] - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        float* pr = a.probs + (long)row * a.probs_bstride + (long)t * a.V;
#pragma unroll
        for (int k = 0; k < 3; ++k) if (lane + 64 * k < a.V) pr[lane + 64 * k] = v[k] - lse;
    }
    const long long gtok = a.gt ? a.gt[(long)row * a.gt_bstride + t] : -1;
    const int tf = (a.teacher_force >> (a.n_clips > 0 ? row / a.n_clips : 0)) & 1;
    const int next_id = (a.gt && tf) ? (int)gtok : mi;
    for (int j = lane; j < a.E; j += 64) {
        float e = a.emb[(long)next_id * a.E + j];
        if (a.drop) e = a.drop[(long)row * a.E + j] ? e * a.inv_keep : 0.f;
        a.xnext[(long)row * a.ldx + j] = e;
    }
    if (lane == 0 && !finished) {
        if (row == 0 && a.steps_exec) *a.steps_exec = t + 1;      // steps run in order on one stream
        if (a.argmax_out) a.argmax_out[(long)row * a.am_bstride + t] = mi;
        const bool hit = a.gt ? (gtok == a.eos_id) : (mi == a.eos_id);
        if (hit) {
            if (!a.eos_seen[row]) { a.eos_seen[row] = 1; atomicAdd(a.n_done, 1); }
            a.lengths[row] = t + 1;
        }
    }
}

__global__ __launch_bounds__(64 * DEC_NW_OUT) void dec_out_step(DecOutArgs a) {
    __shared__ f32x4 part[(DEC_NW_OUT / 2) * 2 * 64];
    __shared__ int last_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.skip && *a.skip) return;
    const int t = a.t + (a.t_base ? *a.t_base : 0);
    if (t >= a.max_t) return;                                   // a replayed chunk may overshoot the step budget
    const int R = a.R, row0 = blockIdx.x * 16;
    const int li = lane & 15, lk = lane >> 4;
    const bool vocab = (int)blockIdx.y < NVW;
    if (!vocab && !a.q_next) return;
    const int n0 = (vocab ? blockIdx.y : blockIdx.y - NVW) * 32;
    const int ncols = vocab ? a.V : a.H;
    const float* Bm = vocab ? a.out_w : a.attn_w;
    const long ldb = vocab ? (long)a.ko : a.ld_aw;
    const int K = vocab ? a.ko : a.H2;
    const float* brow[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) brow[g] = Bm + (long)min(n0 + g * 16 + li, ncols - 1) * ldb;
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    const int arow_p = dec_row(a.rowmap, min(row0 + li, R - 1));
    const float* arow = (vocab ? a.o + (long)arow_p * a.ldo : a.hnew + (long)arow_p * a.H2);
    mfma_rows8<2, DEC_CH_OUT, DEC_NW_OUT>(arow, brow, K / 16, wave, lk, acc);
    reduce_waves<2, DEC_NW_OUT>(acc, part, wave, lane);
    if (wave == 0) {
        const float* bias = vocab ? a.out_b : a.attn_b;
        float* out = vocab ? a.logits : a.q_next;
        const long ldo = vocab ? a.ldl : (long)a.H;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int col = n0 + g * 16 + li;
            if (col >= ncols) continue;
            const float b = bias[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + lk * 4 + r;
                if (row < R) out[(long)(vocab ? row : dec_row(a.rowmap, row)) * ldo + col] = acc[g][r] + b;
            }
        }
    }
    if (!vocab) return;
    // ---- last vocabulary workgroup of this row block: the epilogue
    __syncthreads();
    if (tid == 0) {
#ifndef DEC_X_NOFENCE          /* timing ablation (results may be wrong): what the two device-scope fences of this kernel cost */
        __threadfence();                                           // this workgroup's logits tile is visible device-wide before its ticket
#endif
        const int ticket = atomicAdd(a.tickets + blockIdx.x, 1);
        last_flag = (ticket == NVW - 1);
        if (last_flag) a.tickets[blockIdx.x] = 0;                  // ready for the next launch
    }
    __syncthreads();
    if (!last_flag) return;
#ifndef DEC_X_NOFENCE
    __threadfence();
#endif
    for (int rr = wave; rr < 16; rr += DEC_NW_OUT) {
        const int row = row0 + rr;
        if (row < R) dec_row_epilogue(a, row, t, lane);
    }
}

// ------------------------------------------------------------------------------------------- backward products
// role A (blockIdx.x < nxa): dx[:, n] = dgi . W_ih[:, n]   for 32 columns n of the kx input columns  (B rows = W_ih^T rows)
// role B (the rest):         dh[:, n] += dgh . W_hh[:, n]  for 32 of the H2 state columns          (B rows = W_hh^T rows)
struct DecBwdProdArgs {
    const float* dgi; const float* dgh;                 // (R, 3 H2) each
    const float* wih_t; const float* whh_t;             // (kx, 3 H2), (H2, 3 H2)
    float* dx; long ldx; float* dh;                     // (R, ldx) overwritten; (R, H2) accumulated
    const int* rowmap;                                  // as in DecGruArgs
    int nxa, kx, R, H2;
};

__global__ __launch_bounds__(64 * DEC_NW_PROD) void dec_bwd_products(DecBwdProdArgs a) {
    __shared__ f32x4 part[(DEC_NW_PROD / 2) * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int R = a.R, K = 3 * a.H2, row0 = blockIdx.y * 16;
    const bool role_a = (int)blockIdx.x < a.nxa;
    const int n0 = (role_a ? blockIdx.x : blockIdx.x - a.nxa) * 32;
    const int ncols = role_a ? a.kx : a.H2;
    const float* A = role_a ? a.dgi : a.dgh;
    const float* Bt = role_a ? a.wih_t : a.whh_t;
    const float* brow[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) brow[g] = Bt + (long)min(n0 + g * 16 + li, ncols - 1) * K;
    float c0[2][4];
    if (wave == 0 && !role_a) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) c0[g][r] = a.dh[(long)dec_row(a.rowmap, min(row0 + lk * 4 + r, R - 1)) * a.H2 + min(n0 + g * 16 + li, ncols - 1)];
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    mfma_rows8<2, DEC_CH_PROD, DEC_NW_PROD>(A + (long)dec_row(a.rowmap, min(row0 + li, R - 1)) * K, brow, K / 16, wave, lk, acc);
    reduce_waves<2, DEC_NW_PROD>(acc, part, wave, lane);
    if (wave > 0) return;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int col = n0 + g * 16 + li;
        if (col >= ncols) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (row0 + lk * 4 + r >= R) continue;
            const int row = dec_row(a.rowmap, row0 + lk * 4 + r);
            if (role_a) a.dx[(long)row * a.ldx + col] = acc[g][r];
            else a.dh[(long)row * a.H2 + col] = c0[g][r] + acc[g][r];
        }
    }
}


// ---- 64 rows x 32 columns on the skeleton of dec_gru_mid (round 6): acc[g] (g = 0, 1: the tile's two 16-column n-tiles) += A . B^T over K for the
// 16 rows of the calling wave.  `ar`: this lane's A row (row li of the wave's m-tile) + 4 lk; B rows n0 .. n0 + 31 of Bt (row stride ldb, rows
// >= ncols read as zero) are staged per 64-wide k-chunk in LDS, double-buffered; K % 64 == 0.  All four waves of the workgroup must call it.
__device__ __forceinline__ void mid_product_64x32(const float* __restrict__ ar, const float* __restrict__ Bt, long ldb, int n0, int ncols, int K,
                                                  float (*bs)[32 * MID_LDB], f32x4 (&acc)[2]) {
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int nc = K / MID_KC;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto bload = [&](int c, f32x4 (&v)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i, n = n0 + (e >> 4);
            v[i] = n < ncols ? *reinterpret_cast<const f32x4*>(Bt + (long)n * ldb + c * MID_KC + 4 * (e & 15)) : zero4;
        }
    };
    auto bstore = [&](int buf, const f32x4 (&v)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i;
            *reinterpret_cast<f32x4*>(&bs[buf][(e >> 4) * MID_LDB + 4 * (e & 15)]) = v[i];
        }
    };
    auto aload = [&](int c, f32x4 (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(ar + c * MID_KC + 16 * u);
    };
    f32x4 bv[2], a_cur[4], a_n1[4], a_n2[4];             // (the wave's rows MID_AHEAD chunks ahead, as in dec_gru_mid)
    bload(0, bv);
    aload(0, a_cur);
    if (MID_AHEAD > 1 && nc > 1) aload(1, a_n1);
    bstore(0, bv);
    if (nc > 1) bload(1, bv);
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
#if MID_AHEAD > 1
        if (c + 2 < nc) aload(c + 2, a_n2);
#else
        if (c + 1 < nc) aload(c + 1, a_n1);
#endif
        if (c + 1 < nc) {
            bstore((c + 1) & 1, bv);                       // (that buffer was last read in iteration c - 1, which ended with a barrier)
            if (c + 2 < nc) bload(c + 2, bv);
        }
        const float* const bb = bs[c & 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 b[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) b[g] = *reinterpret_cast<const f32x4*>(&bb[(g * 16 + li) * MID_LDB + 16 * u + 4 * lk]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][jj], b[0][jj], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u][jj], b[1][jj], acc[1], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) { a_cur[u] = a_n1[u]; if (MID_AHEAD > 1) a_n1[u] = a_n2[u]; }
    }
}

// The two backward products for hundreds of rows: a workgroup owns 64 rows x 32 columns of dx (role A) or dh (role B), wave w rows 16 w .. 16 w + 15
// for all of K = 3 H2.  Replaces, in the bulk clip group's backward decode step, the dx product in front of the attention sweep and the dh
// product behind it (two 64 x 32-tile launches of 30-50 us each in the step) by one launch in front of it.
__global__ __launch_bounds__(256, MID_WAVES) void dec_bwd_mid(DecBwdProdArgs a) {
    __shared__ __attribute__((aligned(16))) float bs[2][32 * MID_LDB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int R = a.R, K = 3 * a.H2, row0 = blockIdx.y * 64 + wave * 16;
    const bool role_a = (int)blockIdx.x < a.nxa;
    const int n0 = (role_a ? blockIdx.x : blockIdx.x - a.nxa) * 32;
    const int ncols = role_a ? a.kx : a.H2;
    float c0[2][4];
    if (!role_a) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) c0[g][r] = a.dh[(long)dec_row(a.rowmap, min(row0 + lk * 4 + r, R - 1)) * a.H2 + min(n0 + g * 16 + li, ncols - 1)];
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    mid_product_64x32((role_a ? a.dgi : a.dgh) + (long)dec_row(a.rowmap, min(row0 + li, R - 1)) * K + 4 * lk, role_a ? a.wih_t : a.whh_t, K, n0, ncols, K, bs, acc);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int col = n0 + g * 16 + li;
        if (col >= ncols) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (row0 + lk * 4 + r >= R) continue;
            const int row = dec_row(a.rowmap, row0 + lk * 4 + r);
            if (role_a) a.dx[(long)row * a.ldx + col] = acc[g][r];
            else a.dh[(long)row * a.H2 + col] = c0[g][r] + acc[g][r];
        }
    }
}

// dh[:, n] += dq . W_h[:, n] behind the attention sweep, same tiles (B rows = W_h^T rows, (H2, H); K = H)
__global__ __launch_bounds__(256, MID_WAVES) void dec_bwd_query_mid(const float* __restrict__ dq, const float* __restrict__ wh_t, float* __restrict__ dh, int R, int H, int H2,
                                                         const int* __restrict__ rowmap) {
    __shared__ __attribute__((aligned(16))) float bs[2][32 * MID_LDB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int row0 = blockIdx.y * 64 + wave * 16, n0 = blockIdx.x * 32;
    float c0[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[g][r] = dh[(long)dec_row(rowmap, min(row0 + lk * 4 + r, R - 1)) * H2 + min(n0 + g * 16 + li, H2 - 1)];
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    mid_product_64x32(dq + (long)dec_row(rowmap, min(row0 + li, R - 1)) * H + 4 * lk, wh_t, H, n0, H2, H, bs, acc);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int col = n0 + g * 16 + li;
        if (col >= H2) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (row0 + lk * 4 + r >= R) continue;
            const int row = dec_row(rowmap, row0 + lk * 4 + r);
            dh[(long)row * H2 + col] = c0[g][r] + acc[g][r];
        }
    }
}

// Forward: logits = [h' | ctx] W_out^T + b (role A: 32-column tiles of the vocabulary, K = 2 H2) and the NEXT step's attention query
// q = h' W_h^T + b (role B: 32-column tiles of H, K = H2; absent in the last step) in one launch -- in the bulk loop two split-K products and
// their two reduce launches (~85 us in the step).  The epilogue (log-softmax, token choice, embedding) stays note_step_finalize.
struct DecOutqMidArgs {
    const float* o; long ldo;                           // (R, ldo) rows [h' | ctx]
    const float* out_w; const float* out_b; float* logits; long ldl; int V;
    const float* attn_w; long ld_aw; const float* attn_b; float* q_next; int H;      // q_next NULL: no query role
    const int* rowmap;
    int nva, R, H2;                                     // nva: vocabulary tiles
};
__global__ __launch_bounds__(256, MID_WAVES) void dec_outq_mid(DecOutqMidArgs a) {
    __shared__ __attribute__((aligned(16))) float bs[2][32 * MID_LDB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int R = a.R, row0 = blockIdx.y * 64 + wave * 16;
    const bool role_a = (int)blockIdx.x < a.nva;
    const int n0 = (role_a ? blockIdx.x : blockIdx.x - a.nva) * 32;
    const int ncols = role_a ? a.V : a.H;
    float bias[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) bias[g] = (role_a ? a.out_b : a.attn_b)[min(n0 + g * 16 + li, ncols - 1)];
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    mid_product_64x32(a.o + (long)dec_row(a.rowmap, min(row0 + li, R - 1)) * a.ldo + 4 * lk, role_a ? a.out_w : a.attn_w, role_a ? 2L * a.H2 : a.ld_aw, n0, ncols,
                      role_a ? 2 * a.H2 : a.H2, bs, acc);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int col = n0 + g * 16 + li;
        if (col >= ncols) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (row0 + lk * 4 + r >= R) continue;
            const int row = dec_row(a.rowmap, row0 + lk * 4 + r);
            if (role_a) a.logits[(long)row * a.ldl + col] = acc[g][r] + bias[g];
            else a.q_next[(long)row * a.H + col] = acc[g][r] + bias[g];
        }
    }
}

// dh[:, n] += dq . W_h[:, n]  (B rows = W_h^T rows, (H2, H)); 32 columns per workgroup
__global__ __launch_bounds__(64 * NW) void dec_bwd_query(const float* __restrict__ dq, const float* __restrict__ wh_t, float* __restrict__ dh,
                                                         int R, int H, int H2, const int* __restrict__ rowmap) {
    __shared__ f32x4 part[4 * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int row0 = blockIdx.y * 16, n0 = blockIdx.x * 32;
    const float* brow[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) brow[g] = wh_t + (long)min(n0 + g * 16 + li, H2 - 1) * H;
    float c0[2][4];
    if (wave == 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) c0[g][r] = dh[(long)dec_row(rowmap, min(row0 + lk * 4 + r, R - 1)) * H2 + min(n0 + g * 16 + li, H2 - 1)];
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    mfma_rows8<2, DEC_CH_QUERY>(dq + (long)dec_row(rowmap, min(row0 + li, R - 1)) * H, brow, H / 16, wave, lk, acc);
    reduce_waves<2>(acc, part, wave, lane);
    if (wave > 0) return;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int col = n0 + g * 16 + li;
        if (col >= H2) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (row0 + lk * 4 + r < R) dh[(long)dec_row(rowmap, row0 + lk * 4 + r) * H2 + col] = c0[g][r] + acc[g][r];
        }
    }
}

// out[c][r] = in[r * ld + c] for r < rows, c < cols
__global__ void transpose_ld(const float* __restrict__ in, long ld, float* __restrict__ out, int rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int c = (int)(i / rows), r = (int)(i % rows);
    out[i] = in[(long)r * ld + c];
}

// ------------------------------------------------------------------------------------------- switches / eligibility
static int g_dec_fused = -1, g_dec_fused_max_rows = -1;
void a2s_dec_fused_set(int v) { g_dec_fused = v ? 1 : 0; }
void a2s_dec_fused_max_rows_set(int v) { g_dec_fused_max_rows = v; }
int a2s_dec_fused_enabled(void) {
    if (g_dec_fused < 0) { const char* e = getenv("A2S_DEC_FUSED"); g_dec_fused = (e && e[0] == '0') ? 0 : 1; }      // (documented fallback: INTEGRATION.md)
    return g_dec_fused;
}
int a2s_dec_fused_max_rows(void) {
    if (g_dec_fused_max_rows < 0) g_dec_fused_max_rows = 192;
    return g_dec_fused_max_rows;
}
// scratch layout (floats): [16: flags | FUSED_MAX_RB: tickets | max_rows x 176: logits] then [W_ih^T | W_hh^T | W_h^T] for the backward
#define FUSED_MAX_ROWS_CAP 2048
#define FUSED_HEAD (16 + FUSED_MAX_ROWS_CAP / 16 + (long)FUSED_MAX_ROWS_CAP * 16 * NTV)
size_t a2s_note_step_fused_head_floats(void) { return (size_t)FUSED_HEAD; }
size_t a2s_note_step_workspace_floats_impl(int H, int E) {
    const long H2 = 2L * H, kx = E + H2;
    return (size_t)(FUSED_HEAD + kx * 3 * H2 + H2 * 3 * H2 + H2 * H);
}
static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// The mid-size kernels (dec_gru_mid, dec_bwd_mid: 64-row workgroups, weights staged through LDS) take the launches over more than 160 rows --
// below that the 16-row kernels fill the chip better (32-33 column tiles x rows / 16 workgroups).  a2s_debug_set("dec_mid", 0): never (A/B
// measurements, tests: the bulk calls' per-step products then run as library-style launches, the few-row path on the 16-row kernels only).
static int g_dec_mid = 1;
void a2s_dec_mid_set(int v) { g_dec_mid = v ? 1 : 0; }
int a2s_dec_mid_enabled(void) { return g_dec_mid; }
static int g_dec_mid_launches = 0;          // dec_gru_mid / dec_bwd_mid launches of this process (tests: proof of the path taken)
int a2s_dec_mid_launches(void) { return g_dec_mid_launches; }
static bool dec_use_mid(int nrows, int H2) { return g_dec_mid && nrows > 160 && H2 % MID_KC == 0; }
// greedy: the call is a greedy decode (no ground truth, no backward).  There the 4-launch step wins at every batch size the workspace admits
// (B = 256: 497 -> 536 clips/s, B = 64: 313 -> 317, profiles/r05_infer_variants.txt) -- one stream decodes a staff, nothing runs beside it that
// the weight re-reads of the 16-row tiles could disturb -- so the row limit of the training path (a2s_dec_fused_max_rows) does not apply.
bool a2s_dec_step_fusable(int R, int H, int E, int V, const void* const* ptrs, int nptrs, const float* ws, size_t ws_floats, bool greedy) {
    const int max_rows = greedy ? FUSED_MAX_ROWS_CAP : a2s_dec_fused_max_rows();
    if (!a2s_dec_fused_enabled() || R > max_rows || R > FUSED_MAX_ROWS_CAP || !ws || ws_floats < a2s_note_step_workspace_floats_impl(H, E)) return false;
    if (H % 16 || E % 16 || (V + 15) / 16 != NTV || !aligned16(ws)) return false;
    for (int i = 0; i < nptrs; ++i) if (!aligned16(ptrs[i])) return false;
    return true;
}

// ------------------------------------------------------------------------------------------- forward step (after the attention)
// nrows / rowmap: the step covers rows rowmap[0 .. nrows) of the call's R rows (rowmap NULL: rows 0 .. nrows = R)
int a2s_note_step_fused_fwd(hipStream_t st, const a2s_note_dec_args& a, int si, int so, int sv, int sv_next, int t, const int* t_base, int tf, bool last,
                            int nrows, const int* rowmap, const a2s_attn_deferred* defer) {
    const int H2 = 2 * a.H, ldx = a.E + H2, R = a.R;
    int* flags = reinterpret_cast<int*>(a.step_ws);
    const bool greedy = a.gt == nullptr;
    DecGruArgs g;
    g.x = a.x + (long)si * R * ldx; g.ldx = ldx; g.kx = ldx;
    g.h = a.h + (long)si * R * H2;
    g.w_ih = a.w_ih; g.w_hh = a.w_hh; g.b_ih = a.b_ih; g.b_hh = a.b_hh;
    g.hout = a.h + (long)so * R * H2; g.o = a.o + (long)sv * R * 2 * H2; g.ldo = 2 * H2;
    g.save = a.gates ? a.gates + (long)sv * R * 4 * H2 : nullptr;
    g.n_done = greedy ? a.n_done : nullptr; g.skip = flags;
    g.rowmap = rowmap;
    g.R = nrows; g.H2 = H2;
    if (defer && defer->G > 0) {                 // the attention launch in front of this step left its combine to us
        A2S_REQUIRE(H2 == 64 * NW && defer->G <= 16 && a.E % 16 == 0, "note_step_fused_fwd: deferred combine needs 2 * hidden_size == %d and G <= 16", 64 * NW);
        DecCmbArgs c;
        c.part = defer->part; c.attw = defer->attw; c.xw = a.x + (long)si * R * ldx;
        c.clip_rank = defer->clip_rank; c.row_until = defer->row_until;
        c.G = defer->G; c.groups = defer->groups; c.n_clips = defer->n_clips; c.n_active = defer->n_active; c.step = defer->step; c.T = defer->T;
        c.Rall = R; c.E = a.E;
        hipLaunchKernelGGL(dec_gru_step_cmb, dim3(H2 / 16, a2s_cdiv(nrows, 16)), dim3(64 * NW), dec_cmb_lds_bytes(nrows < 16 ? nrows : 16), st, g, c);
    } else
    if (dec_use_mid(nrows, H2)) { hipLaunchKernelGGL(dec_gru_mid, dim3(H2 / 16, a2s_cdiv(nrows, 64)), dim3(256), 0, st, g); __atomic_fetch_add(&g_dec_mid_launches, 1, __ATOMIC_RELAXED); }
    else hipLaunchKernelGGL(dec_gru_step, dim3(H2 / 16, a2s_cdiv(nrows, 16)), dim3(64 * NW), 0, st, g);
    A2S_CHECK_LAUNCH("dec_gru_step");
    DecOutArgs f;
    f.o = g.o; f.ldo = 2 * H2; f.ko = 2 * H2; f.out_w = a.out_w; f.out_b = a.out_b;
    f.tickets = flags + 16; f.logits = a.step_ws + 16 + FUSED_MAX_ROWS_CAP / 16; f.ldl = 16 * NTV;
    f.probs = a.probs; f.probs_bstride = a.probs_bstride; f.gt = a.gt; f.gt_bstride = a.gt_bstride;
    f.emb = a.emb; f.xnext = a.x + (long)so * R * ldx; f.ldx = ldx;
    f.drop = a.drop ? a.drop + (long)so * R * a.E : nullptr; f.inv_keep = a.inv_keep;
    f.argmax_out = a.argmax_out; f.am_bstride = a.am_bstride;
    f.eos_seen = a.eos_seen; f.lengths = a.lengths; f.n_done = a.n_done; f.steps_exec = a.steps_exec;
    f.t_base = t_base; f.row_until = a.n_active ? a.row_until : nullptr; f.skip = greedy ? flags : nullptr;
    f.hnew = g.hout; f.attn_w = a.attn_w; f.ld_aw = 2 * H2; f.attn_b = a.attn_b;
    f.q_next = last ? nullptr : a.q + (long)sv_next * R * a.H; f.H = a.H;
    f.rowmap = rowmap;
    f.n_clips = a.n_clips > 0 ? a.n_clips : R; f.R = nrows; f.V = a.V; f.E = a.E; f.t = t; f.teacher_force = tf; f.eos_id = a.eos_id;
    f.max_t = a.steps; f.H2 = H2;
    hipLaunchKernelGGL(dec_out_step, dim3(a2s_cdiv(nrows, 16), NVW + a2s_cdiv(a.H, 32)), dim3(64 * DEC_NW_OUT), 0, st, f);
    A2S_CHECK_LAUNCH("dec_out_step");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- backward: once per call, then per step
int a2s_note_step_fused_bwd_prepare(hipStream_t st, const a2s_note_dec_bwd_args& a) {
    const int H2 = 2 * a.H, kx = a.E + H2;
    float* wih_t = a.step_ws + FUSED_HEAD;
    float* whh_t = wih_t + (long)kx * 3 * H2;
    float* wh_t = whh_t + (long)H2 * 3 * H2;
    hipLaunchKernelGGL(transpose_ld, dim3(a2s_cdiv((long)3 * H2 * kx, 256)), dim3(256), 0, st, a.w_ih, (long)kx, wih_t, 3 * H2, kx);
    hipLaunchKernelGGL(transpose_ld, dim3(a2s_cdiv((long)3 * H2 * H2, 256)), dim3(256), 0, st, a.w_hh, (long)H2, whh_t, 3 * H2, H2);
    hipLaunchKernelGGL(transpose_ld, dim3(a2s_cdiv((long)a.H * H2, 256)), dim3(256), 0, st, a.attn_w, (long)2 * H2, wh_t, a.H, H2);
    A2S_CHECK_LAUNCH("transpose_ld");
    return A2S_OK;
}

int a2s_note_step_fused_bwd(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, const float* dh_in, float* dh_out, const a2s_attn_rows* rows,
                            int nrows, const int* rowmap) {
    const int H2 = 2 * a.H, ldx = a.E + H2, R = a.R;
    const float* dos = a.do_all + (long)s * R * 2 * H2;
    float* dgi = a.dgi_all + (long)s * R * 3 * H2;
    float* dgh = a.dgh_all + (long)s * R * 3 * H2;
    float* dxs = a.dx + (long)s * R * ldx;
    float* wih_t = a.step_ws + FUSED_HEAD;
    float* whh_t = wih_t + (long)ldx * 3 * H2;
    float* wh_t = whh_t + (long)H2 * 3 * H2;
    // GRU cell: dh = carry + dh_from_out;  hprev = h[s]; dh_out = dh * z
    DecBwdProdArgs p;
    p.dgi = dgi; p.dgh = dgh; p.wih_t = wih_t; p.whh_t = whh_t; p.dx = dxs; p.ldx = ldx; p.dh = dh_out;
    p.rowmap = rowmap;
    p.nxa = a2s_cdiv(ldx, 32); p.kx = ldx; p.R = nrows; p.H2 = H2;
    const bool mid = dec_use_mid(nrows, H2);
    int rc = a2s_gru_gates_bwd_impl(st, dh_in, H2, dos, 2 * H2, a.gates + (long)s * R * 4 * H2, a.h + (long)s * R * H2, H2,
                                    dgi, 3 * H2, dgh, 3 * H2, nullptr, 0, dh_out, H2, R, H2);
    if (rc) return rc;
    if (mid) { hipLaunchKernelGGL(dec_bwd_mid, dim3(p.nxa + a2s_cdiv(H2, 32), a2s_cdiv(nrows, 64)), dim3(256), 0, st, p); __atomic_fetch_add(&g_dec_mid_launches, 1, __ATOMIC_RELAXED); }
    else hipLaunchKernelGGL(dec_bwd_products, dim3(p.nxa + a2s_cdiv(H2, 32), a2s_cdiv(nrows, 16)), dim3(64 * DEC_NW_PROD), 0, st, p);
    A2S_CHECK_LAUNCH("dec_bwd_products");
    // attention: dctx = dx[:, E:] + do[:, 2H:]
    rc = a2s_attn_step_bwd_impl(st, a.keys, a.enc, a.q + (long)s * R * a.H, a.H, a.attn_v, a.attw + (long)s * R * a.T,
                                a.x + (long)s * R * ldx + a.E, ldx, dxs + a.E, ldx, dos + H2, 2 * H2,
                                a.dctx_all + (long)s * R * H2, H2, a.dq_all + (long)s * R * a.H, a.H,
                                a.ds_all + (long)s * R * a.T, R, a.T, a.H, a.attn_ws, rows);
    if (rc) return rc;
    if (mid) hipLaunchKernelGGL(dec_bwd_query_mid, dim3(a2s_cdiv(H2, 32), a2s_cdiv(nrows, 64)), dim3(256), 0, st, a.dq_all + (long)s * R * a.H, wh_t, dh_out, nrows, a.H, H2, rowmap);
    else hipLaunchKernelGGL(dec_bwd_query, dim3(a2s_cdiv(H2, 32), a2s_cdiv(nrows, 16)), dim3(64 * NW), 0, st, a.dq_all + (long)s * R * a.H, wh_t, dh_out, nrows, a.H, H2,
                            rowmap);
    A2S_CHECK_LAUNCH("dec_bwd_query");
    return A2S_OK;
}


// ------------------------------------------------------------------------------------------- the bulk calls' steps on the mid-size kernels (round 6)
// A decoder call over more rows than the few-row path takes (a2s_dec_fused_max_rows) keeps its launch-per-step loop (a2s_seq.hip / a2s_bwd.hip: query
// and output products, attention, epilogue as library-style launches), but its GRU cell -- two products + the gate kernel -- is ONE dec_gru_mid
// launch behind the attention, and the reverse loop's dx / dh products ONE dec_bwd_mid launch in front of it.
bool a2s_note_step_mid_ok(int H, int E, const void* const* ptrs, int nptrs) {
    if (!g_dec_mid || H % MID_KC || E % 16) return false;          // (K = 2H, 4H, 6H and H are walked in 64-wide chunks)
    for (int i = 0; i < nptrs; ++i) if (!aligned16(ptrs[i])) return false;
    return true;
}
// h' = GRU([token | ctx], h) for rows rowmap[0 .. nrows) (rowmap NULL: rows 0 .. nrows): state slot so, output row o[sv][:, :2H], saved gates;
// then logits of the rows and (sv_next >= 0) the next step's query into slot sv_next
int a2s_note_step_mid_gru(hipStream_t st, const a2s_note_dec_args& a, int si, int so, int sv, int sv_next, int nrows, const int* rowmap) {
    const int H2 = 2 * a.H, ldx = a.E + H2, R = a.R;
    if (nrows <= 0) return A2S_OK;
    DecGruArgs g;
    g.x = a.x + (long)si * R * ldx; g.ldx = ldx; g.kx = ldx;
    g.h = a.h + (long)si * R * H2;
    g.w_ih = a.w_ih; g.w_hh = a.w_hh; g.b_ih = a.b_ih; g.b_hh = a.b_hh;
    g.hout = a.h + (long)so * R * H2; g.o = a.o + (long)sv * R * 2 * H2; g.ldo = 2 * H2;
    g.save = a.gates ? a.gates + (long)sv * R * 4 * H2 : nullptr;
    g.n_done = nullptr; g.skip = nullptr;
    g.rowmap = rowmap;
    g.R = nrows; g.H2 = H2;
    hipLaunchKernelGGL(dec_gru_mid, dim3(H2 / 16, a2s_cdiv(nrows, 64)), dim3(256), 0, st, g);
    __atomic_fetch_add(&g_dec_mid_launches, 1, __ATOMIC_RELAXED);
    A2S_CHECK_LAUNCH("dec_gru_mid");
    DecOutqMidArgs f;
    f.o = g.o; f.ldo = 2 * H2; f.out_w = a.out_w; f.out_b = a.out_b; f.logits = a.logits; f.ldl = a.V; f.V = a.V;
    f.attn_w = a.attn_w; f.ld_aw = 2 * H2; f.attn_b = a.attn_b; f.q_next = sv_next >= 0 ? a.q + (long)sv_next * R * a.H : nullptr; f.H = a.H;
    f.rowmap = rowmap;
    f.nva = a2s_cdiv(a.V, 32); f.R = nrows; f.H2 = H2;
    hipLaunchKernelGGL(dec_outq_mid, dim3(f.nva + (f.q_next ? a2s_cdiv(a.H, 32) : 0), a2s_cdiv(nrows, 64)), dim3(256), 0, st, f);
    A2S_CHECK_LAUNCH("dec_outq_mid");
    return A2S_OK;
}
bool a2s_note_step_mid_bwd_ok(const a2s_note_dec_bwd_args& a) {
    const void* ptrs[] = {a.dgi_all, a.dgh_all, a.dx, a.dh, a.step_ws};
    return a.step_ws && a.step_ws_floats >= a2s_note_step_workspace_floats_impl(a.H, a.E) && a2s_note_step_mid_ok(a.H, a.E, ptrs, 5);
}
// dx[s] = dgi W_ih and dh_out += dgh W_hh for rows rowmap[0 .. nrows); the transposed weights come from a2s_note_step_fused_bwd_prepare
int a2s_note_step_mid_bwd(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, float* dh_out, int nrows, const int* rowmap) {
    const int H2 = 2 * a.H, ldx = a.E + H2, R = a.R;
    if (nrows <= 0) return A2S_OK;
    DecBwdProdArgs p;
    p.dgi = a.dgi_all + (long)s * R * 3 * H2; p.dgh = a.dgh_all + (long)s * R * 3 * H2;
    p.wih_t = a.step_ws + FUSED_HEAD; p.whh_t = p.wih_t + (long)ldx * 3 * H2;
    p.dx = a.dx + (long)s * R * ldx; p.ldx = ldx; p.dh = dh_out;
    p.rowmap = rowmap;
    p.nxa = a2s_cdiv(ldx, 32); p.kx = ldx; p.R = nrows; p.H2 = H2;
    hipLaunchKernelGGL(dec_bwd_mid, dim3(p.nxa + a2s_cdiv(H2, 32), a2s_cdiv(nrows, 64)), dim3(256), 0, st, p);
    __atomic_fetch_add(&g_dec_mid_launches, 1, __ATOMIC_RELAXED);
    A2S_CHECK_LAUNCH("dec_bwd_mid");
    return A2S_OK;
}
// dh_out += dq[s] W_h for rows rowmap[0 .. nrows), behind the attention sweep of the step
int a2s_note_step_mid_bwd_query(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, float* dh_out, int nrows, const int* rowmap) {
    const int H2 = 2 * a.H, ldx = a.E + H2, R = a.R;
    if (nrows <= 0) return A2S_OK;
    const float* wh_t = a.step_ws + FUSED_HEAD + (long)ldx * 3 * H2 + (long)H2 * 3 * H2;
    hipLaunchKernelGGL(dec_bwd_query_mid, dim3(a2s_cdiv(H2, 32), a2s_cdiv(nrows, 64)), dim3(256), 0, st, a.dq_all + (long)s * R * a.H, wh_t, dh_out, nrows, a.H, H2, rowmap);
    A2S_CHECK_LAUNCH("dec_bwd_query_mid");
    return A2S_OK;
}
