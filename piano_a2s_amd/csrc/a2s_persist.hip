// Persistent recurrences: ONE launch runs every step of an encoder-GRU direction (reference Encoder.gru, models.py:63-67,77; forward and BPTT).
//
// The launch-per-step kernels of a2s_seq.hip / a2s_bwd.hip (gru_step_fwd_fused, gru_bptt_step_fused) pay a kernel boundary, a cold fetch of
// their W_hh tile and of the previous state per step: 7.7 / 11.2 us per step against ~0.7 us of matrix work, 2 x 2 x 1201 dependent steps per
// optimizer step.  Here a workgroup keeps its W_hh tile IN REGISTERS for all T steps and the only per-step traffic is the state itself:
//
//   * decomposition as in gru_step_fwd_fused: workgroup (bx, by) owns hidden units [16 bx, 16 bx + 16) (their r, z, n gate columns) of the 16
//     rows of row block by; its 4 waves take the 16-wide k-steps round robin; the B fragments of those k-steps (48 VGPRs) are loaded once;
//   * the recurrence couples only the H / 16 workgroups of ONE row block: each publishes its 16 x 16 slice of the new state as 8-byte
//     {step tag, value} granules and every wave sweeps the granules of the k-columns it multiplies (laid out in the order of its MFMA
//     fragments: 2 KB contiguous per wave and k-step, 16-byte L1-bypassing loads) until all of them carry the current step's tag -- the
//     data is its own flag: no fence, no counter (MI355X_MICROARCH.md "Persistent kernels: hand-off price list"; cdna_hip_programming.md
//     section 6 Guideline 16, form R2).  Two granule buffers alternate: a workgroup can only be one step ahead of the slowest member of its
//     row block (it needs that member's granules of the previous step), so a buffer is never overwritten before every reader is done with it;
//   * where the granules travel.  Written with agent-scope (sc1, write-through) stores they are correct wherever the workgroups run, but
//     every sweep then goes to the memory side of the fabric: measured 8.2 us per forward step and 42 us per BPTT step at B = 256 -- slower
//     than a launch per step.  The observed dispatch deals consecutive workgroup ids round-robin to the 8 XCDs, so the block index is
//     permuted such that the 16 workgroups of a row block land on ONE XCD; each workgroup publishes its XCC_ID first, and only if all 16 of
//     a row block really agree do they switch to PLAIN granule stores, which stay in that XCD's L2 where the peers' L1-bypassing loads find
//     them.  Correctness never depends on the placement: a row block whose members disagree keeps the write-through stores;
//   * every spin is bounded: a workgroup that waits longer than SPIN_LIMIT polls raises the abort word, everybody leaves, and the final state
//     is poisoned with NaN (the training step then skips the update and the recipe counts a non-finite loss) -- a scheduling accident can never
//     hang the GPU.  The granule buffers and the abort word are zeroed by the launcher before every launch.
//
// Residency: grid = (H / 16) x ceil(B / 16) workgroups of 256 threads (B = 256: 256 workgroups, one per CU; two directions run side by
// side on two streams: 2 per CU), ~110 VGPRs, 2.3 KB of LDS -- far inside what a CU admits, so every workgroup of a launch is resident.
#include "a2s_common.h"
#include "../../include/a2s.h"

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define SPIN_LIMIT (1u << 21)          // polls of one wait before giving up (~ seconds)

__device__ __forceinline__ u64 granule_load(const u64* g) {
    return __hip_atomic_load((gu64*)(const_cast<u64*>(g)), RLX_AGENT);
}
__device__ __forceinline__ bool aborted(unsigned* flag) {
    return __hip_atomic_load((gu32*)(flag), RLX_AGENT) != 0;
}
__device__ __forceinline__ void raise_abort(unsigned* flag, unsigned code) {
    __hip_atomic_store((gu32*)(flag), code, RLX_AGENT);
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// granule store: plain (stays in this XCD's L2: only when every reader runs on this XCD) or agent scope (write-through)
__device__ __forceinline__ void granule_put(u64* g, unsigned tag, float v, bool same_xcd) {
    const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
    if (same_xcd) __hip_atomic_store((gu64*)(g), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store((gu64*)(g), x, RLX_AGENT);
}
// granule index of element (row r < 16, column k) of a row block's exchanged tile: the order of the consumers' MFMA A fragments --
// k-step u = k / 16 is one 2 KB run, inside it lane (lk = (k % 16) / 4, li = r) holds the 4 granules of columns 16 u + 4 lk .. + 3
__device__ __forceinline__ long granule_index(int r, int k) { return ((long)((k >> 4) * 64 + ((k & 15) >> 2) * 16 + r) << 2) + (k & 3); }

// One wave gathers its A fragments of the step from the row block's granule tile `tile` (bytes: 16 x K x 8): for its k-steps u = wave + 4 c
// each lane reads its 4 granules (32 contiguous bytes) with two L1-bypassing 16-byte loads, again and again until every granule carries
// `tag`.  Returns false on abort / timeout (wave-uniform).
template <int KS>
__device__ __forceinline__ bool wait_tile(const __amdgpu_buffer_rsrc_t rs, int wave, int lane, unsigned tag, unsigned* abort_flag, unsigned code) {
    constexpr int AUX = (int)(16u | 0x80000000u);        // sc1 + volatile: re-issued on every pass, served below the L1
    // Waiting is done on ONE k-step's granules (32 bytes per lane): the members of a row block publish at about the same time, and a full
    // pass is 16 x K x 8 bytes per workgroup -- polling with full passes made the L2 the bottleneck (BPTT: 96 KB per pass and workgroup).
    for (unsigned spins = 0;; ++spins) {
        const int off = ((wave + 4 * (KS - 1)) * 64 + lane) * 32;
        const u32x4_t x0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
        const u32x4_t x1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, AUX);
        if (__all((x0[1] == tag) & (x0[3] == tag) & (x1[1] == tag) & (x1[3] == tag))) return true;
        if ((spins & 63) == 63) {
            if (aborted(abort_flag)) return false;
            if (spins > SPIN_LIMIT) { raise_abort(abort_flag, code); return false; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// gather the wave's k-steps c in [C0, C0 + NC) (u = wave + 4 c), pass after pass until every granule carries `tag`
template <int C0, int NC>
__device__ __forceinline__ bool gather_tile(const __amdgpu_buffer_rsrc_t rs, int wave, int lane, unsigned tag, float (&a)[NC][4], unsigned* abort_flag, unsigned code) {
    constexpr int AUX = (int)(16u | 0x80000000u);
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int off = ((wave + 4 * (C0 + c)) * 64 + lane) * 32;
            const u32x4_t x0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
            const u32x4_t x1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, AUX);
            a[c][0] = __uint_as_float(x0[0]); a[c][1] = __uint_as_float(x0[2]); a[c][2] = __uint_as_float(x1[0]); a[c][3] = __uint_as_float(x1[2]);
            ok &= (x0[1] == tag) & (x0[3] == tag) & (x1[1] == tag) & (x1[3] == tag);
        }
        if (__all(ok)) return true;
        if ((spins & 63) == 63) {
            if (aborted(abort_flag)) return false;
            if (spins > SPIN_LIMIT) { raise_abort(abort_flag, code); return false; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const u64* tile, unsigned tile_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(tile), 0, tile_bytes, 0x00020000);
}

// Block index -> (unit slice bx, row block by).  With a multiple of 8 row blocks the permutation puts the 16 workgroups of a row block on
// ids that are congruent mod 8 (one XCD under the observed round-robin dispatch); otherwise the plain order.
__device__ __forceinline__ void persist_block(int nslices, int nrb, int& bx, int& by) {
    const int L = blockIdx.x;
    if (nslices == 16 && (nrb & 7) == 0) { const int xcd = L & 7, idx = L >> 3; by = xcd + 8 * (idx >> 4); bx = idx & 15; }
    else { by = L / nslices; bx = L % nslices; }
}
// Do all `n` (<= 64) workgroups of this row block run on one XCD?  Every workgroup publishes XCC_ID + 1 in ids[bx] (agent scope), wave 0
// waits for all of them.  The answer is the same in every member (same data).  Returns false when in doubt (timeout: the sweeps will notice).
__device__ __forceinline__ bool row_block_on_one_xcd(unsigned* ids, int bx, int n, int lane, unsigned* abort_flag) {
    const unsigned mine = (__builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) & 0xf) + 1;      // hwreg(HW_REG_XCC_ID), bits [3:0]
    if (lane == 0) __hip_atomic_store((gu32*)(ids + bx), mine, RLX_AGENT);
    for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
        const unsigned v = lane < n ? __hip_atomic_load((gu32*)(ids + lane), RLX_AGENT) : mine;
        if (__all(v != 0)) return __all(v == mine);
        if ((spins & 63) == 63 && aborted(abort_flag)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// ------------------------------------------------------------------------------------------- forward
struct GruPersistFwd {
    const float* gi; long gi_bstride, gi_tstride;       // (B, T, 3H) input projections of this direction
    const float* w_hh; const float* b_hh;
    float* out; long out_bstride, out_tstride;          // h_t -> out[b][t][col0 ..] (column offset already applied)
    float* save;                                        // (T, B, 4H) [r | z | n | gh_n] or null
    float* hn;                                          // (B, H) final state
    u64* hx;                                            // 2 x RB tiles of 16 x H granules (granule_index order), zeroed before the launch
    unsigned* abort_flag;                               // zeroed before the launch
    unsigned* xcc;                                      // RB x 16 words, zeroed before the launch
    int B, T, reverse, nrb;
    unsigned* latch; unsigned dbg;                      // process-wide abort latch (or null); PERSIST_DBG_* bits
};

template <int H>
__global__ __launch_bounds__(256, 2) void gru_seq_fwd_persist(GruPersistFwd a) {
    constexpr int KS = H / 64;                          // 16-wide k-steps per wave
    __shared__ f32x4 part[3 * 3 * 64];
    int bx, by;
    persist_block(H / 16, a.nrb, bx, by);
    const int j0 = bx * 16, row0 = by * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int j = j0 + li, R = a.B;
    if ((a.dbg & PERSIST_DBG_INJECT_ABORT) && blockIdx.x == 0 && tid == 0) raise_abort(a.abort_flag, 99u);      // test hook: as if a wait had timed out
    const bool same_xcd = wave == 0 && row_block_on_one_xcd(a.xcc + by * 16, bx, H / 16, lane, a.abort_flag) && !(a.dbg & PERSIST_DBG_FORCE_AGENT);
    if (tid == 0 && same_xcd) atomicAdd(a.abort_flag + 1, 1u);          // diagnostic: workgroups that exchange through their XCD's L2
    // B fragments of this wave's k-steps: W_hh rows (g H + j0 + li), columns 16 (wave + 4 c) + 4 lk .. + 3
    f32x4 bw[3][KS];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < KS; ++c)
            bw[g][c] = *reinterpret_cast<const f32x4*>(a.w_hh + ((long)g * H + j0 + li) * H + 16 * (wave + 4 * c) + 4 * lk);
    float br = 0.f, bz = 0.f, bn = 0.f;
    if (wave == 0) { br = a.b_hh[j]; bz = a.b_hh[H + j]; bn = a.b_hh[2 * H + j]; }
    constexpr long TILE = 16L * H;                       // granules per row-block tile
    const long par = (long)a.nrb * TILE;                 // granules per buffer
    float hp[4] = {0.f, 0.f, 0.f, 0.f};                  // wave 0: h_{s-1}[row0 + 4 lk + r][j] (its own outputs of the previous step)
    bool dead = false;
    for (int s = 0; s < a.T; ++s) {
        const int t = a.reverse ? a.T - 1 - s : s;
        float gir[4], giz[4], gin[4];
        if (wave == 0) {                                 // epilogue operands: independent of the recurrence, in flight during the wait
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* gp = a.gi + (long)min(row0 + lk * 4 + r, R - 1) * a.gi_bstride + (long)t * a.gi_tstride;
                gir[r] = gp[j]; giz[r] = gp[H + j]; gin[r] = gp[2 * H + j];
            }
        }
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {                                     // h_0 = 0: the first step has no product
            float av[KS][4];
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(a.hx + (long)((s - 1) & 1) * par + by * TILE, (unsigned)(TILE * 8));
            if (!dead && !(wait_tile<KS>(rs, wave, lane, (unsigned)s, a.abort_flag, 1u) && gather_tile<0, KS>(rs, wave, lane, (unsigned)s, av, a.abort_flag, 1u))) dead = true;
            if (!dead) {
#pragma unroll
                for (int c = 0; c < KS; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][q], bw[g][c][q], acc[g], 0, 0, 0);
            }
            if (wave > 0) {
#pragma unroll
                for (int g = 0; g < 3; ++g) part[((wave - 1) * 3 + g) * 64 + lane] = acc[g];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int w = 0; w < 3; ++w) {
                        const f32x4 o = part[(w * 3 + g) * 64 + lane];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[g][r] += o[r];
                    }
            }
            __syncthreads();                             // `part` is free for the next step's partial tiles
        }
        if (wave == 0) {
            u64* gout = a.hx + (long)(s & 1) * par + by * TILE;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + lk * 4 + r;
                const float ghn = acc[2][r] + bn;
                const float rg = fast_sigmoid(gir[r] + acc[0][r] + br);
                const float zg = fast_sigmoid(giz[r] + acc[1][r] + bz);
                const float ng = fast_tanh(gin[r] + rg * ghn);
                const float hnew = (1.f - zg) * ng + zg * hp[r];
                hp[r] = hnew;
                if (s + 1 < a.T) granule_put(gout + granule_index(lk * 4 + r, j), (unsigned)(s + 1), hnew, same_xcd);      // (padding rows too: their readers wait for them)
                if (row < R) {
                    a.out[(long)row * a.out_bstride + (long)t * a.out_tstride + j] = hnew;
                    if (a.save) {
                        float* sv = a.save + ((long)t * R + row) * 4 * H;
                        sv[j] = rg; sv[H + j] = zg; sv[2 * H + j] = ng; sv[3 * H + j] = ghn;
                    }
                }
            }
        }
    }
    if (wave == 0) {
        const bool bad = dead || aborted(a.abort_flag);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + lk * 4 + r;
            if (row < R) a.hn[(long)row * H + j] = bad ? __builtin_nanf("") : hp[r];
        }
        if (bad && lane == 0 && a.latch) atomicOr(a.latch, 1u);
    }
}

// ------------------------------------------------------------------------------------------- BPTT
// Reverse of the above for one direction (see a2s_gru_seq_bwd_impl for the argument meaning).  Per processed step s (T-1 .. 0, time index t):
//   dh_s = carry + dout_t, carry = dh_{s+1} z_{s+1} + dgh_{s+1} W_hh       (carry of the last step = dhn)
//   gate backward -> dgi_t (for the deferred input-projection gradients), dgh_s (exchanged: the next product's A operand; also written,
//   shifted by one step, for the deferred dW_hh = dgh_shift^T out), carry part dh_s z_s kept in registers.
// The exchanged tile is 16 rows x 3H: workgroup (bx, by) produces the columns of its 16 units (r, z, n) and multiplies all of them with its
// 16 columns of W_hh (B fragments of W_hh^T rows j0 + li in registers: 3H / 64 k-steps per wave).
struct GruPersistBwd {
    const float* dout; long do_bstride, do_tstride;     // gradient wrt this direction's outputs (column offset applied)
    const float* out; long out_bstride, out_tstride;    // forward outputs (h_prev source), column offset applied
    const float* gates;                                 // (T, B, 4H)
    const float* w_hh_t;                                // (H, 3H): W_hh transposed
    const float* dhn;                                   // (B, H) or null
    float* dgi_all; float* dgh_shift; float* dgh_first; // (B, T, 3H), (B, T, 3H), (B, 3H)
    u64* gx;                                            // 2 x RB tiles of 16 x 3H granules (granule_index order), zeroed before the launch
    unsigned* abort_flag;
    unsigned* xcc;                                      // RB x 16 words, zeroed before the launch
    int B, T, reverse, nrb;
    unsigned* latch; unsigned dbg;
};

template <int H>
__global__ __launch_bounds__(256, 2) void gru_seq_bwd_persist(GruPersistBwd a) {
    constexpr int KS = 3 * H / 64;                      // 16-wide k-steps per wave over K = 3H
    __shared__ f32x4 part[3 * 64];
    int bx, by;
    persist_block(H / 16, a.nrb, bx, by);
    const int j0 = bx * 16, row0 = by * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int j = j0 + li, R = a.B, T = a.T;
    if ((a.dbg & PERSIST_DBG_INJECT_ABORT) && blockIdx.x == 0 && tid == 0) raise_abort(a.abort_flag, 99u);      // test hook: as if a wait had timed out
    const bool same_xcd = wave == 0 && row_block_on_one_xcd(a.xcc + by * 16, bx, H / 16, lane, a.abort_flag) && !(a.dbg & PERSIST_DBG_FORCE_AGENT);
    if (tid == 0 && same_xcd) atomicAdd(a.abort_flag + 1, 1u);          // diagnostic: workgroups that exchange through their XCD's L2
    f32x4 bw[KS];
#pragma unroll
    for (int c = 0; c < KS; ++c) bw[c] = *reinterpret_cast<const f32x4*>(a.w_hh_t + (long)(j0 + li) * 3 * H + 16 * (wave + 4 * c) + 4 * lk);
    constexpr long TILE = 16L * 3 * H;
    const long par = (long)a.nrb * TILE;
    float carry[4];                                      // wave 0: dh_{s+1} z_{s+1} (direct path) for (row0 + 4 lk + r, j)
#pragma unroll
    for (int r = 0; r < 4; ++r) carry[r] = (wave == 0 && a.dhn) ? a.dhn[(long)min(row0 + lk * 4 + r, R - 1) * H + j] : 0.f;
    bool dead = false;
    for (int s = T - 1; s >= 0; --s) {
        const int t = a.reverse ? T - 1 - s : s;
        const int tp = a.reverse ? t + 1 : t - 1;        // time index whose output was h_prev of this step (invalid when s == 0)
        float dov[4], rg[4], zg[4], ng[4], ghn[4], hp[4];
        if (wave == 0) {                                 // epilogue operands: in flight during the wait
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(row0 + lk * 4 + r, R - 1);
                dov[r] = a.dout[(long)row * a.do_bstride + (long)t * a.do_tstride + j];
                const float* sv = a.gates + ((long)t * R + row) * 4 * H;
                rg[r] = sv[j]; zg[r] = sv[H + j]; ng[r] = sv[2 * H + j]; ghn[r] = sv[3 * H + j];
                hp[r] = s > 0 ? a.out[(long)row * a.out_bstride + (long)tp * a.out_tstride + j] : 0.f;
            }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s < T - 1) {                                 // the recurrent part of the carry: dgh_{s+1} W_hh
            const __amdgpu_buffer_rsrc_t rs = tile_rsrc(a.gx + (long)((s + 1) & 1) * par + by * TILE, (unsigned)(TILE * 8));
            const unsigned tag = (unsigned)(T - 1 - s);
            if (!dead && !wait_tile<KS>(rs, wave, lane, tag, a.abort_flag, 2u)) dead = true;
            f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};           // two chains: a dependent 16x16x4 waits longer than an independent one
            constexpr int NG = KS / 3;                   // the k-steps in three groups: a third of the operand registers live at a time
#define A2S_BPTT_GROUP(G)                                                                                                   \
            if (!dead) {                                                                                                    \
                float av[NG][4];                                                                                            \
                if (!gather_tile<G * NG, NG>(rs, wave, lane, tag, av, a.abort_flag, 2u)) dead = true;                       \
                else {                                                                                                      \
                    _Pragma("unroll") for (int c = 0; c < NG; ++c) {                                                        \
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][0], bw[G * NG + c][0], acc, 0, 0, 0);              \
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][1], bw[G * NG + c][1], acc2, 0, 0, 0);            \
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][2], bw[G * NG + c][2], acc, 0, 0, 0);              \
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][3], bw[G * NG + c][3], acc2, 0, 0, 0);            \
                    }                                                                                                       \
                }                                                                                                           \
            }
            A2S_BPTT_GROUP(0)
            A2S_BPTT_GROUP(1)
            A2S_BPTT_GROUP(2)
#undef A2S_BPTT_GROUP
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
            if (wave > 0) part[(wave - 1) * 64 + lane] = acc;
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += part[lane][r] + part[64 + lane][r] + part[128 + lane][r];
            }
            __syncthreads();
        }
        if (wave == 0) {
            u64* gout = a.gx + (long)(s & 1) * par + by * TILE;
            const unsigned tag = (unsigned)(T - s);      // readers of step s - 1 expect T - 1 - (s - 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + lk * 4 + r;
                const float dh = carry[r] + acc[r] + dov[r];
                const float dn = dh * (1.f - zg[r]) * (1.f - ng[r] * ng[r]);
                const float dz = dh * (hp[r] - ng[r]) * zg[r] * (1.f - zg[r]);
                const float dr = dn * ghn[r] * rg[r] * (1.f - rg[r]);
                const float dnr = dn * rg[r];
                carry[r] = dh * zg[r];
                if (s > 0) {
                    const int rr = lk * 4 + r;
                    granule_put(gout + granule_index(rr, j), tag, dr, same_xcd);
                    granule_put(gout + granule_index(rr, H + j), tag, dz, same_xcd);
                    granule_put(gout + granule_index(rr, 2 * H + j), tag, dnr, same_xcd);
                }
                if (row < R) {
                    float* gi = a.dgi_all + ((long)row * T + t) * 3 * H;
                    gi[j] = dr; gi[H + j] = dz; gi[2 * H + j] = dn;
                    if (s > 0) { float* g2 = a.dgh_shift + ((long)row * T + tp) * 3 * H; g2[j] = dr; g2[H + j] = dz; g2[2 * H + j] = dnr; }
                    else { float* g1 = a.dgh_first + (long)row * 3 * H; g1[j] = dr; g1[H + j] = dz; g1[2 * H + j] = dnr; }
                }
            }
        }
    }
    if (wave == 0 && (dead || aborted(a.abort_flag))) {   // poison what the deferred weight-gradient products read
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + lk * 4 + r;
            if (row < R) a.dgh_first[(long)row * 3 * H + j] = __builtin_nanf("");
        }
        if (lane == 0 && a.latch) atomicOr(a.latch, 2u);
    }
}

// ------------------------------------------------------------------------------------------- launchers
static int g_gru_persist = -1;                          // A2S_GRU_PERSIST=0 / a2s_debug_set("gru_persist", 0): the launch-per-step kernels
void a2s_gru_persist_set(int v) { g_gru_persist = v ? 1 : 0; }
int a2s_gru_persist_enabled(void) {
    if (g_gru_persist < 0) { const char* e = getenv("A2S_GRU_PERSIST"); g_gru_persist = (e && e[0] == '0') ? 0 : 1; }
    return g_gru_persist;
}
// workspace bytes the persistent recurrences need (granule buffers + abort word); 0: shape not supported
size_t a2s_gru_persist_ws_bytes(int B, int H, int bwd) {
    if (H != 256 || B < 1) return 0;
    const size_t rb = (size_t)(B + 15) / 16;
    return 256 + ((rb * 16 * sizeof(unsigned) + 255) & ~(size_t)255) + sizeof(u64) * 2 * rb * 16 * (size_t)(bwd ? 3 * H : H);
}
// ---- device geometry, abort latch, test hooks (shared with csrc/a2s_dec_persist.hip)
// The persistent kernels are ordinary launches whose workgroups wait for each other, so every workgroup of a launch must be resident at
// once.  What the chip offers is asked of the runtime, once per device: compute units visible to this process (a CPX / DPX partition or a CU
// mask reports fewer than 256), XCDs, and how many workgroups of each kernel one CU admits.  A resident foreign process cannot be seen from
// here; that case ends in a bounded-wait abort, which sets the process-wide latch, and the host then switches the persistent paths off
// (piano_a2s_amd/hip.py check_persist_abort).
static unsigned* g_abort_latch = nullptr;
static unsigned g_persist_dbg = 0;
unsigned* a2s_persist_latch_ptr(void) { return g_abort_latch; }
void a2s_persist_latch_set(void* p) { g_abort_latch = reinterpret_cast<unsigned*>(p); }
unsigned a2s_persist_dbg(void) { return g_persist_dbg; }
void a2s_persist_dbg_set(unsigned bit, int on) { g_persist_dbg = on ? (g_persist_dbg | bit) : (g_persist_dbg & ~bit); }
int a2s_persist_dbg_get(unsigned bit) { return (g_persist_dbg & bit) ? 1 : 0; }
a2s_device_geom a2s_device_geometry(void) {
    static a2s_device_geom cache[16];
    static bool have[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return a2s_device_geom{0, 0};
    if (!have[dev]) {
        int cus = 0, xccs = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess) xccs = 0;
        (void)hipGetLastError();
        cache[dev] = a2s_device_geom{cus, xccs};
        have[dev] = true;
    }
    return cache[dev];
}
// every workgroup of the launch must be resident (they wait for each other) -- and so must those of the OTHER direction of the layer, which
// runs beside it: 2 x grid <= compute units x workgroups of this kernel a CU admits (measured: 2 on gfx950, 256 CUs -> grid <= 256)
template <typename K>
static int persist_blocks_per_cu(K kernel) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    return n;
}
static int g_gru_persist_alone = 0;      // a2s_debug_set("gru_persist_alone", 1): the host runs the directions of a layer one after the other
void a2s_gru_persist_alone_set(int v) { g_gru_persist_alone = v ? 1 : 0; }
int a2s_gru_persist_alone(void) { return g_gru_persist_alone; }
static bool persist_fits(int B, int H, bool bwd) {
    static int occ[2] = {-1, -1};
    if (occ[bwd] < 0) occ[bwd] = bwd ? persist_blocks_per_cu(gru_seq_bwd_persist<256>) : persist_blocks_per_cu(gru_seq_fwd_persist<256>);
    const a2s_device_geom g = a2s_device_geometry();
    return (g_gru_persist_alone ? 1L : 2L) * (H / 16) * ((B + 15) / 16) <= (long)g.cus * occ[bwd];
}

bool a2s_gru_seq_fwd_persist_ok(const float* w_hh, const float* gi, int B, int T, int H, float* ws, size_t ws_bytes) {
    return a2s_gru_persist_enabled() && H == 256 && T >= 2 && persist_fits(B, H, false) && ws && ((uintptr_t)ws % 256 == 0) && ws_bytes >= a2s_gru_persist_ws_bytes(B, H, 0) &&
           ((uintptr_t)w_hh % 16 == 0) && gi;
}
int a2s_gru_seq_fwd_persist_impl(hipStream_t st, const float* gi_all, long gi_bstride, long gi_tstride, const float* w_hh, const float* b_hh, float* out,
                                 long out_bstride, long out_tstride, float* save, float* hn, int B, int T, int H, int reverse, float* ws, size_t ws_bytes) {
    const size_t need = a2s_gru_persist_ws_bytes(B, H, 0);
    A2S_REQUIRE(need && ws_bytes >= need, "gru_seq_fwd_persist: workspace too small");
    hipError_t e = hipMemsetAsync(ws, 0, need, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "gru_seq_fwd_persist memset: %s", hipGetErrorString(e));
    const int nrb = a2s_cdiv(B, 16);
    char* base = reinterpret_cast<char*>(ws);
    const size_t xcc_bytes = ((size_t)nrb * 16 * sizeof(unsigned) + 255) & ~(size_t)255;
    GruPersistFwd a{gi_all, gi_bstride, gi_tstride, w_hh, b_hh, out, out_bstride, out_tstride, save, hn,
                    reinterpret_cast<u64*>(base + 256 + xcc_bytes), reinterpret_cast<unsigned*>(base), reinterpret_cast<unsigned*>(base + 256), B, T, reverse, nrb,
                    g_abort_latch, g_persist_dbg};
    hipLaunchKernelGGL(gru_seq_fwd_persist<256>, dim3((H / 16) * nrb), dim3(256), 0, st, a);
    A2S_CHECK_LAUNCH("gru_seq_fwd_persist");
    return A2S_OK;
}

bool a2s_gru_seq_bwd_persist_ok(int B, int T, int H, float* ws, size_t ws_bytes, size_t ws_used) {
    return a2s_gru_persist_enabled() && H == 256 && T >= 2 && persist_fits(B, H, true) && ws && ((uintptr_t)ws % 256 == 0) && ws_used % 256 == 0 &&
           ws_bytes >= ws_used + a2s_gru_persist_ws_bytes(B, H, 1);
}
// ws_off: bytes at the start of the workspace the caller keeps (W_hh^T)
int a2s_gru_seq_bwd_persist_impl(hipStream_t st, const float* dout, long do_bstride, long do_tstride, const float* out, long out_bstride, long out_tstride,
                                 const float* gates, const float* w_hh_t, const float* dhn, float* dgi_all, float* dgh_shift, float* dgh_first, int B, int T,
                                 int H, int reverse, float* ws, size_t ws_off, size_t ws_bytes) {
    const size_t need = a2s_gru_persist_ws_bytes(B, H, 1);
    A2S_REQUIRE(need && ws_bytes >= ws_off + need, "gru_seq_bwd_persist: workspace too small");
    char* base = reinterpret_cast<char*>(ws) + ws_off;
    hipError_t e = hipMemsetAsync(base, 0, need, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "gru_seq_bwd_persist memset: %s", hipGetErrorString(e));
    const int nrb = a2s_cdiv(B, 16);
    const size_t xcc_bytes = ((size_t)nrb * 16 * sizeof(unsigned) + 255) & ~(size_t)255;
    GruPersistBwd a{dout, do_bstride, do_tstride, out, out_bstride, out_tstride, gates, w_hh_t, dhn, dgi_all, dgh_shift, dgh_first,
                    reinterpret_cast<u64*>(base + 256 + xcc_bytes), reinterpret_cast<unsigned*>(base), reinterpret_cast<unsigned*>(base + 256), B, T, reverse, nrb,
                    g_abort_latch, g_persist_dbg};
    hipLaunchKernelGGL(gru_seq_bwd_persist<256>, dim3((H / 16) * nrb), dim3(256), 0, st, a);
    A2S_CHECK_LAUNCH("gru_seq_bwd_persist");
    return A2S_OK;
}
