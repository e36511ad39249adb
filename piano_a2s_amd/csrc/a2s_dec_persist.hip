// Persistent note decoder for FEW clips: ONE launch runs every step of a NoteDecoder.decode_notes call (reference models.py:366-420) for up
// to 8 clips x up to 5 fused bars, one clip per XCD.
//
// Why.  The clips that hold a full-length bar decode as a group of their own (engine.Engine.forward, clip groups): 398 dependent steps per
// segment on 8-40 rows.  As launches (attention split + combine + dec_gru_step + dec_out_step) a step costs 4 dependent kernels that each
// re-fetch what they need -- the clip's 3.7 MB of keys / encoder outputs, 7.5 MB of weights -- through a memory system the OTHER clip group
// saturates: 125-210 us per step measured, ~150-230 ms per optimizer step, as long as everything else the decoder does.  Here nothing is
// re-fetched:
//   * clip c lives on XCD c: the 32 workgroups (one per CU) of that XCD hold its key image and encoder outputs IN LDS (38 frames each,
//     117 KB) for the whole call, and the GRU / output / query weights as MFMA B fragments IN REGISTERS (every XCD its own copy);
//   * the only per-step traffic is the state of the clip's <= 5 rows, handed between the workgroups of the XCD as 8-byte {step tag, value}
//     granules through that XCD's L2 (plain stores + L1-bypassing loads; see a2s_persist.hip for the protocol, its placement check and the
//     agent-scope fallback that keeps the result independent of where the workgroups run);
//   * a step is four hand-offs on the critical path:  query -> [attention partials of 32 frame chunks] -> [softmax combine, by column
//     slice] -> [GRU cell, by hidden-unit tile] -> [next query | logits -> epilogue], every stage spread over the 32 workgroups.
// The per-step saved tensors (h, x, q, o, gates, attention weights) are written exactly as the launch-per-step path writes them, so the
// backward pass and the deferred weight-gradient products do not care which path ran.  Attention weights are stored as raw scores plus
// the row's (max, 1 / sum) and normalised by a small kernel after the loop.
//
// Every wait is bounded (SPIN_LIMIT polls, then the abort word is raised, everybody leaves and the outputs are poisoned with NaN).
#include "a2s_common.h"
#include "../../include/a2s.h"

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define SPIN_LIMIT (1u << 23)          // failed polls of one wave over the whole call before it gives up
#define NWG 32                      // workgroups per clip (= CUs of an XCD)
#define NTH 512                     // threads per workgroup
#define NWV 8                       // waves per workgroup (the weights: ~140 registers per thread; 16 waves at 128 registers spilled)
#define CHF 38                      // frames per workgroup (32 x 38 >= 1201)
#define MAXR 5                      // rows (fused bars) per clip
#define HH 256                      // hidden_size
#define H2 512
#define EE 16
#define KX (EE + H2)                // GRU input width
#define VV 173
#define VP 176                      // logits row in granules (11 tiles of 16)
#define AUX_SC1V ((int)(16u | 0x80000000u))
#define XLD (EE + 2 * H2 + 4)          // floats per staged row (+4: rows land 4 banks apart)
#define DP_PART_FLOATS (((NWV - 1) * 4 * 80) > (NTH > 512 ? MAXR * H2 : 0) ? ((NWV - 1) * 4 * 80) : (MAXR * H2))
#define DPB_PART_FLOATS (((NWV - 1) * 2 * 80) > (MAXR * HH) ? ((NWV - 1) * 2 * 80) : (MAXR * HH))

__device__ __forceinline__ bool dp_aborted(unsigned* flag) { return __hip_atomic_load((gu32*)(flag), RLX_AGENT) != 0; }
__device__ __forceinline__ void dp_raise(unsigned* flag, unsigned code) { __hip_atomic_store((gu32*)(flag), code, RLX_AGENT); }
__device__ __forceinline__ void put(u64* g, unsigned tag, float v, bool same_xcd) {
    const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
    if (same_xcd) __hip_atomic_store((gu64*)(g), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store((gu64*)(g), x, RLX_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// bounded-wait bookkeeping of one wave
struct Spin {
    unsigned* abort_flag; unsigned code; unsigned n; bool dead;
    __device__ __forceinline__ bool again() {          // call after a failed pass; false: give up (wave-uniform)
        if ((++n & 63) == 0) {
            if (dp_aborted(abort_flag)) { dead = true; return false; }
            if (n > SPIN_LIMIT) { dp_raise(abort_flag, code); dead = true; return false; }
        }
        __builtin_amdgcn_s_sleep(1);
        return true;
    }
};

// A fragments of one 16-wide k-step from a row-major granule matrix: lane (li = row, lk) takes the 4 granules of columns k0 + 4 lk .. + 3 of
// row li (rows >= nrows: zeros).  `rs` covers the matrix, ld = granules per row.  Returns whether every granule read carried `tag`.
__device__ __forceinline__ bool frag_load(const __amdgpu_buffer_rsrc_t rs, int ld, int nrows, int li, int lk, int k0, unsigned tag, float (&a)[4]) {
    if (li >= nrows) { a[0] = a[1] = a[2] = a[3] = 0.f; return true; }
    const int off = (li * ld + k0 + 4 * lk) * 8;
    const u32x4_t x0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX_SC1V);
    const u32x4_t x1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, AUX_SC1V);
    a[0] = __uint_as_float(x0[0]); a[1] = __uint_as_float(x0[2]); a[2] = __uint_as_float(x1[0]); a[3] = __uint_as_float(x1[2]);
    return (x0[1] == tag) & (x0[3] == tag) & (x1[1] == tag) & (x1[3] == tag);
}


// All threads of the workgroup copy `ncols` (even) columns of the ON rows of a row-major granule matrix (`ld` granules per row, region `rs`,
// first column c0 of the matrix) into the LDS image dst[row * dld + dcol0 + col]: every thread issues ALL of its 16-byte loads (two granules
// each) before it looks at a tag, and the whole batch is repeated until every granule of every thread's share carries `tag` -- one memory
// round trip per hand-off instead of one per k-step.  Rows that are off get zeros.  MAXP: loads per thread (>= ceil(nrows * ncols / 2 / NTH)).
template <int MAXP>
__device__ __forceinline__ void stage_rows(const __amdgpu_buffer_rsrc_t rs, int ld, int c0, int nrows, int ncols, int onmask, unsigned tag, float* dst, int dld,
                                           int dcol0, int tid, Spin& spin) {
    const int half = ncols >> 1, total = nrows * half;
    for (;;) {
        u32x4_t x[MAXP];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int p = tid + NTH * i;
            if (p < total) {
                const int row = p / half, c2 = p - row * half;
                if ((onmask >> row) & 1) {
                    x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (row * ld + c0 + 2 * c2) * 8, 0, AUX_SC1V);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int p = tid + NTH * i;
            if (p < total) {
                const int row = p / half, c2 = p - row * half;
                float v0 = 0.f, v1 = 0.f;
                if ((onmask >> row) & 1) { ok &= (x[i][1] == tag) & (x[i][3] == tag); v0 = __uint_as_float(x[i][0]); v1 = __uint_as_float(x[i][2]); }
                dst[row * dld + dcol0 + 2 * c2] = v0; dst[row * dld + dcol0 + 2 * c2 + 1] = v1;
            }
        }
        if (__all(ok)) break;
        if (!spin.again()) break;
    }
}
// A fragment of a 16-wide k-step from an LDS image: lane (li = row, lk) reads 4 consecutive floats at column k0 + 4 lk of row li (rows >= nrows: zeros)
__device__ __forceinline__ f32x4 frag_lds(const float* img, int ld, int nrows, int li, int lk, int k0) {
    return li < nrows ? *reinterpret_cast<const f32x4*>(img + li * ld + k0 + 4 * lk) : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// Sum of the NWV waves' partial tiles in wave 0, fixed order (wave 1, 2, ...), ONE barrier pair.  Only the rows of the clip matter (<= 5 of
// the tile's 16: accumulator element r of lanes lk = 0, and r = 0 of lanes lk = 1), so a wave leaves 5 values per (tile, column) in LDS
// and wave 0 adds them up.  scratch: (NWV - 1) x NT x 5 x 16 floats.
template <int NT>
__device__ __forceinline__ void reduce_rows(f32x4 (&acc)[NT], float* scratch, int wave, int li, int lk) {
    if (wave > 0 && lk < 2) {
        float* p = scratch + (long)(wave - 1) * NT * 80;
#pragma unroll
        for (int g = 0; g < NT; ++g) {
            if (lk == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p[(g * 5 + r) * 16 + li] = acc[g][r];
            } else p[(g * 5 + 4) * 16 + li] = acc[g][0];
        }
    }
    __syncthreads();
    if (wave == 0 && lk < 2) {
        for (int wv = 1; wv < NWV; ++wv) {
            const float* p = scratch + (long)(wv - 1) * NT * 80;
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                if (lk == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[g][r] += p[(g * 5 + r) * 16 + li];
                } else acc[g][0] += p[(g * 5 + 4) * 16 + li];
            }
        }
    }
    __syncthreads();
}

struct DecPersistFwd {
    const float* attn_w; const float* attn_b; const float* attn_v; const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;
    const float* out_w; const float* out_b; const float* emb;
    const float* keys; const float* enc;               // (C, T, 256) key image, (C, T, 512)
    float* h; float* x; float* q; float* o; float* gates; float* attw;     // saved per step, R rows each (see a2s_note_dec_args)
    float* stats;                                       // (steps, R, 2): softmax max and 1 / sum of every (step, row)
    float* probs; long probs_bstride;
    const long long* gt; long gt_bstride;
    const int* flags;                                   // device, per step: bit j = the rows of group j are teacher-forced
    const uint8_t* drop; float inv_keep;
    int* argmax_out; long am_bstride;
    int* eos_seen; long long* lengths; int* n_done; int* steps_exec;
    const int* row_until;                               // device, R ints or null
    int* eos_first;                                     // greedy: step of every row's first <eos> (-1: none yet); R ints
    unsigned* stop;                                     // greedy: per clip, first step NOT to run (0: keep going); 8 words, zeroed
    u64* xg;                                            // granule workspace (zeroed): C regions of DP_REGION granules
    unsigned* abort_flag; unsigned* xcc;                // zeroed; xcc: C x 32 words
    int C, NR, R, T, steps, eos_id;
    unsigned* latch; unsigned dbg;                      // process-wide abort latch (or null); PERSIST_DBG_* bits
};
// granule region of one clip (offsets in granules)
#define G_Q 0                                   // [2][MAXR][HH]
#define G_TOK (G_Q + 2 * MAXR * HH)             // [2][MAXR][EE]
#define G_CTX (G_TOK + 2 * MAXR * EE)           // [2][MAXR][H2]
#define G_H (G_CTX + 2 * MAXR * H2)             // [2][MAXR][H2]
#define G_LG (G_H + 2 * MAXR * H2)              // [MAXR][VP]
#define G_PM (G_LG + MAXR * VP)                 // [NWG][MAXR][2]
#define G_PC (G_PM + NWG * MAXR * 2)            // [NWG][MAXR][H2]
#define DP_REGION (G_PC + NWG * MAXR * H2)

__global__ __launch_bounds__(NTH, 2) void dec_persist_fwd(DecPersistFwd a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Kc = sm;                                     // [CHF][HH]
    float* Ec = Kc + CHF * HH;                          // [CHF][H2]
    float* Eq = Ec + CHF * H2;                          // [MAXR][HH]  exp(2 q)
    float* sc = Eq + MAXR * HH;                         // [MAXR][64]  scores / chunk weights
    float* part = sc + MAXR * 64;                       // (NWV - 1) * 4 * 80 floats: cross-wave reduction; also the second half of the partial contexts
    float* red = part + DP_PART_FLOATS;                 // 64 floats: small broadcasts
    float* XH = red + 64;                               // [MAXR][XLD]: staged rows [token | context | state] of the GRU / output products
    const int L = blockIdx.x, c = L & 7, w = L >> 3;            // clip (XCD under the observed dispatch), member
    if (c >= a.C) return;
    const int tid0 = threadIdx.x;
    const int tid = tid0, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int NR = a.NR, R = a.R, C = a.C, T = a.T;
    u64* G = a.xg + (long)c * DP_REGION;
    // ---- placement check (as a2s_persist.hip): plain granule stores only if all 32 members of this clip run on one XCD
    bool same_xcd;
    {
        const unsigned mine = (__builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) & 0xf) + 1;
        unsigned* ids = a.xcc + c * NWG;
        if (tid == 0) __hip_atomic_store((gu32*)(ids + w), mine, RLX_AGENT);
        int ok = 0;
        if (wave == 0) {
            for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
                const unsigned v = lane < NWG ? __hip_atomic_load((gu32*)(ids + lane), RLX_AGENT) : mine;
                if (__all(v != 0)) { ok = __all(v == mine) ? 1 : 0; break; }
                if ((spins & 63) == 63 && dp_aborted(a.abort_flag)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) red[0] = (float)ok;
        }
        __syncthreads();
        same_xcd = red[0] != 0.f && !(a.dbg & PERSIST_DBG_FORCE_AGENT);
        __syncthreads();
        if (tid == 0 && same_xcd) atomicAdd(a.abort_flag + 1, 1u);
        if ((a.dbg & PERSIST_DBG_INJECT_ABORT) && L == 0 && tid == 0) dp_raise(a.abort_flag, 99u);      // test hook: as if a wait had timed out
    }
    Spin spin{a.abort_flag, 16u + (unsigned)w, 0u, false};
    // ---- resident operands
    const int t0 = w * CHF, nf = max(0, min(T, t0 + CHF) - t0);          // this workgroup's frames
    for (int i = tid; i < CHF * HH / 4; i += NTH) {
        const int f = i / (HH / 4);
        reinterpret_cast<f32x4*>(Kc)[i] = f < nf ? *reinterpret_cast<const f32x4*>(a.keys + ((long)c * T + t0 + f) * HH + 4 * (i % (HH / 4))) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid; i < CHF * H2 / 4; i += NTH) {
        const int f = i / (H2 / 4);
        reinterpret_cast<f32x4*>(Ec)[i] = f < nf ? *reinterpret_cast<const f32x4*>(a.enc + ((long)c * T + t0 + f) * H2 + 4 * (i % (H2 / 4))) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 v4 = {a.attn_v[lane * 4], a.attn_v[lane * 4 + 1], a.attn_v[lane * 4 + 2], a.attn_v[lane * 4 + 3]};
    // GRU tile w: hidden units 16 w .. + 15; k-steps u = wave + 8 cc (u < 65): u = 0 token, 1..32 context, 33..64 previous state
    constexpr int GKS = (65 + NWV - 1) / NWV;
    f32x4 wg_[3][GKS];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int cc = 0; cc < GKS; ++cc) {
            const int u = wave + NWV * cc;
            const long row = (long)g * H2 + 16 * w + li;
            wg_[g][cc] = u < 33 ? *reinterpret_cast<const f32x4*>(a.w_ih + row * KX + 16 * u + 4 * lk)
                                : (u < 65 ? *reinterpret_cast<const f32x4*>(a.w_hh + row * H2 + 16 * (u - 33) + 4 * lk) : (f32x4){0.f, 0.f, 0.f, 0.f});
        }
    // second role: w < 11 vocabulary tile w (K = 1024: 4 k-steps per wave), 11 <= w < 27 query tile w - 11 (K = 512: 32 / NWV per wave), w == 27 epilogue
    const bool role_out = w < 11, role_q = w >= 11 && w < 27, role_epi = w == 27;
    constexpr int OKS = 64 / NWV;
    f32x4 wo_[OKS];
#pragma unroll
    for (int cc = 0; cc < OKS; ++cc) {
        const int u = wave + NWV * cc;
        if (role_out) wo_[cc] = *reinterpret_cast<const f32x4*>(a.out_w + (long)min(16 * w + li, VV - 1) * (2 * H2) + 16 * u + 4 * lk);
        else if (role_q && u < 32) wo_[cc] = *reinterpret_cast<const f32x4*>(a.attn_w + (long)(16 * (w - 11) + li) * (2 * H2) + 16 * u + 4 * lk);
        else wo_[cc] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float gb_r = 0.f, gb_z = 0.f, gb_in = 0.f, gb_hn = 0.f, ob = 0.f;
    if (wave == 0) {
        const int j = 16 * w + li;
        gb_r = a.b_ih[j] + a.b_hh[j]; gb_z = a.b_ih[H2 + j] + a.b_hh[H2 + j]; gb_in = a.b_ih[2 * H2 + j]; gb_hn = a.b_hh[2 * H2 + j];
        if (role_out) ob = a.out_b[min(16 * w + li, VV - 1)];
        else if (role_q) ob = a.attn_b[16 * (w - 11) + li];
    }
    float hreg[4] = {0.f, 0.f, 0.f, 0.f};                // wave 0: h_s[row 4 lk + r][16 w + li] (this lane's own outputs of the previous step)
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (lk * 4 + r < NR) hreg[r] = a.h[(long)((lk * 4 + r) * C + c) * H2 + 16 * w + li];
    }
    __syncthreads();
    const unsigned gbytes = (unsigned)(DP_REGION * 8);
    const __amdgpu_buffer_rsrc_t rsG = rsrc_of(G, gbytes);
    // ---- prologue: publish h[0] (from the caller's state slot 0), the <sos> embedding (x[0][:, :E]) and compute q[0]
    if (wave == 0 && li < 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = lk * 4 + r;
            if (j < NR) put(G + G_H + (long)(0 * MAXR + j) * H2 + 16 * w + li, 1u, a.h[(long)(j * C + c) * H2 + 16 * w + li], same_xcd);
        }
    }
    if (w == 27 && tid < NR * EE) {
        const int j = tid / EE, e = tid % EE;
        put(G + G_TOK + (long)(0 * MAXR + j) * EE + e, 1u, a.x[(long)(j * C + c) * KX + e], same_xcd);
    }
    // -------------------------------------------------------------------------------------------------------------------- step loop
    // (step index s; "slot" = s & 1 for the double-buffered granule arrays; tags: consumer step + 1)
    for (int s = -1; s < a.steps && !spin.dead; ++s) {
        // per-iteration copies of the thread coordinates the optimiser cannot see through: without them every address that only depends on
        // the lane is hoisted out of this (very long) loop body -- a hundred VGPRs of loop-invariant offsets, and the kernel spilled
        int tid_ = tid0;
        asm volatile("" : "+v"(tid_));
        const int tid = tid_, lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const unsigned tag = (unsigned)(s + 1);          // tag of what step s consumes
        int onmask = 0;                                   // bit j: row j of this clip decodes at step s
#pragma unroll
        for (int j = 0; j < MAXR; ++j) onmask |= (j < NR && s >= 0 && (!a.row_until || s < a.row_until[j * C + c])) ? (1 << j) : 0;
        const int n_on = __builtin_popcount(onmask);
#define ON(j) ((onmask >> (j)) & 1)
        if (s >= 0 && n_on > 0) {
            // ================================================================ attention over this workgroup's frames
            {   // q rows -> E_q in LDS (every thread: one float4 of one row, 5 x 64 = 320 threads)
                const int j = tid >> 6;
                if (j < NR) {
                    float qv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (ON(j)) {
                        const int off = (G_Q + ((s & 1) * MAXR + j) * HH + 4 * lane) * 8;
                        for (;;) {
                            const u32x4_t x0 = __builtin_amdgcn_raw_buffer_load_b128(rsG, off, 0, AUX_SC1V);
                            const u32x4_t x1 = __builtin_amdgcn_raw_buffer_load_b128(rsG, off + 16, 0, AUX_SC1V);
                            qv[0] = __uint_as_float(x0[0]); qv[1] = __uint_as_float(x0[2]); qv[2] = __uint_as_float(x1[0]); qv[3] = __uint_as_float(x1[2]);
                            if (__all((x0[1] == tag) & (x0[3] == tag) & (x1[1] == tag) & (x1[3] == tag))) break;
                            if (!spin.again()) break;
                        }
                    }
                    f32x4 e4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) e4[i] = exp2x_clamped(qv[i]);
                    reinterpret_cast<f32x4*>(Eq + j * HH)[lane] = e4;
                }
            }
            __syncthreads();
            // scores: wave handles frames wave, wave + 8, ...; one wave-wide reduction per (frame, row)
            for (int f = wave; f < nf; f += NWV) {
                const f32x4 k4 = reinterpret_cast<const f32x4*>(Kc + f * HH)[lane];
#pragma unroll
                for (int j = 0; j < MAXR; ++j) {
                    if (!ON(j)) continue;
                    const f32x4 e4 = reinterpret_cast<const f32x4*>(Eq + j * HH)[lane];
                    float sj = v4[0] * tanh_ek(k4[0], e4[0]) + v4[1] * tanh_ek(k4[1], e4[1]) + v4[2] * tanh_ek(k4[2], e4[2]) + v4[3] * tanh_ek(k4[3], e4[3]);
                    sj = wave_sum_lane63(sj);
                    if (lane == 63) sc[j * 64 + f] = sj;
                }
            }
            __syncthreads();
            // chunk softmax: wave j handles row j (nf <= 38 < 64 frames: one per lane); raw scores out, (m, l) published
            if (wave < NR && ON(wave)) {
                const int j = wave;
                const float sv = lane < nf ? sc[j * 64 + lane] : -INFINITY;
                const float m = nf > 0 ? wave_max(sv) : -INFINITY;
                const float p = lane < nf ? __expf(sv - m) : 0.f;
                const float l = wave_sum(p);
                sc[j * 64 + lane] = p;
                if (lane < nf && a.attw) a.attw[((long)s * R + j * C + c) * T + t0 + lane] = sv;       // raw score; normalised after the loop
                if (lane == 0) {
                    put(G + G_PM + (long)(w * MAXR + j) * 2 + 0, tag, nf > 0 ? m : -3.0e38f, same_xcd);
                    put(G + G_PM + (long)(w * MAXR + j) * 2 + 1, tag, l, same_xcd);
                }
            }
            __syncthreads();
            // partial contexts: thread (column d = tid & 511, frame residue tid >> 9 of NTH / 512); with 1024 threads the second half hands its
            // sums over through LDS
            {
                constexpr int NHF = NTH / H2;
                const int d = tid & (H2 - 1), hf = tid >> 9;
                float accc[MAXR];
#pragma unroll
                for (int j = 0; j < MAXR; ++j) accc[j] = 0.f;
                for (int f = hf; f < nf; f += NHF) {
                    const float e = Ec[f * H2 + d];
#pragma unroll
                    for (int j = 0; j < MAXR; ++j) if (ON(j)) accc[j] = fmaf(sc[j * 64 + f], e, accc[j]);
                }
                if (NHF > 1) {
                    if (hf == 1) {
#pragma unroll
                        for (int j = 0; j < MAXR; ++j) part[j * H2 + d] = accc[j];
                    }
                    __syncthreads();
                    if (hf == 0) {
#pragma unroll
                        for (int j = 0; j < MAXR; ++j) accc[j] += part[j * H2 + d];
                    }
                    __syncthreads();
                }
                if (hf == 0) {
#pragma unroll
                    for (int j = 0; j < MAXR; ++j) if (ON(j)) put(G + G_PC + (long)(w * MAXR + j) * H2 + d, tag, accc[j], same_xcd);
                }
            }
            // ================================================================ softmax combine, column slice 16 w .. + 15 of every row
            // thread (col = tid / 32, chunk g = tid % 32): half-waves reduce over the 32 chunks
            if (tid < 512) {
                const int col = tid >> 5, g = tid & 31;
                float mg[MAXR], lg[MAXR], pc[MAXR];
                for (;;) {                                       // all rows' granules in flight together
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < MAXR; ++j) {
                        mg[j] = 0.f; lg[j] = 0.f; pc[j] = 0.f;
                        if (!ON(j)) continue;
                        const u64* pm = G + G_PM + (long)(g * MAXR + j) * 2;
                        const u64 xm = __hip_atomic_load((gu64*)(const_cast<u64*>(pm)), RLX_AGENT), xl = __hip_atomic_load((gu64*)(const_cast<u64*>(pm + 1)), RLX_AGENT);
                        const u64 xc = __hip_atomic_load((gu64*)(G + G_PC + (long)(g * MAXR + j) * H2 + 16 * w + col), RLX_AGENT);
                        mg[j] = __uint_as_float((unsigned)xm); lg[j] = __uint_as_float((unsigned)xl); pc[j] = __uint_as_float((unsigned)xc);
                        ok &= ((unsigned)(xm >> 32) == tag) & ((unsigned)(xl >> 32) == tag) & ((unsigned)(xc >> 32) == tag);
                    }
                    if (__all(ok)) break;
                    if (!spin.again()) break;
                }
#pragma unroll
                for (int j = 0; j < MAXR; ++j) {
                    if (!ON(j)) continue;
                    // max over the 32 chunks of this half-wave
                    float M = mg[j];
                    M = fmaxf(M, dpp_take<A2S_DPP_QUAD_1032>(M, M));
                    M = fmaxf(M, dpp_take<A2S_DPP_QUAD_2301>(M, M));
                    M = fmaxf(M, dpp_take<A2S_DPP_ROW_HALF_MIRROR>(M, M));
                    M = fmaxf(M, dpp_take<A2S_DPP_ROW_MIRROR>(M, M));
                    M = fmaxf(M, dpp_take<A2S_DPP_ROW_BCAST15, 0xA, false>(M, M));         // lanes 16..31 / 48..63: the half's max
                    M = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, M), 31));   // (both halves hold the same 32 chunks' statistics)
                    const float eg = __expf(mg[j] - M);
                    float Lsum = lg[j] * eg, cs = pc[j] * eg;
                    Lsum += dpp_take<A2S_DPP_QUAD_1032>(0.f, Lsum); cs += dpp_take<A2S_DPP_QUAD_1032>(0.f, cs);
                    Lsum += dpp_take<A2S_DPP_QUAD_2301>(0.f, Lsum); cs += dpp_take<A2S_DPP_QUAD_2301>(0.f, cs);
                    Lsum += dpp_take<A2S_DPP_ROW_HALF_MIRROR>(0.f, Lsum); cs += dpp_take<A2S_DPP_ROW_HALF_MIRROR>(0.f, cs);
                    Lsum += dpp_take<A2S_DPP_ROW_MIRROR>(0.f, Lsum); cs += dpp_take<A2S_DPP_ROW_MIRROR>(0.f, cs);
                    Lsum += dpp_take<A2S_DPP_ROW_BCAST15, 0xA>(0.f, Lsum); cs += dpp_take<A2S_DPP_ROW_BCAST15, 0xA>(0.f, cs);
                    if ((lane & 31) == 31) {                       // lanes 31 and 63: the sums over the half-wave's 32 chunks
                        const float inv = 1.f / Lsum, cv = cs * inv;
                        const long grow = (long)j * C + c;
                        const int d = 16 * w + col;
                        put(G + G_CTX + (long)((s & 1) * MAXR + j) * H2 + d, tag, cv, same_xcd);
                        a.x[((long)s * R + grow) * KX + EE + d] = cv;
                        a.o[((long)s * R + grow) * (2 * H2) + H2 + d] = cv;
                        if (w == 0 && col == 0) { a.stats[((long)s * R + grow) * 2] = M; a.stats[((long)s * R + grow) * 2 + 1] = inv; }
                    }
                }
            }
            // ================================================================ GRU cell, hidden units 16 w .. + 15 of every row
            {
                f32x4 acc[4];                                    // r, z, n (input part), n (state part)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // [token | context | previous state] rows of the clip -> LDS, one batched pass each (three sources)
                stage_rows<1>(rsrc_of(G + G_TOK + (long)(s & 1) * MAXR * EE, MAXR * EE * 8), EE, 0, NR, EE, onmask, tag, XH, XLD, 0, tid, spin);
                if (!a.gt) {                                       // greedy: did the epilogue of the previous step end the decode?  (uniform over the clip)
                    if (tid == 0) { const unsigned st = __hip_atomic_load((gu32*)(a.stop + c), RLX_AGENT); red[1] = (st != 0 && st <= (unsigned)s) ? 1.f : 0.f; }
                    __syncthreads();
                    const bool stop_now = red[1] != 0.f;
                    __syncthreads();
                    if (stop_now) break;
                }
                stage_rows<3>(rsrc_of(G + G_CTX + (long)(s & 1) * MAXR * H2, MAXR * H2 * 8), H2, 0, NR, H2, onmask, tag, XH, XLD, EE, tid, spin);
                stage_rows<3>(rsrc_of(G + G_H + (long)(s & 1) * MAXR * H2, MAXR * H2 * 8), H2, 0, NR, H2, onmask, tag, XH, XLD, EE + H2, tid, spin);
                __syncthreads();
#pragma unroll
                for (int cc = 0; cc < GKS; ++cc) {
                    const int u = wave + NWV * cc;
                    if (u >= 65) break;
                    const f32x4 av = frag_lds(XH, XLD, NR, li, lk, 16 * u);
                    if (u < 33) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wg_[0][cc][i], acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wg_[1][cc][i], acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wg_[2][cc][i], acc[2], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wg_[0][cc][i], acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wg_[1][cc][i], acc[1], 0, 0, 0);
                            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wg_[2][cc][i], acc[3], 0, 0, 0);
                        }
                    }
                }
                reduce_rows<4>(acc, part, wave, li, lk);
                if (wave == 0) {
                    const int jn = 16 * w + li;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = lk * 4 + r;                  // row of the clip
                        if (j >= NR || !ON(j)) continue;
                        const long grow = (long)j * C + c;
                        const float hp = hreg[r];
                        const float ghn = acc[3][r] + gb_hn;
                        const float rg = fast_sigmoid(acc[0][r] + gb_r);
                        const float zg = fast_sigmoid(acc[1][r] + gb_z);
                        const float ng = fast_tanh(acc[2][r] + gb_in + rg * ghn);
                        const float hn = (1.f - zg) * ng + zg * hp;
                        hreg[r] = hn;
                        put(G + G_H + (long)(((s + 1) & 1) * MAXR + j) * H2 + jn, tag + 1, hn, same_xcd);
                        a.h[((long)(s + 1) * R + grow) * H2 + jn] = hn;
                        a.o[((long)s * R + grow) * (2 * H2) + jn] = hn;
                        if (a.gates) { float* sv = a.gates + ((long)s * R + grow) * 4 * H2; sv[jn] = rg; sv[H2 + jn] = zg; sv[2 * H2 + jn] = ng; sv[3 * H2 + jn] = ghn; }
                    }
                }
            }
        }
        // rows that are switched off at step s but were on before keep their last state in the granule slot the next step reads: nobody reads it
        // ==================================================================== second role: next query | logits | epilogue
        if (role_q || role_out) {
            // A = h[s + 1] rows (K = 512) [| ctx rows of step s (K = 512) for the vocabulary tiles]
            const bool have = s < 0 || n_on > 0;
            if (have && !(role_out && s < 0)) {
                f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
                const unsigned htag = (unsigned)(s + 2);
                // the new state rows h[s + 1] -> LDS behind the context rows staged for the GRU ([token | context | state] image: the output
                // projection's operand [state | context] is read from there; before the first step only the state exists)
                const int hmask = s < 0 ? ((1 << NR) - 1) : onmask;
                __syncthreads();                                   // (the GRU's readers of the image are done)
                stage_rows<3>(rsrc_of(G + G_H + (long)((s + 1) & 1) * MAXR * H2, MAXR * H2 * 8), H2, 0, NR, H2, hmask, htag, XH, XLD, EE + H2, tid, spin);
                __syncthreads();
#pragma unroll
                for (int cc = 0; cc < OKS; ++cc) {
                    const int u = wave + NWV * cc;
                    if (u >= (role_out ? 64 : 32)) break;
                    const f32x4 av = frag_lds(XH, XLD, NR, li, lk, u < 32 ? EE + H2 + 16 * u : EE + 16 * (u - 32));
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wo_[cc][i], acc[0], 0, 0, 0);
                }
                reduce_rows<1>(acc, part, wave, li, lk);
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = lk * 4 + r;
                        if (j >= NR) continue;
                        if (s >= 0 && !ON(j)) continue;
                        const float val = acc[0][r] + ob;
                        if (role_q) {
                            const int n = 16 * (w - 11) + li;
                            if (s + 1 < a.steps) {
                                put(G + G_Q + (long)(((s + 1) & 1) * MAXR + j) * HH + n, tag + 1, val, same_xcd);
                                a.q[((long)(s + 1) * R + (long)j * C + c) * HH + n] = val;
                            }
                        } else {
                            put(G + G_LG + (long)j * VP + 16 * w + li, tag, val, same_xcd);
                        }
                    }
                }
            }
        } else if (role_epi && s >= 0 && n_on > 0) {
            // one wave per row: log-softmax, argmax (lowest index on ties), token choice, its embedding -> next step's input
            const int j = wave;
            if (j < NR && ON(j)) {
                const long grow = (long)j * C + c;
                float v[3];
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int n = lane + 64 * k;
                        if (n < VV) { const u64 xv = __hip_atomic_load((gu64*)(G + G_LG + (long)j * VP + n), RLX_AGENT); v[k] = __uint_as_float((unsigned)xv); ok &= (unsigned)(xv >> 32) == tag; }
                        else v[k] = -INFINITY;
                    }
                    if (__all(ok)) break;
                    if (!spin.again()) break;
                }
                float m = -INFINITY; int mi = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 3; ++k) if (v[k] > m) { m = v[k]; mi = lane + 64 * k; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(mi, o, 64);
                    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
                }
                float sacc = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) if (lane + 64 * k < VV) sacc += expf(v[k] - m);
                sacc = wave_sum(sacc);
                const float lse = m + logf(sacc);
                float* pr = a.probs + grow * a.probs_bstride + (long)s * VV;
#pragma unroll
                for (int k = 0; k < 3; ++k) if (lane + 64 * k < VV) pr[lane + 64 * k] = v[k] - lse;
                const long long gtok = a.gt ? a.gt[grow * a.gt_bstride + s] : -1;
                const int tf = a.flags ? ((a.flags[s] >> j) & 1) : 0;
                const int next_id = (a.gt && tf) ? (int)gtok : mi;
                if (lane == 0) {
                    if (a.argmax_out) a.argmax_out[grow * a.am_bstride + s] = mi;
                    const bool hit = a.gt ? (gtok == a.eos_id) : (mi == a.eos_id);
                    if (hit) {
                        if (!a.eos_seen[grow]) { a.eos_seen[grow] = 1; if (a.eos_first) a.eos_first[grow] = s; atomicAdd(a.n_done, 1); }
                        a.lengths[grow] = s + 1;
                    }
                    if (!a.gt) {
                        // greedy decoding ends (for everybody) once every row of every clip has shown <eos> (reference models.py:389).  The clips
                        // run on their own: whoever sees the count complete tells its clip's workgroups, through a word stored BEFORE this
                        // row's token granules -- the GRU stage of the next step reads it after those granules.  A clip that notices late runs
                        // a few steps too many; dec_persist_greedy_fixup erases them (the true end is a function of the first-<eos> steps).
                        if (__hip_atomic_load((gu32*)(reinterpret_cast<unsigned*>(a.n_done)), RLX_AGENT) >= (unsigned)R) {
                            __hip_atomic_store((gu32*)(a.stop + c), (unsigned)(s + 1), RLX_AGENT);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                }
                if (lane < EE) {                        // (also after the last step: the state slot `steps` exists, as in the launch-per-step path)
                    float e = a.emb[(long)next_id * EE + lane];
                    if (a.drop) e = a.drop[((long)(s + 1) * R + grow) * EE + lane] ? e * a.inv_keep : 0.f;
                    if (s + 1 < a.steps) put(G + G_TOK + (long)(((s + 1) & 1) * MAXR + j) * EE + lane, tag + 1, e, same_xcd);
                    a.x[((long)(s + 1) * R + grow) * KX + lane] = e;
                }
            }
        }
    }
#undef ON
    if (spin.dead || dp_aborted(a.abort_flag)) {
        // Poison what the caller READS: the first log-probability row of every row of this clip (the loss gathers its target there -- a bar
        // has at least one real target --, so the loss becomes non-finite and the update is skipped; greedy decoding: the host reads the
        // abort word with the step count and raises), and the state after the last step.  Nothing hangs.
        for (int i = tid; i < NR * VV; i += NTH) a.probs[(long)((i / VV) * C + c) * a.probs_bstride + (i % VV)] = __builtin_nanf("");
        if (tid < NR) a.h[((long)a.steps * R + (long)tid * C + c) * H2 + 16 * w] = __builtin_nanf("");
        if (tid == 0 && a.latch) atomicOr(a.latch, 4u);
    }
}

// attention weights of the persistent path: raw scores -> exp(s - max) / sum with the (max, 1 / sum) the combine stage stored
__global__ __launch_bounds__(256) void dec_persist_attw_normalise(float* __restrict__ attw, const float* __restrict__ stats, const int* __restrict__ row_until,
                                                                  int steps, int R, int T) {
    const long row = blockIdx.x;                         // (step, row)
    const int s = (int)(row / R), r = (int)(row % R);
    float* aw = attw + row * T;
    if (row_until && s >= row_until[r]) { for (int t = threadIdx.x; t < T; t += 256) aw[t] = 0.f; return; }
    const float m = stats[row * 2], inv = stats[row * 2 + 1];
    for (int t = threadIdx.x; t < T; t += 256) aw[t] = __expf(aw[t] - m) * inv;
}

// =========================================================================================== backward
// Reverse of the above for one call (see a2s_note_decoder_bwd_impl for the per-step math).  Same placement: clip c on XCD c, its keys and
// encoder outputs in LDS, the transposed weight tiles (W_ih^T, W_hh^T, W_h^T: made once per call by a2s_note_step_fused_bwd_prepare) in
// registers.  A step is four hand-offs:
//   gate backward of the workgroup's 16 hidden units (dgi, dgh)  ->  [allgather]  ->  dx = dgi W_ih (context columns 16 w ..) and the
//   recurrent carry dgh W_hh (state columns 16 w ..)  ->  [dctx rows]  ->  attention backward over the workgroup's frames (ds, partial dq)
//   ->  [32 partials]  ->  dq, column slice 8 w ..  ->  [dq rows]  ->  carry += dq W_h.
// The token columns of dx (only the deferred embedding gradient reads them) are one GEMM over all steps after the loop.
struct DecPersistBwd {
    const float* attn_v; const float* wih_t; const float* whh_t; const float* wh_t;      // (KX, 3 H2), (H2, 3 H2), (H2, HH)
    const float* keys; const float* enc;
    const float* h; const float* x; const float* q; const float* gates; const float* attw; const float* do_all;
    float* dgi_all; float* dgh_all; float* dq_all; float* ds_all; float* dctx_all; float* dx; float* dh;
    const int* row_until;
    u64* xg; unsigned* abort_flag; unsigned* xcc;
    int C, NR, R, T, steps;
    unsigned* latch; unsigned dbg;
};
#define B_DG 0                                  // [MAXR][6 * H2]   dgi | dgh rows
#define B_DC (B_DG + MAXR * 6 * H2)             // [MAXR][H2]       dctx rows
#define B_DQP (B_DC + MAXR * H2)                // [NWG][MAXR][HH]  partial dq
#define B_DQ (B_DQP + NWG * MAXR * HH)          // [MAXR][HH]
#define DPB_REGION (B_DQ + MAXR * HH)

__global__ __launch_bounds__(NTH, 2) void dec_persist_bwd(DecPersistBwd a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Kc = sm;                                     // [CHF][HH]
    float* Ec = Kc + CHF * HH;                          // [CHF][H2]
    float* Eq = Ec + CHF * H2;                          // [MAXR][HH]
    float* dcx = Eq + MAXR * HH;                        // [MAXR][H2]  dctx rows
    float* dsv = dcx + MAXR * H2;                       // [MAXR][64]  ds of this workgroup's frames
    float* aw = dsv + MAXR * 64;                        // [MAXR][64]  saved attention weights of these frames
    float* part = aw + MAXR * 64;                       // DP_PART_FLOATS (>= MAXR * HH for the dq halves)
    float* red = part + DPB_PART_FLOATS;                // 64
    const int L = blockIdx.x, c = L & 7, w = L >> 3;
    if (c >= a.C) return;
    const int tid0 = threadIdx.x;
    const int wave = tid0 >> 6;
    const int NR = a.NR, R = a.R, C = a.C, T = a.T, n = a.steps;
    u64* G = a.xg + (long)c * DPB_REGION;
    bool same_xcd;
    {
        const int lane = tid0 & 63;
        const unsigned mine = (__builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) & 0xf) + 1;
        unsigned* ids = a.xcc + c * NWG;
        if (tid0 == 0) __hip_atomic_store((gu32*)(ids + w), mine, RLX_AGENT);
        if (wave == 0) {
            int ok = 0;
            for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
                const unsigned v = lane < NWG ? __hip_atomic_load((gu32*)(ids + lane), RLX_AGENT) : mine;
                if (__all(v != 0)) { ok = __all(v == mine) ? 1 : 0; break; }
                if ((spins & 63) == 63 && dp_aborted(a.abort_flag)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (lane == 0) red[0] = (float)ok;
        }
        __syncthreads();
        same_xcd = red[0] != 0.f && !(a.dbg & PERSIST_DBG_FORCE_AGENT);
        __syncthreads();
        if (tid0 == 0 && same_xcd) atomicAdd(a.abort_flag + 1, 1u);
        if ((a.dbg & PERSIST_DBG_INJECT_ABORT) && L == 0 && tid0 == 0) dp_raise(a.abort_flag, 99u);
    }
    Spin spin{a.abort_flag, 64u + (unsigned)w, 0u, false};
    const int t0 = w * CHF, nf = max(0, min(T, t0 + CHF) - t0);
    for (int i = tid0; i < CHF * HH / 4; i += NTH) {
        const int f = i / (HH / 4);
        reinterpret_cast<f32x4*>(Kc)[i] = f < nf ? *reinterpret_cast<const f32x4*>(a.keys + ((long)c * T + t0 + f) * HH + 4 * (i % (HH / 4))) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid0; i < CHF * H2 / 4; i += NTH) {
        const int f = i / (H2 / 4);
        reinterpret_cast<f32x4*>(Ec)[i] = f < nf ? *reinterpret_cast<const f32x4*>(a.enc + ((long)c * T + t0 + f) * H2 + 4 * (i % (H2 / 4))) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // B fragments: k-steps u = wave + 8 cc over K = 3 H2 (gate columns) for the two product tiles, over K = HH for the query product
    constexpr int PKS = 3 * H2 / 16 / NWV;              // 12
    constexpr int QKS = HH / 16 / NWV;                  // 2
    f32x4 wx_[PKS], wh_[PKS], wq_[QKS];
    {
        const int lane = tid0 & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int cc = 0; cc < PKS; ++cc) {
            const int u = wave + NWV * cc;
            wx_[cc] = *reinterpret_cast<const f32x4*>(a.wih_t + (long)(EE + 16 * w + li) * (3 * H2) + 16 * u + 4 * lk);
            wh_[cc] = *reinterpret_cast<const f32x4*>(a.whh_t + (long)(16 * w + li) * (3 * H2) + 16 * u + 4 * lk);
        }
#pragma unroll
        for (int cc = 0; cc < QKS; ++cc) wq_[cc] = *reinterpret_cast<const f32x4*>(a.wh_t + (long)(16 * w + li) * HH + 16 * (wave + NWV * cc) + 4 * lk);
    }
    float dhc[4] = {0.f, 0.f, 0.f, 0.f};                 // wave 0: carry dh[row 4 lk + r][16 w + li]
    __syncthreads();
    for (int s = n - 1; s >= 0 && !spin.dead; --s) {
        int tid_ = tid0;
        asm volatile("" : "+v"(tid_));
        const int tid = tid_, lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const unsigned tag = (unsigned)(n - s);
        int onmask = 0;
#pragma unroll
        for (int j = 0; j < MAXR; ++j) onmask |= (j < NR && (!a.row_until || s < a.row_until[j * C + c])) ? (1 << j) : 0;
#define ON(j) ((onmask >> (j)) & 1)
        if (onmask == 0) continue;
        // ================================================================ (1) gate backward of units 16 w .. + 15
        float dhz[4] = {0.f, 0.f, 0.f, 0.f}, doc[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave == 0) {
            const int jn = 16 * w + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = lk * 4 + r;
                if (j >= NR || !ON(j)) continue;
                const long grow = (long)j * C + c, sr = (long)s * R + grow;
                const float* sv = a.gates + sr * 4 * H2;
                const float rg = sv[jn], zg = sv[H2 + jn], ng = sv[2 * H2 + jn], ghn = sv[3 * H2 + jn];
                const float hp = a.h[sr * H2 + jn];
                const float dh = dhc[r] + a.do_all[sr * (2 * H2) + jn];
                doc[r] = a.do_all[sr * (2 * H2) + H2 + jn];                  // the output projection's share of dctx, column 16 w + li
                const float dn = dh * (1.f - zg) * (1.f - ng * ng);
                const float dz = dh * (hp - ng) * zg * (1.f - zg);
                const float dr = dn * ghn * rg * (1.f - rg);
                const float dnr = dn * rg;
                dhz[r] = dh * zg;
                u64* gr = G + B_DG + (long)j * 6 * H2;
                put(gr + jn, tag, dr, same_xcd); put(gr + H2 + jn, tag, dz, same_xcd); put(gr + 2 * H2 + jn, tag, dn, same_xcd);
                put(gr + 3 * H2 + jn, tag, dr, same_xcd); put(gr + 4 * H2 + jn, tag, dz, same_xcd); put(gr + 5 * H2 + jn, tag, dnr, same_xcd);
                float* gi = a.dgi_all + sr * 3 * H2; float* gh = a.dgh_all + sr * 3 * H2;
                gi[jn] = dr; gi[H2 + jn] = dz; gi[2 * H2 + jn] = dn;
                gh[jn] = dr; gh[H2 + jn] = dz; gh[2 * H2 + jn] = dnr;
            }
        }
        // saved operands of the attention backward that do not depend on the recurrence: E_q, the frames' attention weights
        {
            const int j = tid >> 6;                           // waves 0 .. NR-1: one row each
            if (j < NR && ON(j)) {
                const long sr = (long)s * R + (long)j * C + c;
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(a.q + sr * HH + 4 * lane);
                f32x4 e4;
#pragma unroll
                for (int i = 0; i < 4; ++i) e4[i] = exp2x_clamped(q4[i]);
                reinterpret_cast<f32x4*>(Eq + j * HH)[lane] = e4;
                aw[j * 64 + lane] = lane < nf ? a.attw[sr * T + t0 + lane] : 0.f;
            }
        }
        // ================================================================ (2) dx context columns 16 w .. and carry columns 16 w ..
        {
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(G + B_DG, MAXR * 6 * H2 * 8);
#pragma unroll
            for (int cc = 0; cc < PKS; ++cc) {
                const int u = wave + NWV * cc;
                float ai[4], ah[4];
                for (;;) {
                    bool ok = frag_load(rs, 6 * H2, NR, li, lk, 16 * u, tag, ai);
                    ok &= frag_load(rs, 6 * H2, NR, li, lk, 3 * H2 + 16 * u, tag, ah);
                    if (li < NR && !ON(li)) { ok = true; ai[0] = ai[1] = ai[2] = ai[3] = 0.f; ah[0] = ah[1] = ah[2] = ah[3] = 0.f; }
                    if (__all(ok)) break;
                    if (!spin.again()) break;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[i], wx_[cc][i], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[i], wh_[cc][i], acc[1], 0, 0, 0);
                }
            }
            reduce_rows<2>(acc, part, wave, li, lk);
            if (wave == 0) {
                const int jn = 16 * w + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = lk * 4 + r;
                    if (j >= NR || !ON(j)) continue;
                    const long sr = (long)s * R + (long)j * C + c;
                    const float dcg = acc[0][r], dct = dcg + doc[r];
                    put(G + B_DC + (long)j * H2 + jn, tag, dct, same_xcd);
                    a.dx[sr * KX + EE + jn] = dcg;
                    a.dctx_all[sr * H2 + jn] = dct;
                    dhc[r] = dhz[r] + acc[1][r];
                }
            }
        }
        // ================================================================ (3) attention backward over this workgroup's frames
        {   // dctx rows -> LDS (thread = column)
#pragma unroll
            for (int j = 0; j < MAXR; ++j) {
                if (!ON(j)) continue;
                float vv = 0.f;
                for (;;) {
                    const u64 xv = __hip_atomic_load((gu64*)(G + B_DC + (long)j * H2 + tid), RLX_AGENT);
                    vv = __uint_as_float((unsigned)xv);
                    if (__all((unsigned)(xv >> 32) == tag)) break;
                    if (!spin.again()) break;
                }
                dcx[j * H2 + tid] = vv;
            }
        }
        __syncthreads();
        if (wave < NR && ON(wave)) {                          // dot = dctx . ctx of row `wave`
            const long sr = (long)s * R + (long)wave * C + c;
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) p = fmaf(dcx[wave * H2 + lane + 64 * i], a.x[sr * KX + EE + lane + 64 * i], p);
            p = wave_sum(p);
            if (lane == 0) red[8 + wave] = p;
        }
        __syncthreads();
        for (int f = wave; f < nf; f += NWV) {                // da_t = dctx . enc_t, one wave per frame
            const f32x4 e0 = reinterpret_cast<const f32x4*>(Ec + f * H2)[lane], e1 = reinterpret_cast<const f32x4*>(Ec + f * H2)[64 + lane];
#pragma unroll
            for (int j = 0; j < MAXR; ++j) {
                if (!ON(j)) continue;
                const f32x4 d0 = reinterpret_cast<const f32x4*>(dcx + j * H2)[lane], d1 = reinterpret_cast<const f32x4*>(dcx + j * H2)[64 + lane];
                float da = d0[0] * e0[0] + d0[1] * e0[1] + d0[2] * e0[2] + d0[3] * e0[3] + d1[0] * e1[0] + d1[1] * e1[1] + d1[2] * e1[2] + d1[3] * e1[3];
                da = wave_sum_lane63(da);
                if (lane == 63) {
                    const float d_s = aw[j * 64 + f] * (da - red[8 + j]);
                    dsv[j * 64 + f] = d_s;
                    a.ds_all[((long)s * R + (long)j * C + c) * T + t0 + f] = d_s;
                }
            }
        }
        __syncthreads();
        {   // partial dq: thread (unit jq = tid & 255, frame parity tid >> 8); the odd half hands over through LDS
            const int jq = tid & (HH - 1), hf = tid >> 8;
            float accq[MAXR];
#pragma unroll
            for (int j = 0; j < MAXR; ++j) accq[j] = 0.f;
            for (int f = hf; f < nf; f += NTH / HH) {
                const float kk = Kc[f * HH + jq];
#pragma unroll
                for (int j = 0; j < MAXR; ++j) if (ON(j)) accq[j] = fmaf(dsv[j * 64 + f], sech2_ek(kk, Eq[j * HH + jq]), accq[j]);
            }
            if (hf == 1) {
#pragma unroll
                for (int j = 0; j < MAXR; ++j) part[j * HH + jq] = accq[j];
            }
            __syncthreads();
            if (hf == 0) {
                const float vj = a.attn_v[jq];
#pragma unroll
                for (int j = 0; j < MAXR; ++j) if (ON(j)) put(G + B_DQP + (long)(w * MAXR + j) * HH + jq, tag, (accq[j] + part[j * HH + jq]) * vj, same_xcd);
            }
            __syncthreads();
        }
        // ================================================================ (4) dq, columns 8 w .. + 7: sum of the 32 partials (half-waves)
        if (tid < 256) {
            const int col = tid >> 5, g = tid & 31;
#pragma unroll
            for (int j = 0; j < MAXR; ++j) {
                if (!ON(j)) continue;
                float pv = 0.f;
                for (;;) {
                    const u64 xv = __hip_atomic_load((gu64*)(G + B_DQP + (long)(g * MAXR + j) * HH + 8 * w + col), RLX_AGENT);
                    pv = __uint_as_float((unsigned)xv);
                    if (__all((unsigned)(xv >> 32) == tag)) break;
                    if (!spin.again()) break;
                }
                pv += dpp_take<A2S_DPP_QUAD_1032>(0.f, pv);
                pv += dpp_take<A2S_DPP_QUAD_2301>(0.f, pv);
                pv += dpp_take<A2S_DPP_ROW_HALF_MIRROR>(0.f, pv);
                pv += dpp_take<A2S_DPP_ROW_MIRROR>(0.f, pv);
                pv += dpp_take<A2S_DPP_ROW_BCAST15, 0xA>(0.f, pv);
                if ((lane & 31) == 31) {
                    put(G + B_DQ + (long)j * HH + 8 * w + col, tag, pv, same_xcd);
                    a.dq_all[((long)s * R + (long)j * C + c) * HH + 8 * w + col] = pv;
                }
            }
        }
        // ================================================================ (5) carry += dq W_h, columns 16 w ..
        {
            f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(G + B_DQ, MAXR * HH * 8);
#pragma unroll
            for (int cc = 0; cc < QKS; ++cc) {
                float av[4];
                for (;;) {
                    bool ok = frag_load(rs, HH, NR, li, lk, 16 * (wave + NWV * cc), tag, av);
                    if (li < NR && !ON(li)) { ok = true; av[0] = av[1] = av[2] = av[3] = 0.f; }
                    if (__all(ok)) break;
                    if (!spin.again()) break;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wq_[cc][i], acc[0], 0, 0, 0);
            }
            reduce_rows<1>(acc, part, wave, li, lk);
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (lk * 4 + r < NR && ON(lk * 4 + r)) dhc[r] += acc[0][r];
            }
        }
#undef ON
    }
    if (wave == 0) {
        const int lane = tid0 & 63, li = lane & 15, lk = lane >> 4;
        const bool bad = spin.dead || dp_aborted(a.abort_flag);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = lk * 4 + r;
            if (j < NR) a.dh[(long)(j * C + c) * H2 + 16 * w + li] = bad ? __builtin_nanf("") : dhc[r];
        }
        if (bad && lane == 0 && a.latch) atomicOr(a.latch, 8u);
    }
}

// Greedy decoding: the true end of the loop is S = 1 + the latest first-<eos> step over all rows (when every row has one; else `steps`); a clip
// may have run past it.  Erase what the reference would not have written, rebuild lengths from the ids below S, report S.
__global__ __launch_bounds__(256) void dec_persist_greedy_fixup(float* __restrict__ probs, long probs_bstride, int* __restrict__ ids, long am_bstride,
                                                                long long* __restrict__ lengths, const int* __restrict__ eos_first, int* __restrict__ steps_exec,
                                                                int R, int steps, int V, int eos_id, int max_steps) {
    __shared__ int S;
    if (threadIdx.x == 0) {
        int m = -1; bool all = true;
        for (int r = 0; r < R; ++r) { all &= eos_first[r] >= 0; m = max(m, eos_first[r]); }
        S = all ? min(m + 1, steps) : steps;
        if (blockIdx.x == 0) *steps_exec = S;
    }
    __syncthreads();
    const int row = blockIdx.x;
    for (long i = (long)S * V + threadIdx.x; i < (long)steps * V; i += 256) probs[row * probs_bstride + i] = 0.f;
    if (threadIdx.x == 0) {
        int last = 0;
        for (int t = 0; t < S; ++t) if (ids[row * am_bstride + t] == eos_id) last = t + 1;
        lengths[row] = last > 0 ? last : max_steps;
    }
    for (int t = S + threadIdx.x; t < steps; t += 256) ids[row * am_bstride + t] = 0;
}

// ------------------------------------------------------------------------------------------- launcher
static int g_dec_persist_launches = 0;                  // persistent forward + backward launches so far (a2s_debug_get("dec_persist_launches"): tests)
int a2s_dec_persist_launches(void) { return g_dec_persist_launches; }
static int g_dec_persist = -1;                          // A2S_DEC_PERSIST=0 / a2s_debug_set("dec_persist", 0): the launch-per-step kernels
void a2s_dec_persist_set(int v) { g_dec_persist = v ? 1 : 0; }
int a2s_dec_persist_enabled(void) {
    if (g_dec_persist < 0) { const char* e = getenv("A2S_DEC_PERSIST"); g_dec_persist = (e && e[0] == '0') ? 0 : 1; }
    return g_dec_persist;
}
static size_t dp_lds_bytes(void) { return sizeof(float) * (CHF * HH + CHF * H2 + MAXR * HH + MAXR * 64 + 64 + DP_PART_FLOATS + MAXR * XLD); }
size_t a2s_note_decoder_persist_ws_bytes(int n_clips, int R, int steps) {
    if (n_clips < 1 || n_clips > 8) return 0;
    return 512 + sizeof(unsigned) * 8 * NWG + sizeof(u64) * (size_t)n_clips * DP_REGION + sizeof(float) * 2 * (size_t)(steps > 0 ? steps : 1) * R
           + sizeof(int) * (size_t)R + 256;
}
static bool aligned16p(const void* p) { return ((uintptr_t)p & 15) == 0; }
// One clip per XCD, its NWG workgroups one per CU, all 8 x NWG resident at once: needs a whole MI355X (8 XCDs x 32 CUs visible to this process --
// a CPX / DPX partition or a CU mask reports fewer) and a CU that admits the kernel's LDS request.  Asked of the runtime once per kernel.
static size_t dpb_lds_bytes(void);
static bool dp_chip_ok(bool bwd) {
    static int occ[2] = {-1, -1};
    if (occ[bwd] < 0) {
        int n = 0;
        hipError_t e = bwd ? hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dpb_lds_bytes())
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds_bytes());
        if (e == hipSuccess) e = bwd ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dec_persist_bwd, NTH, dpb_lds_bytes())
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dec_persist_fwd, NTH, dp_lds_bytes());
        if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
        occ[bwd] = n;
    }
    const a2s_device_geom g = a2s_device_geometry();
    return occ[bwd] >= 1 && g.xccs == 8 && g.cus >= 8 * NWG;
}

bool a2s_note_decoder_fwd_persist_ok(const a2s_note_dec_args& a) {
    if (!a2s_dec_persist_enabled() || !a.persist_ws || a.use_graph || !dp_chip_ok(false)) return false;
    if (!a.gt && (a.gates || a.attw || a.drop || a.n_active)) return false;      // greedy: inference only
    const int C = a.n_clips > 0 ? a.n_clips : a.R;
    if (C < 1 || C > 8 || a.R % C || a.R / C > MAXR) return false;
    if (a.H != HH || a.E != EE || a.V != VV || a.T > NWG * CHF || a.steps < 1) return false;
    if (a.gt && a.tf_flags && !a.tf_flags_dev) return false;
    if (a.persist_ws_bytes < a2s_note_decoder_persist_ws_bytes(C, a.R, a.steps) || ((uintptr_t)a.persist_ws & 255)) return false;
    return aligned16p(a.w_ih) && aligned16p(a.w_hh) && aligned16p(a.out_w) && aligned16p(a.attn_w) && aligned16p(a.keys) && aligned16p(a.enc);
}

int a2s_note_decoder_fwd_persist(hipStream_t st, const a2s_note_dec_args& a, int* steps_done) {
    const int C = a.n_clips > 0 ? a.n_clips : a.R;
    const size_t need = a2s_note_decoder_persist_ws_bytes(C, a.R, a.steps);
    char* base = reinterpret_cast<char*>(a.persist_ws);
    const size_t head = 512 + sizeof(unsigned) * 8 * NWG + sizeof(u64) * (size_t)C * DP_REGION;
    hipError_t e = hipMemsetAsync(base, 0, head, st);
    // what the backward pass reads of rows / steps this call never writes must be finite (as a2s_note_decoder_fwd_impl does for its tail path)
    const long n = a.steps, R = a.R;
    if (e == hipSuccess) e = hipMemsetAsync(a.h + R * H2, 0, sizeof(float) * n * R * H2, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.x + R * KX, 0, sizeof(float) * n * R * KX, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.q, 0, sizeof(float) * n * R * HH, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.o, 0, sizeof(float) * n * R * 2 * H2, st);
    if (e == hipSuccess && a.gates) e = hipMemsetAsync(a.gates, 0, sizeof(float) * n * R * 4 * H2, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_fwd_persist memset: %s", hipGetErrorString(e));
    DecPersistFwd p;
    p.attn_w = a.attn_w; p.attn_b = a.attn_b; p.attn_v = a.attn_v; p.w_ih = a.w_ih; p.w_hh = a.w_hh; p.b_ih = a.b_ih; p.b_hh = a.b_hh;
    p.out_w = a.out_w; p.out_b = a.out_b; p.emb = a.emb; p.keys = a.keys; p.enc = a.enc;
    p.h = a.h; p.x = a.x; p.q = a.q; p.o = a.o; p.gates = a.gates; p.attw = a.attw;
    p.stats = reinterpret_cast<float*>(base + head);
    p.eos_first = reinterpret_cast<int*>(base + head + sizeof(float) * 2 * (size_t)n * R);
    p.stop = reinterpret_cast<unsigned*>(base) + 16;              // (inside the zeroed head: words 16 .. 23)
    if (!a.gt) {
        e = hipMemsetAsync(p.eos_first, 0xff, sizeof(int) * R, st);
        if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_fwd_persist memset: %s", hipGetErrorString(e));
    }
    p.probs = a.probs; p.probs_bstride = a.probs_bstride; p.gt = a.gt; p.gt_bstride = a.gt_bstride; p.flags = a.tf_flags_dev;
    p.drop = a.drop; p.inv_keep = a.inv_keep; p.argmax_out = a.argmax_out; p.am_bstride = a.am_bstride;
    p.eos_seen = a.eos_seen; p.lengths = a.lengths; p.n_done = a.n_done; p.steps_exec = a.steps_exec;
    p.row_until = a.n_active ? a.row_until : nullptr;
    p.xg = reinterpret_cast<u64*>(base + 512 + sizeof(unsigned) * 8 * NWG);
    p.abort_flag = reinterpret_cast<unsigned*>(base); p.xcc = reinterpret_cast<unsigned*>(base + 512);
    p.C = C; p.NR = a.R / C; p.R = a.R; p.T = a.T; p.steps = a.steps; p.eos_id = a.eos_id;
    p.latch = a2s_persist_latch_ptr(); p.dbg = a2s_persist_dbg();
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dp_lds_bytes());
        if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "dec_persist_fwd: LDS attribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(dec_persist_fwd, dim3(8 * NWG), dim3(NTH), dp_lds_bytes(), st, p);
    A2S_CHECK_LAUNCH("dec_persist_fwd");
    __atomic_fetch_add(&g_dec_persist_launches, 1, __ATOMIC_RELAXED);
    if (a.attw) {
        hipLaunchKernelGGL(dec_persist_attw_normalise, dim3((unsigned)(n * R)), dim3(256), 0, st, a.attw, p.stats, p.row_until, a.steps, a.R, a.T);
        A2S_CHECK_LAUNCH("dec_persist_attw_normalise");
    }
    if (!a.gt) {
        hipLaunchKernelGGL(dec_persist_greedy_fixup, dim3(a.R), dim3(256), 0, st, a.probs, a.probs_bstride, a.argmax_out, a.am_bstride, a.lengths, p.eos_first,
                           a.steps_exec, a.R, a.steps, a.V, a.eos_id, (int)a.am_bstride);
        A2S_CHECK_LAUNCH("dec_persist_greedy_fixup");
    }
    if (steps_done) *steps_done = a.steps;
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- backward launcher
int a2s_gemm_impl(hipStream_t st, int M, int N, int K, float alpha, const float* A, long sAm, long sAk,
                  const float* B, long sBk, long sBn, float beta, float* C, long ldc, const float* bias, int act,
                  int batch, long bsA, long bsB, long bsC, int splitk, float* ws, size_t ws_bytes);
int a2s_note_step_fused_bwd_prepare(hipStream_t st, const a2s_note_dec_bwd_args& a);
size_t a2s_note_step_fused_head_floats(void);
static size_t dpb_lds_bytes(void) { return sizeof(float) * (CHF * HH + CHF * H2 + MAXR * HH + MAXR * H2 + 2 * MAXR * 64 + 64 + DPB_PART_FLOATS); }
size_t a2s_note_decoder_bwd_persist_ws_bytes(int n_clips) {
    if (n_clips < 1 || n_clips > 8) return 0;
    return 512 + sizeof(unsigned) * 8 * NWG + sizeof(u64) * (size_t)n_clips * DPB_REGION;
}
bool a2s_note_decoder_bwd_persist_ok(const a2s_note_dec_bwd_args& a) {
    if (!a2s_dec_persist_enabled() || !a.persist_ws || !dp_chip_ok(true)) return false;
    const int C = a.n_clips > 0 ? a.n_clips : a.R;
    if (C < 1 || C > 8 || a.R % C || a.R / C > MAXR) return false;
    if (a.H != HH || a.E != EE || a.T > NWG * CHF || a.steps < 1 || !a.step_ws) return false;
    if (a.persist_ws_bytes < a2s_note_decoder_bwd_persist_ws_bytes(C) || ((uintptr_t)a.persist_ws & 255)) return false;
    const size_t need_ws = a2s_note_step_fused_head_floats() + (size_t)KX * 3 * H2 + (size_t)H2 * 3 * H2 + (size_t)H2 * HH;
    if (a.step_ws_floats < need_ws || ((uintptr_t)(a.step_ws + a2s_note_step_fused_head_floats()) & 15)) return false;
    return aligned16p(a.keys) && aligned16p(a.enc) && aligned16p(a.q);
}
int a2s_note_decoder_bwd_persist(hipStream_t st, const a2s_note_dec_bwd_args& a) {
    const int C = a.n_clips > 0 ? a.n_clips : a.R;
    const long n = a.steps, R = a.R;
    char* base = reinterpret_cast<char*>(a.persist_ws);
    hipError_t e = hipMemsetAsync(base, 0, a2s_note_decoder_bwd_persist_ws_bytes(C), st);
    // rows / steps this call never touches read as zero gradients in the deferred products
    if (e == hipSuccess) e = hipMemsetAsync(a.dgi_all, 0, sizeof(float) * n * R * 3 * H2, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.dgh_all, 0, sizeof(float) * n * R * 3 * H2, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.dq_all, 0, sizeof(float) * n * R * HH, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.ds_all, 0, sizeof(float) * n * R * a.T, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.dctx_all, 0, sizeof(float) * n * R * H2, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.dx, 0, sizeof(float) * n * R * KX, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_bwd_persist memset: %s", hipGetErrorString(e));
    int rc = a2s_note_step_fused_bwd_prepare(st, a);       // W_ih^T, W_hh^T, W_h^T into the step workspace
    if (rc) return rc;
    DecPersistBwd p;
    p.attn_v = a.attn_v;
    p.wih_t = a.step_ws + a2s_note_step_fused_head_floats();
    p.whh_t = p.wih_t + (long)KX * 3 * H2;
    p.wh_t = p.whh_t + (long)H2 * 3 * H2;
    p.keys = a.keys; p.enc = a.enc; p.h = a.h; p.x = a.x; p.q = a.q; p.gates = a.gates; p.attw = a.attw; p.do_all = a.do_all;
    p.dgi_all = a.dgi_all; p.dgh_all = a.dgh_all; p.dq_all = a.dq_all; p.ds_all = a.ds_all; p.dctx_all = a.dctx_all; p.dx = a.dx; p.dh = a.dh;
    p.row_until = a.n_active ? a.row_until : nullptr;
    p.abort_flag = reinterpret_cast<unsigned*>(base); p.xcc = reinterpret_cast<unsigned*>(base + 512);
    p.xg = reinterpret_cast<u64*>(base + 512 + sizeof(unsigned) * 8 * NWG);
    p.C = C; p.NR = a.R / C; p.R = a.R; p.T = a.T; p.steps = a.steps;
    p.latch = a2s_persist_latch_ptr(); p.dbg = a2s_persist_dbg();
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(dec_persist_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dpb_lds_bytes());
        if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "dec_persist_bwd: LDS attribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(dec_persist_bwd, dim3(8 * NWG), dim3(NTH), dpb_lds_bytes(), st, p);
    A2S_CHECK_LAUNCH("dec_persist_bwd");
    __atomic_fetch_add(&g_dec_persist_launches, 1, __ATOMIC_RELAXED);
    // token columns of dx for all steps at once: dx[:, :E] = dgi_all W_ih[:, :E]  (W_ih (3 H2, KX): B(k, n) = w_ih[k * KX + n])
    return a2s_gemm_impl(st, (int)(n * R), EE, 3 * H2, 1.f, a.dgi_all, 3 * H2, 1, a.w_ih, KX, 1, 0.f, a.dx, KX, nullptr, 0, 1, 0, 0, 0, 0, nullptr, 0);
}
