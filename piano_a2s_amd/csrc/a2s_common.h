// Shared definitions for the gfx950 kernels of the piano-a2s hot path (C-ABI library liba2s_hip.so).
// Everything here is device/host plumbing; the public surface is include/a2s.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define A2S_OK 0
#define A2S_ERR_ARG (-1)
#define A2S_ERR_HIP (-2)
#define A2S_ERR_WORKSPACE (-3)

// one message slot per host thread (one host thread per process/GPU under torchrun)
extern thread_local char a2s_err_msg[512];

#define A2S_FAIL(code, ...)                                   \
    do {                                                      \
        snprintf(a2s_err_msg, sizeof(a2s_err_msg), __VA_ARGS__); \
        return (code);                                        \
    } while (0)

#define A2S_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) A2S_FAIL(A2S_ERR_ARG, __VA_ARGS__); \
    } while (0)

// Launch check: never synchronises; reports the launch-time error only.
#define A2S_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// Every kernel launch of the library is counted (a relaxed host-side add): `a2s_launch_count()` lets the bench report launches per optimizer
// step / per decode step without a profiler.
extern long long a2s_launch_counter;
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                              \
    do {                                                                 \
        __atomic_fetch_add(&a2s_launch_counter, 1LL, __ATOMIC_RELAXED);  \
        hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);           \
    } while (0)

static inline int a2s_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- persistent kernels (csrc/a2s_persist.hip, csrc/a2s_dec_persist.hip): what the device offers, the process-wide abort latch, test hooks
struct a2s_device_geom { int cus, xccs; };             // compute units and XCDs visible to this process (current device; cached)
a2s_device_geom a2s_device_geometry(void);
unsigned* a2s_persist_latch_ptr(void);                  // device word the kernels OR a bit into when a bounded wait gave up (null: none registered)
void a2s_persist_latch_set(void* dev_word);
#define PERSIST_DBG_FORCE_AGENT 1u                      // a2s_debug_set("persist_force_agent", 1): never take the plain-store (one-XCD) hand-off
#define PERSIST_DBG_INJECT_ABORT 2u                     // a2s_debug_set("persist_inject_abort", 1): every persistent launch behaves as if a wait had timed out
unsigned a2s_persist_dbg(void);
void a2s_persist_dbg_set(unsigned bit, int on);
int a2s_persist_dbg_get(unsigned bit);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Split of the T frames of a clip over G workgroups for the attention kernels (forward and backward must agree):
// aim at ~768 workgroups (3 per CU) but never more than 16 chunks (64 measured slower overall: with only a handful of clips still
// decoding -- the long-clip group -- a chunk of 76 frames is 5 + 10 dependent load rounds in one workgroup, ~15-35 us per launch; 20
// frames are 2 + 3); chunk is a multiple of 4 frames.
#include <stdlib.h>
static inline int a2s_attn_target_wgs(void) { return 768; }
static inline int a2s_attn_max_split(void) { return 16; }
static inline void a2s_attn_split_geometry(int B, int T, int* G, int* chunk) {
    const int target = a2s_attn_target_wgs();
    int g = (target + B - 1) / B;
    if (g > a2s_attn_max_split()) g = a2s_attn_max_split();
    if (g < 1) g = 1;
    int c = (T + g - 1) / g;
    c = (c + 3) & ~3;
    *G = (T + c - 1) / c;
    *chunk = c;
}

// Which rows one attention launch of a note decoder covers (training only; NULL = every row, one group).
// The R rows of a fused decoder call are `groups` bars of the same n_clips clips, row = group * n_clips + clip: rows of one clip
// share that clip's keys and encoder outputs, so one workgroup serves all of them from a single pass over the clip's chunk.
// A row is skipped (context 0) from step row_until[row] on: its remaining targets are all <pad>.  Clips sorted by the step their
// last row finishes at (latest first) so that the clips still running are a prefix of clip_order.
struct a2s_attn_rows {
    const int* clip_order;   // device, n_clips ints (NULL: identity)
    const int* clip_rank;    // device, n_clips ints: inverse permutation
    const int* row_until;    // device, R ints (NULL: never finished)
    int n_clips;             // clips per group
    int n_active;            // clips with at least one unfinished row at this step (prefix of clip_order)
    int step;
};
#define A2S_ATTN_MAX_GROUPS 5
// Late steps of a large decoder call run their per-step products on the leading m clips of every fused bar (the clips still running) once
// m <= half of the clips (measured over 25 ... 100 %: profiles/r05_prefix_percent.txt)
static inline bool a2s_prefix_rows_ok(int m, int n_clips) { return m > 0 && 2L * m <= (long)n_clips; }
// A forward attention launch whose combine has been left to its consumer (round 5: the few-row GRU step folds it into its prologue -- one
// launch and one dependent-launch gap less per decode step on the long-clip chain, where a launch costs ~20 us + ~18 us of gap under the
// other clip group's traffic: profiles/r05_trace_overlap.txt).  G == 0: nothing deferred, the combine has run.
struct a2s_attn_deferred {
    const float* part;       // partials [m, l, pad, pad, ctx(2H)] of (slot, group, g), as the combine kernel reads them
    float* attw;             // raw scores of the step, (R, T), to be normalised in place (or NULL)
    const int* clip_rank; const int* row_until;
    int G, groups, n_clips, n_active, step, T;
};
// head of the attention workspace: arrival counters of the fused combine (forward: [0, 4096), backward: [4096, 8192)), in floats
#define A2S_ATTN_TICKETS 8192

// Streaming K / enc loads of the split attention kernels: 0 = off, n > 0 = launches covering at least n clips use non-temporal loads
// (a2s_debug_set("attn_nt", n)).
int a2s_attn_nt_enabled(void);
void a2s_attn_nt_set(int v);

// ----------------------------------------------------------------------------- device helpers
#ifdef __HIPCC__
#define A2S_WAVE 64
// 16-byte load of 4 floats, optionally non-temporal (a stream that is read once per launch and is far larger than the caches)
template <bool NT>
__device__ __forceinline__ f32x4 ld_kv(const float* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    else return *reinterpret_cast<const f32x4*>(p);
}

// Wave-wide reductions on the DPP data path (row shifts / mirrors / broadcasts riding on VALU instructions) -- `__shfl_xor` compiles to
// ds_bpermute_b32, i.e. every step of a butterfly goes through the LDS crossbar: ~6 dependent LDS round trips per reduced value, and the
// attention kernels reduce one value per (frame, row).  Six DPP adds leave the total in lane 63; v_readlane broadcasts it.
template <int CTRL, int ROW_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float dpp_take(float oldv, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
#define A2S_DPP_QUAD_1032 0xB1
#define A2S_DPP_QUAD_2301 0x4E
#define A2S_DPP_ROW_HALF_MIRROR 0x141
#define A2S_DPP_ROW_MIRROR 0x140
#define A2S_DPP_ROW_BCAST15 0x142
#define A2S_DPP_ROW_BCAST31 0x143
// sum over the 64 lanes, valid in lane 63 only
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += dpp_take<A2S_DPP_QUAD_1032>(0.f, v);
    v += dpp_take<A2S_DPP_QUAD_2301>(0.f, v);
    v += dpp_take<A2S_DPP_ROW_HALF_MIRROR>(0.f, v);
    v += dpp_take<A2S_DPP_ROW_MIRROR>(0.f, v);                      // every lane: the sum of its row of 16
    v += dpp_take<A2S_DPP_ROW_BCAST15, 0xA>(0.f, v);                // rows 1, 3 += lane 15 of the row below
    v += dpp_take<A2S_DPP_ROW_BCAST31, 0xC>(0.f, v);                // rows 2, 3 += lane 31
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_lane63(v)), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_take<A2S_DPP_QUAD_1032>(v, v));
    v = fmaxf(v, dpp_take<A2S_DPP_QUAD_2301>(v, v));
    v = fmaxf(v, dpp_take<A2S_DPP_ROW_HALF_MIRROR>(v, v));
    v = fmaxf(v, dpp_take<A2S_DPP_ROW_MIRROR>(v, v));
    v = fmaxf(v, dpp_take<A2S_DPP_ROW_BCAST15, 0xA, false>(v, v));
    v = fmaxf(v, dpp_take<A2S_DPP_ROW_BCAST31, 0xC, false>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Merge weights of a row's attention partials, one partial per lane (lanes without one: have = false): e = exp(m_g - m) / l with m the row's
// maximum and l = sum_g l_g exp(m_g - m).  ONE definition with floating-point contraction off, so that the combine kernel (a2s_seq.hip) and the
// GRU step that folds the combine into its prologue (a2s_step.hip) produce the same bits wherever the compiler inlines it.
__device__ __forceinline__ float attn_merge_weight(float mg, float lg, bool have, float& m_out, float& inv_l_out) {
#pragma clang fp contract(off)
    const float m = wave_max(mg);
    const float d = mg - m;
    const float e = have ? __expf(d) : 0.f;
    const float t = lg * e;
    const float l = wave_sum(t);
    m_out = m;
    inv_l_out = 1.f / l;
    return e / l;
}
// GRU cell backward of one (row, hidden unit) (saved [r | z | n | gh_n]; see gru_gates_bwd, a2s_bwd.hip): ONE definition, contraction off, so
// that the elementwise kernel and the product kernel that folds it into its prologue (a2s_step.hip) round alike.
struct GruCellGrad { float dr, dz, dn, dnr, dhz; };
__device__ __forceinline__ GruCellGrad gru_cell_bwd(float dh, float rg, float zg, float ng, float ghn, float hp) {
#pragma clang fp contract(off)
    GruCellGrad o;
    o.dn = dh * (1.f - zg) * (1.f - ng * ng);
    o.dz = dh * (hp - ng) * zg * (1.f - zg);
    o.dr = o.dn * ghn * rg * (1.f - rg);
    o.dnr = o.dn * rg;
    o.dhz = dh * zg;
    return o;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// tanh / sigmoid through the hardware exp2 + rcp (abs error ~2e-7: far inside the 1e-4 parity budget).  The reciprocal is the bare
// v_rcp_f32 (1 ulp): __frcp_rn expands to the full IEEE division sequence (2x v_div_scale, v_rcp, 4 fma, v_div_fmas, v_div_fixup),
// which made tanh 18 instructions instead of 8 -- the attention kernels with several rows per clip are bound by exactly this.
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); clamp keeps exp finite, tanh(+-15) == +-1 in fp32
    x = fminf(fmaxf(x, -15.f), 15.f);
    const float e = __expf(2.f * x);
    return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}
// Additive-attention energies through the KEY IMAGE E_K = exp(2 K) (written once per forward by the key projection's GEMM epilogue,
// act 3) and E_q = exp(2 q) (once per row and step):  tanh(k + q) = 1 - 2 / (1 + E_K E_q)  is 3 instructions with ONE transcendental
// per (frame, unit) instead of 8 with two -- the attention kernels that serve several rows per clip are bound by exactly this.
// The clamp keeps both factors finite and normal (|x| <= 43: e^86 < FLT_MAX); a product that overflows gives rcp(inf) = 0 -> tanh = 1.
__device__ __forceinline__ float exp2x_clamped(float x) { return __expf(2.f * fminf(fmaxf(x, -43.f), 43.f)); }
__device__ __forceinline__ float tanh_ek(float ek, float eq) { return fmaf(-2.f, __builtin_amdgcn_rcpf(fmaf(ek, eq, 1.f)), 1.f); }
__device__ __forceinline__ float sech2_ek(float ek, float eq) {      // 1 - tanh^2(k + q) = 4 r (1 - r), r = 1 / (1 + E_K E_q)
    const float r = __builtin_amdgcn_rcpf(fmaf(ek, eq, 1.f));
    return 4.f * r * (1.f - r);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
    x = fminf(fmaxf(x, -30.f), 30.f);
    return __builtin_amdgcn_rcpf(1.f + __expf(-x));
}
// ---- fp32 operands on the bf16 matrix pipes (conv3x3_bf16x3, gemm split path)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// Two fp32 values -> their three bf16 terms, packed (x0's term in the low half): v_cvt_pk_bf16_f32 rounds to nearest even, the
// residuals are exact in fp32 and the third term is exact, so t0 + t1 + t2 == x and the dropped products have no sign bias.
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    f32x2 v = {x0, x1};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    v[0] -= __uint_as_float(p0 << 16); v[1] -= __uint_as_float(p0 & 0xffff0000u);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    v[0] -= __uint_as_float(p1 << 16); v[1] -= __uint_as_float(p1 & 0xffff0000u);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// ---- fp32 operands on the fp16 matrix pipes, TWO terms (conv3x3_split<.., 2>): x = t0 + t1 with fp16 t0 = rn(x), t1 = rn(x - t0); the
// residual x - t0 is exact in fp32, so |x - t0 - t1| <= 2^-23 |x| as long as t1 is a NORMAL fp16 number (|x| >= 2^-3 after the caller's
// power-of-two scaling; below that the absolute error is the fp16 subnormal quantum 2^-25).  Three term products (t0 t0', t0 t1', t1 t0')
// instead of the six of the bf16 three-term split: each is exact in the fp32 accumulator (11 x 11 bits), the dropped t1 t1' <= 2^-22 |x x'|.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ void split2_pair_f16(float x0, float x1, unsigned& p0, unsigned& p1) {
    f32x2 v = {x0, x1};
    const f16x2 t0 = __builtin_convertvector(v, f16x2);              // v_cvt_f16_f32: round to nearest even
    p0 = __builtin_bit_cast(unsigned, t0);
    // x - t0 with the fp16 -> fp32 extension riding on the operand (one instruction per element instead of convert + subtract; the
    // compiler does not form it from fpext + fsub)
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[0]) : "v"(p0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[1]) : "v"(p0));
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// Fold a workgroup's running max |x| (am >= 0, per thread) into a device scalar: one candidate per workgroup, and the atomic only when
// it would raise the value (non-negative floats order like their bit patterns).  Half a million unconditional same-address atomics -- one
// per wave of bn_bwd_apply_planes -- serialise at ~30 ns each: 5 ms on a 10 ms kernel.  `red` = 16 floats of LDS, blockDim.x a multiple of 64.
__device__ __forceinline__ void block_absmax_to(float* __restrict__ out, float am, float* red) {
    am = block_max(am, red);
    if (threadIdx.x == 0) {
        const unsigned bits = __float_as_uint(am);
        if (bits > __hip_atomic_load(reinterpret_cast<unsigned*>(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned*>(out), bits);
    }
}
// power-of-two scale that brings a tensor whose largest magnitude is `amax` to [2^target, 2^(target+1)): exact to apply and to undo
__device__ __forceinline__ int pow2_scale_exp(float amax, int target) {
    if (!(amax > 0.f) || !isfinite(amax)) return 0;
    const int k = target - ilogbf(amax);
    return min(max(k, -96), 96);
}

#endif
