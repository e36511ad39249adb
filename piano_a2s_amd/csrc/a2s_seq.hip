// Sequential part of the hot path, forward: GRU cells, additive attention, note-decoder step epilogue,
// packed staff-embedding bi-GRU, and the C++ step loops that drive them (so no Python runs per step).
// Reference: Encoder.forward models.py:75-82 (a-3), AttentionLayer models.py:452-461 (a-7), context bmm
// :242,:394 (a-8), NoteDecoder.decode_notes :366-420 (a-9), get_staff_token_* :164-189 (a-11).
//
// Attention is restructured algebraically but not numerically re-ordered beyond fp32 round-off:
//   energy = tanh(W [h ; enc_t] + b) = tanh(W_h h + b  +  W_e enc_t) = tanh(q + K_t)
// K = enc W_e^T is step-invariant and computed once per layer by the GEMM; per step only
// q (a skinny GEMM), the score/softmax and the context remain -- one pass over K and one over enc.
#include "a2s_common.h"
#include "../../include/a2s.h"

int a2s_gemm_impl(hipStream_t st, int M, int N, int K, float alpha, const float* A, long sAm, long sAk,
                  const float* B, long sBk, long sBn, float beta, float* C, long ldc, const float* bias, int act,
                  int batch, long bsA, long bsB, long bsC, int splitk, float* ws, size_t ws_bytes);

// ------------------------------------------------------------------------------------------- GRU cell
// PyTorch packing [r; z; n].  gi = W_ih x + b_ih, gh = W_hh h + b_hh (both (R, 3H), row strides given).
// h' = (1-z)*n + z*h ;  optional `live` mask (packed-sequence semantics): dead rows keep h.
// When `save` != null stores r, z, n, gh_n per row ((R, 4H)) for the backward pass.
__global__ void gru_gates_fwd(const float* __restrict__ gi, long ldgi, const float* __restrict__ gh, long ldgh,
                              const float* __restrict__ hprev, long ldhp, float* __restrict__ hout, long ldho,
                              float* __restrict__ hout2, long ldho2, float* __restrict__ save, int R, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * H) return;
    const int r_ = (int)(idx / H), j = (int)(idx % H);
    const float* a = gi + (long)r_ * ldgi;
    const float* b = gh + (long)r_ * ldgh;
    const float rg = fast_sigmoid(a[j] + b[j]);
    const float zg = fast_sigmoid(a[H + j] + b[H + j]);
    const float ghn = b[2 * H + j];
    const float ng = fast_tanh(a[2 * H + j] + rg * ghn);
    const float hp = hprev[(long)r_ * ldhp + j];
    const float hn = (1.f - zg) * ng + zg * hp;
    hout[(long)r_ * ldho + j] = hn;
    if (hout2) hout2[(long)r_ * ldho2 + j] = hn;
    if (save) {
        float* s = save + (long)r_ * 4 * H;
        s[j] = rg; s[H + j] = zg; s[2 * H + j] = ng; s[3 * H + j] = ghn;
    }
}

int a2s_gru_gates_fwd_impl(hipStream_t st, const float* gi, long ldgi, const float* gh, long ldgh, const float* hprev,
                           long ldhp, float* hout, long ldho, float* hout2, long ldho2, float* save, int R, int H) {
    hipLaunchKernelGGL(gru_gates_fwd, dim3(a2s_cdiv((long)R * H, 256)), dim3(256), 0, st, gi, ldgi, gh, ldgh, hprev, ldhp,
                       hout, ldho, hout2, ldho2, save, R, H);
    A2S_CHECK_LAUNCH("gru_gates_fwd");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- fused recurrent step
// The encoder recurrences are 1201 dependent steps of a tiny product (B x H) x (H x 3H): as GEMM + split-K reduce + gate kernel each
// step costs three launches and ~19 us whatever the batch.  Here one launch does the step: a workgroup owns 16 rows x 16 hidden
// units (their r, z, n gate columns = 3 n-tiles); both MFMA operands are K-contiguous rows (h and W_hh), so every lane fetches its
// fragments straight from L2 with 16-byte loads -- no LDS staging, one round trip -- the 4 waves take the 16-wide k-steps round
// robin, their partial tiles are summed through LDS and the gate math runs on the accumulators (its operands are fetched before the
// product so that their HBM latency hides behind it).  k order inside a 16-wide step: lane group lk supplies k = 4*lk + j for the
// j-th MFMA of the step (any partition is valid as long as A and B agree).
template <int NT, int CH>
__device__ __forceinline__ void mfma_rows(const float* __restrict__ arow, const float* const (&brow)[NT], int ksteps, int wave, int lk,
                                          f32x4 (&acc)[NT]) {
    // arow / brow[g]: this lane's operand rows; k-step u covers k in [16u, 16u+16), this lane reads 4 floats at 16u + 4*lk
    for (int u0 = wave; u0 < ksteps; u0 += 4 * CH) {
        f32x4 a[CH], b[NT][CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int u = u0 + 4 * c;
            const bool ok = u < ksteps;
            a[c] = ok ? *reinterpret_cast<const f32x4*>(arow + 16 * u + 4 * lk) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < NT; ++g)
                b[g][c] = ok ? *reinterpret_cast<const f32x4*>(brow[g] + 16 * u + 4 * lk) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (u0 + 4 * c >= ksteps) break;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < NT; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], b[g][c][j], acc[g], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(256) void gru_step_fwd_fused(const float* __restrict__ gi, long ldgi, const float* __restrict__ w_hh,
                                                          const float* __restrict__ b_hh, const float* __restrict__ hprev,
                                                          float* __restrict__ hout, float* __restrict__ hout2, long ldho2,
                                                          float* __restrict__ save, int R, int H) {
    __shared__ f32x4 part[3 * 3 * 64];
    const int j0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    // accumulator layout: lane holds column n = li (hidden unit j0 + li), rows lk*4 + r of the tile
    const int j = j0 + li;
    float gir[4], giz[4], gin[4], hp[4], br = 0.f, bz = 0.f, bn = 0.f;
    if (wave == 0) {                       // gate operands of the epilogue: in flight while the product runs
        br = b_hh[j]; bz = b_hh[H + j]; bn = b_hh[2 * H + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(row0 + lk * 4 + r, R - 1);
            const float* a = gi + (long)row * ldgi;
            gir[r] = a[j]; giz[r] = a[H + j]; gin[r] = a[2 * H + j];
            hp[r] = hprev[(long)row * H + j];
        }
    }
    const float* arow = hprev + (long)min(row0 + li, R - 1) * H;
    const float* brow[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) brow[g] = w_hh + ((long)g * H + j0 + li) * H;
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_rows<3, 4>(arow, brow, H / 16, wave, lk, acc);
    if (wave > 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g) part[((wave - 1) * 3 + g) * 64 + lane] = acc[g];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const f32x4 o = part[(w * 3 + g) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[g][r] += o[r];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + lk * 4 + r;
        if (row >= R) continue;
        const float ghn = acc[2][r] + bn;
        const float rg = fast_sigmoid(gir[r] + acc[0][r] + br);
        const float zg = fast_sigmoid(giz[r] + acc[1][r] + bz);
        const float ng = fast_tanh(gin[r] + rg * ghn);
        const float hn = (1.f - zg) * ng + zg * hp[r];
        hout[(long)row * H + j] = hn;
        if (hout2) hout2[(long)row * ldho2 + j] = hn;
        if (save) {
            float* sv = save + (long)row * 4 * H;
            sv[j] = rg; sv[H + j] = zg; sv[2 * H + j] = ng; sv[3 * H + j] = ghn;
        }
    }
}

// C[R x N] += A[R x K] Bt[N x K]^T for a skinny recurrent product (both operands K-contiguous rows): same scheme with one n-tile
// per workgroup.  Used by the encoder BPTT: dh_prev += dgh W_hh with Bt = W_hh^T (H x 3H).
__global__ __launch_bounds__(256) void skinny_gemm_acc(const float* __restrict__ A, long lda, const float* __restrict__ Bt, long ldb,
                                                       float* __restrict__ Cm, long ldc, int R, int K) {
    __shared__ f32x4 part[3 * 64];
    const int n0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    float c0[4];
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c0[r] = Cm[(long)min(row0 + lk * 4 + r, R - 1) * ldc + n0 + li];
    }
    const float* arow = A + (long)min(row0 + li, R - 1) * lda;
    const float* brow[1] = {Bt + (long)(n0 + li) * ldb};
    f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    mfma_rows<1, 12>(arow, brow, K / 16, wave, lk, acc);
    if (wave > 0) part[(wave - 1) * 64 + lane] = acc[0];
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + lk * 4 + r;
        if (row < R) Cm[(long)row * ldc + n0 + li] = c0[r] + acc[0][r] + part[lane][r] + part[64 + lane][r] + part[128 + lane][r];
    }
}

// One launch per BPTT step of an encoder direction: the carry  dh_{s-1} = dhz_s + dgh_s W_hh  (tile: 16 rows x 16 hidden units, the
// skinny product above) and, on the accumulators, the gate backward of step s-1 for exactly those (row, unit) pairs
// (gru_gates_bwd): dgi_{s-1}, dgh_{s-1} (next launch's A operand, written to the OTHER scratch buffer), its shifted copy for the
// deferred dW_hh, and dhz_{s-1} = (dh_{s-1} + dout_{s-1}) z_{s-1}.  The epilogue operands are fetched before the product.
struct GruBpttStep {
    const float* dgh; const float* w_hh_t;              // (R, 3H) of step s; (H, 3H)
    const float* dhz_in;                                // (R, H): (dh_s + dout_s) z_s
    const float* dout; long ld_dout;                    // step s-1 slice of the layer-output gradient
    const float* save;                                  // (R, 4H) [r|z|n|gh_n] of step s-1
    const float* hprev; long ld_hprev;                  // h_{s-2} (null: zeros)
    float* dgi; long ld_dgi; float* dgh_out; float* dgh2; long ld_dgh2;   // dgh2 may be null
    float* dhz_out;                                     // (R, H)
    int R, H;
};
__global__ __launch_bounds__(256) void gru_bptt_step_fused(GruBpttStep a) {
    __shared__ f32x4 part[3 * 64];
    const int H = a.H, R = a.R;
    const int j0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int j = j0 + li;
    float carry[4], dov[4], rg[4], zg[4], ng[4], ghn[4], hp[4];
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(row0 + lk * 4 + r, R - 1);
            carry[r] = a.dhz_in[(long)row * H + j];
            dov[r] = a.dout[(long)row * a.ld_dout + j];
            const float* sv = a.save + (long)row * 4 * H;
            rg[r] = sv[j]; zg[r] = sv[H + j]; ng[r] = sv[2 * H + j]; ghn[r] = sv[3 * H + j];
            hp[r] = a.hprev ? a.hprev[(long)row * a.ld_hprev + j] : 0.f;
        }
    }
    const float* arow = a.dgh + (long)min(row0 + li, R - 1) * 3 * H;
    const float* brow[1] = {a.w_hh_t + (long)(j0 + li) * 3 * H};
    f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    mfma_rows<1, 12>(arow, brow, 3 * H / 16, wave, lk, acc);
    if (wave > 0) part[(wave - 1) * 64 + lane] = acc[0];
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + lk * 4 + r;
        if (row >= R) continue;
        const float dh = carry[r] + acc[0][r] + part[lane][r] + part[64 + lane][r] + part[128 + lane][r] + dov[r];
        const float dn = dh * (1.f - zg[r]) * (1.f - ng[r] * ng[r]);
        const float dz = dh * (hp[r] - ng[r]) * zg[r] * (1.f - zg[r]);
        const float dr = dn * ghn[r] * rg[r] * (1.f - rg[r]);
        float* gi = a.dgi + (long)row * a.ld_dgi;
        gi[j] = dr; gi[H + j] = dz; gi[2 * H + j] = dn;
        float* gh = a.dgh_out + (long)row * 3 * H;
        gh[j] = dr; gh[H + j] = dz; gh[2 * H + j] = dn * rg[r];
        if (a.dgh2) { float* g2 = a.dgh2 + (long)row * a.ld_dgh2; g2[j] = dr; g2[H + j] = dz; g2[2 * H + j] = dn * rg[r]; }
        a.dhz_out[(long)row * H + j] = dh * zg[r];
    }
}

int a2s_gru_bptt_step_impl(hipStream_t st, const float* dgh, const float* w_hh_t, const float* dhz_in, const float* dout, long ld_dout,
                           const float* save, const float* hprev, long ld_hprev, float* dgi, long ld_dgi, float* dgh_out, float* dgh2,
                           long ld_dgh2, float* dhz_out, int R, int H) {
    GruBpttStep a{dgh, w_hh_t, dhz_in, dout, ld_dout, save, hprev, ld_hprev, dgi, ld_dgi, dgh_out, dgh2, ld_dgh2, dhz_out, R, H};
    hipLaunchKernelGGL(gru_bptt_step_fused, dim3(H / 16, a2s_cdiv(R, 16)), dim3(256), 0, st, a);
    A2S_CHECK_LAUNCH("gru_bptt_step_fused");
    return A2S_OK;
}

int a2s_skinny_gemm_acc_impl(hipStream_t st, const float* A, long lda, const float* Bt, long ldb, float* Cm, long ldc, int R, int N, int K) {
    A2S_REQUIRE(N % 16 == 0 && K % 16 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A | (uintptr_t)Bt) % 16 == 0,
                "skinny_gemm_acc: N %% 16, K %% 16 and 16-byte aligned K-contiguous operands required");
    hipLaunchKernelGGL(skinny_gemm_acc, dim3(N / 16, a2s_cdiv(R, 16)), dim3(256), 0, st, A, lda, Bt, ldb, Cm, ldc, R, K);
    A2S_CHECK_LAUNCH("skinny_gemm_acc");
    return A2S_OK;
}

static int g_gru_fused = -1;                                 // a2s_debug_set("gru_fused", 0): the three-launch step (A/B measurements)
void a2s_gru_step_fused_set(int v) { g_gru_fused = v ? 1 : 0; }
bool a2s_gru_step_fused_enabled(void) {
    if (g_gru_fused < 0) g_gru_fused = 1;
    return g_gru_fused != 0;
}
static bool gru_step_fusable(const float* w_hh, int H) { return a2s_gru_step_fused_enabled() && H % 16 == 0 && ((uintptr_t)w_hh % 16 == 0); }

bool a2s_gru_seq_fwd_persist_ok(const float* w_hh, const float* gi, int B, int T, int H, float* ws, size_t ws_bytes);
int a2s_gru_seq_fwd_persist_impl(hipStream_t st, const float* gi_all, long gi_bstride, long gi_tstride, const float* w_hh, const float* b_hh, float* out,
                                 long out_bstride, long out_tstride, float* save, float* hn, int B, int T, int H, int reverse, float* ws, size_t ws_bytes);

// One direction of one encoder GRU layer over all T steps (h0 = 0).
//   gi_all : (B, T, 3H) = x W_ih^T + b_ih for this direction (row stride ld_gi between time steps of a clip)
//   out    : (B, T, ldo) -- h_t is written at column offset `col0` (fwd dir 0, reverse dir H)
//   hbuf   : (2, B, H) ping-pong state, gh: (B, 3H) scratch, save: (T, B, 4H) or null
//   hn     : (B, H) final state
int a2s_gru_seq_fwd_impl(hipStream_t st, const float* gi_all, long gi_bstride, long gi_tstride, const float* w_hh,
                         const float* b_hh, float* out, long out_bstride, long out_tstride, float* hbuf, float* gh,
                         float* save, float* hn, int B, int T, int H, int reverse, float* ws, size_t ws_bytes) {
    A2S_REQUIRE(gi_all && w_hh && b_hh && out && hbuf && gh && hn, "gru_seq_fwd: null tensor");
    // one persistent launch for all T steps (a2s_persist.hip) when the shape allows it
    if (a2s_gru_seq_fwd_persist_ok(w_hh, gi_all, B, T, H, ws, ws_bytes))
        return a2s_gru_seq_fwd_persist_impl(st, gi_all, gi_bstride, gi_tstride, w_hh, b_hh, out, out_bstride, out_tstride, save, hn, B, T, H, reverse, ws, ws_bytes);
    hipError_t e = hipMemsetAsync(hbuf, 0, sizeof(float) * B * H, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "gru_seq_fwd memset: %s", hipGetErrorString(e));
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        const float* hp = hbuf + (long)(s & 1) * B * H;
        float* hq = (s == T - 1) ? hn : hbuf + (long)((s + 1) & 1) * B * H;
        if (gru_step_fusable(w_hh, H)) {        // one launch per step (see gru_step_fwd_fused)
            hipLaunchKernelGGL(gru_step_fwd_fused, dim3(H / 16, a2s_cdiv(B, 16)), dim3(256), 0, st, gi_all + (long)t * gi_tstride, gi_bstride, w_hh, b_hh,
                               hp, hq, out + (long)t * out_tstride, out_bstride, save ? save + (long)t * B * 4 * H : nullptr, B, H);
            A2S_CHECK_LAUNCH("gru_step_fwd_fused");
            continue;
        }
        // gh = h W_hh^T + b_hh   (M=B, N=3H, K=H; W_hh is (3H, H): B(k,n) = W[n*H + k])
        int rc = a2s_gemm_impl(st, B, 3 * H, H, 1.f, hp, H, 1, w_hh, 1, H, 0.f, gh, 3 * H, b_hh, 0, 1, 0, 0, 0, 0, ws, ws_bytes);
        if (rc) return rc;
        rc = a2s_gru_gates_fwd_impl(st, gi_all + (long)t * gi_tstride, gi_bstride, gh, 3 * H, hp, H, hq, H,
                                    out + (long)t * out_tstride, out_bstride, save ? save + (long)t * B * 4 * H : nullptr, B, H);
        if (rc) return rc;
    }
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- attention
// One workgroup per row (clip).  Pass 1 streams K (T x H): score_t = v . tanh(K_t + q); softmax over T in LDS;
// pass 2 streams enc (T x 2H): ctx = sum_t a_t enc_t.  Every byte of K and enc is read exactly once.
// (round 5: the width H is a run-time argument -- the reference constructor takes any hidden_size -- and only the number of key-row elements a lane
// holds, PER = ceil(H / 64), is a template parameter)
template <int PER>
__global__ __launch_bounds__(256) void attn_step_fwd(const float* __restrict__ Kmat, const float* __restrict__ enc,
                                                     const float* __restrict__ q, long ldq, const float* __restrict__ v,
                                                     float* __restrict__ ctx, long ldctx, float* __restrict__ ctx2, long ldctx2,
                                                     float* __restrict__ attw, int T, const int* __restrict__ n_done, int n_rows_total,
                                                     int n_clips, int H) {
    if (n_done && *n_done >= n_rows_total) return;      // greedy decode: every clip already emitted <eos>
    extern __shared__ __attribute__((aligned(16))) float sm[];   // T scores + 16 reduction slots
    float* sc = sm;
    float* red = sm + ((T + 3) & ~3);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = b % n_clips;                       // fused bars: row = bar * n_clips + clip
    const float* Kb = Kmat + (long)clip * T * H;
    const float* Eb = enc + (long)clip * T * 2 * H;
    float qv[PER], vv[PER];                               // PER = ceil(H / 64) elements of a K row per lane
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int j = lane + i * 64;
        qv[i] = j < H ? exp2x_clamped(q[(long)b * ldq + j]) : 0.f;      // E_q; Kmat holds the key image E_K = exp(2K)
        vv[i] = j < H ? v[j] : 0.f;
    }
    for (int t = wave; t < T; t += 4) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = lane + i * 64;
            if (j < H) s = fmaf(vv[i], tanh_ek(Kb[(long)t * H + j], qv[i]), s);
        }
        s = wave_sum(s);
        if (lane == 0) sc[t] = s;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int t = tid; t < T; t += 256) m = fmaxf(m, sc[t]);
    m = block_max(m, red);
    float l = 0.f;
    for (int t = tid; t < T; t += 256) { const float p = __expf(sc[t] - m); sc[t] = p; l += p; }
    l = block_sum(l, red);
    const float inv = 1.f / l;
    __syncthreads();
    for (int t = tid; t < T; t += 256) {
        const float w = sc[t] * inv;
        sc[t] = w;
        if (attw) attw[(long)b * T + t] = w;
    }
    __syncthreads();
    for (int d = tid; d < 2 * H; d += 256) {
        float acc = 0.f;
        int t = 0;
        for (; t + 4 <= T; t += 4) {
            const float e0 = Eb[(long)(t + 0) * 2 * H + d], e1 = Eb[(long)(t + 1) * 2 * H + d];
            const float e2 = Eb[(long)(t + 2) * 2 * H + d], e3 = Eb[(long)(t + 3) * 2 * H + d];
            acc = fmaf(sc[t], e0, acc); acc = fmaf(sc[t + 1], e1, acc);
            acc = fmaf(sc[t + 2], e2, acc); acc = fmaf(sc[t + 3], e3, acc);
        }
        for (; t < T; ++t) acc = fmaf(sc[t], Eb[(long)t * 2 * H + d], acc);
        ctx[(long)b * ldctx + d] = acc;
        if (ctx2) ctx2[(long)b * ldctx2 + d] = acc;
    }
}

int a2s_attn_step_fwd_split_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                                 float* ctx, long ldctx, float* ctx2, long ldctx2, float* attw, float* ws, int B, int T, int H,
                                 const int* n_done, int n_rows_total, const a2s_attn_rows* rows, a2s_attn_deferred* defer = nullptr);

int a2s_attn_step_fwd_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                           float* ctx, long ldctx, float* ctx2, long ldctx2, float* attw, int B, int T, int H,
                           const int* n_done, int n_rows_total, float* ws, const a2s_attn_rows* rows = nullptr, a2s_attn_deferred* defer = nullptr);
// rows (optional, split kernels only): which rows are computed and how they group by clip -- see a2s_attn_rows.  Used by the fused
// training step: once a row's remaining targets are all <pad> nothing that reaches the loss depends on it any more.
int a2s_attn_step_fwd_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                           float* ctx, long ldctx, float* ctx2, long ldctx2, float* attw, int B, int T, int H,
                           const int* n_done, int n_rows_total, float* ws, const a2s_attn_rows* rows, a2s_attn_deferred* defer) {
    if (defer) defer->G = 0;
    if (H == 256 && ws)
        return a2s_attn_step_fwd_split_impl(st, Kmat, enc, q, ldq, v, ctx, ldctx, ctx2, ldctx2, attw, ws, B, T, H, n_done, n_rows_total, rows, defer);
    // one-workgroup-per-row kernels: every row is computed (no skipping); fused bars only change which clip a row reads
    const int n_clips = rows ? rows->n_clips : B;
    const size_t shm = (((T + 3) & ~3) + 16) * sizeof(float);
    A2S_REQUIRE(H >= 1 && H <= 512, "attn_step_fwd: hidden_size must be in 1 .. 512 (got %d)", H);
#define A2S_ATTN_FWD(P) hipLaunchKernelGGL(attn_step_fwd<P>, dim3(B), dim3(256), shm, st, Kmat, enc, q, ldq, v, ctx, ldctx, ctx2, ldctx2, attw, T, n_done, n_rows_total, n_clips, H)
    switch ((H + 63) / 64) {
        case 1: A2S_ATTN_FWD(1); break;
        case 2: A2S_ATTN_FWD(2); break;
        case 3: A2S_ATTN_FWD(3); break;
        case 4: A2S_ATTN_FWD(4); break;
        case 5: A2S_ATTN_FWD(5); break;
        case 6: A2S_ATTN_FWD(6); break;
        case 7: A2S_ATTN_FWD(7); break;
        default: A2S_ATTN_FWD(8); break;
    }
#undef A2S_ATTN_FWD
    A2S_CHECK_LAUNCH("attn_step_fwd");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- step epilogue
// One wave per row: log_softmax over V logits -> probs[b, t, :]; argmax (lowest index on ties, as torch);
// next input token = gt[b,t] when teacher-forced else the argmax; its embedding (optionally dropped out)
// goes to the first E columns of the next step's GRU input row; EOS bookkeeping of reference
// models.py:411-419: every hit overwrites lengths[b] = t+1; n_done counts rows that have hit at least once.
struct StepFinArgs {
    const float* logits; long ldl;        // (R, V)
    float* probs; long probs_bstride;     // row b, step t at probs + b*probs_bstride + t*V
    const long long* gt; long gt_bstride; // ground-truth ids (row b at gt + b*gt_bstride), null in inference
    const float* emb;                     // (V, E) embedding table
    float* xnext; long ldx;               // next GRU input rows; token embedding -> columns [0, E)
    const uint8_t* drop; float inv_keep;  // (R, E) keep mask for the NEXT token or null
    int* argmax_out; long am_bstride;     // ids[b*am_bstride + t] (int32) or null
    int* eos_seen; long long* lengths; int* n_done; int* steps_exec;
    const int* t_base;                    // graph replay: step index = t + *t_base (null: t)
    const int* row_until;                 // training: rows finished at this step (t >= row_until[row]) keep their outputs untouched
    int n_clips;                          // rows per group (fused bars); teacher_force bit g applies to the rows of group g
    int R, V, E, t, teacher_force, eos_id, max_t;
};

__global__ __launch_bounds__(256) void note_step_finalize(StepFinArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.R) return;
    if (*a.n_done >= a.R && a.gt == nullptr) return;
    const int t = a.t + (a.t_base ? *a.t_base : 0);
    if (t >= a.max_t) return;                                   // a replayed chunk may overshoot the step budget
    const float* lg = a.logits + (long)row * a.ldl;
    const bool finished = a.row_until && t >= a.row_until[row];    // its bar's loop has ended in the reference (outputs stay zero) or only <pad> targets remain
    float m = -INFINITY; int mi = 0x7fffffff;
    for (int j = lane; j < a.V; j += 64) { const float x = lg[j]; if (x > m) { m = x; mi = j; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (!finished) {
        float s = 0.f;
        for (int j = lane; j < a.V; j += 64) s += expf(lg[j] - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        float* pr = a.probs + (long)row * a.probs_bstride + (long)t * a.V;
        for (int j = lane; j < a.V; j += 64) pr[j] = lg[j] - lse;
    }
    const long long g = a.gt ? a.gt[(long)row * a.gt_bstride + t] : -1;
    const int tf = (a.teacher_force >> (a.n_clips > 0 ? row / a.n_clips : 0)) & 1;
    const int next_id = (a.gt && tf) ? (int)g : mi;
    for (int j = lane; j < a.E; j += 64) {
        float e = a.emb[(long)next_id * a.E + j];
        if (a.drop) e = a.drop[(long)row * a.E + j] ? e * a.inv_keep : 0.f;
        a.xnext[(long)row * a.ldx + j] = e;
    }
    if (lane == 0 && !finished) {
        if (row == 0 && a.steps_exec) *a.steps_exec = t + 1;      // steps run in order on one stream
        if (a.argmax_out) a.argmax_out[(long)row * a.am_bstride + t] = mi;
        const bool hit = a.gt ? (g == a.eos_id) : (mi == a.eos_id);
        if (hit) {
            if (!a.eos_seen[row]) { a.eos_seen[row] = 1; atomicAdd(a.n_done, 1); }
            a.lengths[row] = t + 1;
        }
    }
}

int a2s_note_step_finalize_impl(hipStream_t st, const StepFinArgs& a) {
    hipLaunchKernelGGL(note_step_finalize, dim3(a2s_cdiv(a.R, 4)), dim3(256), 0, st, a);
    A2S_CHECK_LAUNCH("note_step_finalize");
    return A2S_OK;
}

// Row-wise log_softmax (+ argmax) for the time-signature / key heads: (R, V) -> (R, V) with row stride.
__global__ __launch_bounds__(256) void log_softmax_rows(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy,
                                                        int* __restrict__ argmax_out, int R, int V) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* lg = x + (long)row * ldx;
    float m = -INFINITY; int mi = 0x7fffffff;
    for (int j = lane; j < V; j += 64) { const float v = lg[j]; if (v > m) { m = v; mi = j; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    float s = 0.f;
    for (int j = lane; j < V; j += 64) s += expf(lg[j] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    for (int j = lane; j < V; j += 64) y[(long)row * ldy + j] = lg[j] - lse;
    if (argmax_out && lane == 0) argmax_out[row] = mi;
}

int a2s_log_softmax_rows_impl(hipStream_t st, const float* x, long ldx, float* y, long ldy, int* argmax_out, int R, int V) {
    hipLaunchKernelGGL(log_softmax_rows, dim3(a2s_cdiv(R, 4)), dim3(256), 0, st, x, ldx, y, ldy, argmax_out, R, V);
    A2S_CHECK_LAUNCH("log_softmax_rows");
    return A2S_OK;
}

// out[r, col0 + j] = table[ids[r]][j] (optionally dropped out) -- embedding rows into a wider row buffer.
__global__ void embed_rows(const float* __restrict__ table, const long long* __restrict__ ids64, const int* __restrict__ ids32,
                           long id_stride, int const_id, float* __restrict__ out, long ldo, int col0, int R, int E,
                           const uint8_t* __restrict__ drop, float inv_keep) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * E) return;
    const int r = (int)(idx / E), j = (int)(idx % E);
    const long id = ids64 ? ids64[(long)r * id_stride] : (ids32 ? ids32[(long)r * id_stride] : const_id);
    float e = table[id * E + j];
    if (drop) e = drop[idx] ? e * inv_keep : 0.f;
    out[(long)r * ldo + col0 + j] = e;
}

int a2s_embed_rows_impl(hipStream_t st, const float* table, const long long* ids64, const int* ids32, long id_stride,
                        int const_id, float* out, long ldo, int col0, int R, int E, const uint8_t* drop, float inv_keep) {
    hipLaunchKernelGGL(embed_rows, dim3(a2s_cdiv((long)R * E, 256)), dim3(256), 0, st, table, ids64, ids32, id_stride, const_id,
                       out, ldo, col0, R, E, drop, inv_keep);
    A2S_CHECK_LAUNCH("embed_rows");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- note decoder loop
// reference NoteDecoder.decode_notes (models.py:366-420) for one (bar, staff): `steps` iterations of
//   [q | gh] GEMMs -> attention -> gi GEMM -> GRU gates -> out GEMM -> step epilogue.
// Training: `steps` and the per-step teacher-forcing flags come from the host plan (they are functions of the
// ground truth and of Python's random stream only).  Greedy: steps = max_steps; every kernel of a step is a
// no-op once n_done == R, and the host polls n_done every `poll` steps to stop launching.
typedef a2s_note_dec_args NoteDecArgs;   // one definition only: the public C struct (include/a2s.h)

bool a2s_dec_step_fusable(int R, int H, int E, int V, const void* const* ptrs, int nptrs, const float* ws, size_t ws_floats, bool greedy = false);
int a2s_note_step_fused_fwd(hipStream_t st, const a2s_note_dec_args& a, int si, int so, int sv, int sv_next, int t, const int* t_base, int tf, bool last,
                            int nrows, const int* rowmap, const a2s_attn_deferred* defer = nullptr);
bool a2s_note_step_mid_ok(int H, int E, const void* const* ptrs, int nptrs);
int a2s_note_step_mid_gru(hipStream_t st, const a2s_note_dec_args& a, int si, int so, int sv, int sv_next, int nrows, const int* rowmap);
// rows the fused step of step t would cover: all R, or (training, finished rows skipped) the rows still running, a prefix of row_list
// both staves' sweeps of a step in one launch (round 6, further down): the pair's clip bookkeeping at this step and the geometry its partials use
struct AttnPairStep { const int* clip_order; const int* clip_rank; int n_clips; int n_active; int step; int G; int chunk; };
static int attn_pair_sweep(hipStream_t st, const NoteDecArgs& au, const NoteDecArgs& al, int sv, AttnPairStep& p);
static int attn_pair_combine(hipStream_t st, const NoteDecArgs& a, int si, int sv, const AttnPairStep& p);
int a2s_attn_pair_enabled(void);
static int note_step_rows(const NoteDecArgs& a, int t) { return (a.row_list && a.n_rows_active && t >= 0) ? a.n_rows_active[t] : a.R; }
// inside the pair loop (a2s_note_decoder_fwd_pair_impl) the few-row kernels take over later: at a2s_debug_set("attn_pair_fused_rows") rows (32) instead of
// "dec_fused_max_rows" (192) -- above that, a lockstep step with its shared sweep beats two few-row steps (192 -> 64: +1.2 ms per step, 64 -> 32: +2.2, 32 -> 16: +0.1,
// never: -16.8; profiles/r06_pair_fused_rows_ab.txt)
int a2s_attn_pair_fused_rows(void);
static thread_local int t_pair_rows_limit = -1;
struct PairRowsLimit { PairRowsLimit(int v) { t_pair_rows_limit = v; } ~PairRowsLimit() { t_pair_rows_limit = -1; } };
static bool note_step_fusable(const NoteDecArgs& a, int t = -1) {
    const void* ptrs[] = {a.x, a.h, a.o, a.q, a.w_ih, a.w_hh, a.out_w, a.attn_w};
    const int n = note_step_rows(a, t);
    if (t_pair_rows_limit >= 0 && n > t_pair_rows_limit) return false;
    return n > 0 && a2s_dec_step_fusable(n, a.H, a.E, a.V, ptrs, 8, a.step_ws, a.step_ws_floats, a.gt == nullptr && !a.gates);
}
// the launch-per-step loop's steps on the mid-size kernels (round 6)?  Not in graph-replay mode: its captured chunk computes the query at the start of a step
static bool note_step_mid(const NoteDecArgs& a, const int* t_base) {
    const void* ptrs[] = {a.x, a.h, a.o, a.q, a.w_ih, a.w_hh, a.out_w, a.attn_w};
    return !t_base && a2s_note_step_mid_ok(a.H, a.E, ptrs, 8);
}
// q of slot `sv` from the state in slot `si` (the fused and the mid-size paths compute every later query in the previous step's last launch)
static int enqueue_query(hipStream_t st, const NoteDecArgs& a, int si, int sv) {
    const int H2 = 2 * a.H;
    return a2s_gemm_impl(st, a.R, a.H, H2, 1.f, a.h + (long)si * a.R * H2, H2, 1, a.attn_w, 1, 2 * H2, 0.f, a.q + (long)sv * a.R * a.H, a.H, a.attn_b, 0, 1, 0,
                         0, 0, 0, a.gemm_ws, a.gemm_ws_bytes);
}

// one decode step: state read from slot `si`, written to slot `so` (slot = step index, or step parity in graph mode);
// per-step saved tensors (q, o, gates, attention weights) go to index `sv`.  fused: the few-row path of a2s_step.hip -- the query
// of slot sv must already be there (enqueue_query / the previous step), this step leaves the next one's in slot sv_next (!last).
static int enqueue_note_step(hipStream_t st, const NoteDecArgs& a, int si, int so, int sv, int t, const int* t_base, int tf,
                             bool fused = false, int sv_next = 0, bool last = false, const AttnPairStep* pair = nullptr) {
    const int H2 = 2 * a.H, ldx = a.E + H2;
    if (fused) {
        float* xs = a.x + (long)si * a.R * ldx;
        float* os = a.o + (long)sv * a.R * 2 * H2;
        a2s_attn_rows rows_v = {a.clip_order, a.clip_rank, a.row_until, a.n_clips > 0 ? a.n_clips : a.R, a.n_active ? a.n_active[t] : 0, t};
        a2s_attn_deferred defer;
        defer.G = 0;
        int rc = a2s_attn_step_fwd_impl(st, a.keys, a.enc, a.q + (long)sv * a.R * a.H, a.H, a.attn_v, xs + a.E, ldx, os + H2, 2 * H2,
                                        a.attw ? a.attw + (long)sv * a.R * a.T : nullptr, a.R, a.T, a.H, a.gt ? nullptr : a.n_done, a.R, a.attn_ws,
                                        a.n_active ? &rows_v : nullptr, (a.gt && a.n_active && !t_base) ? &defer : nullptr);
        if (rc) return rc;
        const int nrows = note_step_rows(a, t_base ? -1 : t);
        return a2s_note_step_fused_fwd(st, a, si, so, sv, sv_next, t, t_base, tf, last, nrows, nrows < a.R ? a.row_list : nullptr, defer.G > 0 ? &defer : nullptr);
    }
    // Rows the per-step products run on: all R, or -- late in a large call, when only a few leading clips still have an unfinished row --
    // the first m clips of every fused bar (batch = bars, row stride n_clips).  The elementwise kernels below keep running over all rows:
    // finished rows are never read again (models.py:401-419 stops writing them, their targets are <pad>).
    int gM = a.R, gB = 1;
    long gS = 0;
    if (a.m_active && a.n_clips > 0 && a2s_prefix_rows_ok(a.m_active[t], a.n_clips)) { gM = a.m_active[t]; gB = a.R / a.n_clips; gS = a.n_clips; }
    const float* hp = a.h + (long)si * a.R * H2;
    float* hq = a.h + (long)so * a.R * H2;
    float* xs = a.x + (long)si * a.R * ldx;
    float* qs = a.q + (long)sv * a.R * a.H;
    float* os = a.o + (long)sv * a.R * 2 * H2;
    int rc;
    a2s_attn_rows rows_v = {a.clip_order, a.clip_rank, a.row_until, a.n_clips > 0 ? a.n_clips : a.R, a.n_active ? a.n_active[t] : 0, t};
    const a2s_attn_rows* rows = a.n_active ? &rows_v : nullptr;
    if (note_step_mid(a, t_base)) {
        // round 6 (a2s_step.hip): the query of slot sv is already there (enqueue_query / the previous step, as on the few-row path); behind the
        // attention ONE launch for the GRU cell (dec_gru_mid: gh, gi, gates) and ONE for the logits and the next step's query (dec_outq_mid), over
        // the rows still running
        // (pair: the sweep of this step has been launched for both staves at once -- only this staff's combine is left)
        rc = pair ? attn_pair_combine(st, a, si, sv, *pair)
                  : a2s_attn_step_fwd_impl(st, a.keys, a.enc, qs, a.H, a.attn_v, xs + a.E, ldx, os + H2, 2 * H2,
                                           a.attw ? a.attw + (long)sv * a.R * a.T : nullptr, a.R, a.T, a.H, a.gt ? nullptr : a.n_done, a.R, a.attn_ws, rows);
        if (rc) return rc;
        const int nrows = note_step_rows(a, t);
        rc = a2s_note_step_mid_gru(st, a, si, so, sv, last ? -1 : sv_next, nrows, nrows < a.R ? a.row_list : nullptr);
        if (rc) return rc;
    } else {
    // q = h W_h^T + b   (W = [W_h | W_e], W_h = first 2H columns of the (H, 4H) matrix)
    rc = a2s_gemm_impl(st, gM, a.H, H2, 1.f, hp, H2, 1, a.attn_w, 1, 2 * H2, 0.f, qs, a.H, a.attn_b, 0, gB, gS * H2, 0, gS * a.H, 0, a.gemm_ws, a.gemm_ws_bytes);
    if (rc) return rc;
    // gh = h W_hh^T + b_hh
    rc = a2s_gemm_impl(st, gM, 3 * H2, H2, 1.f, hp, H2, 1, a.w_hh, 1, H2, 0.f, a.gh, 3 * H2, a.b_hh, 0, gB, gS * H2, 0, gS * 3 * H2, 0, a.gemm_ws, a.gemm_ws_bytes);
    if (rc) return rc;
    // attention -> ctx into x[si][:, E:] and o[sv][:, 2H:]
    rc = a2s_attn_step_fwd_impl(st, a.keys, a.enc, qs, a.H, a.attn_v, xs + a.E, ldx, os + H2, 2 * H2,
                                a.attw ? a.attw + (long)sv * a.R * a.T : nullptr, a.R, a.T, a.H, a.gt ? nullptr : a.n_done, a.R, a.attn_ws,
                                rows);
    if (rc) return rc;
    // gi = x W_ih^T + b_ih
    rc = a2s_gemm_impl(st, gM, 3 * H2, ldx, 1.f, xs, ldx, 1, a.w_ih, 1, ldx, 0.f, a.gi, 3 * H2, a.b_ih, 0, gB, gS * ldx, 0, gS * 3 * H2, 0, a.gemm_ws, a.gemm_ws_bytes);
    if (rc) return rc;
    // h' -> h[so] and o[sv][:, :2H]
    rc = a2s_gru_gates_fwd_impl(st, a.gi, 3 * H2, a.gh, 3 * H2, hp, H2, hq, H2, os, 2 * H2,
                                a.gates ? a.gates + (long)sv * a.R * 4 * H2 : nullptr, a.R, H2);
    if (rc) return rc;
    // logits = o W_out^T + b_out
    rc = a2s_gemm_impl(st, gM, a.V, 2 * H2, 1.f, os, 2 * H2, 1, a.out_w, 1, 2 * H2, 0.f, a.logits, a.V, a.out_b, 0, gB, gS * 2 * H2, 0, gS * a.V, 0, a.gemm_ws, a.gemm_ws_bytes);
    if (rc) return rc;
    }
    StepFinArgs f;
    f.logits = a.logits; f.ldl = a.V; f.probs = a.probs; f.probs_bstride = a.probs_bstride;
    f.gt = a.gt; f.gt_bstride = a.gt_bstride; f.emb = a.emb;
    f.xnext = a.x + (long)so * a.R * ldx; f.ldx = ldx;
    f.drop = a.drop ? a.drop + (long)so * a.R * a.E : nullptr; f.inv_keep = a.inv_keep;
    f.argmax_out = a.argmax_out; f.am_bstride = a.am_bstride;
    f.eos_seen = a.eos_seen; f.lengths = a.lengths; f.n_done = a.n_done; f.steps_exec = a.steps_exec;
    f.t_base = t_base;
    f.row_until = a.n_active ? a.row_until : nullptr; f.n_clips = a.n_clips > 0 ? a.n_clips : a.R;
    f.R = a.R; f.V = a.V; f.E = a.E; f.t = t; f.teacher_force = tf; f.eos_id = a.eos_id; f.max_t = a.steps;
    return a2s_note_step_finalize_impl(st, f);
}

__global__ void advance_counter(int* p, int inc) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += inc; }

// Greedy decode as a replayed hipGraph: the state ping-pongs between two slots (nothing is kept for a backward pass), the step
// index comes from a device counter, so ONE captured chunk of `chunk` steps serves the whole sequence; the host replays it and
// looks at the done counter after every replay.  Removes the per-launch host cost that dominates small-batch decoding.
static int note_decoder_greedy_graph(hipStream_t st, const NoteDecArgs& a, int* steps_done) {
    int chunk = a.poll > 0 ? a.poll : 16;
    if (chunk & 1) ++chunk;                                   // even: the state is back in slot 0 after every replay
    hipError_t e = hipMemsetAsync(a.t_base, 0, sizeof(int), st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "greedy graph memset: %s", hipGetErrorString(e));
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const bool fused = note_step_fusable(a);
    int rc = fused ? enqueue_query(st, a, 0, 0) : A2S_OK;     // the very first query; every later one is left behind by the previous step
    if (rc) return rc;
    e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    for (int j = 0; j < chunk && rc == A2S_OK; ++j) rc = enqueue_note_step(st, a, j & 1, (j + 1) & 1, 0, j, a.t_base, 0, fused, 0, false);
    if (rc == A2S_OK) hipLaunchKernelGGL(advance_counter, dim3(1), dim3(64), 0, st, a.t_base, chunk);
    e = hipStreamEndCapture(st, &graph);
    if (rc != A2S_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(graph); A2S_FAIL(A2S_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    int launched = 0;
    while (launched < a.steps) {
        e = hipGraphLaunch(exec, st);
        if (e != hipSuccess) break;
        launched += chunk;
        int done = 0;
        e = hipMemcpyAsync(&done, a.n_done, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess || done >= a.R) break;
    }
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "greedy graph replay: %s", hipGetErrorString(e));
    if (steps_done) *steps_done = launched < a.steps ? launched : a.steps;
    return A2S_OK;
}

bool a2s_note_decoder_fwd_persist_ok(const a2s_note_dec_args& a);
int a2s_note_decoder_fwd_persist(hipStream_t st, const a2s_note_dec_args& a, int* steps_done);

// Tail steps on the few-row kernels write only the rows still running.  What the backward pass reads of the others (operands of its
// weight-gradient products over all rows and steps, the saved gates) must be finite: everything behind slot 0 starts as zeros
// (~6 GB per training step at B = 256, ~1.3 ms; issued here and not by the Python host: see engine.Engine._decode_staff).
static int note_decoder_zero_fill(hipStream_t st, const NoteDecArgs& a) {
    if (!(a.row_list && a.n_rows_active && a.steps > 0)) return A2S_OK;
    const long H2 = 2L * a.H, ldx = a.E + H2, n = a.steps, R = a.R;
    hipError_t e = hipMemsetAsync(a.h + R * H2, 0, sizeof(float) * n * R * H2, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.x + R * ldx, 0, sizeof(float) * n * R * ldx, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.q, 0, sizeof(float) * n * R * a.H, st);
    if (e == hipSuccess) e = hipMemsetAsync(a.o, 0, sizeof(float) * n * R * 2 * H2, st);
    if (e == hipSuccess && a.gates) e = hipMemsetAsync(a.gates, 0, sizeof(float) * n * R * 4 * H2, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder memset: %s", hipGetErrorString(e));
    return A2S_OK;
}

int a2s_note_decoder_fwd_impl(hipStream_t st, const NoteDecArgs& a, int* steps_done) {
    // few clips: one persistent launch for the whole call (a2s_dec_persist.hip)
    if (a2s_note_decoder_fwd_persist_ok(a)) return a2s_note_decoder_fwd_persist(st, a, steps_done);
    // stream capture is not allowed on the legacy default stream: callers that want the graph path run on a created stream
    if (!a.gt && a.use_graph && a.t_base && !a.gates && !a.attw && !a.drop && st != nullptr) return note_decoder_greedy_graph(st, a, steps_done);
    { const int rc0 = note_decoder_zero_fill(st, a); if (rc0) return rc0; }
    int s = 0;
    // The few-row step kernels take over as soon as the rows still running fit them (the whole call when it is small; the tail of a large
    // training call otherwise: the handful of full-length rows then decode in 4 launches per step instead of 12 over every row).  The
    // first fused step finds no query left behind by a fused predecessor: it is computed for all rows first.
    bool prev_q = false;                     // did the previous step leave this step's query behind?
    const bool mid = note_step_mid(a, nullptr);
    for (; s < a.steps; ++s) {
        const bool fused = note_step_fusable(a, s);
        if ((fused || mid) && !prev_q) { int rc = enqueue_query(st, a, s, s); if (rc) return rc; }
        prev_q = fused || mid;
        int rc = enqueue_note_step(st, a, s, s + 1, s, s, nullptr, a.tf_flags ? a.tf_flags[s] : 0, fused, s + 1, s + 1 == a.steps);
        if (rc) return rc;
        if (!a.gt && a.poll > 0 && ((s + 1) % a.poll == 0) && s + 1 < a.steps) {
            int done = 0;   // greedy only: one small D2H + sync per `poll` steps
            hipError_t e = hipMemcpyAsync(&done, a.n_done, sizeof(int), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder poll: %s", hipGetErrorString(e));
            if (done >= a.R) { ++s; break; }
        }
    }
    if (steps_done) *steps_done = s;
    return A2S_OK;
}

// The two NoteDecoders of a segment (models.py:261-275) decoded by ONE host loop on their two streams: while both staves run a step on the
// mid-size kernels, the step's attention sweep is one launch for both (attn_fwd_split256_pair on the upper staff's stream: it waits for the lower
// staff's query, the lower staff's stream waits for it); everything else of a step stays per staff on the staff's own stream, and every step that
// does not qualify (one staff has ended, or runs its tail on the few-row kernels) is the single-staff step unchanged.  pair_n_active: HOST array,
// max(steps) ints -- clips with an unfinished row of either staff at step t, a prefix of pair_order.
int a2s_note_decoder_fwd_pair_impl(hipStream_t su, hipStream_t sl, const NoteDecArgs& au, const NoteDecArgs& al, const int* pair_order,
                                   const int* pair_rank, const int* pair_n_active, int* done_u, int* done_l) {
    const NoteDecArgs* as[2] = {&au, &al};
    hipStream_t sts[2] = {su, sl};
    const bool can_pair = a2s_attn_pair_enabled() && su != sl && pair_order && pair_rank && pair_n_active && au.gt && al.gt && au.n_active && al.n_active &&
                          au.n_clips > 0 && au.n_clips == al.n_clips && au.R == al.R && au.T == al.T && au.H == 256 && al.H == 256 && au.enc == al.enc &&
                          au.attn_ws && al.attn_ws && !a2s_note_decoder_fwd_persist_ok(au) && !a2s_note_decoder_fwd_persist_ok(al) &&
                          note_step_mid(au, nullptr) && note_step_mid(al, nullptr);
    if (!can_pair) {
        int rc = a2s_note_decoder_fwd_impl(su, au, done_u);
        return rc ? rc : a2s_note_decoder_fwd_impl(sl, al, done_l);
    }
    static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k)
        if (!ev[k]) { const hipError_t e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming); if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_fwd_pair: hipEventCreate: %s", hipGetErrorString(e)); }
    const PairRowsLimit limit(a2s_attn_pair_fused_rows());
    for (int k = 0; k < 2; ++k) { const int rc = note_decoder_zero_fill(sts[k], *as[k]); if (rc) return rc; }
    bool prev_q[2] = {false, false};
    const int nmax = au.steps > al.steps ? au.steps : al.steps;
    for (int s = 0; s < nmax; ++s) {
        bool fused[2] = {false, false}, in[2];
        for (int k = 0; k < 2; ++k) {
            in[k] = s < as[k]->steps;
            if (!in[k]) continue;
            fused[k] = note_step_fusable(*as[k], s);
            if (!prev_q[k]) { const int rc = enqueue_query(sts[k], *as[k], s, s); if (rc) return rc; }       // (every step here is a fused or a mid-size one)
            prev_q[k] = true;
        }
        AttnPairStep p = {pair_order, pair_rank, au.n_clips, pair_n_active[s], s, 1, au.T};
        const bool joint = in[0] && in[1] && !fused[0] && !fused[1] && p.n_active > 0 && au.n_active[s] > 0 && al.n_active[s] > 0;
        if (joint) {
            hipError_t e = hipEventRecord(ev[1], sl);                       // the lower staff's query of this step
            if (e == hipSuccess) e = hipStreamWaitEvent(su, ev[1], 0);
            if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_fwd_pair: event: %s", hipGetErrorString(e));
            const int rc = attn_pair_sweep(su, au, al, s, p);
            if (rc) return rc;
            e = hipEventRecord(ev[0], su);
            if (e == hipSuccess) e = hipStreamWaitEvent(sl, ev[0], 0);
            if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_fwd_pair: event: %s", hipGetErrorString(e));
        }
        for (int k = 0; k < 2; ++k) {
            if (!in[k]) continue;
            const NoteDecArgs& a = *as[k];
            const int rc = enqueue_note_step(sts[k], a, s, s + 1, s, s, nullptr, a.tf_flags ? a.tf_flags[s] : 0, fused[k], s + 1, s + 1 == a.steps, joint ? &p : nullptr);
            if (rc) return rc;
        }
    }
    if (done_u) *done_u = au.steps;
    if (done_l) *done_l = al.steps;
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- staff embedding
static int g_staff_emb_fast = -1;       // the E = 16, S = 32 kernels (a2s_debug_set("staff_emb_fast", 0): the generic ones)
void a2s_staff_emb_fast_set(int v) { g_staff_emb_fast = v ? 1 : 0; }
int a2s_staff_emb_fast_enabled(void) {
    if (g_staff_emb_fast < 0) g_staff_emb_fast = 1;
    return g_staff_emb_fast;
}
// reference get_staff_token_* (models.py:164-189): packed bi-GRU (E -> S) final states.  One workgroup per
// (row, direction); the 3S x (E+S) weights live in LDS and the whole (<= 398 step) recurrence runs in-kernel.
// ids come as int32 (argmax buffer) or int64 (ground truth).  Saves per-step h for the backward pass.
__global__ __launch_bounds__(128) void staff_emb_fwd(const float* __restrict__ note_emb, const float* __restrict__ w_ih_f,
                                                     const float* __restrict__ w_hh_f, const float* __restrict__ b_ih_f,
                                                     const float* __restrict__ b_hh_f, const float* __restrict__ w_ih_r,
                                                     const float* __restrict__ w_hh_r, const float* __restrict__ b_ih_r,
                                                     const float* __restrict__ b_hh_r, const long long* __restrict__ ids64,
                                                     const int* __restrict__ ids32, long id_bstride,
                                                     const long long* __restrict__ lengths, long len_stride,
                                                     float* __restrict__ out, long ldo, int col0, float* __restrict__ hsave,
                                                     int maxlen, int E, int S) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const float* w_ih = dir ? w_ih_r : w_ih_f; const float* w_hh = dir ? w_hh_r : w_hh_f;
    const float* b_ih = dir ? b_ih_r : b_ih_f; const float* b_hh = dir ? b_hh_r : b_hh_f;
    float* Wi = sm;                       // (3S, E)
    float* Wh = Wi + 3 * S * E;           // (3S, S)
    float* bi = Wh + 3 * S * S;           // 3S
    float* bh = bi + 3 * S;               // 3S
    float* h = bh + 3 * S;                // S
    float* xe = h + S;                    // E
    float* g = xe + E;                    // 6S : gi | gh
    for (int i = tid; i < 3 * S * E; i += blockDim.x) Wi[i] = w_ih[i];
    for (int i = tid; i < 3 * S * S; i += blockDim.x) Wh[i] = w_hh[i];
    for (int i = tid; i < 3 * S; i += blockDim.x) { bi[i] = b_ih[i]; bh[i] = b_hh[i]; }
    for (int i = tid; i < S; i += blockDim.x) h[i] = 0.f;
    int len = (int)lengths[(long)b * len_stride];
    len = max(0, min(len, maxlen));
    __syncthreads();
    for (int s = 0; s < len; ++s) {
        const int t = dir ? len - 1 - s : s;
        const long id = ids64 ? ids64[(long)b * id_bstride + t] : ids32[(long)b * id_bstride + t];
        for (int i = tid; i < E; i += blockDim.x) xe[i] = note_emb[id * E + i];
        __syncthreads();
        for (int r = tid; r < 6 * S; r += blockDim.x) {
            float acc;
            if (r < 3 * S) { acc = bi[r]; for (int k = 0; k < E; ++k) acc = fmaf(Wi[r * E + k], xe[k], acc); }
            else { const int rr = r - 3 * S; acc = bh[rr]; for (int k = 0; k < S; ++k) acc = fmaf(Wh[rr * S + k], h[k], acc); }
            g[r] = acc;
        }
        __syncthreads();
        if (tid < S) {
            const float rg = fast_sigmoid(g[tid] + g[3 * S + tid]);
            const float zg = fast_sigmoid(g[S + tid] + g[4 * S + tid]);
            const float ng = fast_tanh(g[2 * S + tid] + rg * g[5 * S + tid]);
            const float hn = (1.f - zg) * ng + zg * h[tid];
            h[tid] = hn;
            if (hsave) hsave[(((long)b * 2 + dir) * maxlen + s) * S + tid] = hn;
        }
        __syncthreads();
    }
    for (int i = tid; i < S; i += blockDim.x) out[(long)b * ldo + col0 + dir * S + i] = h[i];
}

// The same recurrence for the model's sizes (E = 16, S = 32), 192 threads: thread r < 96 owns row r of W_ih (16 registers), thread
// 96 + r row r of W_hh (32 registers) -- the weights never pass through LDS (the generic kernel reads W[r * E + k] with r = thread:
// a 16- or 32-way bank conflict on every load, 3.9 us per step of a <= 398-step chain) --, the token ids of the row sit in LDS, the
// embedding row of step s + 1 is fetched while step s computes, x_s and h broadcast from LDS as 16-byte reads.  Two barriers per step.
__global__ __launch_bounds__(192) void staff_emb_fwd_e16s32(const float* __restrict__ note_emb, const float* __restrict__ w_ih_f,
                                                            const float* __restrict__ w_hh_f, const float* __restrict__ b_ih_f,
                                                            const float* __restrict__ b_hh_f, const float* __restrict__ w_ih_r,
                                                            const float* __restrict__ w_hh_r, const float* __restrict__ b_ih_r,
                                                            const float* __restrict__ b_hh_r, const long long* __restrict__ ids64,
                                                            const int* __restrict__ ids32, long id_bstride,
                                                            const long long* __restrict__ lengths, long len_stride,
                                                            float* __restrict__ out, long ldo, int col0, float* __restrict__ hsave, int maxlen) {
    constexpr int E = 16, S = 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xe = sm;                        // 16
    float* hb = xe + E;                    // 32
    float* g = hb + S;                     // 192: gi | gh
    int* lid = reinterpret_cast<int*>(g + 6 * S);      // maxlen
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const bool gi_role = tid < 3 * S;
    const int r = gi_role ? tid : tid - 3 * S;
    const float* wsrc = gi_role ? (dir ? w_ih_r : w_ih_f) + r * E : (dir ? w_hh_r : w_hh_f) + r * S;
    float w[S];
#pragma unroll
    for (int k = 0; k < S; ++k) w[k] = (gi_role && k >= E) ? 0.f : wsrc[k];
    const float bias = gi_role ? (dir ? b_ih_r : b_ih_f)[r] : (dir ? b_hh_r : b_hh_f)[r];
    int len = (int)lengths[(long)b * len_stride];
    len = max(0, min(len, maxlen));
    for (int i = tid; i < len; i += 192) {
        const int t = dir ? len - 1 - i : i;
        lid[i] = ids64 ? (int)ids64[(long)b * id_bstride + t] : ids32[(long)b * id_bstride + t];
    }
    if (tid < S) hb[tid] = 0.f;
    float h = 0.f;                          // thread j < 32: h[j]
    __syncthreads();
    float xe_n = (tid < E && len > 0) ? note_emb[(long)lid[0] * E + tid] : 0.f;
    for (int s = 0; s < len; ++s) {
        if (tid < E) {
            xe[tid] = xe_n;
            if (s + 1 < len) xe_n = note_emb[(long)lid[s + 1] * E + tid];
        }
        __syncthreads();                    // x_s and h_{s-1} are in LDS
        float acc = bias;
        if (gi_role) {
#pragma unroll
            for (int k4 = 0; k4 < E / 4; ++k4) {
                const f32x4 x4 = reinterpret_cast<const f32x4*>(xe)[k4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = fmaf(w[4 * k4 + c], x4[c], acc);
            }
        } else {
#pragma unroll
            for (int k4 = 0; k4 < S / 4; ++k4) {
                const f32x4 h4 = reinterpret_cast<const f32x4*>(hb)[k4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = fmaf(w[4 * k4 + c], h4[c], acc);
            }
        }
        g[tid] = acc;
        __syncthreads();
        if (tid < S) {
            const float rg = fast_sigmoid(g[tid] + g[3 * S + tid]);
            const float zg = fast_sigmoid(g[S + tid] + g[4 * S + tid]);
            const float ng = fast_tanh(g[2 * S + tid] + rg * g[5 * S + tid]);
            h = (1.f - zg) * ng + zg * h;
            hb[tid] = h;
            if (hsave) hsave[(((long)b * 2 + dir) * maxlen + s) * S + tid] = h;
        }
    }
    if (tid < S) out[(long)b * ldo + col0 + dir * S + tid] = h;
}

int a2s_staff_emb_fwd_impl(hipStream_t st, const float* note_emb, const float* const* w /* 8 GRU tensors f then r */,
                           const long long* ids64, const int* ids32, long id_bstride, const long long* lengths,
                           long len_stride, float* out, long ldo, int col0, float* hsave, int R, int maxlen, int E, int S) {
    A2S_REQUIRE((ids64 != nullptr) != (ids32 != nullptr), "staff_emb_fwd: exactly one of ids64/ids32");
    if (E == 16 && S == 32 && a2s_staff_emb_fast_enabled()) {
        const size_t shm16 = sizeof(float) * (E + S + 6 * S) + sizeof(int) * (size_t)maxlen;
        hipLaunchKernelGGL(staff_emb_fwd_e16s32, dim3(R, 2), dim3(192), shm16, st, note_emb, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7],
                           ids64, ids32, id_bstride, lengths, len_stride, out, ldo, col0, hsave, maxlen);
        A2S_CHECK_LAUNCH("staff_emb_fwd_e16s32");
        return A2S_OK;
    }
    const size_t shm = sizeof(float) * (3 * S * E + 3 * S * S + 6 * S + S + E + 6 * S);
    hipLaunchKernelGGL(staff_emb_fwd, dim3(R, 2), dim3(128), shm, st, note_emb, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7],
                       ids64, ids32, id_bstride, lengths, len_stride, out, ldo, col0, hsave, maxlen, E, S);
    A2S_CHECK_LAUNCH("staff_emb_fwd");
    return A2S_OK;
}

// ---- combine of a row's G partials, shared by the stand-alone combine kernel and by the split kernels' fused tail (the workgroup that
// finishes LAST for its clip -- ticket counter, release / acquire through L2 as in dec_out_step -- merges the partials itself: one launch
// and one dependent-launch gap less per decode step; on the long-clip chain the combine was 7 us + ~4.5 us of gap in a 92 us step).
// pb: the row's G partials [m, l, pad, pad, ctx(2H)]; wgt: 80 floats of LDS.
struct AttnCombineOut { float* ctx; float* ctx2; float* attw; int T; };
__device__ __forceinline__ void attn_combine_row(const volatile float* pb, int G, const AttnCombineOut& o, float* wgt) {
    constexpr int H = 256, PS = 2 * H + 4;
    const int tid = threadIdx.x;
    // Round 4: EVERYTHING the row needs is requested before anything is waited for -- the partial statistics, the first 16 partial contexts
    // (G <= 16 by default: all of them) and the thread's share of the saved scores.  The kernel sits on every decode chain and under the other
    // streams' traffic a dependent round trip costs ~7 us: as statistics -> barrier -> contexts -> scores it took 23.8 us (median on the
    // long-clip group's queue, profiles/r04_trace_overlap.txt) for a few hundred KB.
    const bool have = tid < 64 && tid < G;
    const float mg = have ? pb[(long)tid * PS] : -INFINITY, lg = have ? pb[(long)tid * PS + 1] : 0.f;
    float p0[16], p1[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const bool ok = u < G;
        p0[u] = ok ? pb[(long)u * PS + 4 + tid] : 0.f;
        p1[u] = ok ? pb[(long)u * PS + 4 + 256 + tid] : 0.f;
    }
    volatile float* aw = o.attw;                      // raw scores written by the clip's other workgroups
    float sv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sv[u] = (o.attw && tid + 256 * u < o.T) ? aw[tid + 256 * u] : 0.f;
    __syncthreads();                                  // wgt free (previous row / the caller's scratch)
    if (tid < 64) {                                   // G <= 64 (a2s_attn_max_split)
        float m, inv_l;
        wgt[tid] = attn_merge_weight(mg, lg, have, m, inv_l);
        if (tid < 16) wgt[64 + tid] = 0.f;
        if (tid == 0) { wgt[79] = m; wgt[78] = inv_l; }
    }
    __syncthreads();
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc0 = fmaf(p0[u], wgt[u], acc0); acc1 = fmaf(p1[u], wgt[u], acc1); }
    for (int g0 = 16; g0 < G; g0 += 16) {             // (only if a2s_attn_max_split() > 16)
        float q0[16], q1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool ok = g0 + u < G;
            q0[u] = ok ? pb[(long)(g0 + u) * PS + 4 + tid] : 0.f;
            q1[u] = ok ? pb[(long)(g0 + u) * PS + 4 + 256 + tid] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { acc0 = fmaf(q0[u], wgt[g0 + u], acc0); acc1 = fmaf(q1[u], wgt[g0 + u], acc1); }
    }
    o.ctx[tid] = acc0; o.ctx[256 + tid] = acc1;
    if (o.ctx2) { o.ctx2[tid] = acc0; o.ctx2[256 + tid] = acc1; }
    if (o.attw) {
        const float m = wgt[79], inv = wgt[78];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (tid + 256 * u < o.T) aw[tid + 256 * u] = __expf(sv[u] - m) * inv;
        for (int t0 = tid + 8 * 256; t0 < o.T; t0 += 8 * 256) {       // (T > 2048 only)
            float sw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) sw[u] = (t0 + 256 * u < o.T) ? aw[t0 + 256 * u] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (t0 + 256 * u < o.T) aw[t0 + 256 * u] = __expf(sw[u] - m) * inv;
        }
    }
}
__device__ __forceinline__ void attn_zero_row(const AttnCombineOut& o) {       // skipped row: a finite, well-defined context (zeros)
    for (int d = threadIdx.x; d < 512; d += 256) { o.ctx[d] = 0.f; if (o.ctx2) o.ctx2[d] = 0.f; }
    if (o.attw) for (int t = threadIdx.x; t < o.T; t += 256) o.attw[t] = 0.f;
}
// true in exactly one of the G workgroups of a slot: the last one to arrive, after everybody's partials are visible device-wide
__device__ __forceinline__ bool attn_last_arrival(int* ticket, int G, int* flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int t = atomicAdd(ticket, 1);
        *flag = (t == G - 1);
        if (t == G - 1) *ticket = 0;                  // ready for the next launch
    }
    __syncthreads();
    const bool last = *flag != 0;
    if (last) __threadfence();
    return last;
}
// what the fused tail needs besides the partials (tickets == nullptr: the stand-alone combine kernel follows)
struct AttnFusedTail { int* tickets; float* ctx; long ldctx; float* ctx2; long ldctx2; int n_slots_active; int n_rows; };

// =========================================================================================== split-T attention (H = 256)
// The one-workgroup-per-clip kernel above is latency-bound (3.69 MB per workgroup, measured 15 GB/s per workgroup, 253 us per
// launch whatever the batch).  Here the 1201 frames of a clip are split over G workgroups (flash-decoding style): each
// streams its chunk of K and enc ONCE with 16-byte loads and produces a partial softmax (m_g, l_g, ctx_g); a small combine
// kernel merges the G partials per clip.  Grid = B*G workgroups, G chosen so the grid is a few waves of the 256 CUs.
//   partial layout per (clip, g): [m, l, pad, pad, ctx(2H)]  -> (2H + 4) floats
template <bool NT>
__global__ __launch_bounds__(256) void attn_fwd_split256(const float* __restrict__ Kmat, const float* __restrict__ enc,
                                                         const float* __restrict__ q, long ldq, const float* __restrict__ v,
                                                         float* __restrict__ partial, float* __restrict__ scores, int T, int G, int chunk,
                                                         const int* __restrict__ n_done, int n_rows_total,
                                                         const int* __restrict__ row_order, AttnFusedTail ft) {
    if (n_done && *n_done >= n_rows_total) return;
    constexpr int H = 256;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int last_flag;
    float* pw = sm;                                   // chunk weights exp(s - m_g)
    float* red = sm + chunk;                          // 16 + 2 * 128 * 4 floats (reduction scratch)
    if (ft.tickets && (int)blockIdx.x >= ft.n_slots_active * G) {        // fused tail: one extra workgroup per skipped row zero-fills it
        const int zs = ft.n_slots_active + (int)blockIdx.x - ft.n_slots_active * G;
        const int zb = row_order ? row_order[zs] : zs;
        attn_zero_row(AttnCombineOut{ft.ctx + (long)zb * ft.ldctx, ft.ctx2 ? ft.ctx2 + (long)zb * ft.ldctx2 : nullptr, scores ? scores + (long)zb * T : nullptr, T});
        return;
    }
    const int slot = blockIdx.x / G, g = blockIdx.x % G;          // slot: position among the rows this launch covers
    const int b = row_order ? row_order[slot] : slot;
    const int t0 = g * chunk, t1 = min(T, t0 + chunk), n = t1 - t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Kb = Kmat + ((long)b * T + t0) * H;
    const float* Eb = enc + ((long)b * T + t0) * 2 * H;
    f32x4 q4 = *reinterpret_cast<const f32x4*>(q + (long)b * ldq + lane * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = exp2x_clamped(q4[c]);                               // E_q; Kmat holds the key image E_K = exp(2K)
    const f32x4 v4 = {v[lane * 4], v[lane * 4 + 1], v[lane * 4 + 2], v[lane * 4 + 3]};      // parameter (view of the flat buffer): 4-byte aligned only
    // ---- pass 1: scores of the chunk; one wave per frame, 4 frames in flight per wave
    for (int r = wave * 4; r < n; r += 16) {
        f32x4 k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            k[u] = (r + u < n) ? ld_kv<NT>(Kb + (long)(r + u) * H + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        float s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[u] = v4[0] * tanh_ek(k[u][0], q4[0]) + v4[1] * tanh_ek(k[u][1], q4[1])
                 + v4[2] * tanh_ek(k[u][2], q4[2]) + v4[3] * tanh_ek(k[u][3], q4[3]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = wave_sum_lane63(s[u]);
        if (lane == 63) {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (r + u < n) pw[r + u] = s[u];
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, pw[i]);
    m = block_max(m, red);
    float l = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float sc = pw[i];
        if (scores) scores[(long)b * T + t0 + i] = sc;       // raw score; normalised by the combine kernel
        const float p = __expf(sc - m);
        pw[i] = p; l += p;
    }
    l = block_sum(l, red);
    __syncthreads();
    // ---- pass 2: ctx_g = sum_i pw[i] * enc[i, :]; thread = (float4 column c4, row parity)
    const int c4 = tid & 127, rp = tid >> 7;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int i = rp;
    for (; i + 6 < n; i += 8) {
        const f32x4 e0 = ld_kv<NT>(Eb + (long)(i + 0) * 2 * H + c4 * 4);
        const f32x4 e1 = ld_kv<NT>(Eb + (long)(i + 2) * 2 * H + c4 * 4);
        const f32x4 e2 = ld_kv<NT>(Eb + (long)(i + 4) * 2 * H + c4 * 4);
        const f32x4 e3 = ld_kv<NT>(Eb + (long)(i + 6) * 2 * H + c4 * 4);
        const float w0 = pw[i], w1 = pw[i + 2], w2 = pw[i + 4], w3 = pw[i + 6];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += w0 * e0[c] + w1 * e1[c] + w2 * e2[c] + w3 * e3[c];
    }
    for (; i < n; i += 2) {
        const f32x4 e0 = ld_kv<NT>(Eb + (long)i * 2 * H + c4 * 4);
        const float w0 = pw[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += w0 * e0[c];
    }
    f32x4* red4 = reinterpret_cast<f32x4*>(red + 16);
    if (rp == 1) red4[c4] = acc;
    __syncthreads();
    float* pout = partial + ((long)slot * G + g) * (2 * H + 4);
    if (rp == 0) {
        const f32x4 o = red4[c4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += o[c];
        *reinterpret_cast<f32x4*>(pout + 4 + c4 * 4) = acc;
    }
    if (tid == 0) { pout[0] = m; pout[1] = l; }
    if (!ft.tickets) return;
    if (!attn_last_arrival(ft.tickets + slot, G, &last_flag)) return;
    attn_combine_row(partial + (long)slot * G * (2 * H + 4), G,
                     AttnCombineOut{ft.ctx + (long)b * ft.ldctx, ft.ctx2 ? ft.ctx2 + (long)b * ft.ldctx2 : nullptr, scores ? scores + (long)b * T : nullptr, T}, red);
}

// Fused bars: NQ rows (bars) of one clip per workgroup -- the clip's K and enc chunk is streamed ONCE and applied to every unfinished
// row of the clip, so the HBM bytes per decoded row drop by the number of rows sharing the clip.  Same two passes and the same
// partial layout as attn_fwd_split256; partial of (slot, j, g) at ((slot * NQ + j) * G + g).
template <int NQ, bool NT>
__global__ __launch_bounds__(256) void attn_fwd_split256_mq(const float* __restrict__ Kmat, const float* __restrict__ enc,
                                                            const float* __restrict__ q, long ldq, const float* __restrict__ v,
                                                            float* __restrict__ partial, float* __restrict__ scores, int T, int G, int chunk,
                                                            const int* __restrict__ clip_order, const int* __restrict__ row_until,
                                                            int step, int n_clips, AttnFusedTail ft) {
    constexpr int H = 256;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ int last_flag;
    float* pw = sm;                                   // NQ x chunk weights
    float* red = sm + NQ * chunk;                     // 16 + NQ * 128 * 4 floats
    if (ft.tickets && (int)blockIdx.x >= ft.n_slots_active * G) {        // fused tail: one extra workgroup per clip without unfinished rows
        const int zs = ft.n_slots_active + (int)blockIdx.x - ft.n_slots_active * G;
        const int zb = clip_order ? clip_order[zs] : zs;
        for (int j = 0; j < NQ; ++j) {
            const long row = (long)j * n_clips + zb;
            attn_zero_row(AttnCombineOut{ft.ctx + row * ft.ldctx, ft.ctx2 ? ft.ctx2 + row * ft.ldctx2 : nullptr, scores ? scores + row * T : nullptr, T});
        }
        return;
    }
    const int slot = blockIdx.x / G, g = blockIdx.x % G;
    const int b = clip_order ? clip_order[slot] : slot;
    const int t0 = g * chunk, t1 = min(T, t0 + chunk), n = t1 - t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool on[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) on[j] = !row_until || step < row_until[j * n_clips + b];
    const float* Kb = Kmat + ((long)b * T + t0) * H;
    const float* Eb = enc + ((long)b * T + t0) * 2 * H;
    f32x4 q4[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
        q4[j] = on[j] ? *reinterpret_cast<const f32x4*>(q + ((long)j * n_clips + b) * ldq + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) q4[j][c] = exp2x_clamped(q4[j][c]);                    // E_q; Kmat holds the key image E_K = exp(2K)
    const f32x4 v4 = {v[lane * 4], v[lane * 4 + 1], v[lane * 4 + 2], v[lane * 4 + 3]};
    // ---- pass 1: scores; one wave per frame, 4 frames in flight, every frame scored against the NQ queries
    for (int r = wave * 4; r < n; r += 16) {
        f32x4 k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            k[u] = (r + u < n) ? ld_kv<NT>(Kb + (long)(r + u) * H + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (!on[j]) continue;
            float sj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                sj[u] = v4[0] * tanh_ek(k[u][0], q4[j][0]) + v4[1] * tanh_ek(k[u][1], q4[j][1])
                      + v4[2] * tanh_ek(k[u][2], q4[j][2]) + v4[3] * tanh_ek(k[u][3], q4[j][3]);
#pragma unroll
            for (int u = 0; u < 4; ++u) sj[u] = wave_sum_lane63(sj[u]);
            if (lane == 63) {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (r + u < n) pw[j * chunk + r + u] = sj[u];
            }
        }
    }
    __syncthreads();
    float mj[NQ], lj[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        mj[j] = 0.f; lj[j] = 0.f;
        if (!on[j]) continue;                          // uniform over the workgroup
        float m = -INFINITY;
        for (int i = tid; i < n; i += 256) m = fmaxf(m, pw[j * chunk + i]);
        m = block_max(m, red);
        float l = 0.f;
        for (int i = tid; i < n; i += 256) {
            const float sc = pw[j * chunk + i];
            if (scores) scores[((long)j * n_clips + b) * T + t0 + i] = sc;
            const float p = __expf(sc - m);
            pw[j * chunk + i] = p; l += p;
        }
        l = block_sum(l, red);
        mj[j] = m; lj[j] = l;
    }
    __syncthreads();
    // ---- pass 2: ctx_j = sum_i pw[j][i] * enc[i, :]; thread = (float4 column c4, row parity)
    const int c4 = tid & 127, rp = tid >> 7;
    f32x4 acc[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int i = rp;
    for (; i + 6 < n; i += 8) {
        const f32x4 e0 = ld_kv<NT>(Eb + (long)(i + 0) * 2 * H + c4 * 4);
        const f32x4 e1 = ld_kv<NT>(Eb + (long)(i + 2) * 2 * H + c4 * 4);
        const f32x4 e2 = ld_kv<NT>(Eb + (long)(i + 4) * 2 * H + c4 * 4);
        const f32x4 e3 = ld_kv<NT>(Eb + (long)(i + 6) * 2 * H + c4 * 4);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (!on[j]) continue;
            const float* pj = pw + j * chunk;
            const float w0 = pj[i], w1 = pj[i + 2], w2 = pj[i + 4], w3 = pj[i + 6];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] += w0 * e0[c] + w1 * e1[c] + w2 * e2[c] + w3 * e3[c];
        }
    }
    for (; i < n; i += 2) {
        const f32x4 e0 = ld_kv<NT>(Eb + (long)i * 2 * H + c4 * 4);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (!on[j]) continue;
            const float w0 = pw[j * chunk + i];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] += w0 * e0[c];
        }
    }
    f32x4* red4 = reinterpret_cast<f32x4*>(red + 16);
    if (rp == 1) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) red4[j * 128 + c4] = acc[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        if (!on[j]) continue;
        float* pout = partial + (((long)slot * NQ + j) * G + g) * (2 * H + 4);
        if (rp == 0) {
            const f32x4 o = red4[j * 128 + c4];
            f32x4 a = acc[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] += o[c];
            *reinterpret_cast<f32x4*>(pout + 4 + c4 * 4) = a;
        }
        if (tid == 0) { pout[0] = mj[j]; pout[1] = lj[j]; }
    }
    if (!ft.tickets) return;
    if (!attn_last_arrival(ft.tickets + slot, G, &last_flag)) return;
    for (int j = 0; j < NQ; ++j) {
        const long row = (long)j * n_clips + b;
        const AttnCombineOut o{ft.ctx + row * ft.ldctx, ft.ctx2 ? ft.ctx2 + row * ft.ldctx2 : nullptr, scores ? scores + row * T : nullptr, T};
        if (on[j]) attn_combine_row(partial + (((long)slot * NQ + j) * G) * (2 * H + 4), G, o, red);
        else attn_zero_row(o);
    }
}



// ---- both staves on one pass over the encoder outputs (round 6).  The upper and the lower NoteDecoder of a segment attend over the SAME encoder
// outputs with key images of their own (models.py:261-275: both decode_notes calls get `encoder_outputs`); per decode step a clip's sweep is
// 1.23 MB of keys + 2.46 MB of encoder outputs per staff.  While both staves of a clip still decode, this kernel scores the chunk against the
// upper staff's key image and queries, then against the lower staff's, and forms all 2 * NQ partial contexts from ONE pass over the chunk's
// encoder rows: 4.92 MB per clip and step instead of 7.38.  Same passes, partial layout and raw scores as attn_fwd_split256_mq, per staff into
// that staff's workspace -- the staves' own combine launches follow unchanged.  Clips in the order of the PAIR (the step the clip's last row of
// either staff finishes at); a staff without an unfinished row in a clip is skipped there (its key image is not read).
struct AttnPairSide { const float* Kmat; const float* q; const float* v; float* partial; float* scores; const int* row_until; };
template <int NQ, bool NT>
__global__ __launch_bounds__(256) void attn_fwd_split256_pair(AttnPairSide s0, AttnPairSide s1, const float* __restrict__ enc, long ldq, int T, int G, int chunk,
                                                              const int* __restrict__ clip_order, int step, int n_clips) {
    constexpr int H = 256, NJ = 2 * NQ;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pw = sm;                                   // NJ x chunk weights
    float* red = sm + NJ * chunk;                     // 16 + NJ * 128 * 4 floats
    const int slot = blockIdx.x / G, g = blockIdx.x % G;
    const int b = clip_order ? clip_order[slot] : slot;
    const int t0 = g * chunk, t1 = min(T, t0 + chunk), n = t1 - t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool on[NJ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        on[j] = !s0.row_until || step < s0.row_until[j * n_clips + b];
        on[NQ + j] = !s1.row_until || step < s1.row_until[j * n_clips + b];
    }
    // ---- pass 1, once per staff: scores of the chunk against that staff's key image
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const AttnPairSide& sd = side ? s1 : s0;
        bool any = false;
#pragma unroll
        for (int j = 0; j < NQ; ++j) any = any || on[side * NQ + j];
        if (!any) continue;                            // uniform over the workgroup
        const float* Kb = sd.Kmat + ((long)b * T + t0) * H;
        f32x4 q4[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j)
            q4[j] = on[side * NQ + j] ? *reinterpret_cast<const f32x4*>(sd.q + ((long)j * n_clips + b) * ldq + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) q4[j][c] = exp2x_clamped(q4[j][c]);
        const f32x4 v4 = {sd.v[lane * 4], sd.v[lane * 4 + 1], sd.v[lane * 4 + 2], sd.v[lane * 4 + 3]};
        for (int r = wave * 4; r < n; r += 16) {
            f32x4 k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                k[u] = (r + u < n) ? ld_kv<NT>(Kb + (long)(r + u) * H + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (!on[side * NQ + j]) continue;
                float sj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    sj[u] = v4[0] * tanh_ek(k[u][0], q4[j][0]) + v4[1] * tanh_ek(k[u][1], q4[j][1])
                          + v4[2] * tanh_ek(k[u][2], q4[j][2]) + v4[3] * tanh_ek(k[u][3], q4[j][3]);
#pragma unroll
                for (int u = 0; u < 4; ++u) sj[u] = wave_sum_lane63(sj[u]);
                if (lane == 63) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (r + u < n) pw[(side * NQ + j) * chunk + r + u] = sj[u];
                }
            }
        }
    }
    __syncthreads();
    float mj[NJ], lj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        mj[j] = 0.f; lj[j] = 0.f;
        if (!on[j]) continue;
        float* scores = j < NQ ? s0.scores : s1.scores;
        const int jr = j < NQ ? j : j - NQ;
        float m = -INFINITY;
        for (int i = tid; i < n; i += 256) m = fmaxf(m, pw[j * chunk + i]);
        m = block_max(m, red);
        float l = 0.f;
        for (int i = tid; i < n; i += 256) {
            const float sc = pw[j * chunk + i];
            if (scores) scores[((long)jr * n_clips + b) * T + t0 + i] = sc;
            const float p = __expf(sc - m);
            pw[j * chunk + i] = p; l += p;
        }
        l = block_sum(l, red);
        mj[j] = m; lj[j] = l;
    }
    __syncthreads();
    // ---- pass 2, once: every partial context from one pass over the chunk's encoder rows
    const float* Eb = enc + ((long)b * T + t0) * 2 * H;
    const int c4 = tid & 127, rp = tid >> 7;
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int i = rp;
    for (; i + 6 < n; i += 8) {
        const f32x4 e0 = ld_kv<NT>(Eb + (long)(i + 0) * 2 * H + c4 * 4);
        const f32x4 e1 = ld_kv<NT>(Eb + (long)(i + 2) * 2 * H + c4 * 4);
        const f32x4 e2 = ld_kv<NT>(Eb + (long)(i + 4) * 2 * H + c4 * 4);
        const f32x4 e3 = ld_kv<NT>(Eb + (long)(i + 6) * 2 * H + c4 * 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!on[j]) continue;
            const float* pj = pw + j * chunk;
            const float w0 = pj[i], w1 = pj[i + 2], w2 = pj[i + 4], w3 = pj[i + 6];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] += w0 * e0[c] + w1 * e1[c] + w2 * e2[c] + w3 * e3[c];
        }
    }
    for (; i < n; i += 2) {
        const f32x4 e0 = ld_kv<NT>(Eb + (long)i * 2 * H + c4 * 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!on[j]) continue;
            const float w0 = pw[j * chunk + i];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] += w0 * e0[c];
        }
    }
    f32x4* red4 = reinterpret_cast<f32x4*>(red + 16);
    if (rp == 1) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) red4[j * 128 + c4] = acc[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (!on[j]) continue;
        const int jr = j < NQ ? j : j - NQ;
        float* pout = (j < NQ ? s0.partial : s1.partial) + (((long)slot * NQ + jr) * G + g) * (2 * H + 4);
        if (rp == 0) {
            const f32x4 o = red4[j * 128 + c4];
            f32x4 a = acc[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] += o[c];
            *reinterpret_cast<f32x4*>(pout + 4 + c4 * 4) = a;
        }
        if (tid == 0) { pout[0] = mj[j]; pout[1] = lj[j]; }
    }
}

// ---- few-clip form of the sweep (round 5).  The kernels above stream a chunk in rounds of 4 frames per wave / 4 rows per thread -- 5 + 10 dependent
// load rounds for a 76-frame chunk -- which is right for a launch that fills the chip many times over and wrong for the long-clip chain, where a launch
// covers a handful of clips, every workgroup has a CU to itself and the kernel's time IS its dependent round trips (each 2-3x longer while the bulk group's
// sweeps saturate the memory: DESIGN.md section 3.4).  Here every load of the chunk is issued before anything is waited for: 8 waves, a wave holds the
// key rows of its <= 10 frames, a thread its float4 column of <= 20 encoder rows (116 registers), ONE round trip.  Same partial layout and raw scores
// as attn_fwd_split256[_mq]: the combine kernel does not change.  Launches over at most a2s_debug_set("attn_deep") (default 24) clips, chunks of <= 80 frames.
template <int NQ>
__global__ __launch_bounds__(512) void attn_fwd_split256_deep(const float* __restrict__ Kmat, const float* __restrict__ enc,
                                                              const float* __restrict__ q, long ldq, const float* __restrict__ v,
                                                              float* __restrict__ partial, float* __restrict__ scores, int T, int G, int chunk,
                                                              const int* __restrict__ clip_order, const int* __restrict__ row_until,
                                                              int step, int n_clips) {
    constexpr int H = 256, KF = 10, ER = 20;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pw = sm;                                   // NQ x chunk scores / weights
    float* red = sm + NQ * chunk;                     // 16 + NQ * 3 * 128 * 4 floats
    const int slot = blockIdx.x / G, g = blockIdx.x % G;
    const int b = clip_order ? clip_order[slot] : slot;
    const int t0 = g * chunk, t1 = min(T, t0 + chunk), n = t1 - t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 127, rp = tid >> 7;          // pass 2: float4 column, row residue mod 4
    const float* Kb = Kmat + ((long)b * T + t0) * H;
    const float* Eb = enc + ((long)b * T + t0) * 2 * H;
    bool on[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) on[j] = !row_until || step < row_until[j * n_clips + b];
    // ---- every load of the chunk, then the queries
    f32x4 k[KF], e[ER];
#pragma unroll
    for (int u = 0; u < KF; ++u) {
        const int f = wave + 8 * u;
        k[u] = f < n ? *reinterpret_cast<const f32x4*>(Kb + (long)f * H + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int m = 0; m < ER; ++m) {
        const int i = rp + 4 * m;
        e[m] = i < n ? *reinterpret_cast<const f32x4*>(Eb + (long)i * 2 * H + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 q4[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
        q4[j] = on[j] ? *reinterpret_cast<const f32x4*>(q + ((long)j * n_clips + b) * ldq + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 v4 = {v[lane * 4], v[lane * 4 + 1], v[lane * 4 + 2], v[lane * 4 + 3]};
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) q4[j][c] = exp2x_clamped(q4[j][c]);
    // ---- pass 1: scores (one wave per frame)
#pragma unroll
    for (int u = 0; u < KF; ++u) {
        const int f = wave + 8 * u;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (!on[j]) continue;
            float sj = v4[0] * tanh_ek(k[u][0], q4[j][0]) + v4[1] * tanh_ek(k[u][1], q4[j][1])
                     + v4[2] * tanh_ek(k[u][2], q4[j][2]) + v4[3] * tanh_ek(k[u][3], q4[j][3]);
            sj = wave_sum_lane63(sj);
            if (lane == 63 && f < n) pw[j * chunk + f] = sj;
        }
    }
    __syncthreads();
    float mj[NQ], lj[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        mj[j] = 0.f; lj[j] = 0.f;
        if (!on[j]) continue;                          // uniform over the workgroup
        const float sc = tid < n ? pw[j * chunk + tid] : -INFINITY;          // (n <= 80 < 512: one score per thread)
        const float m = block_max(sc, red);
        const float p = tid < n ? __expf(sc - m) : 0.f;
        if (tid < n) {
            if (scores) scores[((long)j * n_clips + b) * T + t0 + tid] = sc;
            pw[j * chunk + tid] = p;
        }
        lj[j] = block_sum(p, red);
        mj[j] = m;
    }
    __syncthreads();
    // ---- pass 2: partial contexts from the rows already in registers
    f32x4 acc[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < ER; ++m) {
        const int i = rp + 4 * m;
        if (i < n) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (!on[j]) continue;
                const float w = pw[j * chunk + i];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[j][c] = fmaf(w, e[m][c], acc[j][c]);
            }
        }
    }
    f32x4* red4 = reinterpret_cast<f32x4*>(red + 16);
    if (rp > 0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) red4[(j * 3 + rp - 1) * 128 + c4] = acc[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        if (!on[j]) continue;
        float* pout = partial + (((long)slot * NQ + j) * G + g) * (2 * H + 4);
        if (rp == 0) {
            f32x4 a = acc[j];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const f32x4 o = red4[(j * 3 + w) * 128 + c4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] += o[c];
            }
            *reinterpret_cast<f32x4*>(pout + 4 + c4 * 4) = a;
        }
        if (tid == 0) { pout[0] = mj[j]; pout[1] = lj[j]; }
    }
}
static int g_attn_deep = -1;
void a2s_attn_deep_set(int v) { g_attn_deep = v < 0 ? 0 : v; }
int a2s_attn_deep_max_clips(void) {
    if (g_attn_deep < 0) g_attn_deep = 24;          // launches over at most this many active clips (a2s_debug_set("attn_deep", n))
    return g_attn_deep;
}
// Combine of the few-clip training launches folded into the GRU step (a2s_debug_set("attn_defer_combine", 0): off)
static int g_attn_defer = -1;
void a2s_attn_defer_combine_set(int v) { g_attn_defer = v ? 1 : 0; }
int a2s_attn_defer_combine_enabled(void) {
    if (g_attn_defer < 0) g_attn_defer = 1;
    return g_attn_defer;
}
template <int NQ>
static void launch_fwd_deep(hipStream_t st, int nwg, const float* Kmat, const float* enc, const float* q, long ldq, const float* v, float* ws, float* attw,
                            int T, int G, int chunk, const a2s_attn_rows& r) {
    const size_t shm = ((size_t)NQ * chunk + 16 + (size_t)NQ * 3 * 128 * 4) * sizeof(float);
    hipLaunchKernelGGL((attn_fwd_split256_deep<NQ>), dim3(nwg), dim3(512), shm, st, Kmat, enc, q, ldq, v, ws, attw, T, G, chunk, r.clip_order, r.row_until, r.step,
                       r.n_clips);
}


// merge the G partials of a row: ctx = sum_g ctx_g e^{m_g-m} / l ; optional normalisation of the saved weights.  One workgroup per
// row (= group * n_clips + clip); rows that are skipped this step get zeros.
__global__ __launch_bounds__(256) void attn_fwd_combine256(const float* __restrict__ partial, float* __restrict__ ctx, long ldctx,
                                                           float* __restrict__ ctx2, long ldctx2, float* __restrict__ attw, int T, int G,
                                                           const int* __restrict__ n_done, int n_rows_total,
                                                           const int* __restrict__ clip_rank, const int* __restrict__ row_until,
                                                           int n_clips, int groups, int n_active, int step) {
    if (n_done && *n_done >= n_rows_total) return;
    constexpr int H = 256;
    const int b = blockIdx.x;
    const int clip = b % n_clips, grp = b / n_clips;
    const int slot = clip_rank ? clip_rank[clip] : clip;
    if (slot >= n_active || (row_until && step >= row_until[b])) {   // skipped row
        attn_zero_row(AttnCombineOut{ctx + (long)b * ldctx, ctx2 ? ctx2 + (long)b * ldctx2 : nullptr, attw ? attw + (long)b * T : nullptr, T});
        return;
    }
    __shared__ float wgt[80];
    attn_combine_row(partial + ((long)slot * groups + grp) * G * (2 * H + 4), G,
                     AttnCombineOut{ctx + (long)b * ldctx, ctx2 ? ctx2 + (long)b * ldctx2 : nullptr, attw ? attw + (long)b * T : nullptr, T}, wgt);
}

size_t a2s_attn_workspace_floats_impl(int B, int T, int H, int groups) {
    // a launch may cover any n <= B clips (finished rows skipped), each split G(n) ways: size for the largest n * G(n)
    size_t rows = 0;
    for (int n = 1; n <= B; ++n) {
        int G, chunk;
        a2s_attn_split_geometry(n, T, &G, &chunk);
        if ((size_t)n * G > rows) rows = (size_t)n * G;
    }
    return A2S_ATTN_TICKETS + rows * (groups > 0 ? groups : 1) * (2 * H + 4);
}

// Fused combine (OFF by default; a2s_debug_set("attn_fused_combine", 1)): the split kernels' last-arriving
// workgroup per clip merges the partials, no separate combine launch.  Parity-tested, and measured SLOWER in the training step (632 ->
// 692 ms): every one of the ~770 workgroups of a launch pays a device-scope release (L2 write-back) before its ticket and the last one
// an acquire (L2 invalidate) -- on an 8-XCD part that costs more than the 7 us launch it saves and evicts the other streams' lines.
static int g_attn_fused_combine = -1;
void a2s_attn_fused_combine_set(int v) { g_attn_fused_combine = v < 0 ? 0 : v; }      // 0 never, 1 always, n >= 2: launches over at most n clips
int a2s_attn_fused_combine_enabled(void) {
    if (g_attn_fused_combine < 0) g_attn_fused_combine = 0;
    return g_attn_fused_combine;
}
static int g_attn_nt = -1;
void a2s_attn_nt_set(int v) { g_attn_nt = v < 0 ? 0 : v; }
int a2s_attn_nt_enabled(void) {
    if (g_attn_nt < 0) g_attn_nt = 64;      // launches over >= 64 clips
    return g_attn_nt;
}
// Occupancy cap of the bulk launches.  A launch over >= 64 clips asks for at least this much LDS -- 64 KB forward (2 workgroups per CU),
// 32 KB backward (5) -- although its kernel needs ~2-3 KB: the loaded HBM latency that every OTHER kernel on the chip sees goes with the
// bytes the bulk launches keep in flight (8 workgroups x 256 lanes x several 16-byte loads per CU = ~30 MB against the ~6 MB that saturate the
// memory), and the long-clip group's chain of short dependent kernels -- the step's critical path -- pays that latency several times per
// decode step.  Measured (profiles/r04_attn_occupancy_cap.txt, B = 256, same box): forward launch alone 155 -> 159 us at 2 per CU, backward
// 152 -> 203 us (hence 5 there); in the step the long-clip group finishes 11 ms earlier, the bulk group 8 ms later, the step 507 -> 498 ms.
// The cap only pays while ANOTHER clip group decodes beside the bulk one: the host switches it on for exactly those passes
// (a2s_debug_set("attn_bulk_cap", 1); off by default: greedy decoding of 256 clips 495 -> 468 clips/s with it).  The sizes were swept in rounds 4-6
// (profiles/r04_attn_occupancy_cap.txt, r05_prefix_percent.txt, r06_cap_sweep.txt: flat within +-2 ms around these values; ONE forward workgroup
// per CU -- 80 KB and more -- loses 5-10 ms).
static int g_attn_bulk_cap = 0;
void a2s_attn_bulk_cap_set(int on) { g_attn_bulk_cap = on > 0 ? 1 : 0; }
int a2s_attn_bulk_cap_enabled(void) { return g_attn_bulk_cap; }
size_t a2s_attn_bulk_lds(size_t shm, int n_active, int backward) {
    if (!g_attn_bulk_cap) return shm;
    // forward 64 KB: 2 workgroups per CU.  Backward 28 KB: 5 per CU with 17 KB of LDS left free on every CU -- with 5 x 32 KB the long-clip chain's
    // kernels, which all need 8-16 KB for their cross-wave reduction, could only start where a bulk workgroup had just left (round 5: 452.3 -> 449.5 ms)
    const size_t c = (backward & 1) ? 28672 : 65536;
    return (n_active >= 64 && c > shm) ? c : shm;
}
template <int NQ>
static void launch_fwd_mq(hipStream_t st, int nwg, size_t shm, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                          float* ws, float* attw, int T, int G, int chunk, const a2s_attn_rows& r, const AttnFusedTail& ft, bool nt) {
    if (nt && shm > 65536) {
        static bool raised = false;
        if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_split256_mq<NQ, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304); raised = true; }
    }
    if (nt) hipLaunchKernelGGL((attn_fwd_split256_mq<NQ, true>), dim3(nwg), dim3(256), shm, st, Kmat, enc, q, ldq, v, ws, attw, T, G, chunk,
                               r.clip_order, r.row_until, r.step, r.n_clips, ft);
    else hipLaunchKernelGGL((attn_fwd_split256_mq<NQ, false>), dim3(nwg), dim3(256), shm, st, Kmat, enc, q, ldq, v, ws, attw, T, G, chunk,
                            r.clip_order, r.row_until, r.step, r.n_clips, ft);
}

int a2s_attn_step_fwd_split_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                                 float* ctx, long ldctx, float* ctx2, long ldctx2, float* attw, float* ws, int B, int T, int H,
                                 const int* n_done, int n_rows_total, const a2s_attn_rows* rows, a2s_attn_deferred* defer) {
    A2S_REQUIRE(H == 256 && ws, "attn_step_fwd_split: needs hidden_size 256 and a workspace");
    A2S_REQUIRE(ldq % 4 == 0 && ((uintptr_t)q % 16 == 0) && ((uintptr_t)Kmat % 16 == 0) && ((uintptr_t)enc % 16 == 0), "attn_step_fwd_split: 16-byte alignment");
    a2s_attn_rows r = {nullptr, nullptr, nullptr, B, B, 0};
    if (rows) r = *rows;
    A2S_REQUIRE(r.n_clips > 0 && B % r.n_clips == 0, "attn_step_fwd_split: rows (%d) must be a multiple of the clips (%d)", B, r.n_clips);
    const int groups = B / r.n_clips;
    A2S_REQUIRE(groups <= A2S_ATTN_MAX_GROUPS, "attn_step_fwd_split: at most %d fused bars (got %d)", A2S_ATTN_MAX_GROUPS, groups);
    A2S_REQUIRE(r.n_active >= 0 && r.n_active <= r.n_clips && (!r.clip_order || r.clip_rank), "attn_step_fwd_split: bad row compaction");
    A2S_REQUIRE(groups == 1 || !n_done, "attn_step_fwd_split: fused bars are a training-only path");
    A2S_REQUIRE(a2s_attn_max_split() <= 64, "attn_step_fwd_split: the split must be <= 64");
    int G = 1, chunk = T;
    // workspace: [A2S_ATTN_TICKETS ints: arrival counters, zero between launches (the allocation must be zero-initialised once)] [partials]
    int* tickets = reinterpret_cast<int*>(ws);
    float* part = ws + A2S_ATTN_TICKETS;
    // (mode n >= 2: only launches over at most n active clips -- the long-clip group's latency chain, where one launch and one dependent round
    // trip less per step count and the per-workgroup release is paid by a few dozen workgroups instead of ~770)
    const int fmode = a2s_attn_fused_combine_enabled();
    const bool fused = fmode && (fmode == 1 || r.n_active <= fmode) && r.n_clips <= A2S_ATTN_TICKETS / 2;
    // fused tail: one extra workgroup per clip (row) WITHOUT unfinished rows zero-fills its outputs
    const int n_zero = fused ? r.n_clips - r.n_active : 0;
    const AttnFusedTail ft = {fused ? tickets : nullptr, ctx, ldctx, ctx2, ldctx2, r.n_active, B};
    // streaming (non-temporal) K / enc loads when many clips are active: the sweep is far larger than any cache, and the lines of a
    // concurrently decoding few-clip group (its K / enc and its weights) then survive in L2 / Infinity Cache ("attn_nt")
    const bool nt = a2s_attn_nt_enabled() > 0 && r.n_active >= a2s_attn_nt_enabled();
    if (r.n_active > 0 || n_zero > 0) {
        // the grid covers the clips that still have unfinished rows, re-split so that it still fills the chip
        if (r.n_active > 0) a2s_attn_split_geometry(r.n_active, T, &G, &chunk);
        const int nwg = r.n_active * G + n_zero;
        if (!fused && !n_done && groups <= 4 && r.n_active <= a2s_attn_deep_max_clips() && chunk <= 80) {       // the few-clip form: every load of a chunk in one round trip
            // (five fused bars: the instantiation needs scratch; measured slower on the one-segment steps, which are the only ones that fuse five)
            switch (groups) {
                case 1: launch_fwd_deep<1>(st, nwg, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, r); break;
                case 2: launch_fwd_deep<2>(st, nwg, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, r); break;
                case 3: launch_fwd_deep<3>(st, nwg, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, r); break;
                default: launch_fwd_deep<4>(st, nwg, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, r); break;
            }
        } else if (groups == 1) {
            const size_t shm = a2s_attn_bulk_lds((chunk + 16 + 128 * 4) * sizeof(float), r.n_active, 0);
            if (nt && shm > 65536) {
                static bool raised = false;
                if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_split256<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304); raised = true; }
            }
            if (nt) hipLaunchKernelGGL(attn_fwd_split256<true>, dim3(nwg), dim3(256), shm, st, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, n_done, n_rows_total, r.clip_order, ft);
            else hipLaunchKernelGGL(attn_fwd_split256<false>, dim3(nwg), dim3(256), shm, st, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, n_done, n_rows_total, r.clip_order, ft);
        } else {
            const size_t shm = a2s_attn_bulk_lds(((size_t)groups * chunk + 16 + (size_t)groups * 128 * 4) * sizeof(float), r.n_active, 2);
            switch (groups) {
                case 2: launch_fwd_mq<2>(st, nwg, shm, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, r, ft, nt); break;
                case 3: launch_fwd_mq<3>(st, nwg, shm, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, r, ft, nt); break;
                case 4: launch_fwd_mq<4>(st, nwg, shm, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, r, ft, nt); break;
                default: launch_fwd_mq<5>(st, nwg, shm, Kmat, enc, q, ldq, v, part, attw, T, G, chunk, r, ft, nt); break;
            }
        }
        A2S_CHECK_LAUNCH("attn_fwd_split256");
    }
    if (fused) return A2S_OK;
    // the few-clip launches of a training step leave the combine to the GRU step that consumes the contexts (a2s_step.hip: dec_gru_step_cmb)
    if (defer && !n_done && r.n_active > 0 && r.n_active <= a2s_attn_deep_max_clips() && groups <= 4 && G <= 16 && a2s_attn_defer_combine_enabled()) {
        *defer = a2s_attn_deferred{part, attw, r.clip_rank, r.row_until, G, groups, r.n_clips, r.n_active, r.step, T};
        return A2S_OK;
    }
    hipLaunchKernelGGL(attn_fwd_combine256, dim3(B), dim3(256), 0, st, part, ctx, ldctx, ctx2, ldctx2, attw, T, G, n_done, n_rows_total,
                       r.clip_rank, r.row_until, r.n_clips, groups, r.n_active, r.step);
    A2S_CHECK_LAUNCH("attn_fwd_combine256");
    return A2S_OK;
}

// ---- the two staves' sweeps of one decode step as ONE launch (attn_fwd_split256_pair), then each staff's own combine (attn_pair_combine, on
// that staff's stream).  rows_u / rows_l: the staves' row bookkeeping (row_until of their own); pair: clip order / rank / active count of the
// PAIR at this step.  Both staves' partials use the pair's geometry.
static int g_attn_pair = 1;                  // a2s_debug_set("attn_pair", 0): every staff sweeps on its own (the A/B and the parity tests)
static long g_attn_pair_launches = 0;
void a2s_attn_pair_set(int on) { g_attn_pair = on ? 1 : 0; }
int a2s_attn_pair_enabled(void) { return g_attn_pair; }
long a2s_attn_pair_launches(void) { return g_attn_pair_launches; }
int a2s_dec_fused_max_rows(void);
static int g_attn_pair_fused_rows = 32;
void a2s_attn_pair_fused_rows_set(int v) { g_attn_pair_fused_rows = v; }
int a2s_attn_pair_fused_rows(void) { const int cap = a2s_dec_fused_max_rows(); return g_attn_pair_fused_rows < cap ? g_attn_pair_fused_rows : cap; }

template <int NQ>
static void launch_fwd_pair(hipStream_t st, int nwg, size_t shm, const AttnPairSide& s0, const AttnPairSide& s1, const float* enc, long ldq, int T,
                            const AttnPairStep& p, bool nt) {
    if (nt && shm > 65536) {
        static bool raised = false;
        if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_split256_pair<NQ, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304); raised = true; }
    }
    if (nt) hipLaunchKernelGGL((attn_fwd_split256_pair<NQ, true>), dim3(nwg), dim3(256), shm, st, s0, s1, enc, ldq, T, p.G, p.chunk, p.clip_order, p.step, p.n_clips);
    else hipLaunchKernelGGL((attn_fwd_split256_pair<NQ, false>), dim3(nwg), dim3(256), shm, st, s0, s1, enc, ldq, T, p.G, p.chunk, p.clip_order, p.step, p.n_clips);
}

static int attn_pair_sweep(hipStream_t st, const NoteDecArgs& au, const NoteDecArgs& al, int sv, AttnPairStep& p) {
    const int T = au.T, H = au.H, groups = au.R / p.n_clips;
    A2S_REQUIRE(H == 256 && au.attn_ws && al.attn_ws && au.enc == al.enc && au.R == al.R && au.T == al.T && groups >= 1 && groups <= A2S_ATTN_MAX_GROUPS,
                "attn_pair_sweep: the staves must decode the same rows over the same encoder outputs");
    a2s_attn_split_geometry(p.n_active, T, &p.G, &p.chunk);
    const AttnPairSide s0 = {au.keys, au.q + (long)sv * au.R * H, au.attn_v, au.attn_ws + A2S_ATTN_TICKETS, au.attw ? au.attw + (long)sv * au.R * T : nullptr, au.row_until};
    const AttnPairSide s1 = {al.keys, al.q + (long)sv * al.R * H, al.attn_v, al.attn_ws + A2S_ATTN_TICKETS, al.attw ? al.attw + (long)sv * al.R * T : nullptr, al.row_until};
    const bool nt = a2s_attn_nt_enabled() > 0 && p.n_active >= a2s_attn_nt_enabled();
    const size_t shm = a2s_attn_bulk_lds(((size_t)2 * groups * p.chunk + 16 + (size_t)2 * groups * 128 * 4) * sizeof(float), p.n_active, 2);
    const int nwg = p.n_active * p.G;
    switch (groups) {
        case 1: launch_fwd_pair<1>(st, nwg, shm, s0, s1, au.enc, H, T, p, nt); break;
        case 2: launch_fwd_pair<2>(st, nwg, shm, s0, s1, au.enc, H, T, p, nt); break;
        case 3: launch_fwd_pair<3>(st, nwg, shm, s0, s1, au.enc, H, T, p, nt); break;
        case 4: launch_fwd_pair<4>(st, nwg, shm, s0, s1, au.enc, H, T, p, nt); break;
        default: launch_fwd_pair<5>(st, nwg, shm, s0, s1, au.enc, H, T, p, nt); break;
    }
    A2S_CHECK_LAUNCH("attn_fwd_split256_pair");
    __atomic_fetch_add(&g_attn_pair_launches, 1, __ATOMIC_RELAXED);
    return A2S_OK;
}

// one staff's combine behind a pair sweep: contexts into x[si][:, E:] and o[sv][:, 2H:], weights normalised in place
static int attn_pair_combine(hipStream_t st, const NoteDecArgs& a, int si, int sv, const AttnPairStep& p) {
    const int H2 = 2 * a.H, ldx = a.E + H2;
    float* xs = a.x + (long)si * a.R * ldx;
    float* os = a.o + (long)sv * a.R * 2 * H2;
    hipLaunchKernelGGL(attn_fwd_combine256, dim3(a.R), dim3(256), 0, st, a.attn_ws + A2S_ATTN_TICKETS, xs + a.E, (long)ldx, os + H2, (long)(2 * H2),
                       a.attw ? a.attw + (long)sv * a.R * a.T : nullptr, a.T, p.G, (const int*)nullptr, a.R, p.clip_rank, a.row_until, p.n_clips,
                       a.R / p.n_clips, p.n_active, p.step);
    A2S_CHECK_LAUNCH("attn_fwd_combine256");
    return A2S_OK;
}
