// Backward kernels of the sequential part of the hot path + the reverse step loops (no Python per step).
// Gradient flow follows what torch.autograd does for the reference graph (models.py:191-420): argmax-selected
// tokens are constants (only their embedding rows receive gradient), log_softmax rows that were never decoded
// receive nothing, attention keys are hoisted so dK/dEnc contributions of all steps are accumulated once per
// (bar, staff) instead of once per step.
#include "a2s_common.h"
#include "../../include/a2s.h"

int a2s_gemm_impl(hipStream_t st, int M, int N, int K, float alpha, const float* A, long sAm, long sAk,
                  const float* B, long sBk, long sBn, float beta, float* C, long ldc, const float* bias, int act,
                  int batch, long bsA, long bsB, long bsC, int splitk, float* ws, size_t ws_bytes);

// ------------------------------------------------------------------------------------------- log_softmax bwd
// y = log_softmax(x) row-wise; given g = dL/dy and y: dx = g - exp(y) * sum_j g_j.
// Row r of g / y lives at base + (r / inner) * outer_stride + (r % inner) * V  (covers (B,5,U,V)[:,bar,:steps]
// with r = b*steps + s as well as plain matrices); dx is written TIME-MAJOR: row (s*B + b) when time_major.
__global__ __launch_bounds__(256) void log_softmax_bwd_rows(const float* __restrict__ g, const float* __restrict__ y,
                                                            long outer_stride, int inner, float* __restrict__ dx,
                                                            int R, int V, int n_outer, int time_major) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const int ob = row / inner, in = row % inner;
    const float* gr = g + (long)ob * outer_stride + (long)in * V;
    const float* yr = y + (long)ob * outer_stride + (long)in * V;
    float s = 0.f;
    for (int j = lane; j < V; j += 64) s += gr[j];
    s = wave_sum(s);
    const long orow = time_major ? ((long)in * n_outer + ob) : row;
    for (int j = lane; j < V; j += 64) dx[orow * V + j] = gr[j] - expf(yr[j]) * s;
}

int a2s_log_softmax_bwd_rows_impl(hipStream_t st, const float* g, const float* y, long outer_stride, int inner, float* dx,
                                  int R, int V, int n_outer, int time_major) {
    hipLaunchKernelGGL(log_softmax_bwd_rows, dim3(a2s_cdiv(R, 4)), dim3(256), 0, st, g, y, outer_stride, inner, dx, R, V, n_outer, time_major);
    A2S_CHECK_LAUNCH("log_softmax_bwd_rows");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- GRU cell bwd
// saved = [r | z | n | gh_n] per row (4H).  dh: grad wrt the cell output (dh = dh_a + dh_b, dh_b optional).
//   dn = dh (1-z)(1-n^2) ; dz = dh (hprev - n) z(1-z) ; dr = dn gh_n r(1-r)
//   dgi = [dr, dz, dn] ; dgh = [dr, dz, dn*r] ; dh_prev_direct = dh*z
__global__ void gru_gates_bwd(const float* __restrict__ dh_a, long lda, const float* __restrict__ dh_b, long ldb,
                              const float* __restrict__ save, const float* __restrict__ hprev, long ldhp,
                              float* __restrict__ dgi, long ldgi, float* __restrict__ dgh, long ldgh,
                              float* __restrict__ dgh2, long ldgh2, float* __restrict__ dhprev, long lddp, int R, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * H) return;
    const int r_ = (int)(idx / H), j = (int)(idx % H);
    const float* s = save + (long)r_ * 4 * H;
    const float rg = s[j], zg = s[H + j], ng = s[2 * H + j], ghn = s[3 * H + j];
    float dh = dh_a[(long)r_ * lda + j];
    if (dh_b) dh += dh_b[(long)r_ * ldb + j];
    const float hp = hprev ? hprev[(long)r_ * ldhp + j] : 0.f;
    const GruCellGrad cg = gru_cell_bwd(dh, rg, zg, ng, ghn, hp);
    const float dn = cg.dn, dz = cg.dz, dr = cg.dr;
    float* a = dgi + (long)r_ * ldgi;
    a[j] = dr; a[H + j] = dz; a[2 * H + j] = dn;
    float* b = dgh + (long)r_ * ldgh;
    b[j] = dr; b[H + j] = dz; b[2 * H + j] = cg.dnr;
    if (dgh2) { float* c = dgh2 + (long)r_ * ldgh2; c[j] = dr; c[H + j] = dz; c[2 * H + j] = cg.dnr; }
    dhprev[(long)r_ * lddp + j] = cg.dhz;
}

int a2s_gru_gates_bwd_impl(hipStream_t st, const float* dh_a, long lda, const float* dh_b, long ldb, const float* save,
                           const float* hprev, long ldhp, float* dgi, long ldgi, float* dgh, long ldgh, float* dgh2, long ldgh2,
                           float* dhprev, long lddp, int R, int H) {
    hipLaunchKernelGGL(gru_gates_bwd, dim3(a2s_cdiv((long)R * H, 256)), dim3(256), 0, st, dh_a, lda, dh_b, ldb, save, hprev, ldhp,
                       dgi, ldgi, dgh, ldgh, dgh2, ldgh2, dhprev, lddp, R, H);
    A2S_CHECK_LAUNCH("gru_gates_bwd");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- attention bwd (per step)
// Given dctx (= dctx_a + dctx_b) for one step:  da_t = dctx . enc_t ; ds_t = a_t (da_t - dctx . ctx)   [softmax bwd,
// sum_t a_t da_t = dctx . ctx];  dq_j = sum_t ds_t v_j (1 - e_tj^2), e = tanh(K_tj + q_j).
// Writes dq (for the W_h / hidden gradient), ds (T per row, for the deferred dK / dv) and the summed dctx.
__global__ __launch_bounds__(256) void attn_step_bwd(const float* __restrict__ Kmat, const float* __restrict__ enc,
                                                     const float* __restrict__ q, long ldq, const float* __restrict__ v,
                                                     const float* __restrict__ attw, const float* __restrict__ ctx, long ldctx,
                                                     const float* __restrict__ dctx_a, long ldda, const float* __restrict__ dctx_b, long lddb,
                                                     float* __restrict__ dctx_out, long lddo, float* __restrict__ dq, long lddq,
                                                     float* __restrict__ ds_out, int T, int n_clips, int H) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dsv = sm;                               // T
    float* dc = sm + ((T + 3) & ~3);               // 2H
    float* red = dc + 2 * H;                       // 16
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int clip = b % n_clips;                  // fused bars: row = bar * n_clips + clip
    const float* Kb = Kmat + (long)clip * T * H;
    const float* Eb = enc + (long)clip * T * 2 * H;
    float part = 0.f;
    for (int d = tid; d < 2 * H; d += 256) {
        float g = dctx_a[(long)b * ldda + d];
        if (dctx_b) g += dctx_b[(long)b * lddb + d];
        dc[d] = g;
        if (dctx_out) dctx_out[(long)b * lddo + d] = g;
        part += g * ctx[(long)b * ldctx + d];
    }
    const float dot_ctx = block_sum(part, red);    // also orders the dc[] writes before the reads below
    // pass over enc: one wave per frame, 2H/64 elements per lane
    for (int t = wave; t < T; t += 4) {
        float s = 0.f;
        for (int d = lane; d < 2 * H; d += 64) s = fmaf(dc[d], Eb[(long)t * 2 * H + d], s);
        s = wave_sum(s);
        if (lane == 0) {
            const float d_s = attw[(long)b * T + t] * (s - dot_ctx);
            dsv[t] = d_s;
            if (ds_out) ds_out[(long)b * T + t] = d_s;
        }
    }
    __syncthreads();
    // pass over K: thread j accumulates dq_j over all frames
    for (int j = tid; j < H; j += 256) {
        const float eq = exp2x_clamped(q[(long)b * ldq + j]), vj = v[j];      // Kmat holds the key image E_K = exp(2K)
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = fmaf(dsv[t], sech2_ek(Kb[(long)t * H + j], eq), acc);
        dq[(long)b * lddq + j] = acc * vj;
    }
}

int a2s_attn_step_bwd_split_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                                 const float* attw, const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b,
                                 long lddb, float* dctx_out, long lddo, float* dq, long lddq, float* ds_out, float* ws, int B, int T, int H,
                                 const a2s_attn_rows* rows);
int a2s_attn_step_bwd_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                           const float* attw, const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b,
                           long lddb, float* dctx_out, long lddo, float* dq, long lddq, float* ds_out, int B, int T, int H, float* ws,
                           const a2s_attn_rows* rows = nullptr);

int a2s_attn_step_bwd_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                           const float* attw, const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b,
                           long lddb, float* dctx_out, long lddo, float* dq, long lddq, float* ds_out, int B, int T, int H, float* ws,
                           const a2s_attn_rows* rows) {
    if (H == 256 && ws)
        return a2s_attn_step_bwd_split_impl(st, Kmat, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb, dctx_out, lddo, dq, lddq, ds_out, ws, B, T, H, rows);
    const int n_clips = rows ? rows->n_clips : B;
    const size_t shm = (((T + 3) & ~3) + 2 * H + 16) * sizeof(float);
    A2S_REQUIRE(H >= 1 && H <= 512, "attn_step_bwd: hidden_size must be in 1 .. 512 (got %d)", H);
    hipLaunchKernelGGL(attn_step_bwd, dim3(B), dim3(256), shm, st, Kmat, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb, dctx_out, lddo, dq, lddq, ds_out, T, n_clips, H);
    A2S_CHECK_LAUNCH("attn_step_bwd");
    return A2S_OK;
}

// Deferred key / v gradients of one (bar, staff): for every (b, t, j)
//   dK[b,t,j] += v_j * sum_s ds[s,b,t] (1 - e^2),  dv_j += sum_{s,b,t} ds[s,b,t] e,   e = tanh(K[b,t,j] + q[s,b,j]).
// One workgroup per (b, tile of 16 frames); thread j keeps K[t,j] for its 16 frames in registers and streams the
// S queries -- K is read once, traffic is S*(H + 16) floats per workgroup.  dv partials: [nblocks][H].
__global__ __launch_bounds__(256) void attn_dk_accum(const float* __restrict__ Kmat, const float* __restrict__ q_all,
                                                     const float* __restrict__ ds_all, const float* __restrict__ v,
                                                     float* __restrict__ dK, float* __restrict__ dv_partial, int B, int T, int S,
                                                     const int* __restrict__ row_until, int groups, int H) {
    constexpr int TT = 16;
    const int tiles = (T + TT - 1) / TT;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * TT;
    __shared__ float dss[TT];
    for (int j0 = 0; j0 < H; j0 += blockDim.x) {          // (one pass for H <= the block size: every width the recipes use)
        const int j = j0 + threadIdx.x;
        float kreg[TT], acc[TT];
        float dvj = 0.f;
        if (j < H) {
#pragma unroll
            for (int i = 0; i < TT; ++i) { kreg[i] = (t0 + i < T) ? Kmat[((long)b * T + t0 + i) * H + j] : 0.f; acc[i] = 0.f; }
        }
        // rows of a step: `groups` bars of the same B clips (row = group * B + b); ds is exactly zero from step row_until[row] on
        for (int sg = 0; sg < S * groups; ++sg) {
            const int s = sg / groups, grp = sg % groups;
            if (row_until && s >= row_until[grp * B + b]) continue;          // uniform over the workgroup
            const long row = (long)sg * B + b;
            __syncthreads();
            if (threadIdx.x < TT) dss[threadIdx.x] = (t0 + threadIdx.x < T) ? ds_all[row * T + t0 + threadIdx.x] : 0.f;
            __syncthreads();
            if (j < H) {
                const float eq = exp2x_clamped(q_all[row * H + j]);               // kreg holds the key image E_K = exp(2K)
#pragma unroll
                for (int i = 0; i < TT; ++i) {
                    const float e = tanh_ek(kreg[i], eq);
                    acc[i] = fmaf(dss[i], 1.f - e * e, acc[i]);
                    dvj = fmaf(dss[i], e, dvj);
                }
            }
        }
        if (j < H) {
            const float vj = v[j];
#pragma unroll
            for (int i = 0; i < TT; ++i)
                if (t0 + i < T) dK[((long)b * T + t0 + i) * H + j] += vj * acc[i];
            dv_partial[(long)blockIdx.x * H + j] = dvj;
        }
        __syncthreads();
    }
}

int a2s_attn_dk_accum_impl(hipStream_t st, const float* Kmat, const float* q_all, const float* ds_all, const float* v,
                           float* dK, float* dv_partial, int B, int T, int S, int H, const int* row_until, int groups) {
    const int nblk = B * a2s_cdiv(T, 16);
    if (groups < 1) groups = 1;
    A2S_REQUIRE(H >= 1, "attn_dk_accum: hidden_size must be positive (got %d)", H);
    const int nth = H >= 256 ? 256 : (H > 128 ? 256 : (H > 64 ? 128 : 64));
    hipLaunchKernelGGL(attn_dk_accum, dim3(nblk), dim3(nth), 0, st, Kmat, q_all, ds_all, v, dK, dv_partial, B, T, S, row_until, groups, H);
    A2S_CHECK_LAUNCH("attn_dk_accum");
    return A2S_OK;
}

// out[c] (+)= alpha * sum_r x[r*ld + c]   -- bias gradients / reduction of partial slabs, fixed order per column.
__global__ __launch_bounds__(256) void col_sum(const float* __restrict__ x, long ld, float* __restrict__ out, long rows, int C,
                                               float alpha, float beta) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;          // 4 row-partitions per column block
    __shared__ float red[4][64];
    float s = 0.f;
    if (c < C) for (long r = part; r < rows; r += 4) s += x[r * ld + c];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < C) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        out[c] = alpha * t + (beta != 0.f ? beta * out[c] : 0.f);
    }
}

// stage 1 of the two-stage form: partial[rb][c] = sum of rows [rb*rpb, (rb+1)*rpb) -- grid (column blocks, row blocks)
__global__ __launch_bounds__(256) void col_sum_partial(const float* __restrict__ x, long ld, float* __restrict__ partial, long rows, int C, long rpb) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    __shared__ float red[4][64];
    float s = 0.f;
    if (c < C) for (long r = r0 + part; r < r1; r += 4) s += x[r * ld + c];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < C)
        partial[(long)blockIdx.y * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

int a2s_col_sum_impl(hipStream_t st, const float* x, long ld, float* out, long rows, int C, float alpha, float beta, float* ws, size_t ws_floats) {
    // long matrices: spread the rows over many workgroups first (fixed partition -> deterministic), then sum the partials
    if (ws && rows >= 2048 && ws_floats >= (size_t)2 * C) {
        long nb = (rows + 511) / 512;
        if (nb > 1024) nb = 1024;
        if ((size_t)nb * C > ws_floats) nb = (long)(ws_floats / C);
        const long rpb = (rows + nb - 1) / nb;
        nb = (rows + rpb - 1) / rpb;
        hipLaunchKernelGGL(col_sum_partial, dim3(a2s_cdiv(C, 64), (unsigned)nb), dim3(256), 0, st, x, ld, ws, rows, C, rpb);
        A2S_CHECK_LAUNCH("col_sum_partial");
        hipLaunchKernelGGL(col_sum, dim3(a2s_cdiv(C, 64)), dim3(256), 0, st, ws, (long)C, out, nb, C, alpha, beta);
        A2S_CHECK_LAUNCH("col_sum");
        return A2S_OK;
    }
    hipLaunchKernelGGL(col_sum, dim3(a2s_cdiv(C, 64)), dim3(256), 0, st, x, ld, out, rows, C, alpha, beta);
    A2S_CHECK_LAUNCH("col_sum");
    return A2S_OK;
}

// table_grad[id[r]][j] += g[r*ldg + col0 + j] * (keep mask) -- embedding gradient (atomic: duplicate ids collide).
__global__ void embed_scatter_add(float* __restrict__ table_grad, const long long* __restrict__ ids64, const int* __restrict__ ids32,
                                  long id_stride, int const_id, const float* __restrict__ g, long ldg, int col0, int R, int E,
                                  const uint8_t* __restrict__ drop, float inv_keep) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * E) return;
    const int r = (int)(idx / E), j = (int)(idx % E);
    const long id = ids64 ? ids64[(long)r * id_stride] : (ids32 ? ids32[(long)r * id_stride] : const_id);
    float v = g[(long)r * ldg + col0 + j];
    if (drop) v = drop[idx] ? v * inv_keep : 0.f;
    atomicAdd(table_grad + id * E + j, v);
}

int a2s_embed_scatter_add_impl(hipStream_t st, float* table_grad, const long long* ids64, const int* ids32, long id_stride,
                               int const_id, const float* g, long ldg, int col0, int R, int E, const uint8_t* drop, float inv_keep) {
    hipLaunchKernelGGL(embed_scatter_add, dim3(a2s_cdiv((long)R * E, 256)), dim3(256), 0, st, table_grad, ids64, ids32, id_stride,
                       const_id, g, ldg, col0, R, E, drop, inv_keep);
    A2S_CHECK_LAUNCH("embed_scatter_add");
    return A2S_OK;
}

// elementwise helpers: dx = g * (1 - y^2) (tanh) / dx = g * [y > 0] (relu); optional in-place (dx == g)
__global__ void ew_act_bwd(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ dx, long n, int act) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float yy = y[i];
        dx[i] = act == 2 ? g[i] * (1.f - yy * yy) : (yy > 0.f ? g[i] : 0.f);
    }
}
int a2s_ew_act_bwd_impl(hipStream_t st, const float* g, const float* y, float* dx, long n, int act) {
    hipLaunchKernelGGL(ew_act_bwd, dim3(min((long)2048, (n + 255) / 256)), dim3(256), 0, st, g, y, dx, n, act);
    A2S_CHECK_LAUNCH("ew_act_bwd");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- note decoder reverse loop
// Reverse of a2s_note_decoder_fwd for one (bar, staff); `steps` = steps the forward executed.
// argument block: a2s_note_dec_bwd_args (single definition in include/a2s.h)

bool a2s_dec_step_fusable(int R, int H, int E, int V, const void* const* ptrs, int nptrs, const float* ws, size_t ws_floats, bool greedy = false);
int a2s_note_step_fused_bwd_prepare(hipStream_t st, const a2s_note_dec_bwd_args& a);
int a2s_note_step_fused_bwd(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, const float* dh_in, float* dh_out, const a2s_attn_rows* rows,
                            int nrows, const int* rowmap);

bool a2s_note_decoder_bwd_persist_ok(const a2s_note_dec_bwd_args& a);
int a2s_note_decoder_bwd_persist(hipStream_t st, const a2s_note_dec_bwd_args& a);
bool a2s_note_step_mid_bwd_ok(const a2s_note_dec_bwd_args& a);
int a2s_note_step_mid_bwd(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, float* dh_out, int nrows, const int* rowmap);
int a2s_note_step_mid_bwd_query(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, float* dh_out, int nrows, const int* rowmap);

// the pair's clip bookkeeping at one step and the geometry its dq partials use (both staves' sweeps of the step in one launch: attn_bwd_split256_pair)
struct AttnPairBwdStep { const int* clip_order; const int* clip_rank; int n_clips; int n_active; int step; int G; int chunk; };
static int attn_pair_bwd_sweep(hipStream_t st, const a2s_note_dec_bwd_args& au, const a2s_note_dec_bwd_args& al, int s, AttnPairBwdStep& p);
static int attn_pair_bwd_combine(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, const AttnPairBwdStep& p);
int a2s_attn_pair_enabled(void);

static int note_bwd_step_rows(const a2s_note_dec_bwd_args& a, int s) { return (a.row_list && a.n_rows_active) ? a.n_rows_active[s] : a.R; }
int a2s_attn_pair_fused_rows(void);             // (a2s_seq.hip: the pair loops hand over to the few-row kernels at fewer rows)
static thread_local int t_pair_rows_limit = -1;
struct PairRowsLimit { PairRowsLimit(int v) { t_pair_rows_limit = v; } ~PairRowsLimit() { t_pair_rows_limit = -1; } };
static bool note_bwd_step_fused(const a2s_note_dec_bwd_args& a, int s) {
    const void* ptrs[] = {a.dgi_all, a.dgh_all, a.dq_all, a.dx, a.dh, a.w_ih, a.w_hh, a.attn_w};
    const int n = note_bwd_step_rows(a, s);
    if (t_pair_rows_limit >= 0 && n > t_pair_rows_limit) return false;
    return n > 0 && a2s_dec_step_fusable(n, a.H, a.E, 173, ptrs, 8, a.step_ws, a.step_ws_floats);      // (the vocabulary size plays no role here)
}

// One reverse step of a note decoder; `cur` = which half of a.dh holds the incoming carry (flipped on return).  part: 0 = the whole step; 1 = what
// comes before the attention sweep of a launch-per-step step (GRU cell, dx products), 2 = what comes behind it (dq reduction, dh products) -- the
// pair loop runs 1, the two staves' sweep as one launch, then 2 with `pair` set.
static int note_bwd_step(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, int& cur, bool mid, int part, const AttnPairBwdStep* pair) {
    const int H2 = 2 * a.H, ldx = a.E + H2, R = a.R;
    const bool fused = note_bwd_step_fused(a, s);
    a2s_attn_rows rows_v = {a.clip_order, a.clip_rank, a.row_until, a.n_clips > 0 ? a.n_clips : R, a.n_active ? a.n_active[s] : 0, s};
    const a2s_attn_rows* rows = a.n_active ? &rows_v : nullptr;
    float* dh_in = a.dh + (long)cur * R * H2;
    float* dh_out = a.dh + (long)(cur ^ 1) * R * H2;
    if (fused) {
        const int nrows = note_bwd_step_rows(a, s);
        int rc = a2s_note_step_fused_bwd(st, a, s, dh_in, dh_out, rows, nrows, nrows < R ? a.row_list : nullptr);
        if (rc) return rc;
        cur ^= 1;
        return A2S_OK;
    }
    int gM = R, gB = 1;                  // rows of the per-step products: see enqueue_note_step (a2s_seq.hip); rows left out carry zero gradients
    long gS = 0;
    if (a.m_active && a.n_clips > 0 && a2s_prefix_rows_ok(a.m_active[s], a.n_clips)) { gM = a.m_active[s]; gB = R / a.n_clips; gS = a.n_clips; }
    const float* dos = a.do_all + (long)s * R * 2 * H2;
    float* dgi = a.dgi_all + (long)s * R * 3 * H2;
    float* dgh = a.dgh_all + (long)s * R * 3 * H2;
    float* dxs = a.dx + (long)s * R * ldx;
    int rc = A2S_OK;
    if (part != 2) {
        // GRU cell: dh = carry + dh_from_out;  hprev = h[s]
        rc = a2s_gru_gates_bwd_impl(st, dh_in, H2, dos, 2 * H2, a.gates + (long)s * R * 4 * H2, a.h + (long)s * R * H2, H2,
                                    dgi, 3 * H2, dgh, 3 * H2, nullptr, 0, dh_out, H2, R, H2);
        if (rc) return rc;
        if (mid) {
            const int nrows = note_bwd_step_rows(a, s);
            rc = a2s_note_step_mid_bwd(st, a, s, dh_out, nrows, nrows < R ? a.row_list : nullptr);
        } else {
            // dx = dgi W_ih   (R x ldx): [dtok | dctx_from_gru]
            rc = a2s_gemm_impl(st, gM, ldx, 3 * H2, 1.f, dgi, 3 * H2, 1, a.w_ih, ldx, 1, 0.f, dxs, ldx, nullptr, 0, gB, gS * 3 * H2, 0, gS * ldx, 0, a.gemm_ws, a.gemm_ws_bytes);
        }
        if (rc) return rc;
        if (part == 1) return A2S_OK;
    }
    // attention: dctx = dx[:, E:] + do[:, 2H:]
    rc = pair ? attn_pair_bwd_combine(st, a, s, *pair)
              : a2s_attn_step_bwd_impl(st, a.keys, a.enc, a.q + (long)s * R * a.H, a.H, a.attn_v, a.attw + (long)s * R * a.T,
                                       a.x + (long)s * R * ldx + a.E, ldx, dxs + a.E, ldx, dos + H2, 2 * H2,
                                       a.dctx_all + (long)s * R * H2, H2, a.dq_all + (long)s * R * a.H, a.H,
                                       a.ds_all + (long)s * R * a.T, R, a.T, a.H, a.attn_ws, rows);
    if (rc) return rc;
    // dh_prev += dgh W_hh + dq W_h   (W_h = first 2H columns of attn_w (H, 4H))
    if (!mid) {
        rc = a2s_gemm_impl(st, gM, H2, 3 * H2, 1.f, dgh, 3 * H2, 1, a.w_hh, H2, 1, 1.f, dh_out, H2, nullptr, 0, gB, gS * 3 * H2, 0, gS * H2, 0, a.gemm_ws, a.gemm_ws_bytes);
        if (rc) return rc;
    }
    if (mid) {
        const int nrows = note_bwd_step_rows(a, s);
        rc = a2s_note_step_mid_bwd_query(st, a, s, dh_out, nrows, nrows < R ? a.row_list : nullptr);
    } else
    rc = a2s_gemm_impl(st, gM, H2, a.H, 1.f, a.dq_all + (long)s * R * a.H, a.H, 1, a.attn_w, 2 * H2, 1, 1.f, dh_out, H2, nullptr, 0, gB, gS * a.H, 0, gS * H2, 0, a.gemm_ws, a.gemm_ws_bytes);
    if (rc) return rc;
    cur ^= 1;
    return A2S_OK;
}

// what a reverse loop needs before its first step: zeroed carries / dx, the transposed weight copies of the few-row and mid-size kernels
static int note_bwd_prepare(hipStream_t st, const a2s_note_dec_bwd_args& a, bool* mid_out) {
    const int H2 = 2 * a.H, ldx = a.E + H2, R = a.R;
    hipError_t e = hipMemsetAsync(a.dh, 0, sizeof(float) * 2 * R * H2, st);
    // (steps that skip finished rows leave their dx rows unwritten: they must read as zero gradients)
    if (e == hipSuccess && (a.m_active || a.row_list)) e = hipMemsetAsync(a.dx, 0, sizeof(float) * (size_t)a.steps * R * ldx, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_bwd memset: %s", hipGetErrorString(e));
    // transposed weight copies for the few-row kernels: needed as soon as ANY step of the call runs on them (the last step has the fewest rows,
    // but it may have none at all -- a row without <eos> whose last targets are <pad> -- while earlier steps still have 1 .. max_rows)
    bool any_fused = false;
    for (int s = a.steps - 1; s >= 0 && !any_fused; --s) any_fused = note_bwd_step_fused(a, s);
    // round 6: the dx / dh products of the steps that stay on this loop as ONE launch in front of the attention (dec_bwd_mid, a2s_step.hip), over the
    // rows still running; it reads the same transposed weight copies
    const bool mid = a2s_note_step_mid_bwd_ok(a);
    bool any_mid = false;
    if (mid) for (int s = a.steps - 1; s >= 0 && !any_mid; --s) any_mid = !note_bwd_step_fused(a, s);
    if (any_fused || any_mid) { int rc = a2s_note_step_fused_bwd_prepare(st, a); if (rc) return rc; }
    *mid_out = mid;
    return A2S_OK;
}
static int note_bwd_finish(hipStream_t st, const a2s_note_dec_bwd_args& a, int cur) {
    if (cur != 0) {   // leave the final carry in dh[0]
        const hipError_t e = hipMemcpyAsync(a.dh, a.dh + (long)a.R * 2 * a.H, sizeof(float) * a.R * 2 * a.H, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_bwd copy: %s", hipGetErrorString(e));
    }
    return A2S_OK;
}

int a2s_note_decoder_bwd_impl(hipStream_t st, const a2s_note_dec_bwd_args& a) {
    // few clips: one persistent launch for the whole reverse loop (a2s_dec_persist.hip)
    if (a2s_note_decoder_bwd_persist_ok(a)) return a2s_note_decoder_bwd_persist(st, a);
    bool mid = false;
    int rc = note_bwd_prepare(st, a, &mid);
    if (rc) return rc;
    int cur = 0;
    // rows step s covers on the few-row kernels: all R, or the rows still running (a prefix of row_list); the kernels take over for the steps
    // whose rows fit them (see a2s_note_decoder_fwd_impl)
    for (int s = a.steps - 1; s >= 0; --s) {
        rc = note_bwd_step(st, a, s, cur, mid, 0, nullptr);
        if (rc) return rc;
    }
    return note_bwd_finish(st, a, cur);
}

// The reverse loops of a segment's two NoteDecoders issued by ONE host loop on their two streams (forward: a2s_note_decoder_fwd_pair_impl,
// a2s_seq.hip): the longer staff runs its last steps alone, then both staves step together and the attention sweep of a step is one launch for both.
int a2s_note_decoder_bwd_pair_impl(hipStream_t su, hipStream_t sl, const a2s_note_dec_bwd_args& au, const a2s_note_dec_bwd_args& al, const int* pair_order,
                                   const int* pair_rank, const int* pair_n_active) {
    const a2s_note_dec_bwd_args* as[2] = {&au, &al};
    hipStream_t sts[2] = {su, sl};
    const bool can_pair = a2s_attn_pair_enabled() && su != sl && pair_order && pair_rank && pair_n_active && au.n_active && al.n_active && au.n_clips > 0 &&
                          au.n_clips == al.n_clips && au.R == al.R && au.T == al.T && au.H == 256 && al.H == 256 && au.enc == al.enc && au.attn_ws && al.attn_ws &&
                          !a2s_note_decoder_bwd_persist_ok(au) && !a2s_note_decoder_bwd_persist_ok(al) && a2s_note_step_mid_bwd_ok(au) && a2s_note_step_mid_bwd_ok(al);
    if (!can_pair) {
        const int rc = a2s_note_decoder_bwd_impl(su, au);
        return rc ? rc : a2s_note_decoder_bwd_impl(sl, al);
    }
    static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k)
        if (!ev[k]) { const hipError_t e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming); if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_bwd_pair: hipEventCreate: %s", hipGetErrorString(e)); }
    bool mid[2] = {false, false};
    int cur[2] = {0, 0};
    const PairRowsLimit limit(a2s_attn_pair_fused_rows());
    for (int k = 0; k < 2; ++k) { const int rc = note_bwd_prepare(sts[k], *as[k], &mid[k]); if (rc) return rc; }
    const int nmax = au.steps > al.steps ? au.steps : al.steps;
    for (int s = nmax - 1; s >= 0; --s) {
        bool in[2], fused[2] = {false, false};
        for (int k = 0; k < 2; ++k) { in[k] = s < as[k]->steps; if (in[k]) fused[k] = note_bwd_step_fused(*as[k], s); }
        AttnPairBwdStep p = {pair_order, pair_rank, au.n_clips, pair_n_active[s], s, 1, au.T};
        const bool joint = in[0] && in[1] && mid[0] && mid[1] && !fused[0] && !fused[1] && p.n_active > 0 && au.n_active[s] > 0 && al.n_active[s] > 0;
        if (!joint) {
            for (int k = 0; k < 2; ++k)
                if (in[k]) { const int rc = note_bwd_step(sts[k], *as[k], s, cur[k], mid[k], 0, nullptr); if (rc) return rc; }
            continue;
        }
        for (int k = 0; k < 2; ++k) { const int rc = note_bwd_step(sts[k], *as[k], s, cur[k], mid[k], 1, nullptr); if (rc) return rc; }
        hipError_t e = hipEventRecord(ev[1], sl);                           // the lower staff's dx of this step
        if (e == hipSuccess) e = hipStreamWaitEvent(su, ev[1], 0);
        if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_bwd_pair: event: %s", hipGetErrorString(e));
        const int rc = attn_pair_bwd_sweep(su, au, al, s, p);
        if (rc) return rc;
        e = hipEventRecord(ev[0], su);
        if (e == hipSuccess) e = hipStreamWaitEvent(sl, ev[0], 0);
        if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "note_decoder_bwd_pair: event: %s", hipGetErrorString(e));
        for (int k = 0; k < 2; ++k) { const int rc2 = note_bwd_step(sts[k], *as[k], s, cur[k], mid[k], 2, &p); if (rc2) return rc2; }
    }
    for (int k = 0; k < 2; ++k) { const int rc = note_bwd_finish(sts[k], *as[k], cur[k]); if (rc) return rc; }
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- encoder GRU BPTT
// Reverse of a2s_gru_seq_fwd for one direction.  dout: (B,T,*) gradient wrt this direction's outputs (column
// offset already applied); out: the forward outputs (hprev source); dhn: gradient wrt the final state.
//   dgi_all (B,T,3H) <- per-step input-projection gradients (caller: dW_ih, db_ih, dX via GEMMs)
//   dgh_shift (B,T,3H) <- dgh of the step whose h_prev is out[:,t]  (row (b,t) pairs with out[b,t]: dW_hh = dgh_shift^T out)
//   dgh_first (B,3H)   <- dgh of the first processed step (h_prev = 0): only contributes to db_hh
bool a2s_gru_step_fused_enabled(void);
int a2s_skinny_gemm_acc_impl(hipStream_t st, const float* A, long lda, const float* Bt, long ldb, float* Cm, long ldc, int R, int N, int K);
int a2s_gru_bptt_step_impl(hipStream_t st, const float* dgh, const float* w_hh_t, const float* dhz_in, const float* dout, long ld_dout,
                           const float* save, const float* hprev, long ld_hprev, float* dgi, long ld_dgi, float* dgh_out, float* dgh2,
                           long ld_dgh2, float* dhz_out, int R, int H);

bool a2s_gru_seq_bwd_persist_ok(int B, int T, int H, float* ws, size_t ws_bytes, size_t ws_used);
int a2s_gru_seq_bwd_persist_impl(hipStream_t st, const float* dout, long do_bstride, long do_tstride, const float* out, long out_bstride, long out_tstride,
                                 const float* gates, const float* w_hh_t, const float* dhn, float* dgi_all, float* dgh_shift, float* dgh_first, int B, int T,
                                 int H, int reverse, float* ws, size_t ws_off, size_t ws_bytes);

// out[c][r] = in[r][c]  (rows x cols -> cols x rows); small parameter matrices only
__global__ void transpose_f32(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int c = idx / rows, r = idx % rows;          // consecutive threads write consecutive out elements
    out[idx] = in[(long)r * cols + c];
}

int a2s_gru_seq_bwd_impl(hipStream_t st, const float* dout, long do_bstride, long do_tstride, const float* out, long out_bstride,
                         long out_tstride, const float* gates, const float* w_hh, const float* dhn, float* dgi_all, float* dgh_shift,
                         float* dgh_first, float* dhbuf, float* dgh_tmp, int B, int T, int H, int reverse, float* ws, size_t ws_bytes) {
    A2S_REQUIRE(dout && out && gates && w_hh && dgi_all && dgh_shift && dgh_first && dhbuf && dgh_tmp, "gru_seq_bwd: null tensor");
    hipError_t e;
    if (dhn) e = hipMemcpyAsync(dhbuf, dhn, sizeof(float) * B * H, hipMemcpyDeviceToDevice, st);
    else e = hipMemsetAsync(dhbuf, 0, sizeof(float) * B * H, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "gru_seq_bwd init: %s", hipGetErrorString(e));
    e = hipMemsetAsync(dgh_shift, 0, sizeof(float) * (size_t)B * T * 3 * H, st);
    if (e != hipSuccess) A2S_FAIL(A2S_ERR_HIP, "gru_seq_bwd memset: %s", hipGetErrorString(e));
    // one launch for dh_prev += dgh W_hh (skinny_gemm_acc, a2s_seq.hip): needs W_hh^T (H x 3H) so that both operands are
    // K-contiguous rows; the transposed copy lives in the workspace
    const bool fused = a2s_gru_step_fused_enabled() && H % 32 == 0 && ws && ws_bytes >= sizeof(float) * 3 * H * H && ((uintptr_t)ws % 16 == 0);
    if (fused) {
        hipLaunchKernelGGL(transpose_f32, dim3(a2s_cdiv(3 * H * H, 256)), dim3(256), 0, st, w_hh, ws, 3 * H, H);
        A2S_CHECK_LAUNCH("transpose_f32");
    }
    // one persistent launch for all T steps (a2s_persist.hip); its granule buffers live behind W_hh^T in the workspace
    if (fused && a2s_gru_seq_bwd_persist_ok(B, T, H, ws, ws_bytes, sizeof(float) * 3 * (size_t)H * H))
        return a2s_gru_seq_bwd_persist_impl(st, dout, do_bstride, do_tstride, out, out_bstride, out_tstride, gates, ws, dhn, dgi_all, dgh_shift, dgh_first, B, T, H,
                                            reverse, ws, sizeof(float) * 3 * (size_t)H * H, ws_bytes);
    if (fused && ws_bytes >= sizeof(float) * (3 * (size_t)H * H + 3 * (size_t)B * H) && (3 * H * H) % 4 == 0) {
        // ONE launch per step: the carry product of step s with the gate backward of step s-1 on its accumulators
        // (gru_bptt_step_fused); the two dgh scratch buffers alternate (this launch reads one as its A operand and writes the other)
        float* tmp[2] = {dgh_tmp, ws + 3 * (size_t)H * H};
        auto tix = [&](int s) { return reverse ? T - 1 - s : s; };
        int cur = 0, p = 0;
        {   // last processed step first: plain gate backward
            const int s = T - 1, t = tix(s), tp = reverse ? t + 1 : t - 1;
            float* dgh = (s == 0) ? dgh_first : tmp[p];
            int rc = a2s_gru_gates_bwd_impl(st, dhbuf, H, dout + (long)t * do_tstride, do_bstride, gates + (long)t * B * 4 * H,
                                            s == 0 ? nullptr : out + (long)tp * out_tstride, out_bstride,
                                            dgi_all + (long)t * 3 * H, (long)T * 3 * H, dgh, 3 * H,
                                            s == 0 ? nullptr : dgh_shift + (long)tp * 3 * H, (long)T * 3 * H, dhbuf + (long)B * H, H, B, H);
            if (rc) return rc;
            cur = 1;
        }
        for (int s = T - 1; s >= 1; --s) {
            const int sp = s - 1, t = tix(sp), tp = reverse ? t + 1 : t - 1;      // the epilogue handles step s-1
            float* dgh_out = (sp == 0) ? dgh_first : tmp[p ^ 1];
            int rc = a2s_gru_bptt_step_impl(st, tmp[p], ws, dhbuf + (long)cur * B * H, dout + (long)t * do_tstride, do_bstride,
                                            gates + (long)t * B * 4 * H, sp == 0 ? nullptr : out + (long)tp * out_tstride, out_bstride,
                                            dgi_all + (long)t * 3 * H, (long)T * 3 * H, dgh_out,
                                            sp == 0 ? nullptr : dgh_shift + (long)tp * 3 * H, (long)T * 3 * H,
                                            dhbuf + (long)(cur ^ 1) * B * H, B, H);
            if (rc) return rc;
            cur ^= 1; p ^= 1;
        }
        return A2S_OK;
    }
    int cur = 0;
    for (int s = T - 1; s >= 0; --s) {                 // s = processing index of the forward pass
        const int t = reverse ? T - 1 - s : s;          // time index of this step
        const int tp = reverse ? t + 1 : t - 1;         // time index whose output was h_prev (invalid when s == 0)
        float* dh_in = dhbuf + (long)cur * B * H;
        float* dh_out = dhbuf + (long)(cur ^ 1) * B * H;
        float* dgh = (s == 0) ? dgh_first : dgh_tmp;
        int rc = a2s_gru_gates_bwd_impl(st, dh_in, H, dout + (long)t * do_tstride, do_bstride, gates + (long)t * B * 4 * H,
                                        s == 0 ? nullptr : out + (long)tp * out_tstride, out_bstride,
                                        dgi_all + (long)t * 3 * H, (long)T * 3 * H, dgh, 3 * H,
                                        s == 0 ? nullptr : dgh_shift + (long)tp * 3 * H, (long)T * 3 * H, dh_out, H, B, H);
        if (rc) return rc;
        if (s > 0) {   // dh_prev += dgh W_hh
            if (fused) rc = a2s_skinny_gemm_acc_impl(st, dgh, 3 * H, ws, 3 * H, dh_out, H, B, H, 3 * H);
            else rc = a2s_gemm_impl(st, B, H, 3 * H, 1.f, dgh, 3 * H, 1, w_hh, H, 1, 1.f, dh_out, H, nullptr, 0, 1, 0, 0, 0, 0, ws, ws_bytes);
            if (rc) return rc;
        }
        cur ^= 1;
    }
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- staff embedding bwd
// BPTT of staff_emb_fwd for one (row, direction) per workgroup; hsave[(b*2+dir)*maxlen + s] = h after processing
// step s.  Weight gradients are accumulated per workgroup in LDS and added atomically once at the end; the
// embedding-table gradient is added atomically per step (duplicate ids).
__global__ __launch_bounds__(128) void staff_emb_bwd(const float* __restrict__ note_emb, const float* __restrict__ w_ih_f,
                                                     const float* __restrict__ w_hh_f, const float* __restrict__ b_ih_f,
                                                     const float* __restrict__ b_hh_f, const float* __restrict__ w_ih_r,
                                                     const float* __restrict__ w_hh_r, const float* __restrict__ b_ih_r,
                                                     const float* __restrict__ b_hh_r, float* const* __restrict__ grads /* 8 */,
                                                     float* __restrict__ note_emb_grad, const long long* __restrict__ ids64,
                                                     const int* __restrict__ ids32, long id_bstride, const long long* __restrict__ lengths,
                                                     long len_stride, const float* __restrict__ dout, long lddo, int col0,
                                                     const float* __restrict__ hsave, int maxlen, int E, int S) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const float* w_ih = dir ? w_ih_r : w_ih_f; const float* w_hh = dir ? w_hh_r : w_hh_f;
    const float* b_ih = dir ? b_ih_r : b_ih_f; const float* b_hh = dir ? b_hh_r : b_hh_f;
    float* Wi = sm;                         // 3S*E
    float* Wh = Wi + 3 * S * E;             // 3S*S
    float* gWi = Wh + 3 * S * S;            // 3S*E
    float* gWh = gWi + 3 * S * E;           // 3S*S
    float* gbi = gWh + 3 * S * S;           // 3S
    float* gbh = gbi + 3 * S;               // 3S
    float* bi = gbh + 3 * S; float* bh = bi + 3 * S;
    float* xe = bh + 3 * S;                 // E
    float* hp = xe + E;                     // S   h_prev
    float* g = hp + S;                      // 6S  gi | gh
    float* dgi = g + 6 * S;                 // 3S
    float* dgh = dgi + 3 * S;               // 3S
    float* dh = dgh + 3 * S;                // S
    for (int i = tid; i < 3 * S * E; i += nt) { Wi[i] = w_ih[i]; gWi[i] = 0.f; }
    for (int i = tid; i < 3 * S * S; i += nt) { Wh[i] = w_hh[i]; gWh[i] = 0.f; }
    for (int i = tid; i < 3 * S; i += nt) { bi[i] = b_ih[i]; bh[i] = b_hh[i]; gbi[i] = 0.f; gbh[i] = 0.f; }
    for (int i = tid; i < S; i += nt) dh[i] = dout[(long)b * lddo + col0 + dir * S + i];
    int len = (int)lengths[(long)b * len_stride];
    len = max(0, min(len, maxlen));
    // Latency: the recurrence is up to 398 dependent steps of a tiny cell, so anything that waits on global memory inside the loop is
    // paid 398 times (round 1: token id -> embedding row -> h_prev, three dependent round trips per step, ~10 us per step, 4 ms per
    // full-length row).  The row's token ids go to LDS once; the embedding row and h_prev of step s-1 are fetched into registers
    // while step s computes; the index arithmetic of the gradient accumulations uses shifts when E and S are powers of two.
    int* lid = reinterpret_cast<int*>(dh + S);          // maxlen ints behind the float scratch (the launcher sizes the LDS for it)
    for (int i = tid; i < len; i += nt) {
        const int t = dir ? len - 1 - i : i;
        lid[i] = ids64 ? (int)ids64[(long)b * id_bstride + t] : ids32[(long)b * id_bstride + t];
    }
    const bool pre = E <= nt && S <= nt;                // one element per thread: register prefetch
    const bool pow2 = (E & (E - 1)) == 0 && (S & (S - 1)) == 0;
    const int esh = 31 - __clz(E), ssh = 31 - __clz(S);
    __syncthreads();
    float xe_n = 0.f, hp_n = 0.f;
    if (pre && len > 0) {
        if (tid < E) xe_n = note_emb[(long)lid[len - 1] * E + tid];
        if (tid < S && len > 1) hp_n = hsave[(((long)b * 2 + dir) * maxlen + len - 2) * S + tid];
    }
    for (int s = len - 1; s >= 0; --s) {
        const long id = lid[s];
        if (pre) {
            if (tid < E) xe[tid] = xe_n;
            if (tid < S) hp[tid] = hp_n;
            if (s > 0) {                                 // next step's operands: in flight during this step's arithmetic
                if (tid < E) xe_n = note_emb[(long)lid[s - 1] * E + tid];
                if (tid < S) hp_n = s > 1 ? hsave[(((long)b * 2 + dir) * maxlen + s - 2) * S + tid] : 0.f;
            }
        } else {
            for (int i = tid; i < E; i += nt) xe[i] = note_emb[id * E + i];
            for (int i = tid; i < S; i += nt) hp[i] = s > 0 ? hsave[(((long)b * 2 + dir) * maxlen + s - 1) * S + i] : 0.f;
        }
        __syncthreads();
        for (int r = tid; r < 6 * S; r += nt) {          // recompute the pre-activations of this step
            float acc;
            if (r < 3 * S) { acc = bi[r]; for (int k = 0; k < E; ++k) acc = fmaf(Wi[r * E + k], xe[k], acc); }
            else { const int rr = r - 3 * S; acc = bh[rr]; for (int k = 0; k < S; ++k) acc = fmaf(Wh[rr * S + k], hp[k], acc); }
            g[r] = acc;
        }
        __syncthreads();
        if (tid < S) {
            const float rg = fast_sigmoid(g[tid] + g[3 * S + tid]);
            const float zg = fast_sigmoid(g[S + tid] + g[4 * S + tid]);
            const float ghn = g[5 * S + tid];
            const float ng = fast_tanh(g[2 * S + tid] + rg * ghn);
            const float d = dh[tid];
            const float dn = d * (1.f - zg) * (1.f - ng * ng);
            const float dz = d * (hp[tid] - ng) * zg * (1.f - zg);
            const float dr = dn * ghn * rg * (1.f - rg);
            dgi[tid] = dr; dgi[S + tid] = dz; dgi[2 * S + tid] = dn;
            dgh[tid] = dr; dgh[S + tid] = dz; dgh[2 * S + tid] = dn * rg;
            dh[tid] = d * zg;                              // direct path; recurrent path added below
        }
        __syncthreads();
        if (pow2) {
            for (int i = tid; i < 3 * S * E; i += nt) gWi[i] = fmaf(dgi[i >> esh], xe[i & (E - 1)], gWi[i]);
            for (int i = tid; i < 3 * S * S; i += nt) gWh[i] = fmaf(dgh[i >> ssh], hp[i & (S - 1)], gWh[i]);
        } else {
            for (int i = tid; i < 3 * S * E; i += nt) gWi[i] = fmaf(dgi[i / E], xe[i % E], gWi[i]);
            for (int i = tid; i < 3 * S * S; i += nt) gWh[i] = fmaf(dgh[i / S], hp[i % S], gWh[i]);
        }
        for (int i = tid; i < 3 * S; i += nt) { gbi[i] += dgi[i]; gbh[i] += dgh[i]; }
        // dx -> embedding row of this token, and the recurrent part of dh: two transposed matvecs of 3S terms each, dealt to ALL threads
        // (thread -> (output k, quarter of the 3S rows), partial sums meet in LDS) instead of E + S threads running 96-term chains
        float part = 0.f;
        const int nout = E + S;
        const int parts = max(1, min(nt / nout, (6 * S) / nout));      // 128 / 48 = 2 row partitions (and the partials must fit the 6S scratch)
        if (tid < nout * parts) {
            const int k = tid % nout, pr = tid / nout;
            const int r0 = (3 * S * pr) / parts, r1 = (3 * S * (pr + 1)) / parts;
            if (k < E) { for (int r = r0; r < r1; ++r) part = fmaf(dgi[r], Wi[r * E + k], part); }
            else { const int kk = k - E; for (int r = r0; r < r1; ++r) part = fmaf(dgh[r], Wh[r * S + kk], part); }
            g[tid] = part;                                 // g (6S floats >= nt) is free again after the gate math
        }
        __syncthreads();
        if (tid < nout) {
            float tot = 0.f;
            for (int pr = 0; pr < parts; ++pr) tot += g[pr * nout + tid];
            if (tid < E) atomicAdd(note_emb_grad + id * E + tid, tot);
            else dh[tid - E] += tot;
        }
        __syncthreads();
    }
    float* gwi = grads[dir * 4 + 0]; float* gwh = grads[dir * 4 + 1]; float* gb1 = grads[dir * 4 + 2]; float* gb2 = grads[dir * 4 + 3];
    for (int i = tid; i < 3 * S * E; i += nt) atomicAdd(gwi + i, gWi[i]);
    for (int i = tid; i < 3 * S * S; i += nt) atomicAdd(gwh + i, gWh[i]);
    for (int i = tid; i < 3 * S; i += nt) { atomicAdd(gb1 + i, gbi[i]); atomicAdd(gb2 + i, gbh[i]); }
}

// BPTT for the model's sizes (E = 16, S = 32), 192 threads, nothing but vectors in LDS (the generic kernel above spends 10 us per step
// of a <= 398-step chain: 16- / 32-way bank conflicts on W[r * E + k] with r = thread, and 4608 read-modify-writes of the weight-gradient
// images in LDS per step).  Thread r < 96 owns row r of W_ih AND of its gradient (16 + 16 registers), thread 96 + r row r of W_hh and of
// its gradient (32 + 32): the recomputed pre-activation and the rank-1 gradient update of a row are register arithmetic on operands
// broadcast from LDS.  For the transposed products (dx = W_ih^T dgi, dh_prev += W_hh^T dgh) every thread also holds a quarter of a
// COLUMN (24 registers): thread = (output k of 48, quarter of the 96 rows), the four partials meet in LDS.  The embedding row and
// h_{s-2} of the next step are in flight during the current one.  Four barriers per step.
int a2s_staff_emb_fast_enabled(void);
__global__ __launch_bounds__(192) void staff_emb_bwd_e16s32(const float* __restrict__ note_emb, const float* __restrict__ w_ih_f,
                                                            const float* __restrict__ w_hh_f, const float* __restrict__ b_ih_f,
                                                            const float* __restrict__ b_hh_f, const float* __restrict__ w_ih_r,
                                                            const float* __restrict__ w_hh_r, const float* __restrict__ b_ih_r,
                                                            const float* __restrict__ b_hh_r, float* const* __restrict__ grads /* 8 */,
                                                            float* __restrict__ note_emb_grad, const long long* __restrict__ ids64,
                                                            const int* __restrict__ ids32, long id_bstride, const long long* __restrict__ lengths,
                                                            long len_stride, const float* __restrict__ dout, long lddo, int col0,
                                                            const float* __restrict__ hsave, int maxlen) {
    constexpr int E = 16, S = 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xe = sm;                        // 16  x_s
    float* hp = xe + E;                    // 32  h_{s-1}
    float* g = hp + S;                     // 192 gi | gh, then the 4 x 48 partials of the transposed products
    float* dgi = g + 6 * S;                // 96
    float* dgh = dgi + 3 * S;              // 96
    int* lid = reinterpret_cast<int*>(dgh + 3 * S);    // maxlen
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const float* w_ih = dir ? w_ih_r : w_ih_f; const float* w_hh = dir ? w_hh_r : w_hh_f;
    const bool gi_role = tid < 3 * S;
    const int r = gi_role ? tid : tid - 3 * S;
    float w[S], gw[S];
#pragma unroll
    for (int k = 0; k < S; ++k) { w[k] = gi_role ? (k < E ? w_ih[r * E + k] : 0.f) : w_hh[r * S + k]; gw[k] = 0.f; }
    const float bias = gi_role ? (dir ? b_ih_r : b_ih_f)[r] : (dir ? b_hh_r : b_hh_f)[r];
    float gbias = 0.f;
    // transposed products: output k = tid % 48 (k < 16: dx[k] from W_ih / dgi; else dh_prev[k - 16] from W_hh / dgh), rows 24 q .. 24 q + 23
    const int tk = tid % 48, tq = tid / 48;
    float wc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) wc[i] = tk < E ? w_ih[(24 * tq + i) * E + tk] : w_hh[(24 * tq + i) * S + tk - E];
    const float* dgsrc = (tk < E ? dgi : dgh) + 24 * tq;
    int len = (int)lengths[(long)b * len_stride];
    len = max(0, min(len, maxlen));
    for (int i = tid; i < len; i += 192) {
        const int t = dir ? len - 1 - i : i;
        lid[i] = ids64 ? (int)ids64[(long)b * id_bstride + t] : ids32[(long)b * id_bstride + t];
    }
    float dh = tid < S ? dout[(long)b * lddo + col0 + dir * S + tid] : 0.f;      // thread j < 32: dh[j]
    __syncthreads();
    const float* hrow = hsave + ((long)b * 2 + dir) * maxlen * S;
    float xe_n = 0.f, hp_n = 0.f;
    if (len > 0) {
        if (tid < E) xe_n = note_emb[(long)lid[len - 1] * E + tid];
        if (tid < S && len > 1) hp_n = hrow[(long)(len - 2) * S + tid];
    }
    for (int s = len - 1; s >= 0; --s) {
        const long id = lid[s];
        if (tid < E) xe[tid] = xe_n;
        if (tid < S) hp[tid] = hp_n;
        const float hp_own = hp_n;                          // thread j < 32: h_{s-1}[j]
        if (s > 0) {                                         // the next step's operands: in flight during this step's arithmetic
            if (tid < E) xe_n = note_emb[(long)lid[s - 1] * E + tid];
            if (tid < S) hp_n = s > 1 ? hrow[(long)(s - 2) * S + tid] : 0.f;
        }
        __syncthreads();                                     // (A) x_s, h_{s-1} staged
        f32x4 op[S / 4];                                     // this role's operand vector (x_s: 4 quads, h_{s-1}: 8), kept for the rank-1 update
        float acc = bias;
#pragma unroll
        for (int k4 = 0; k4 < S / 4; ++k4) {
            if (gi_role && k4 >= E / 4) { op[k4] = (f32x4){0.f, 0.f, 0.f, 0.f}; continue; }
            op[k4] = gi_role ? reinterpret_cast<const f32x4*>(xe)[k4] : reinterpret_cast<const f32x4*>(hp)[k4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = fmaf(w[4 * k4 + c], op[k4][c], acc);
        }
        g[tid] = acc;
        __syncthreads();                                     // (B) pre-activations
        if (tid < S) {
            const float rg = fast_sigmoid(g[tid] + g[3 * S + tid]);
            const float zg = fast_sigmoid(g[S + tid] + g[4 * S + tid]);
            const float ghn = g[5 * S + tid];
            const float ng = fast_tanh(g[2 * S + tid] + rg * ghn);
            const float d = dh;
            const float dn = d * (1.f - zg) * (1.f - ng * ng);
            const float dz = d * (hp_own - ng) * zg * (1.f - zg);
            const float dr = dn * ghn * rg * (1.f - rg);
            dgi[tid] = dr; dgi[S + tid] = dz; dgi[2 * S + tid] = dn;
            dgh[tid] = dr; dgh[S + tid] = dz; dgh[2 * S + tid] = dn * rg;
            dh = d * zg;                                     // direct path; the recurrent path is added below
        }
        __syncthreads();                                     // (C) dgi, dgh
        {
            const float dg = gi_role ? dgi[r] : dgh[r];      // rank-1 update of this thread's gradient row
            gbias += dg;
#pragma unroll
            for (int k4 = 0; k4 < S / 4; ++k4) {
                if (gi_role && k4 >= E / 4) continue;
#pragma unroll
                for (int c = 0; c < 4; ++c) gw[4 * k4 + c] = fmaf(dg, op[k4][c], gw[4 * k4 + c]);
            }
            float part = 0.f;                                // quarter of a transposed product
#pragma unroll
            for (int i4 = 0; i4 < 6; ++i4) {
                const f32x4 d4 = reinterpret_cast<const f32x4*>(dgsrc)[i4];
#pragma unroll
                for (int c = 0; c < 4; ++c) part = fmaf(d4[c], wc[4 * i4 + c], part);
            }
            g[tid] = part;                                   // (g is free again after the gate math)
        }
        __syncthreads();                                     // (D) partials
        if (tid < S) dh += (g[E + tid] + g[48 + E + tid]) + (g[96 + E + tid] + g[144 + E + tid]);
        else if (tid >= 64 && tid < 64 + E) {
            const int k = tid - 64;
            atomicAdd(note_emb_grad + id * E + k, (g[k] + g[48 + k]) + (g[96 + k] + g[144 + k]));
        }
    }
    float* gwt = grads[dir * 4 + (gi_role ? 0 : 1)];
    float* gbt = grads[dir * 4 + (gi_role ? 2 : 3)];
    const int ncol = gi_role ? E : S;
#pragma unroll
    for (int k = 0; k < S; ++k) if (k < ncol) atomicAdd(gwt + r * ncol + k, gw[k]);
    atomicAdd(gbt + r, gbias);
}

int a2s_staff_emb_bwd_impl(hipStream_t st, const float* note_emb, const float* const* w, float* const* grads_dev, float* note_emb_grad,
                           const long long* ids64, const int* ids32, long id_bstride, const long long* lengths, long len_stride,
                           const float* dout, long lddo, int col0, const float* hsave, int R, int maxlen, int E, int S) {
    A2S_REQUIRE((ids64 != nullptr) != (ids32 != nullptr), "staff_emb_bwd: exactly one of ids64/ids32");
    A2S_REQUIRE(hsave && grads_dev && note_emb_grad && dout, "staff_emb_bwd: null tensor");
    A2S_REQUIRE(E <= 5 * S && E + S <= 128, "staff_emb_bwd: note_emb_size <= 5 * staff_emb_size and note_emb_size + staff_emb_size <= 128 expected");
    if (E == 16 && S == 32 && a2s_staff_emb_fast_enabled()) {
        const size_t shm16 = sizeof(float) * (E + S + 6 * S + 6 * S) + sizeof(int) * (size_t)maxlen;
        hipLaunchKernelGGL(staff_emb_bwd_e16s32, dim3(R, 2), dim3(192), shm16, st, note_emb, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7],
                           grads_dev, note_emb_grad, ids64, ids32, id_bstride, lengths, len_stride, dout, lddo, col0, hsave, maxlen);
        A2S_CHECK_LAUNCH("staff_emb_bwd_e16s32");
        return A2S_OK;
    }
    const size_t shm = sizeof(float) * (2 * (3 * S * E + 3 * S * S) + 4 * 3 * S + E + S + 6 * S + 6 * S + S) + sizeof(int) * (size_t)maxlen;
    hipLaunchKernelGGL(staff_emb_bwd, dim3(R, 2), dim3(128), shm, st, note_emb, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7],
                       grads_dev, note_emb_grad, ids64, ids32, id_bstride, lengths, len_stride, dout, lddo, col0, hsave, maxlen, E, S);
    A2S_CHECK_LAUNCH("staff_emb_bwd");
    return A2S_OK;
}

// =========================================================================================== split-T attention backward (H = 256)
// Same decomposition as attn_fwd_split256: G workgroups per clip, each streams its chunk of enc (for da_t) and of K (for dq) once
// with 16-byte loads; partial dq per (clip, g) merged by a small combine kernel.
template <bool NT>
__global__ __launch_bounds__(256) void attn_bwd_split256(const float* __restrict__ Kmat, const float* __restrict__ enc,
                                                         const float* __restrict__ q, long ldq, const float* __restrict__ v,
                                                         const float* __restrict__ attw, const float* __restrict__ ctx, long ldctx,
                                                         const float* __restrict__ dctx_a, long ldda, const float* __restrict__ dctx_b, long lddb,
                                                         float* __restrict__ dctx_out, long lddo, float* __restrict__ dq_partial,
                                                         float* __restrict__ ds_out, int T, int G, int chunk,
                                                         const int* __restrict__ row_order) {
    constexpr int H = 256;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dsv = sm;                                   // chunk
    f32x4* red4 = reinterpret_cast<f32x4*>(sm + chunk);   // 3 * 64 float4
    const int slot = blockIdx.x / G, g = blockIdx.x % G;
    const int b = row_order ? row_order[slot] : slot;
    const int t0 = g * chunk, t1 = min(T, t0 + chunk), n = t1 - t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Kb = Kmat + ((long)b * T + t0) * H;
    const float* Eb = enc + ((long)b * T + t0) * 2 * H;
    // dctx held by every wave: lane owns columns [4*lane, 4*lane+4) and [256 + 4*lane, ...)
    f32x4 dc0 = *reinterpret_cast<const f32x4*>(dctx_a + (long)b * ldda + lane * 4);
    f32x4 dc1 = *reinterpret_cast<const f32x4*>(dctx_a + (long)b * ldda + H + lane * 4);
    if (dctx_b) {
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(dctx_b + (long)b * lddb + lane * 4);
        const f32x4 o1 = *reinterpret_cast<const f32x4*>(dctx_b + (long)b * lddb + H + lane * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { dc0[c] += o0[c]; dc1[c] += o1[c]; }
    }
    if (dctx_out && g == 0 && wave == 0) {
        *reinterpret_cast<f32x4*>(dctx_out + (long)b * lddo + lane * 4) = dc0;
        *reinterpret_cast<f32x4*>(dctx_out + (long)b * lddo + H + lane * 4) = dc1;
    }
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(ctx + (long)b * ldctx + lane * 4);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(ctx + (long)b * ldctx + H + lane * 4);
    float dot_ctx = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) dot_ctx += dc0[c] * c0[c] + dc1[c] * c1[c];
    dot_ctx = wave_sum(dot_ctx);
    // ---- pass A: da_t = dctx . enc_t, one wave per frame, 4 frames in flight
    for (int r = wave * 4; r < n; r += 16) {
        float s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[u] = 0.f;
            if (r + u < n) {
                const f32x4 e0 = ld_kv<NT>(Eb + (long)(r + u) * 2 * H + lane * 4);
                const f32x4 e1 = ld_kv<NT>(Eb + (long)(r + u) * 2 * H + H + lane * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) s[u] += dc0[c] * e0[c] + dc1[c] * e1[c];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = wave_sum_lane63(s[u]);
        if (lane == 63) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + u < n) {
                    const float d_s = attw[(long)b * T + t0 + r + u] * (s[u] - dot_ctx);
                    dsv[r + u] = d_s;
                    if (ds_out) ds_out[(long)b * T + t0 + r + u] = d_s;
                }
        }
    }
    __syncthreads();
    // ---- pass B: dq_j += ds_t (1 - tanh^2(K_tj + q_j)); thread = (float4 column, row group of 4)
    const int c4 = tid & 63, rg = tid >> 6;
    f32x4 q4 = *reinterpret_cast<const f32x4*>(q + (long)b * ldq + c4 * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = exp2x_clamped(q4[c]);                 // E_q; Kmat holds the key image E_K = exp(2K)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int i = rg;
    for (; i + 12 < n; i += 16) {
        f32x4 k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = ld_kv<NT>(Kb + (long)(i + 4 * u) * H + c4 * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float w = dsv[i + 4 * u];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(w, sech2_ek(k[u][c], q4[c]), acc[c]);
        }
    }
    for (; i < n; i += 4) {
        const f32x4 k0 = ld_kv<NT>(Kb + (long)i * H + c4 * 4);
        const float w = dsv[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(w, sech2_ek(k0[c], q4[c]), acc[c]);
    }
    if (rg > 0) red4[(rg - 1) * 64 + c4] = acc;
    __syncthreads();
    if (rg == 0) {
        const f32x4 v4 = {v[c4 * 4], v[c4 * 4 + 1], v[c4 * 4 + 2], v[c4 * 4 + 3]};            // parameter: 4-byte aligned only
#pragma unroll
        for (int u = 0; u < 3; ++u) { const f32x4 o = red4[u * 64 + c4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += o[c]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] *= v4[c];
        *reinterpret_cast<f32x4*>(dq_partial + ((long)slot * G + g) * H + c4 * 4) = acc;
    }
}

// Fused bars (see attn_fwd_split256_mq): NQ rows of one clip per workgroup, the clip's enc and K chunk streamed once for all of them.
#define ATT_DCS 516             // floats per row of the dctx image in LDS (2H + 4: the rows of a clip land 4 banks apart)
#ifndef ATT_BWD_MQ_WAVES
// waves per SIMD the register budget is sized for (launch bounds); ATT_BWD_MQ_KPRE: K tiles of pass B requested ahead of pass A.  Round 6 measured
// (6, 1) -- 74-80 registers instead of 92, so that five bulk workgroups per CU leave a SIMD 112 registers for the long-clip chain's waves instead
// of 32 -- against (4, 2): 443.7 against 443.7 ms per step over three rounds (profiles/r06_lib_ab_variants.txt): the chain does not wait for registers.
#define ATT_BWD_MQ_WAVES 4
#define ATT_BWD_MQ_KPRE 2
#endif
template <int NQ, bool NT>
__global__ __launch_bounds__(256, ATT_BWD_MQ_WAVES) void attn_bwd_split256_mq(const float* __restrict__ Kmat, const float* __restrict__ enc,
                                                            const float* __restrict__ q, long ldq, const float* __restrict__ v,
                                                            const float* __restrict__ attw, const float* __restrict__ ctx, long ldctx,
                                                            const float* __restrict__ dctx_a, long ldda, const float* __restrict__ dctx_b, long lddb,
                                                            float* __restrict__ dctx_out, long lddo, float* __restrict__ dq_partial,
                                                            float* __restrict__ ds_out, int T, int G, int chunk,
                                                            const int* __restrict__ clip_order, const int* __restrict__ row_until,
                                                            int step, int n_clips) {
    constexpr int H = 256;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dsv = sm;                                         // NQ x chunk
    f32x4* red4 = reinterpret_cast<f32x4*>(sm + NQ * chunk);    // NQ * 3 * 64 float4
    const int slot = blockIdx.x / G, g = blockIdx.x % G;
    const int b = clip_order ? clip_order[slot] : slot;
    const int t0 = g * chunk, t1 = min(T, t0 + chunk), n = t1 - t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool on[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) on[j] = !row_until || step < row_until[j * n_clips + b];
    const float* Kb = Kmat + ((long)b * T + t0) * H;
    const float* Eb = enc + ((long)b * T + t0) * 2 * H;
    // dctx of the clip's rows (dctx_a + dctx_b) -> LDS as the B operand of pass A, their dot products with the saved contexts beside them
    // (round 5: the dctx image and the cross-group reduction scratch SHARE their LDS -- the image is last read in pass A, the scratch first written after
    // pass B, a barrier apart: 13.6 KB instead of 22 at four rows, which fits the 17 KB a CU has left beside the bulk group's backward sweeps)
    float* dcT = reinterpret_cast<float*>(red4);                           // NQ x ATT_DCS floats (8.3 KB at NQ = 4) inside red4's 12.3 KB
    float* dots = dcT + NQ * ATT_DCS;                                      // 16
    int onmask = 0;
#pragma unroll
    for (int j = 0; j < NQ; ++j) onmask |= on[j] ? (1 << j) : 0;
    // (one row per wave -- rows 0 .. 3 on waves 0 .. 3, a fifth on wave 0 again: the rows' operands are requested side by side instead of one
    // dependent round trip per row on wave 0 while three waves wait at the barrier)
    {
        for (int j = wave; j < NQ; j += 4) {
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
            float dot = 0.f;
            if ((onmask >> j) & 1) {
                const long row = (long)j * n_clips + b;
                d0 = *reinterpret_cast<const f32x4*>(dctx_a + row * ldda + lane * 4);
                d1 = *reinterpret_cast<const f32x4*>(dctx_a + row * ldda + H + lane * 4);
                if (dctx_b) {
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(dctx_b + row * lddb + lane * 4);
                    const f32x4 o1 = *reinterpret_cast<const f32x4*>(dctx_b + row * lddb + H + lane * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { d0[c] += o0[c]; d1[c] += o1[c]; }
                }
                if (dctx_out && g == 0) {
                    *reinterpret_cast<f32x4*>(dctx_out + row * lddo + lane * 4) = d0;
                    *reinterpret_cast<f32x4*>(dctx_out + row * lddo + H + lane * 4) = d1;
                }
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(ctx + row * ldctx + lane * 4);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(ctx + row * ldctx + H + lane * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) dot += d0[c] * c0[c] + d1[c] * c1[c];
                dot = wave_sum(dot);
            }
            *reinterpret_cast<f32x4*>(dcT + j * ATT_DCS + lane * 4) = d0;
            *reinterpret_cast<f32x4*>(dcT + j * ATT_DCS + H + lane * 4) = d1;
            if (lane == 0) dots[j] = dot;
        }
    }
    __syncthreads();
    // the first two K tiles of pass B (frames rg + 4 u + 16 p) are requested HERE: they arrive while pass A multiplies
    const int c4 = tid & 63, rg = tid >> 6;
    constexpr int KPRE = ATT_BWD_MQ_KPRE;
    f32x4 kpre[KPRE][4];
#pragma unroll
    for (int p = 0; p < KPRE; ++p)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            kpre[p][u] = (rg + 16 * p + 12 < n) ? ld_kv<NT>(Kb + (long)(rg + 16 * p + 4 * u) * H + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- pass A: da[t][j] = enc_t . dctx_j on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation): a wave
    // takes 16 frames, lane (i, g) feeds frame i's columns 32 u + 8 g .. + 7 as the A operand and row min(i, NQ - 1) of dctx at the same
    // columns as the B operand -- 128 MFMAs per 32 KB of enc instead of 8 FMAs + a 6-step cross-lane reduction per lane, frame and row
    // (with 2-5 rows per clip that arithmetic, not HBM, set the launch time: 4.5 / 3.1 TB/s at 2 / 5 rows against 5.4 at one).
    {
        const int li = lane & 15, lg = lane >> 4;
        const float* brow = dcT + min(li, NQ - 1) * ATT_DCS + 8 * lg;
        const float dotn = dots[min(li, NQ - 1)];
        const bool mine = li < NQ && ((onmask >> li) & 1);
        const long arow = ((long)min(li, NQ - 1) * n_clips + b) * T + t0;
        for (int blk = wave; blk * 16 < n; blk += 4) {
            const int fr = blk * 16 + li;
            const bool valid = fr < n;
            const float* ep = Eb + (long)min(fr, n - 1) * 2 * H + 8 * lg;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};       // two chains: a dependent 16x16x4 waits 40 cycles, an independent one 32
            float aw[4];                                                          // the saved weights of this lane's outputs: requested with the block, not after it
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f2 = blk * 16 + 4 * lg + r;
                aw[r] = (mine && f2 < n) ? attw[arow + f2] : 0.f;
            }
#pragma unroll 4
            for (int u = 0; u < 16; ++u) {
                f32x4 e0 = ld_kv<NT>(ep + 32 * u);
                f32x4 e1 = ld_kv<NT>(ep + 32 * u + 4);
                if (!valid) e0 = e1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(brow + 32 * u);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(brow + 32 * u + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[c], b0[c], acc, 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[c], b1[c], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
            // acc[r] = da[frame 16 blk + 4 lg + r][row li]
            if (mine) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f2 = blk * 16 + 4 * lg + r;
                    if (f2 < n) {
                        const float d_s = aw[r] * (acc[r] - dotn);
                        dsv[li * chunk + f2] = d_s;
                        if (ds_out) ds_out[arow + f2] = d_s;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- pass B: dq_j += ds_t (1 - tanh^2(K_tj + q_j)); thread = (float4 column, row group of 4)
    f32x4 q4[NQ], acc[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        q4[j] = on[j] ? *reinterpret_cast<const f32x4*>(q + ((long)j * n_clips + b) * ldq + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) q4[j][c] = exp2x_clamped(q4[j][c]);       // E_q; Kmat holds the key image E_K = exp(2K)
    }
    int i = rg;
    auto tile = [&](const f32x4 (&k)[4]) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (!on[j]) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float w = dsv[j * chunk + i + 4 * u];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[j][c] = fmaf(w, sech2_ek(k[u][c], q4[j][c]), acc[j][c]);
            }
        }
    };
#pragma unroll
    for (int p = 0; p < KPRE; ++p)
        if (i + 12 < n) { tile(kpre[p]); i += 16; }
    for (; i + 12 < n; i += 16) {
        f32x4 k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = ld_kv<NT>(Kb + (long)(i + 4 * u) * H + c4 * 4);
        tile(k);
    }
    for (; i < n; i += 4) {
        const f32x4 k0 = ld_kv<NT>(Kb + (long)i * H + c4 * 4);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (!on[j]) continue;
            const float w = dsv[j * chunk + i];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = fmaf(w, sech2_ek(k0[c], q4[j][c]), acc[j][c]);
        }
    }
    if (rg > 0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) red4[(j * 3 + rg - 1) * 64 + c4] = acc[j];
    }
    __syncthreads();
    if (rg == 0) {
        const f32x4 v4 = {v[c4 * 4], v[c4 * 4 + 1], v[c4 * 4 + 2], v[c4 * 4 + 3]};            // parameter: 4-byte aligned only
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (!on[j]) continue;
            f32x4 a = acc[j];
#pragma unroll
            for (int u = 0; u < 3; ++u) { const f32x4 o = red4[(j * 3 + u) * 64 + c4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] += o[c]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] *= v4[c];
            *reinterpret_cast<f32x4*>(dq_partial + (((long)slot * NQ + j) * G + g) * H + c4 * 4) = a;
        }
    }
}

// Both staves on one pass over the encoder outputs (round 6; forward: attn_fwd_split256_pair in a2s_seq.hip).  Pass A's matrix products have 16
// output columns and a clip has at most 5 fused bars: the lower staff's dctx rows ride in columns NQ .. 2 NQ - 1 of the SAME MFMAs, so the chunk's
// encoder rows are read once and multiplied once for both staves; pass B runs per staff over that staff's key image.  Per-staff outputs (ds, dq
// partials, summed dctx) exactly as attn_bwd_split256_mq writes them; the staves' own attn_bwd_combine256 launches follow.
struct AttnPairBwdSide { const float* Kmat; const float* q; const float* v; const float* attw; const float* ctx; const float* dctx_a; const float* dctx_b;
                         float* dctx_out; float* dq_partial; float* ds_out; const int* row_until; };
template <int NQ, bool NT>
__global__ __launch_bounds__(256, ATT_BWD_MQ_WAVES) void attn_bwd_split256_pair(AttnPairBwdSide s0, AttnPairBwdSide s1, const float* __restrict__ enc, long ldq,
                                                                                 long ldctx, long ldda, long lddb, long lddo, int T, int G, int chunk,
                                                                                 const int* __restrict__ clip_order, int step, int n_clips) {
    constexpr int H = 256, NJ = 2 * NQ;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* dsv = sm;                                            // NJ x chunk
    f32x4* red4 = reinterpret_cast<f32x4*>(sm + NJ * chunk);    // NQ * 3 * 64 float4 (one staff at a time) | the dctx image
    const int slot = blockIdx.x / G, g = blockIdx.x % G;
    const int b = clip_order ? clip_order[slot] : slot;
    const int t0 = g * chunk, t1 = min(T, t0 + chunk), n = t1 - t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int onmask = 0;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        onmask |= (!s0.row_until || step < s0.row_until[j * n_clips + b]) ? (1 << j) : 0;
        onmask |= (!s1.row_until || step < s1.row_until[j * n_clips + b]) ? (1 << (NQ + j)) : 0;
    }
    const float* Eb = enc + ((long)b * T + t0) * 2 * H;
    float* dcT = reinterpret_cast<float*>(red4);                // NJ x ATT_DCS floats
    float* dots = dcT + NJ * ATT_DCS;                           // 16
    for (int j = wave; j < NJ; j += 4) {
        const AttnPairBwdSide& sd = j < NQ ? s0 : s1;
        const int jr = j < NQ ? j : j - NQ;
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        float dot = 0.f;
        if ((onmask >> j) & 1) {
            const long row = (long)jr * n_clips + b;
            d0 = *reinterpret_cast<const f32x4*>(sd.dctx_a + row * ldda + lane * 4);
            d1 = *reinterpret_cast<const f32x4*>(sd.dctx_a + row * ldda + H + lane * 4);
            if (sd.dctx_b) {
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(sd.dctx_b + row * lddb + lane * 4);
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(sd.dctx_b + row * lddb + H + lane * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) { d0[c] += o0[c]; d1[c] += o1[c]; }
            }
            if (sd.dctx_out && g == 0) {
                *reinterpret_cast<f32x4*>(sd.dctx_out + row * lddo + lane * 4) = d0;
                *reinterpret_cast<f32x4*>(sd.dctx_out + row * lddo + H + lane * 4) = d1;
            }
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(sd.ctx + row * ldctx + lane * 4);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(sd.ctx + row * ldctx + H + lane * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) dot += d0[c] * c0[c] + d1[c] * c1[c];
            dot = wave_sum(dot);
        }
        *reinterpret_cast<f32x4*>(dcT + j * ATT_DCS + lane * 4) = d0;
        *reinterpret_cast<f32x4*>(dcT + j * ATT_DCS + H + lane * 4) = d1;
        if (lane == 0) dots[j] = dot;
    }
    __syncthreads();
    const int c4 = tid & 63, rg = tid >> 6;
    constexpr int KPRE = ATT_BWD_MQ_KPRE;
    const int pre_side = (onmask & ((1 << NQ) - 1)) ? 0 : 1;          // the staff pass B starts with: its first K tiles are requested before pass A
    f32x4 kpre[KPRE][4];
    {
        const float* Kp = (pre_side ? s1.Kmat : s0.Kmat) + ((long)b * T + t0) * H;
#pragma unroll
        for (int p = 0; p < KPRE; ++p)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                kpre[p][u] = (rg + 16 * p + 12 < n) ? ld_kv<NT>(Kp + (long)(rg + 16 * p + 4 * u) * H + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // ---- pass A (see attn_bwd_split256_mq): da[t][j] = enc_t . dctx_j for the rows of BOTH staves in one set of matrix products
    {
        const int li = lane & 15, lg = lane >> 4;
        const int lr = min(li, NJ - 1);
        const float* brow = dcT + lr * ATT_DCS + 8 * lg;
        const float dotn = dots[lr];
        const bool mine = li < NJ && ((onmask >> li) & 1);
        const AttnPairBwdSide& sd = lr < NQ ? s0 : s1;
        const long arow = ((long)(lr < NQ ? lr : lr - NQ) * n_clips + b) * T + t0;
        const float* attw = sd.attw;
        float* ds_out = sd.ds_out;
        for (int blk = wave; blk * 16 < n; blk += 4) {
            const int fr = blk * 16 + li;
            const bool valid = fr < n;
            const float* ep = Eb + (long)min(fr, n - 1) * 2 * H + 8 * lg;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
            float aw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f2 = blk * 16 + 4 * lg + r;
                aw[r] = (mine && f2 < n) ? attw[arow + f2] : 0.f;
            }
#pragma unroll 4
            for (int u = 0; u < 16; ++u) {
                f32x4 e0 = ld_kv<NT>(ep + 32 * u);
                f32x4 e1 = ld_kv<NT>(ep + 32 * u + 4);
                if (!valid) e0 = e1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(brow + 32 * u);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(brow + 32 * u + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[c], b0[c], acc, 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[c], b1[c], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
            if (mine) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f2 = blk * 16 + 4 * lg + r;
                    if (f2 < n) {
                        const float d_s = aw[r] * (acc[r] - dotn);
                        dsv[li * chunk + f2] = d_s;
                        if (ds_out) ds_out[arow + f2] = d_s;
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- pass B, per staff: dq_j += ds_t (1 - tanh^2(K_tj + q_j)) over that staff's key image
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int side = pass == 0 ? pre_side : 1 - pre_side;
        const int smask = (onmask >> (side * NQ)) & ((1 << NQ) - 1);
        if (!smask) continue;                                      // uniform over the workgroup
        const AttnPairBwdSide& sd = side ? s1 : s0;
        const float* Kb = sd.Kmat + ((long)b * T + t0) * H;
        const float* dsj = dsv + side * NQ * chunk;
        f32x4 q4[NQ], acc[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            q4[j] = ((smask >> j) & 1) ? *reinterpret_cast<const f32x4*>(sd.q + ((long)j * n_clips + b) * ldq + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) q4[j][c] = exp2x_clamped(q4[j][c]);
        }
        int i = rg;
        auto tile = [&](const f32x4 (&k)[4]) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (!((smask >> j) & 1)) continue;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float w = dsj[j * chunk + i + 4 * u];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[j][c] = fmaf(w, sech2_ek(k[u][c], q4[j][c]), acc[j][c]);
                }
            }
        };
        if (pass == 0) {
#pragma unroll
            for (int p = 0; p < KPRE; ++p)
                if (i + 12 < n) { tile(kpre[p]); i += 16; }
        }
        for (; i + 12 < n; i += 16) {
            f32x4 k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = ld_kv<NT>(Kb + (long)(i + 4 * u) * H + c4 * 4);
            tile(k);
        }
        for (; i < n; i += 4) {
            const f32x4 k0 = ld_kv<NT>(Kb + (long)i * H + c4 * 4);
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (!((smask >> j) & 1)) continue;
                const float w = dsj[j * chunk + i];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[j][c] = fmaf(w, sech2_ek(k0[c], q4[j][c]), acc[j][c]);
            }
        }
        if (pass == 1) __syncthreads();                            // the first staff's reduction is done with the scratch
        if (rg > 0) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) red4[(j * 3 + rg - 1) * 64 + c4] = acc[j];
        }
        __syncthreads();
        if (rg == 0) {
            const f32x4 v4 = {sd.v[c4 * 4], sd.v[c4 * 4 + 1], sd.v[c4 * 4 + 2], sd.v[c4 * 4 + 3]};
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                if (!((smask >> j) & 1)) continue;
                f32x4 a = acc[j];
#pragma unroll
                for (int u = 0; u < 3; ++u) { const f32x4 o = red4[(j * 3 + u) * 64 + c4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) a[c] += o[c]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] *= v4[c];
                *reinterpret_cast<f32x4*>(sd.dq_partial + (((long)slot * NQ + j) * G + g) * H + c4 * 4) = a;
            }
        }
    }
}

// one workgroup (256 threads) per row: dq = sum of the G partials; rows the forward pass skipped (upstream gradient exactly zero) get
// zeros in everything the deferred GEMMs read (dq, ds, dctx)
__global__ __launch_bounds__(256) void attn_bwd_combine256(const float* __restrict__ dq_partial, float* __restrict__ dq, long lddq, int G,
                                                           const int* __restrict__ clip_rank, const int* __restrict__ row_until,
                                                           int n_clips, int groups, int n_active, int step, float* __restrict__ ds_out,
                                                           int T, float* __restrict__ dctx_out, long lddo) {
    const int b = blockIdx.x, j = threadIdx.x;
    const int clip = b % n_clips, grp = b / n_clips;
    const int slot = clip_rank ? clip_rank[clip] : clip;
    if (slot >= n_active || (row_until && step >= row_until[b])) {
        dq[(long)b * lddq + j] = 0.f;
        if (ds_out) for (int t = j; t < T; t += 256) ds_out[(long)b * T + t] = 0.f;
        if (dctx_out) for (int d = j; d < 512; d += 256) dctx_out[(long)b * lddo + d] = 0.f;
        return;
    }
    const float* pb = dq_partial + ((long)slot * groups + grp) * G * 256;
    float s = 0.f;
    for (int g0 = 0; g0 < G; g0 += 8) {               // 8 independent loads per round trip
        float p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = (g0 + u < G) ? pb[(long)(g0 + u) * 256 + j] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += p[u];
    }
    dq[(long)b * lddq + j] = s;
}

size_t a2s_attn_bulk_lds(size_t shm, int n_active, int backward);
template <int NQ>
static void launch_bwd_mq(hipStream_t st, int nwg, size_t shm, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                          const float* attw, const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b, long lddb,
                          float* dctx_out, long lddo, float* ws, float* ds_out, int T, int G, int chunk, const a2s_attn_rows& r, bool nt) {
    if (nt) hipLaunchKernelGGL((attn_bwd_split256_mq<NQ, true>), dim3(nwg), dim3(256), shm, st, Kmat, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb,
                               dctx_out, lddo, ws, ds_out, T, G, chunk, r.clip_order, r.row_until, r.step, r.n_clips);
    else hipLaunchKernelGGL((attn_bwd_split256_mq<NQ, false>), dim3(nwg), dim3(256), shm, st, Kmat, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb,
                            dctx_out, lddo, ws, ds_out, T, G, chunk, r.clip_order, r.row_until, r.step, r.n_clips);
}

int a2s_attn_step_bwd_split_impl(hipStream_t st, const float* Kmat, const float* enc, const float* q, long ldq, const float* v,
                                 const float* attw, const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b,
                                 long lddb, float* dctx_out, long lddo, float* dq, long lddq, float* ds_out, float* ws, int B, int T, int H,
                                 const a2s_attn_rows* rows) {
    A2S_REQUIRE(H == 256 && ws, "attn_step_bwd_split: needs hidden_size 256 and a workspace");
    A2S_REQUIRE(ldq % 4 == 0 && ldctx % 4 == 0 && ldda % 4 == 0 && (!dctx_b || lddb % 4 == 0) && (!dctx_out || lddo % 4 == 0),
                "attn_step_bwd_split: row strides must be multiples of 4 floats");
    A2S_REQUIRE(((uintptr_t)q | (uintptr_t)ctx | (uintptr_t)dctx_a | (uintptr_t)dctx_b | (uintptr_t)dctx_out | (uintptr_t)Kmat | (uintptr_t)enc) % 16 == 0,
                "attn_step_bwd_split: 16-byte alignment");
    a2s_attn_rows r = {nullptr, nullptr, nullptr, B, B, 0};
    if (rows) r = *rows;
    A2S_REQUIRE(r.n_clips > 0 && B % r.n_clips == 0, "attn_step_bwd_split: rows (%d) must be a multiple of the clips (%d)", B, r.n_clips);
    const int groups = B / r.n_clips;
    A2S_REQUIRE(groups <= A2S_ATTN_MAX_GROUPS, "attn_step_bwd_split: at most %d fused bars (got %d)", A2S_ATTN_MAX_GROUPS, groups);
    A2S_REQUIRE(r.n_active >= 0 && r.n_active <= r.n_clips && (!r.clip_order || r.clip_rank), "attn_step_bwd_split: bad row compaction");
    int G = 1, chunk = T;
    ws += A2S_ATTN_TICKETS;                 // the head of the workspace holds the arrival counters of the fused combines (a2s_seq.hip)
    // streaming loads as in a2s_attn_step_fwd_split_impl -- single-row launches only: measured +12 % on attn_bwd_split256, -0 .. 5 % on the
    // fused-rows kernels, whose enc pass feeds the matrix cores (profiles/r04_attn_mq_bench.txt)
    const bool nt = a2s_attn_nt_enabled() > 0 && r.n_active >= a2s_attn_nt_enabled() && groups == 1;
    if (r.n_active > 0) {
        a2s_attn_split_geometry(r.n_active, T, &G, &chunk);
        // a handful of clips (late in the long-clip group's chain: the one or two clips that hold a full-length bar): a finer split -- the launch
        // is a chain of dependent passes over the chunk, not bandwidth; the dq partials of 2 G chunks still fit the workspace the forward sized
        // (2 G x 256 <= G x 516 floats per row).  Up to 4 clips (measured 458.1 / 457.8 -> 454.7 / 456.1 ms per step, tools/step_time.py,
        // alternating processes: profiles/r05_prefix_percent.txt)
        {
            const int fine = 4;
            if (fine > 0 && r.n_active <= fine && G == a2s_attn_max_split() && 2 * G <= 64) {
                int c = (T + 2 * G - 1) / (2 * G);
                c = (c + 3) & ~3;
                G = (T + c - 1) / c;
                chunk = c;
            }
        }
        const int nwg = r.n_active * G;
        if (groups == 1) {
            const size_t shm = a2s_attn_bulk_lds((chunk + 3 * 64 * 4) * sizeof(float), r.n_active, 1);
            if (nt) hipLaunchKernelGGL(attn_bwd_split256<true>, dim3(nwg), dim3(256), shm, st, Kmat, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb,
                                       dctx_out, lddo, ws, ds_out, T, G, chunk, r.clip_order);
            else hipLaunchKernelGGL(attn_bwd_split256<false>, dim3(nwg), dim3(256), shm, st, Kmat, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb,
                                    dctx_out, lddo, ws, ds_out, T, G, chunk, r.clip_order);
        } else {
            // [ds rows][max(reduction scratch: groups x 3 x 64 float4, dctx image + dots: groups x ATT_DCS + 16 floats)]
            const size_t scratch = (size_t)groups * 3 * 64 * 4 > (size_t)groups * ATT_DCS + 16 ? (size_t)groups * 3 * 64 * 4 : (size_t)groups * ATT_DCS + 16;
            const size_t shm = a2s_attn_bulk_lds(((size_t)groups * chunk + scratch) * sizeof(float), r.n_active, 3);
#define A2S_BWD_MQ(N) launch_bwd_mq<N>(st, nwg, shm, Kmat, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb, dctx_out, lddo, ws, ds_out, T, G, chunk, r, nt)
            switch (groups) {
                case 2: A2S_BWD_MQ(2); break;
                case 3: A2S_BWD_MQ(3); break;
                case 4: A2S_BWD_MQ(4); break;
                default: A2S_BWD_MQ(5); break;
            }
#undef A2S_BWD_MQ
        }
        A2S_CHECK_LAUNCH("attn_bwd_split256");
    }
    hipLaunchKernelGGL(attn_bwd_combine256, dim3(B), dim3(256), 0, st, ws, dq, lddq, G, r.clip_rank, r.row_until, r.n_clips, groups, r.n_active, r.step,
                       ds_out, T, dctx_out, lddo);
    A2S_CHECK_LAUNCH("attn_bwd_combine256");
    return A2S_OK;
}

template <int NQ>
static void launch_bwd_pair(hipStream_t st, int nwg, size_t shm, const AttnPairBwdSide& s0, const AttnPairBwdSide& s1, const float* enc, long ldq, long ldctx,
                            long ldda, long lddb, long lddo, int T, const AttnPairBwdStep& p) {
    hipLaunchKernelGGL((attn_bwd_split256_pair<NQ, false>), dim3(nwg), dim3(256), shm, st, s0, s1, enc, ldq, ldctx, ldda, lddb, lddo, T, p.G, p.chunk,
                       p.clip_order, p.step, p.n_clips);
}
static long g_attn_pair_bwd_launches = 0;
long a2s_attn_pair_bwd_launches(void) { return g_attn_pair_bwd_launches; }

static AttnPairBwdSide attn_pair_bwd_side(const a2s_note_dec_bwd_args& a, int s) {
    const int H2 = 2 * a.H, ldx = a.E + H2;
    const long R = a.R;
    return AttnPairBwdSide{a.keys, a.q + (long)s * R * a.H, a.attn_v, a.attw + (long)s * R * a.T, a.x + (long)s * R * ldx + a.E, a.dx + (long)s * R * ldx + a.E,
                           a.do_all + (long)s * R * 2 * H2 + H2, a.dctx_all + (long)s * R * H2, a.attn_ws + A2S_ATTN_TICKETS, a.ds_all + (long)s * R * a.T, a.row_until};
}

static int attn_pair_bwd_sweep(hipStream_t st, const a2s_note_dec_bwd_args& au, const a2s_note_dec_bwd_args& al, int s, AttnPairBwdStep& p) {
    const int T = au.T, H2 = 2 * au.H, ldx = au.E + H2, groups = au.R / p.n_clips;
    A2S_REQUIRE(au.H == 256 && au.E == al.E && au.enc == al.enc && au.R == al.R && groups >= 1 && groups <= A2S_ATTN_MAX_GROUPS && ldx % 4 == 0,
                "attn_pair_bwd_sweep: the staves must decode the same rows over the same encoder outputs");
    a2s_attn_split_geometry(p.n_active, T, &p.G, &p.chunk);
    const AttnPairBwdSide s0 = attn_pair_bwd_side(au, s), s1 = attn_pair_bwd_side(al, s);
    const size_t red = (size_t)groups * 3 * 64 * 4, img = (size_t)2 * groups * ATT_DCS + 16;
    const size_t shm = a2s_attn_bulk_lds(((size_t)2 * groups * p.chunk + (red > img ? red : img)) * sizeof(float), p.n_active, 3);
    const int nwg = p.n_active * p.G;
#define A2S_BWD_PAIR(N) launch_bwd_pair<N>(st, nwg, shm, s0, s1, au.enc, au.H, ldx, ldx, 2 * H2, H2, T, p)
    switch (groups) {
        case 1: A2S_BWD_PAIR(1); break;
        case 2: A2S_BWD_PAIR(2); break;
        case 3: A2S_BWD_PAIR(3); break;
        case 4: A2S_BWD_PAIR(4); break;
        default: A2S_BWD_PAIR(5); break;
    }
#undef A2S_BWD_PAIR
    A2S_CHECK_LAUNCH("attn_bwd_split256_pair");
    __atomic_fetch_add(&g_attn_pair_bwd_launches, 1, __ATOMIC_RELAXED);
    return A2S_OK;
}

// one staff's dq reduction behind a pair sweep (rows the forward skipped: zeros in dq, ds, dctx)
static int attn_pair_bwd_combine(hipStream_t st, const a2s_note_dec_bwd_args& a, int s, const AttnPairBwdStep& p) {
    const int H2 = 2 * a.H;
    const long R = a.R;
    hipLaunchKernelGGL(attn_bwd_combine256, dim3(a.R), dim3(256), 0, st, a.attn_ws + A2S_ATTN_TICKETS, a.dq_all + (long)s * R * a.H, (long)a.H, p.G, p.clip_rank,
                       a.row_until, p.n_clips, a.R / p.n_clips, p.n_active, p.step, a.ds_all + (long)s * R * a.T, a.T, a.dctx_all + (long)s * R * H2, (long)H2);
    A2S_CHECK_LAUNCH("attn_bwd_combine256");
    return A2S_OK;
}
