// extern "C" surface of liba2s_hip.so (declared in include/a2s.h): thin, exception-free trampolines onto the
// *_impl launchers of a2s_gemm.hip / a2s_conv.hip / a2s_seq.hip.
#include "a2s_common.h"
#include "../../include/a2s.h"

thread_local char a2s_err_msg[512] = {0};
long long a2s_launch_counter = 0;

// ---- launchers implemented in the kernel translation units
int a2s_gemm_impl(hipStream_t, int, int, int, float, const float*, long, long, const float*, long, long, float, float*, long,
                  const float*, int, int, long, long, long, int, float*, size_t);
size_t a2s_gemm_workspace_bytes_impl(int, int, int, int);
int a2s_gemm_affine_impl(hipStream_t, int, int, int, float, const float*, long, long, const float*, long, long, float, float*, long, const float*,
                         int, int, long, long, long, int, float*, size_t, const float*, const float*, int, const float*, const float*, int,
                         const float*, const float*, const float*, const float*, const float*, float*, int, int, const float*, const float*);
int a2s_absmax_impl(hipStream_t, const float*, long, float*);
void a2s_gemm_f16x2_set(int);
int a2s_gemm_f16x2_enabled(void);
int a2s_gemm_bnstats_slots(int);
int a2s_conv3x3_impl(hipStream_t, const float*, const float*, float*, const float*, const float*, float*, int, int, int, int, int, int, float*,
                     const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*);
void a2s_conv_rows_set(int);
void a2s_conv_c1_fast_set(int);
int a2s_conv_c1_fast_enabled(void);
int a2s_conv_rows_enabled(void);
void a2s_wgrad_rows_set(int);
int a2s_wgrad_rows_enabled(void);
void a2s_staff_emb_fast_set(int);
int a2s_staff_emb_fast_enabled(void);
void a2s_conv_f16x2_set(int);
int a2s_conv_f16x2_enabled(void);
size_t a2s_conv3x3_workspace_floats_impl(int);
void a2s_conv_bf16x3_set(int);
void a2s_gemm_split_set(int);
void a2s_wgrad_split_set(int);
int a2s_wgrad_split_enabled(void);
int a2s_gemm_split_enabled(void);
int a2s_conv_bf16x3_enabled(void);
int a2s_conv3x3_stat_blocks_impl(int, int, int, int);
int a2s_bn_finalize_impl(hipStream_t, const float*, int, int, double, const float*, const float*, float*, float*, long long*,
                         float*, float*, float*, float*, float, float, int);
int a2s_bn_relu_apply_impl(hipStream_t, const float*, float*, const float*, const float*, long, int, int);
int a2s_col_stats_impl(hipStream_t, const float*, float*, long, int, int);
int a2s_bn1d_relu_dropout_impl(hipStream_t, const float*, float*, const float*, const float*, const uint8_t*, float, long, int);
int a2s_gru_gates_fwd_impl(hipStream_t, const float*, long, const float*, long, const float*, long, float*, long, float*, long, float*, int, int);
int a2s_gru_seq_fwd_impl(hipStream_t, const float*, long, long, const float*, const float*, float*, long, long, float*, float*,
                         float*, float*, int, int, int, int, float*, size_t);
int a2s_attn_step_fwd_impl(hipStream_t, const float*, const float*, const float*, long, const float*, float*, long, float*, long,
                           float*, int, int, int, const int*, int, float*, const a2s_attn_rows*, a2s_attn_deferred* = nullptr);
size_t a2s_attn_workspace_floats_impl(int, int, int, int);
int a2s_log_softmax_rows_impl(hipStream_t, const float*, long, float*, long, int*, int, int);
int a2s_embed_rows_impl(hipStream_t, const float*, const long long*, const int*, long, int, float*, long, int, int, int, const uint8_t*, float);
int a2s_staff_emb_fwd_impl(hipStream_t, const float*, const float* const*, const long long*, const int*, long, const long long*, long,
                           float*, long, int, float*, int, int, int, int);
int a2s_gemm_pick_splitk_impl(int M, int N, int K, int batch);
void a2s_gemm_debug_tile_impl(int);
void a2s_gru_step_fused_set(int);
void a2s_gru_persist_set(int);
int a2s_gru_persist_enabled(void);
void a2s_dec_persist_set(int);
void a2s_gru_persist_alone_set(int);
int a2s_gru_persist_alone(void);
void a2s_attn_deep_set(int);
void a2s_attn_defer_combine_set(int);
void a2s_dec_mid_set(int);
int a2s_dec_mid_enabled(void);
int a2s_dec_mid_launches(void);
int a2s_attn_defer_combine_enabled(void);
int a2s_attn_deep_max_clips(void);
int a2s_dec_persist_launches(void);
int a2s_dec_persist_enabled(void);
size_t a2s_note_decoder_persist_ws_bytes(int n_clips, int R, int steps);
size_t a2s_note_decoder_bwd_persist_ws_bytes(int n_clips);
int a2s_nll_grad_impl(hipStream_t st, float* dlogp, const long long* target, const float* loss_out, float gscale, long rows, int V, long long ignore_index);
void a2s_attn_fused_combine_set(int v);
int a2s_attn_fused_combine_enabled(void);
void a2s_dec_fused_set(int v);
void a2s_dec_fused_max_rows_set(int v);
int a2s_dec_fused_enabled(void);
int a2s_dec_fused_max_rows(void);
size_t a2s_note_step_workspace_floats_impl(int H, int E);

bool a2s_gru_step_fused_enabled(void);
int a2s_note_decoder_fwd_impl(hipStream_t st, const a2s_note_dec_args& a, int* steps_done);
int a2s_note_decoder_fwd_pair_impl(hipStream_t su, hipStream_t sl, const a2s_note_dec_args& au, const a2s_note_dec_args& al, const int* pair_order,
                                   const int* pair_rank, const int* pair_n_active, int* done_u, int* done_l);
void a2s_attn_pair_set(int);
int a2s_attn_pair_enabled(void);
long a2s_attn_pair_launches(void);
long a2s_attn_pair_bwd_launches(void);
void a2s_attn_pair_fused_rows_set(int);
int a2s_attn_pair_fused_rows(void);
int a2s_note_decoder_bwd_pair_impl(hipStream_t su, hipStream_t sl, const a2s_note_dec_bwd_args& au, const a2s_note_dec_bwd_args& al, const int* pair_order,
                                   const int* pair_rank, const int* pair_n_active);

int a2s_log_softmax_bwd_rows_impl(hipStream_t, const float*, const float*, long, int, float*, int, int, int, int);
int a2s_gru_gates_bwd_impl(hipStream_t, const float*, long, const float*, long, const float*, const float*, long, float*, long, float*, long,
                           float*, long, float*, long, int, int);
int a2s_attn_step_bwd_impl(hipStream_t, const float*, const float*, const float*, long, const float*, const float*, const float*, long,
                           const float*, long, const float*, long, float*, long, float*, long, float*, int, int, int, float*, const a2s_attn_rows*);
int a2s_attn_dk_accum_impl(hipStream_t, const float*, const float*, const float*, const float*, float*, float*, int, int, int, int, const int*, int);
int a2s_col_sum_impl(hipStream_t, const float*, long, float*, long, int, float, float, float*, size_t);
int a2s_embed_scatter_add_impl(hipStream_t, float*, const long long*, const int*, long, int, const float*, long, int, int, int, const uint8_t*, float);
int a2s_ew_act_bwd_impl(hipStream_t, const float*, const float*, float*, long, int);
int a2s_note_decoder_bwd_impl(hipStream_t, const a2s_note_dec_bwd_args&);
int a2s_gru_seq_bwd_impl(hipStream_t, const float*, long, long, const float*, long, long, const float*, const float*, const float*, float*,
                         float*, float*, float*, float*, int, int, int, int, float*, size_t);
int a2s_staff_emb_bwd_impl(hipStream_t, const float*, const float* const*, float* const*, float*, const long long*, const int*, long,
                           const long long*, long, const float*, long, int, const float*, int, int, int, int);

int a2s_bn_bwd_impl(hipStream_t, const float*, const float*, const float*, const float*, const float*, const float*, const uint8_t*, float,
                    float*, float*, float*, float*, float*, long, int, int, float*);
size_t a2s_bn_bwd_partial_floats_impl(long, int, int);
int a2s_bn_bwd_from_partial_impl(hipStream_t, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*,
                                 const float*, int, float*, long, int, int, float*);

int a2s_conv3x3_wgrad_impl(hipStream_t, const float*, const float*, const float*, const float*, float*, float*, size_t, int, int, int, int, int,
                           const float*, const float*, const float*, const float*, const float*, const float*, float*, const float*, const float*);
int a2s_act_bound_impl(hipStream_t, const float*, const float*, const float*, int, float*);
void a2s_wgrad_f16x2_set(int);
int a2s_wgrad_f16x2_enabled(void);
size_t a2s_conv3x3_wgrad_workspace_bytes_impl(int, int);

int a2s_nll_loss_impl(hipStream_t, const float*, const long long*, long, int, long long, float*, float*, float, double*, int);
int a2s_clip_adadelta_impl(hipStream_t, float*, float*, float*, float*, long, const float*, float, float, float, float, float*, double*, int, int);

int a2s_vqt_logmag_impl(hipStream_t, const float*, float*, float*, int, long, int, float);
int a2s_vqt_logmag_octaves_impl(hipStream_t, const float*, float*, float*, int, long, int, int, float);
int a2s_vqt_decimate_impl(hipStream_t, const float*, long, const float*, int, float*, long, int);

int a2s_bn_bwd_stats_impl(hipStream_t, const float*, const float*, const float*, const float*, const float*, const float*, const uint8_t*, float,
                          float*, float*, long, int, int);
int a2s_bn_bwd_apply_impl(hipStream_t, const float*, const float*, const float*, const float*, const float*, const float*, const uint8_t*, float,
                          const float*, const float*, double, float*, float*, float*, float*, long, int, int);
int a2s_bn_bwd_sums_from_partial_impl(hipStream_t, const float*, int, int, float*);
int a2s_bn_bwd_c12_from_sums_impl(hipStream_t, const float*, const float*, double, float*, float*, float*, int);

int a2s_linear_dgrad_bnstats_impl(hipStream_t st, int M, int N, int K, const float* A, long lda, const float* Wt, long sBk, long sBn, float* C, long ldc,
                                  const float* ep_y, const float* mean, const float* invstd, const float* scale, const float* shift, int period,
                                  float* partial, const float* a_absmax, const float* b_absmax, float* ws, size_t ws_bytes, float* c_absmax_out);
size_t a2s_linear_dgrad_ws_bytes_impl(int N, int K);
int a2s_linear_fwd_impl(hipStream_t st, int M, int N, int K, const float* A, long lda, const float* W, float* C, long ldc, const float* a_scale,
                        const float* a_shift, int period, const float* a_absmax, const float* w_absmax, float* ws, size_t ws_bytes);
bool a2s_linear_fwd_ok(int M, int N, int K, long lda, long ldc, int period, const void* A, const void* W, const void* C);
int a2s_linear_wgrad_impl(hipStream_t st, int M, int N, int K, const float* dz, long ldz, const float* A, long lda, float* G, long ldg, const float* a_scale,
                          const float* a_shift, int period, const float* dz_absmax, const float* a_absmax, float* ws, size_t ws_bytes);
size_t a2s_linear_wgrad_ws_bytes_impl(int M, int K);
bool a2s_linear_wgrad_ok(int M, int N, int K, long ldz, long lda, long ldg, int period, const void* dz, const void* A, const void* G);
int a2s_linear_dgrad_blocks_impl(int M);
bool a2s_linear_dgrad_ok(int M, int N, int K, long lda, long sBk, long sBn, long ldc, int period, const void* A, const void* B, const void* C, const void* y);

bool a2s_wgrad_rows_eligible(int F, int Cin, int Cout);
int a2s_conv3x3_wgrad_rows_bn_impl(hipStream_t st, const float* g, const float* y, const float* mean, const float* invstd, const float* scale,
                                   const float* shift, const float* c12, const float* g_absmax, int g_absmax_n, const float* y_absmax, float* dz_out,
                                   float* dz_absmax_out, const float* x, const float* in_scale, const float* in_shift, float* dW, float* ws, size_t ws_bytes,
                                   int B, int T, int F, int Cin, int Cout, const float* act_absmax);

void a2s_attn_bulk_cap_set(int on);
int a2s_attn_bulk_cap_enabled(void);
#define ST ((hipStream_t)stream)

extern "C" {

const char* a2s_last_error(void) { return a2s_err_msg; }
int a2s_version(void) { return 1; }
long long a2s_launch_count(void) { return __atomic_load_n(&a2s_launch_counter, __ATOMIC_RELAXED); }

int a2s_gemm_f32(void* stream, int M, int N, int K, float alpha, const float* A, long sAm, long sAk, const float* B, long sBk,
                 long sBn, float beta, float* C, long ldc, const float* bias, int act, int batch, long bsA, long bsB, long bsC,
                 int splitk, float* workspace, size_t workspace_bytes) {
    return a2s_gemm_impl(ST, M, N, K, alpha, A, sAm, sAk, B, sBk, sBn, beta, C, ldc, bias, act, batch, bsA, bsB, bsC, splitk,
                         workspace, workspace_bytes);
}
int a2s_gemm_f32_affine(void* stream, int M, int N, int K, float alpha, const float* A, long sAm, long sAk, const float* B, long sBk,
                        long sBn, float beta, float* C, long ldc, const float* bias, int act, int batch, long bsA, long bsB, long bsC,
                        int splitk, float* workspace, size_t workspace_bytes, const float* a_scale, const float* a_shift, int a_period,
                        const float* b_scale, const float* b_shift, int b_period) {
    return a2s_gemm_affine_impl(ST, M, N, K, alpha, A, sAm, sAk, B, sBk, sBn, beta, C, ldc, bias, act, batch, bsA, bsB, bsC, splitk,
                                workspace, workspace_bytes, a_scale, a_shift, a_period, b_scale, b_shift, b_period,
                                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr);
}
int a2s_gemm_f32_affine_scaled(void* stream, int M, int N, int K, float alpha, const float* A, long sAm, long sAk, const float* B, long sBk,
                               long sBn, float beta, float* C, long ldc, const float* bias, int act, int batch, long bsA, long bsB, long bsC,
                               int splitk, float* workspace, size_t workspace_bytes, const float* a_scale, const float* a_shift, int a_period,
                               const float* b_scale, const float* b_shift, int b_period, const float* a_absmax, const float* b_absmax) {
    return a2s_gemm_affine_impl(ST, M, N, K, alpha, A, sAm, sAk, B, sBk, sBn, beta, C, ldc, bias, act, batch, bsA, bsB, bsC, splitk,
                                workspace, workspace_bytes, a_scale, a_shift, a_period, b_scale, b_shift, b_period,
                                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, a_absmax, b_absmax);
}
int a2s_gemm_f32_bnstats_scaled(void* stream, int M, int N, int K, const float* A, long sAm, long sAk, const float* B, long sBk, long sBn, float* C, long ldc,
                                const float* y, const float* mean, const float* invstd, const float* scale, const float* shift, int period, float* partial,
                                const float* a_absmax, const float* b_absmax) {
    return a2s_gemm_affine_impl(ST, M, N, K, 1.f, A, sAm, sAk, B, sBk, sBn, 0.f, C, ldc, nullptr, 0, 1, 0, 0, 0, 1, nullptr, 0,
                                nullptr, nullptr, 0, nullptr, nullptr, 0, y, mean, invstd, scale, shift, partial, period, 1, a_absmax, b_absmax);
}
int a2s_absmax(void* stream, const float* x, long n, float* out) { return a2s_absmax_impl(ST, x, n, out); }
int a2s_linear_dgrad_bnstats(void* stream, int M, int N, int K, const float* dz, long lda, const float* Wt, float* da, long ldc, const float* y,
                             const float* mean, const float* invstd, const float* scale, const float* shift, int period, float* partial,
                             const float* dz_absmax, const float* w_absmax, float* workspace, size_t workspace_bytes, float* da_absmax_out) {
    return a2s_linear_dgrad_bnstats_impl(ST, M, N, K, dz, lda, Wt, 1, K, da, ldc, y, mean, invstd, scale, shift, period, partial, dz_absmax, w_absmax,
                                         workspace, workspace_bytes, da_absmax_out);
}
size_t a2s_linear_dgrad_ws_bytes(int N, int K) { return a2s_linear_dgrad_ws_bytes_impl(N, K); }
int a2s_linear_fwd(void* stream, int M, int N, int K, const float* y, long lda, const float* W, float* z, long ldc, const float* scale, const float* shift,
                   int period, const float* y_absmax, const float* w_absmax, float* workspace, size_t workspace_bytes) {
    return a2s_linear_fwd_impl(ST, M, N, K, y, lda, W, z, ldc, scale, shift, period, y_absmax, w_absmax, workspace, workspace_bytes);
}
int a2s_linear_wgrad(void* stream, int M, int N, int K, const float* dz, long ldz, const float* y, long lda, float* G, long ldg, const float* scale,
                     const float* shift, int period, const float* dz_absmax, const float* y_absmax, float* workspace, size_t workspace_bytes) {
    return a2s_linear_wgrad_impl(ST, M, N, K, dz, ldz, y, lda, G, ldg, scale, shift, period, dz_absmax, y_absmax, workspace, workspace_bytes);
}
size_t a2s_linear_wgrad_ws_bytes(int M, int K) { return a2s_linear_wgrad_ws_bytes_impl(M, K); }
int a2s_linear_wgrad_eligible(int M, int N, int K, int period) { return a2s_linear_wgrad_ok(M, N, K, 256, 4, 4, period, nullptr, nullptr, nullptr) ? 1 : 0; }
int a2s_linear_fwd_eligible(int M, int N, int K, int period) { return a2s_linear_fwd_ok(M, N, K, 4, 4, period, nullptr, nullptr, nullptr) ? 1 : 0; }
int a2s_linear_dgrad_blocks(int M) { return a2s_linear_dgrad_blocks_impl(M); }
int a2s_linear_dgrad_eligible(int M, int N, int K, int period) {
    return a2s_linear_dgrad_ok(M, N, K, 4, 1, K, 4, period, nullptr, nullptr, nullptr, nullptr) ? 1 : 0;
}
int a2s_gemm_f32_bnstats(void* stream, int M, int N, int K, const float* A, long sAm, long sAk, const float* B, long sBk, long sBn, float* C, long ldc,
                         const float* y, const float* mean, const float* invstd, const float* scale, const float* shift, int period, float* partial) {
    return a2s_gemm_affine_impl(ST, M, N, K, 1.f, A, sAm, sAk, B, sBk, sBn, 0.f, C, ldc, nullptr, 0, 1, 0, 0, 0, 1, nullptr, 0,
                                nullptr, nullptr, 0, nullptr, nullptr, 0, y, mean, invstd, scale, shift, partial, period, 0, nullptr, nullptr);
}
int a2s_gemm_bnstats_blocks(int M, int period) { return a2s_cdiv(M, 128) * a2s_gemm_bnstats_slots(period); }
size_t a2s_gemm_workspace_bytes(int M, int N, int batch, int splitk) { return a2s_gemm_workspace_bytes_impl(M, N, batch, splitk); }
int a2s_gemm_pick_splitk(int M, int N, int K, int batch) { return a2s_gemm_pick_splitk_impl(M, N, K, batch); }
void a2s_gemm_debug_tile(int cfg) { a2s_gemm_debug_tile_impl(cfg); }
size_t a2s_note_step_workspace_floats(int H, int E) { return a2s_note_step_workspace_floats_impl(H, E); }
int a2s_debug_set(const char* key, int value) {
    if (!key) return A2S_ERR_ARG;
    if (!strcmp(key, "dec_fused")) { a2s_dec_fused_set(value); return A2S_OK; }
    if (!strcmp(key, "attn_fused_combine")) { a2s_attn_fused_combine_set(value); return A2S_OK; }
    if (!strcmp(key, "attn_nt")) { a2s_attn_nt_set(value); return A2S_OK; }
    if (!strcmp(key, "attn_bulk_cap")) { a2s_attn_bulk_cap_set(value); return A2S_OK; }
    if (!strcmp(key, "attn_deep")) { a2s_attn_deep_set(value); return A2S_OK; }
    if (!strcmp(key, "attn_defer_combine")) { a2s_attn_defer_combine_set(value); return A2S_OK; }
    if (!strcmp(key, "dec_mid")) { a2s_dec_mid_set(value); return A2S_OK; }
    if (!strcmp(key, "attn_pair")) { a2s_attn_pair_set(value); return A2S_OK; }
    if (!strcmp(key, "attn_pair_fused_rows")) { a2s_attn_pair_fused_rows_set(value); return A2S_OK; }
    if (!strcmp(key, "dec_fused_max_rows")) { a2s_dec_fused_max_rows_set(value); return A2S_OK; }
    if (!strcmp(key, "gru_fused")) { a2s_gru_step_fused_set(value); return A2S_OK; }
    if (!strcmp(key, "gru_persist")) { a2s_gru_persist_set(value); return A2S_OK; }
    if (!strcmp(key, "gru_persist_alone")) { a2s_gru_persist_alone_set(value); return A2S_OK; }
    if (!strcmp(key, "dec_persist")) { a2s_dec_persist_set(value); return A2S_OK; }
    if (!strcmp(key, "persist_force_agent")) { a2s_persist_dbg_set(PERSIST_DBG_FORCE_AGENT, value); return A2S_OK; }
    if (!strcmp(key, "persist_inject_abort")) { a2s_persist_dbg_set(PERSIST_DBG_INJECT_ABORT, value); return A2S_OK; }
    if (!strcmp(key, "gemm_tile")) { a2s_gemm_debug_tile_impl(value); return A2S_OK; }
    if (!strcmp(key, "conv_bf16x3")) { a2s_conv_bf16x3_set(value); return A2S_OK; }
    if (!strcmp(key, "conv_rows")) { a2s_conv_rows_set(value); return A2S_OK; }
    if (!strcmp(key, "conv_c1_fast")) { a2s_conv_c1_fast_set(value); return A2S_OK; }
    if (!strcmp(key, "wgrad_rows")) { a2s_wgrad_rows_set(value); return A2S_OK; }
    if (!strcmp(key, "staff_emb_fast")) { a2s_staff_emb_fast_set(value); return A2S_OK; }
    if (!strcmp(key, "conv_f16x2")) { a2s_conv_f16x2_set(value); return A2S_OK; }
    if (!strcmp(key, "wgrad_f16x2")) { a2s_wgrad_f16x2_set(value); return A2S_OK; }
    if (!strcmp(key, "gemm_bf16x3")) { a2s_gemm_split_set(value); return A2S_OK; }
    if (!strcmp(key, "gemm_f16x2")) { a2s_gemm_f16x2_set(value); return A2S_OK; }
    if (!strcmp(key, "wgrad_bf16x3")) { a2s_wgrad_split_set(value); return A2S_OK; }
    snprintf(a2s_err_msg, sizeof(a2s_err_msg), "a2s_debug_set: unknown key %s", key);
    return A2S_ERR_ARG;
}

int a2s_persist_abort_latch(void* device_word) { a2s_persist_latch_set(device_word); return A2S_OK; }

int a2s_debug_get(const char* key) {
    if (key && !strcmp(key, "attn_bulk_cap")) return a2s_attn_bulk_cap_enabled();
    if (key && !strcmp(key, "attn_deep")) return a2s_attn_deep_max_clips();
    if (key && !strcmp(key, "attn_defer_combine")) return a2s_attn_defer_combine_enabled();
    if (key && !strcmp(key, "dec_mid")) return a2s_dec_mid_enabled();
    if (key && !strcmp(key, "attn_pair")) return a2s_attn_pair_enabled();
    if (key && !strcmp(key, "attn_pair_fused_rows")) return a2s_attn_pair_fused_rows();
    if (key && !strcmp(key, "attn_pair_launches")) return (int)a2s_attn_pair_launches();
    if (key && !strcmp(key, "attn_pair_bwd_launches")) return (int)a2s_attn_pair_bwd_launches();
    if (key && !strcmp(key, "dec_mid_launches")) return a2s_dec_mid_launches();
    if (key && !strcmp(key, "conv_bf16x3")) return a2s_conv_bf16x3_enabled();
    if (key && !strcmp(key, "conv_rows")) return a2s_conv_rows_enabled();
    if (key && !strcmp(key, "conv_c1_fast")) return a2s_conv_c1_fast_enabled();
    if (key && !strcmp(key, "wgrad_rows")) return a2s_wgrad_rows_enabled();
    if (key && !strcmp(key, "staff_emb_fast")) return a2s_staff_emb_fast_enabled();
    if (key && !strcmp(key, "conv_f16x2")) return a2s_conv_f16x2_enabled();
    if (key && !strcmp(key, "wgrad_f16x2")) return a2s_wgrad_f16x2_enabled();
    if (key && !strcmp(key, "gemm_bf16x3")) return a2s_gemm_split_enabled();
    if (key && !strcmp(key, "gemm_f16x2")) return a2s_gemm_f16x2_enabled();
    if (key && !strcmp(key, "wgrad_bf16x3")) return a2s_wgrad_split_enabled();
    if (key && !strcmp(key, "gru_fused")) return a2s_gru_step_fused_enabled();
    if (key && !strcmp(key, "gru_persist")) return a2s_gru_persist_enabled();
    if (key && !strcmp(key, "gru_persist_alone")) return a2s_gru_persist_alone();
    if (key && !strcmp(key, "dec_persist")) return a2s_dec_persist_enabled();
    if (key && !strcmp(key, "dec_persist_launches")) return a2s_dec_persist_launches();
    if (key && !strcmp(key, "persist_force_agent")) return a2s_persist_dbg_get(PERSIST_DBG_FORCE_AGENT);
    if (key && !strcmp(key, "persist_inject_abort")) return a2s_persist_dbg_get(PERSIST_DBG_INJECT_ABORT);
    if (key && !strcmp(key, "device_cus")) return a2s_device_geometry().cus;
    if (key && !strcmp(key, "device_xccs")) return a2s_device_geometry().xccs;
    if (key && !strcmp(key, "dec_fused")) return a2s_dec_fused_enabled();
    if (key && !strcmp(key, "attn_fused_combine")) return a2s_attn_fused_combine_enabled();
    if (key && !strcmp(key, "attn_nt")) return a2s_attn_nt_enabled();
    if (key && !strcmp(key, "dec_fused_max_rows")) return a2s_dec_fused_max_rows();
    return -1;
}

int a2s_conv3x3(void* stream, const float* x, const float* w, float* y, const float* in_scale, const float* in_shift,
                float* stat_partial, int B, int T, int F, int Cin, int Cout, int flip, float* workspace) {
    return a2s_conv3x3_impl(ST, x, w, y, in_scale, in_shift, stat_partial, B, T, F, Cin, Cout, flip, workspace, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr);
}
int a2s_conv3x3_ranged(void* stream, const float* x, const float* w, float* y, const float* in_scale, const float* in_shift, const float* in_absmax,
                       float* stat_partial, float* out_absmax, int B, int T, int F, int Cin, int Cout, float* workspace) {
    return a2s_conv3x3_impl(ST, x, w, y, in_scale, in_shift, stat_partial, B, T, F, Cin, Cout, 0, workspace, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            in_absmax, out_absmax);
}
int a2s_conv3x3_dgrad_bnstats(void* stream, const float* dy, const float* w, float* g, const float* yl, const float* yl_mean, const float* yl_invstd,
                              const float* yl_scale, const float* yl_shift, float* stat_partial, int B, int T, int F, int Cin, int Cout,
                              float* workspace) {
    if (!yl) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "conv3x3_dgrad_bnstats: yl is required"); return A2S_ERR_ARG; }
    return a2s_conv3x3_impl(ST, dy, w, g, nullptr, nullptr, stat_partial, B, T, F, Cin, Cout, 1, workspace, yl, yl_mean, yl_invstd, yl_scale, yl_shift, nullptr, nullptr, nullptr);
}
int a2s_conv3x3_dgrad_bnstats_scaled(void* stream, const float* dy, const float* w, float* g, const float* yl, const float* yl_mean, const float* yl_invstd,
                                     const float* yl_scale, const float* yl_shift, float* stat_partial, int B, int T, int F, int Cin, int Cout,
                                     float* workspace, const float* dy_absmax) {
    if (!yl) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "conv3x3_dgrad_bnstats_scaled: yl is required"); return A2S_ERR_ARG; }
    return a2s_conv3x3_impl(ST, dy, w, g, nullptr, nullptr, stat_partial, B, T, F, Cin, Cout, 1, workspace, yl, yl_mean, yl_invstd, yl_scale, yl_shift, dy_absmax, nullptr, nullptr);
}
int a2s_conv3x3_dgrad_bnstats_ranged(void* stream, const float* dy, const float* w, float* g, const float* yl, const float* yl_mean, const float* yl_invstd,
                                     const float* yl_scale, const float* yl_shift, float* stat_partial, int B, int T, int F, int Cin, int Cout,
                                     float* workspace, const float* dy_absmax, float* g_absmax_out) {
    if (!yl) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "conv3x3_dgrad_bnstats_ranged: yl is required"); return A2S_ERR_ARG; }
    return a2s_conv3x3_impl(ST, dy, w, g, nullptr, nullptr, stat_partial, B, T, F, Cin, Cout, 1, workspace, yl, yl_mean, yl_invstd, yl_scale, yl_shift, dy_absmax, nullptr, g_absmax_out);
}
int a2s_bn_bwd_from_partial(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale, const float* shift,
                            float* dgamma, float* dbeta, float* dx, const float* partial, int nblocks, float* c12, long rows, int C, int F) {
    return a2s_bn_bwd_from_partial_impl(ST, g, x, mean, invstd, scale, shift, dgamma, dbeta, dx, partial, nblocks, c12, rows, C, F, nullptr);
}
int a2s_bn_bwd_from_partial_amax(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale, const float* shift,
                                 float* dgamma, float* dbeta, float* dx, const float* partial, int nblocks, float* c12, long rows, int C, int F,
                                 float* dx_absmax) {
    return a2s_bn_bwd_from_partial_impl(ST, g, x, mean, invstd, scale, shift, dgamma, dbeta, dx, partial, nblocks, c12, rows, C, F, dx_absmax);
}
size_t a2s_conv3x3_workspace_floats(int Cin) { return a2s_conv3x3_workspace_floats_impl(Cin); }
int a2s_conv3x3_stat_blocks(int B, int T, int F, int Cin) { return a2s_conv3x3_stat_blocks_impl(B, T, F, Cin); }
int a2s_bn_finalize(void* stream, const float* partial, int nblocks, int C, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, long long* nbt, float* mean, float* invstd, float* scale,
                    float* shift, float eps, float momentum, int training) {
    return a2s_bn_finalize_impl(ST, partial, nblocks, C, count, gamma, beta, running_mean, running_var, nbt, mean, invstd, scale,
                                shift, eps, momentum, training);
}
int a2s_bn_relu_apply(void* stream, const float* x, float* y, const float* scale, const float* shift, long n, int C, int F) {
    return a2s_bn_relu_apply_impl(ST, x, y, scale, shift, n, C, F);
}
int a2s_col_stats(void* stream, const float* x, float* partial, long rows, int C, int rows_per_block) {
    return a2s_col_stats_impl(ST, x, partial, rows, C, rows_per_block);
}
int a2s_bn1d_relu_dropout(void* stream, const float* x, float* y, const float* scale, const float* shift, const uint8_t* keep_mask,
                          float inv_keep, long n, int C) {
    return a2s_bn1d_relu_dropout_impl(ST, x, y, scale, shift, keep_mask, inv_keep, n, C);
}
int a2s_gru_gates_fwd(void* stream, const float* gi, long ldgi, const float* gh, long ldgh, const float* hprev, long ldhp,
                      float* hout, long ldho, float* hout2, long ldho2, float* save, int R, int H) {
    return a2s_gru_gates_fwd_impl(ST, gi, ldgi, gh, ldgh, hprev, ldhp, hout, ldho, hout2, ldho2, save, R, H);
}
int a2s_gru_seq_fwd(void* stream, const float* gi_all, long gi_bstride, long gi_tstride, const float* w_hh, const float* b_hh,
                    float* out, long out_bstride, long out_tstride, float* hbuf, float* gh, float* save, float* hn, int B, int T,
                    int H, int reverse, float* workspace, size_t workspace_bytes) {
    return a2s_gru_seq_fwd_impl(ST, gi_all, gi_bstride, gi_tstride, w_hh, b_hh, out, out_bstride, out_tstride, hbuf, gh, save, hn,
                                B, T, H, reverse, workspace, workspace_bytes);
}
int a2s_attn_step_fwd(void* stream, const float* keys, const float* enc, const float* q, long ldq, const float* v, float* ctx,
                      long ldctx, float* ctx2, long ldctx2, float* attw, int B, int T, int H, const int* n_done, int n_rows_total, float* workspace) {
    return a2s_attn_step_fwd_impl(ST, keys, enc, q, ldq, v, ctx, ldctx, ctx2, ldctx2, attw, B, T, H, n_done, n_rows_total, workspace, nullptr);
}
int a2s_attn_step_fwd_rows(void* stream, const float* keys, const float* enc, const float* q, long ldq, const float* v, float* ctx,
                           long ldctx, float* ctx2, long ldctx2, float* attw, int R, int T, int H, float* workspace, int n_clips,
                           const int* clip_order, const int* clip_rank, const int* row_until, int n_active, int step) {
    a2s_attn_rows rows = {clip_order, clip_rank, row_until, n_clips, n_active, step};
    return a2s_attn_step_fwd_impl(ST, keys, enc, q, ldq, v, ctx, ldctx, ctx2, ldctx2, attw, R, T, H, nullptr, 0, workspace, &rows);
}
size_t a2s_attn_workspace_floats(int B, int T, int H) { return a2s_attn_workspace_floats_impl(B, T, H, 1); }
size_t a2s_attn_workspace_floats_fused(int n_clips, int T, int H, int groups) { return a2s_attn_workspace_floats_impl(n_clips, T, H, groups); }
int a2s_log_softmax_rows(void* stream, const float* x, long ldx, float* y, long ldy, int* argmax_out, int R, int V) {
    return a2s_log_softmax_rows_impl(ST, x, ldx, y, ldy, argmax_out, R, V);
}
int a2s_embed_rows(void* stream, const float* table, const long long* ids64, const int* ids32, long id_stride, int const_id,
                   float* out, long ldo, int col0, int R, int E, const uint8_t* keep_mask, float inv_keep) {
    return a2s_embed_rows_impl(ST, table, ids64, ids32, id_stride, const_id, out, ldo, col0, R, E, keep_mask, inv_keep);
}
int a2s_note_decoder_fwd(void* stream, const a2s_note_dec_args* args, int* steps_done) {
    if (!args) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "note_decoder_fwd: null args"); return A2S_ERR_ARG; }
    return a2s_note_decoder_fwd_impl(ST, *args, steps_done);
}
int a2s_note_decoder_fwd_pair(void* stream_upper, void* stream_lower, const a2s_note_dec_args* upper, const a2s_note_dec_args* lower,
                              const int* pair_order, const int* pair_rank, const int* pair_n_active, int* steps_done_upper, int* steps_done_lower) {
    if (!upper || !lower) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "note_decoder_fwd_pair: null args"); return A2S_ERR_ARG; }
    return a2s_note_decoder_fwd_pair_impl((hipStream_t)stream_upper, (hipStream_t)stream_lower, *upper, *lower, pair_order, pair_rank, pair_n_active,
                                          steps_done_upper, steps_done_lower);
}
int a2s_staff_emb_fwd(void* stream, const float* note_emb, const float* const* gru_w, const long long* ids64, const int* ids32,
                      long id_bstride, const long long* lengths, long len_stride, float* out, long ldo, int col0, float* hsave,
                      int R, int maxlen, int E, int S) {
    if (!gru_w) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "staff_emb_fwd: null weight table"); return A2S_ERR_ARG; }
    return a2s_staff_emb_fwd_impl(ST, note_emb, gru_w, ids64, ids32, id_bstride, lengths, len_stride, out, ldo, col0, hsave, R,
                                  maxlen, E, S);
}

int a2s_log_softmax_bwd_rows(void* stream, const float* g, const float* y, long outer_stride, int inner, float* dx, int R, int V,
                             int n_outer, int time_major) {
    return a2s_log_softmax_bwd_rows_impl(ST, g, y, outer_stride, inner, dx, R, V, n_outer, time_major);
}
int a2s_gru_gates_bwd(void* stream, const float* dh_a, long lda, const float* dh_b, long ldb, const float* save, const float* hprev,
                      long ldhp, float* dgi, long ldgi, float* dgh, long ldgh, float* dgh2, long ldgh2, float* dhprev, long lddp, int R, int H) {
    return a2s_gru_gates_bwd_impl(ST, dh_a, lda, dh_b, ldb, save, hprev, ldhp, dgi, ldgi, dgh, ldgh, dgh2, ldgh2, dhprev, lddp, R, H);
}
int a2s_attn_step_bwd(void* stream, const float* keys, const float* enc, const float* q, long ldq, const float* v, const float* attw,
                      const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b, long lddb, float* dctx_out,
                      long lddo, float* dq, long lddq, float* ds_out, int B, int T, int H, float* workspace) {
    return a2s_attn_step_bwd_impl(ST, keys, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb, dctx_out, lddo, dq, lddq, ds_out, B, T, H, workspace, nullptr);
}
int a2s_attn_step_bwd_rows(void* stream, const float* keys, const float* enc, const float* q, long ldq, const float* v, const float* attw,
                           const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b, long lddb, float* dctx_out,
                           long lddo, float* dq, long lddq, float* ds_out, int R, int T, int H, float* workspace, int n_clips,
                           const int* clip_order, const int* clip_rank, const int* row_until, int n_active, int step) {
    a2s_attn_rows rows = {clip_order, clip_rank, row_until, n_clips, n_active, step};
    return a2s_attn_step_bwd_impl(ST, keys, enc, q, ldq, v, attw, ctx, ldctx, dctx_a, ldda, dctx_b, lddb, dctx_out, lddo, dq, lddq, ds_out, R, T, H, workspace, &rows);
}
int a2s_attn_dk_accum(void* stream, const float* keys, const float* q_all, const float* ds_all, const float* v, float* dK,
                      float* dv_partial, int B, int T, int S, int H, const int* row_until, int groups) {
    return a2s_attn_dk_accum_impl(ST, keys, q_all, ds_all, v, dK, dv_partial, B, T, S, H, row_until, groups);
}
int a2s_attn_dk_blocks(int B, int T) { return B * ((T + 15) / 16); }
int a2s_col_sum(void* stream, const float* x, long ld, float* out, long rows, int C, float alpha, float beta, float* workspace, size_t workspace_floats) {
    return a2s_col_sum_impl(ST, x, ld, out, rows, C, alpha, beta, workspace, workspace_floats);
}
int a2s_embed_scatter_add(void* stream, float* table_grad, const long long* ids64, const int* ids32, long id_stride, int const_id,
                          const float* g, long ldg, int col0, int R, int E, const uint8_t* keep_mask, float inv_keep) {
    return a2s_embed_scatter_add_impl(ST, table_grad, ids64, ids32, id_stride, const_id, g, ldg, col0, R, E, keep_mask, inv_keep);
}
int a2s_ew_act_bwd(void* stream, const float* g, const float* y, float* dx, long n, int act) { return a2s_ew_act_bwd_impl(ST, g, y, dx, n, act); }
int a2s_note_decoder_bwd(void* stream, const a2s_note_dec_bwd_args* args) {
    if (!args) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "note_decoder_bwd: null args"); return A2S_ERR_ARG; }
    return a2s_note_decoder_bwd_impl(ST, *args);
}
int a2s_note_decoder_bwd_pair(void* stream_upper, void* stream_lower, const a2s_note_dec_bwd_args* upper, const a2s_note_dec_bwd_args* lower,
                              const int* pair_order, const int* pair_rank, const int* pair_n_active) {
    if (!upper || !lower) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "note_decoder_bwd_pair: null args"); return A2S_ERR_ARG; }
    return a2s_note_decoder_bwd_pair_impl((hipStream_t)stream_upper, (hipStream_t)stream_lower, *upper, *lower, pair_order, pair_rank, pair_n_active);
}
int a2s_gru_seq_bwd(void* stream, const float* dout, long do_bstride, long do_tstride, const float* out, long out_bstride, long out_tstride,
                    const float* gates, const float* w_hh, const float* dhn, float* dgi_all, float* dgh_shift, float* dgh_first,
                    float* dhbuf, float* dgh_tmp, int B, int T, int H, int reverse, float* workspace, size_t workspace_bytes) {
    return a2s_gru_seq_bwd_impl(ST, dout, do_bstride, do_tstride, out, out_bstride, out_tstride, gates, w_hh, dhn, dgi_all, dgh_shift,
                                dgh_first, dhbuf, dgh_tmp, B, T, H, reverse, workspace, workspace_bytes);
}
int a2s_staff_emb_bwd(void* stream, const float* note_emb, const float* const* gru_w, float* const* grads, float* note_emb_grad,
                      const long long* ids64, const int* ids32, long id_bstride, const long long* lengths, long len_stride,
                      const float* dout, long lddo, int col0, const float* hsave, int R, int maxlen, int E, int S) {
    if (!gru_w) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "staff_emb_bwd: null weight table"); return A2S_ERR_ARG; }
    return a2s_staff_emb_bwd_impl(ST, note_emb, gru_w, grads, note_emb_grad, ids64, ids32, id_bstride, lengths, len_stride, dout, lddo,
                                  col0, hsave, R, maxlen, E, S);
}

int a2s_bn_bwd(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale, const float* shift,
               const uint8_t* keep_mask, float inv_keep, float* dgamma, float* dbeta, float* dx, float* partial, float* c12, long rows, int C, int F) {
    return a2s_bn_bwd_impl(ST, g, x, mean, invstd, scale, shift, keep_mask, inv_keep, dgamma, dbeta, dx, partial, c12, rows, C, F, nullptr);
}
int a2s_bn_bwd_amax(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale, const float* shift,
                    const uint8_t* keep_mask, float inv_keep, float* dgamma, float* dbeta, float* dx, float* partial, float* c12, long rows, int C, int F,
                    float* dx_absmax) {
    return a2s_bn_bwd_impl(ST, g, x, mean, invstd, scale, shift, keep_mask, inv_keep, dgamma, dbeta, dx, partial, c12, rows, C, F, dx_absmax);
}
size_t a2s_bn_bwd_partial_floats(long rows, int C, int F) { return a2s_bn_bwd_partial_floats_impl(rows, C, F); }
int a2s_conv3x3_wgrad(void* stream, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW, float* workspace,
                      size_t workspace_bytes, int B, int T, int F, int Cin, int Cout) {
    return a2s_conv3x3_wgrad_impl(ST, dy, x, in_scale, in_shift, dW, workspace, workspace_bytes, B, T, F, Cin, Cout, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, nullptr, nullptr);
}
int a2s_conv3x3_wgrad_ranged(void* stream, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW, float* workspace,
                             size_t workspace_bytes, int B, int T, int F, int Cin, int Cout, const float* dy_absmax, const float* act_absmax) {
    return a2s_conv3x3_wgrad_impl(ST, dy, x, in_scale, in_shift, dW, workspace, workspace_bytes, B, T, F, Cin, Cout, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, dy_absmax, act_absmax);
}
int a2s_act_bound(void* stream, const float* scale, const float* shift, const float* absmax, int C, float* out) {
    return a2s_act_bound_impl(ST, scale, shift, absmax, C, out);
}
int a2s_conv3x3_wgrad_scaled(void* stream, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW, float* workspace,
                             size_t workspace_bytes, int B, int T, int F, int Cin, int Cout, const float* dy_absmax) {
    return a2s_conv3x3_wgrad_impl(ST, dy, x, in_scale, in_shift, dW, workspace, workspace_bytes, B, T, F, Cin, Cout, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, dy_absmax, nullptr);
}
int a2s_conv3x3_wgrad_bn(void* stream, const float* g, const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                         const float* c12, float* dy_out, const float* x, const float* in_scale, const float* in_shift, float* dW, float* workspace,
                         size_t workspace_bytes, int B, int T, int F, int Cin, int Cout) {
    if (!y) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "conv3x3_wgrad_bn: y is required"); return A2S_ERR_ARG; }
    return a2s_conv3x3_wgrad_impl(ST, g, x, in_scale, in_shift, dW, workspace, workspace_bytes, B, T, F, Cin, Cout, y, mean, invstd, scale, shift, c12, dy_out, nullptr, nullptr);
}
size_t a2s_conv3x3_wgrad_workspace_bytes(int Cin, int Cout) { return a2s_conv3x3_wgrad_workspace_bytes_impl(Cin, Cout); }
int a2s_conv3x3_wgrad_bn_ranged_eligible(int F, int Cin, int Cout) { return (a2s_wgrad_rows_eligible(F, Cin, Cout) && !(Cin == 40 && Cout == 20)) ? 1 : 0; }
int a2s_conv3x3_wgrad_bn_ranged(void* stream, const float* g, const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                                const float* c12, const float* g_absmax, int g_absmax_n, const float* y_absmax, float* dy_out, float* dy_absmax_out, const float* x,
                                const float* in_scale, const float* in_shift, float* dW, float* workspace, size_t workspace_bytes, int B, int T, int F,
                                int Cin, int Cout, const float* act_absmax) {
    if (!a2s_conv3x3_wgrad_bn_ranged_eligible(F, Cin, Cout)) { snprintf(a2s_err_msg, sizeof(a2s_err_msg), "conv3x3_wgrad_bn_ranged: shape not eligible"); return A2S_ERR_ARG; }
    return a2s_conv3x3_wgrad_rows_bn_impl(ST, g, y, mean, invstd, scale, shift, c12, g_absmax, g_absmax_n, y_absmax, dy_out, dy_absmax_out, x, in_scale, in_shift, dW,
                                          workspace, workspace_bytes, B, T, F, Cin, Cout, act_absmax);
}

int a2s_nll_grad(void* stream, float* dlogp, const long long* target, const float* loss_out, float gscale, long rows, int V, long long ignore_index) {
    return a2s_nll_grad_impl(ST, dlogp, target, loss_out, gscale, rows, V, ignore_index);
}
int a2s_nll_loss(void* stream, const float* logp, const long long* target, long rows, int V, long long ignore_index, float* loss_out,
                 float* dlogp, float gscale, double* partial, int nblocks) {
    return a2s_nll_loss_impl(ST, logp, target, rows, V, ignore_index, loss_out, dlogp, gscale, partial, nblocks);
}
int a2s_clip_adadelta(void* stream, float* params, float* grads, float* square_avg, float* acc_delta, long n, const float* loss, float max_norm,
                      float lr, float rho, float eps, float* ctl, double* partial, int nblocks, int zero_grad) {
    return a2s_clip_adadelta_impl(ST, params, grads, square_avg, acc_delta, n, loss, max_norm, lr, rho, eps, ctl, partial, nblocks, zero_grad);
}

int a2s_vqt_logmag(void* stream, const float* C, float* out, float* partial, int B, long rows, int bins, float top_db) {
    return a2s_vqt_logmag_impl(ST, C, out, partial, B, rows, bins, top_db);
}

int a2s_bn_bwd_stats(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale, const float* shift,
                     const uint8_t* keep_mask, float inv_keep, float* partial, float* sums, long rows, int C, int F) {
    return a2s_bn_bwd_stats_impl(ST, g, x, mean, invstd, scale, shift, keep_mask, inv_keep, partial, sums, rows, C, F);
}
int a2s_bn_bwd_apply(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale, const float* shift,
                     const uint8_t* keep_mask, float inv_keep, const float* sums_local, const float* sums_global, double count_global,
                     float* dgamma, float* dbeta, float* dx, float* c12, long rows, int C, int F) {
    return a2s_bn_bwd_apply_impl(ST, g, x, mean, invstd, scale, shift, keep_mask, inv_keep, sums_local, sums_global, count_global, dgamma, dbeta,
                                 dx, c12, rows, C, F);
}
int a2s_vqt_logmag_octaves(void* stream, const float* C, float* out, float* partial, int B, long rows, int bins, int bins_per_octave, float top_db) {
    return a2s_vqt_logmag_octaves_impl(ST, C, out, partial, B, rows, bins, bins_per_octave, top_db);
}
int a2s_vqt_decimate(void* stream, const float* ypad, long padded_len, const float* taps, int ntaps, float* out, long n_out, int B) {
    return a2s_vqt_decimate_impl(ST, ypad, padded_len, taps, ntaps, out, n_out, B);
}
int a2s_bn_bwd_sums_from_partial(void* stream, const float* partial, int nblocks, int C, float* sums) {
    return a2s_bn_bwd_sums_from_partial_impl(ST, partial, nblocks, C, sums);
}
int a2s_bn_bwd_c12_from_sums(void* stream, const float* sums_local, const float* sums_global, double count_global, float* dgamma, float* dbeta,
                             float* c12, int C) {
    return a2s_bn_bwd_c12_from_sums_impl(ST, sums_local, sums_global, count_global, dgamma, dbeta, c12, C);
}

}  // extern "C"
