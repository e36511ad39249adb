/* a2s.h -- C ABI of liba2s_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the data-parallel training
 * hot path of piano-a2s (models.ScoreTranscription forward/backward, loss, clip + Adadelta).
 *
 * The reference has no native layer: every operation below replaces a torch.nn call made by
 * /root/reference/models.py (cited per function) that would otherwise run through ATen -> cuDNN/cuBLAS.
 * The boundary a maintainer binds is this file; INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every tensor is a plain device pointer to fp32 unless stated (ids: int64 `long long` or int32);
 *     the library BORROWS pointers for the duration of the call and owns no memory;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); calls only
 *     enqueue work -- they never allocate and never synchronise, with ONE documented exception:
 *     a2s_note_decoder_fwd in greedy mode polls a device counter every `poll` steps;
 *   - return value 0 = ok, negative = error (A2S_ERR_*), message via a2s_last_error(); never throws;
 *   - scratch memory is caller-allocated; *_workspace_bytes() say how much;
 *   - one host thread per process / GPU (torchrun model); ordering is by stream.
 */
#ifndef A2S_H
#define A2S_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define A2S_OK 0
#define A2S_ERR_ARG (-1)
#define A2S_ERR_HIP (-2)
#define A2S_ERR_WORKSPACE (-3)

const char* a2s_last_error(void);
int a2s_version(void);
/* kernel launches issued by this process through the library so far (diagnostic: launches per optimizer / decode step in bench.py;
 * the reference has no native layer, nothing replaced) */
long long a2s_launch_count(void);

/* ---- dense contraction: every nn.Linear / GRU projection of models.py (:68,:123-132,:359,:444-445,:504) and
 * their backward forms.  C[m,n] = act(alpha * sum_k A(m,k) B(k,n) + beta*C + bias[n]);
 * A(m,k)=A[m*sAm+k*sAk], B(k,n)=B[k*sBk+n*sBn]; act 0 none / 1 relu / 2 tanh / 3 exp(2x) (the attention key image below); split-K is deterministic
 * (splitk 0 = pick automatically for skinny problems when a workspace is supplied). */
int a2s_gemm_f32(void* stream, int M, int N, int K, float alpha, const float* A, long sAm, long sAk,
                 const float* B, long sBk, long sBn, float beta, float* C, long ldc, const float* bias, int act,
                 int batch, long bsA, long bsB, long bsC, int splitk, float* workspace, size_t workspace_bytes);
/* The same with BatchNorm + ReLU of an operand applied while it is staged: element -> max(0, e*scale[c] + shift[c]), c = (index along
 * the operand's unit-stride dimension) / period; NULL scale = operand taken as is.  The 19200->256 Linear (models.py:537-539) reads the
 * last convolution's pre-BN output this way (forward: A; weight gradient: B), so the activated tensor is never materialised. */
int a2s_gemm_f32_affine(void* stream, int M, int N, int K, float alpha, const float* A, long sAm, long sAk, const float* B, long sBk,
                        long sBn, float beta, float* C, long ldc, const float* bias, int act, int batch, long bsA, long bsB, long bsC,
                        int splitk, float* workspace, size_t workspace_bytes, const float* a_scale, const float* a_shift, int a_period,
                        const float* b_scale, const float* b_shift, int b_period);
/* C = A B whose output (M rows x N = channels * period columns, same layout as y) is the gradient wrt relu(bn(y)) of a ConvStack layer (the
 * data gradient of the 19200->256 Linear): its BatchNorm-backward statistics (sum g', sum g' xhat per channel) are accumulated in the
 * epilogue into partial[a2s_gemm_bnstats_blocks(M, period)][N / period][2] -- the layout a2s_bn_bwd_from_partial reads. */
int a2s_gemm_f32_bnstats(void* stream, int M, int N, int K, const float* A, long sAm, long sAk, const float* B, long sBk, long sBn, float* C, long ldc,
                         const float* y, const float* mean, const float* invstd, const float* scale, const float* shift, int period, float* partial);
/* The two entries above on the two-term fp16 split (three matrix-core products instead of the six of the three-term bf16 split, same
 * fp32-level accuracy; 128x128 tiles with k- or row-contiguous operands, otherwise the call runs as the unscaled entry): a_absmax /
 * b_absmax are device scalars holding max |A| / max |B| (a2s_absmax, or the BatchNorm backward's a2s_bn_bwd_amax), from which the kernel
 * derives exact power-of-two operand scales; NULL = that operand is O(1) (post-BatchNorm activations) and is used as is.  Switch:
 * a2s_debug_set("gemm_f16x2", 0/1) (default 1; environment: A2S_ARITH). */
int a2s_gemm_f32_affine_scaled(void* stream, int M, int N, int K, float alpha, const float* A, long sAm, long sAk, const float* B, long sBk,
                               long sBn, float beta, float* C, long ldc, const float* bias, int act, int batch, long bsA, long bsB, long bsC,
                               int splitk, float* workspace, size_t workspace_bytes, const float* a_scale, const float* a_shift, int a_period,
                               const float* b_scale, const float* b_shift, int b_period, const float* a_absmax, const float* b_absmax);
int a2s_gemm_f32_bnstats_scaled(void* stream, int M, int N, int K, const float* A, long sAm, long sAk, const float* B, long sBk, long sBn, float* C, long ldc,
                                const float* y, const float* mean, const float* invstd, const float* scale, const float* shift, int period, float* partial,
                                const float* a_absmax, const float* b_absmax);
/* Round 4: the data gradient of the ConvStack's 19200 -> 256 Linear (reference models.py:68; backward of y = relu(bn4(y4)) W^T) as a kernel of
 * its own (csrc/a2s_linear.hip): da (M x N) = dz (M x 256, leading dimension lda) * Wt^T with Wt (N x 256) the k-contiguous copy of the weight,
 * plus the BatchNorm-backward statistics of a2s_gemm_f32_bnstats_scaled into partial[a2s_linear_dgrad_blocks(M)][N / period][2] (same
 * consumer).  K must be 256, N % 32 == 0, period % 32 == 0; workspace: a2s_linear_dgrad_ws_bytes(N, 256) bytes (the weight as fp16 term
 * planes), 16-byte aligned.  da_absmax_out (device scalar, may be NULL): max |da| is folded into it by atomic max (zero it before the call).
 * a2s_linear_dgrad_eligible says whether a shape qualifies (otherwise call a2s_gemm_f32_bnstats_scaled). */
int a2s_linear_dgrad_bnstats(void* stream, int M, int N, int K, const float* dz, long lda, const float* Wt, float* da, long ldc, const float* y,
                             const float* mean, const float* invstd, const float* scale, const float* shift, int period, float* partial,
                             const float* dz_absmax, const float* w_absmax, float* workspace, size_t workspace_bytes, float* da_absmax_out);
/* ... and its forward (reference models.py:68,537-539): z (M x 256, leading dimension ldc) = relu(y * scale[k / period] + shift[k / period]) W^T
 * with y (M x K, leading dimension lda) and W (256 x K) row-major; scale / shift NULL: y as it is.  N must be 256, K % 64 == 0, period % 32 == 0;
 * y_absmax: device scalar bounding the ACTIVATED operand (a2s_act_bound; NULL: O(1)); workspace: a2s_linear_dgrad_ws_bytes(256, K) bytes.
 * a2s_linear_fwd_eligible says whether a shape qualifies (otherwise a2s_gemm_f32_affine_scaled). */
int a2s_linear_fwd(void* stream, int M, int N, int K, const float* y, long lda, const float* W, float* z, long ldc, const float* scale, const float* shift,
                   int period, const float* y_absmax, const float* w_absmax, float* workspace, size_t workspace_bytes);
int a2s_linear_fwd_eligible(int M, int N, int K, int period);
/* ... and its weight gradient: G (256 x K, leading dimension ldg) += dz^T relu(y * scale[k / period] + shift[k / period]) with dz (M x 256, leading
 * dimension ldz); N must be 256, K % 128 == 0, period % 4 == 0, M >= 64; dz_absmax: device scalar max |dz|; y_absmax as in a2s_linear_fwd;
 * workspace: a2s_linear_wgrad_ws_bytes(M, K) bytes (dz as fp16 term planes + the split-K slabs).  Deterministic (fixed-order reduction). */
int a2s_linear_wgrad(void* stream, int M, int N, int K, const float* dz, long ldz, const float* y, long lda, float* G, long ldg, const float* scale,
                     const float* shift, int period, const float* dz_absmax, const float* y_absmax, float* workspace, size_t workspace_bytes);
size_t a2s_linear_wgrad_ws_bytes(int M, int K);
int a2s_linear_wgrad_eligible(int M, int N, int K, int period);
size_t a2s_linear_dgrad_ws_bytes(int N, int K);
int a2s_linear_dgrad_blocks(int M);
int a2s_linear_dgrad_eligible(int M, int N, int K, int period);
/* out[0] = max |x[i]| over n floats (device scalar) */
int a2s_absmax(void* stream, const float* x, long n, float* out);
int a2s_gemm_bnstats_blocks(int M, int period);
size_t a2s_gemm_workspace_bytes(int M, int N, int batch, int splitk);
int a2s_gemm_pick_splitk(int M, int N, int K, int batch);
/* tuning aid (tools/gemm_sweep.py): force the tile configuration for M > 64 (1: 32x64, 2: 64x32, 3: 64x64, 4: 128x128; 0: heuristic) */
void a2s_gemm_debug_tile(int cfg);
/* measurement switches for A/B runs (tools/ab_step.py): "gru_fused" 0/1 (one-launch recurrent step), "gemm_tile" as above,
 * "conv_bf16x3" bit mask -- which 3x3 convolutions run on the bf16 matrix pipes with every fp32 operand split exactly into three bf16
 * terms (six products, fp32-level accuracy; csrc/a2s_conv.hip conv3x3_bf16x3): bit 0 forward launches, bit 1 data-gradient launches
 * (default 3; 0 = the fp32-input MFMA kernel everywhere); "gemm_bf16x3" 0/1 -- the same split for 128x128 GEMM tiles whose two
 * operands are k- or row-contiguous (default 1); "wgrad_bf16x3" 0/1/2 -- the split-operand weight-gradient convolution: 1 (default) where it is
 * faster than conv3x3_wgrad (40 -> 40 channels), 2 every eligible launch;
 * "attn_defer_combine" 0/1 (default 1): inside a2s_note_decoder_fwd, the few-clip attention launches of a training call
 * leave their softmax combine to the GRU-step kernel that consumes the contexts (csrc/a2s_step.hip dec_gru_step_cmb; same bits as the combine
 * kernel); "dec_mid" 0/1 (default 1, round 6): the launch-per-step loop's GRU cell, output + next query and backward products on the mid-size
 * fused kernels (0: library-style products; read-only "dec_mid_launches" counts them);
 * "attn_deep" n (default 24): training launches over at most n clips use the one-round-trip forward sweep */
int a2s_debug_set(const char* key, int value);
int a2s_debug_get(const char* key);   /* current value of "conv_bf16x3" / "gemm_bf16x3" / "wgrad_bf16x3" / "gru_fused"; -1 for an unknown key;
                                        * also "device_cus" / "device_xccs": compute units and XCDs the runtime reports for the current device */
/* Persistent kernels (encoder recurrences, few-clip note decoder: one launch whose workgroups wait for each other, replacing the per-step
 * launches of nn.GRU / NoteDecoder.decode_notes, models.py:63-67,388-419).  Their waits are bounded; a launch that gives up poisons its outputs
 * with NaN (the loss becomes non-finite, the update is skipped) and ORs a bit into *device_word (a 4-byte device word the caller owns, zeroed by
 * the caller; NULL unregisters).  The host reads the word at a synchronisation point it has anyway and switches the persistent paths off
 * (a2s_debug_set("gru_persist" / "dec_persist", 0)) for the rest of the process.  The paths are only taken when the runtime reports a chip
 * they fit (compute units x workgroups per CU >= the launch; 8 XCDs x 32 CUs for the decoder).
 * Test hooks: a2s_debug_set("persist_force_agent", 1) = never use the one-XCD plain-store hand-off, "persist_inject_abort", 1 = every
 * persistent launch behaves as if a wait had timed out. */
int a2s_persist_abort_latch(void* device_word);

/* ---- ConvStack (models.py:475-502,:523-534).  Activations are (B, T, C, F); see csrc/a2s_conv.hip.
 * conv3x3: y = conv(relu(x*in_scale+in_shift)) (scale/shift NULL: plain x), zero padding 1, no bias;
 * stat_partial [blocks][Cout][2] receives per-block sum / sum-of-squares of y (NULL to skip);
 * flip=1 runs the data-gradient form with w read as w'[ci][co][2-dt][2-df]. */
int a2s_conv3x3(void* stream, const float* x, const float* w, float* y, const float* in_scale, const float* in_shift,
                float* stat_partial, int B, int T, int F, int Cin, int Cout, int flip, float* workspace);
/* The same forward convolution with operand RANGES (round 3; replaces nothing new in the reference: models.py:525-534 as a2s_conv3x3):
 * in_absmax[Cin] = max |x| per input channel (as written into `out_absmax` by the launch that produced x; NULL: measured here by an
 * extra pass), out_absmax[Cout] = max |y| per output channel, written by this launch (NULL: not wanted).  The row-streaming kernel
 * (csrc/a2s_conv_rows.hip) derives exact power-of-two operand scales from them, so that its fp16 operand terms never overflow or lose
 * precision whatever BatchNorm's scale is. */
int a2s_conv3x3_ranged(void* stream, const float* x, const float* w, float* y, const float* in_scale, const float* in_shift, const float* in_absmax,
                       float* stat_partial, float* out_absmax, int B, int T, int F, int Cin, int Cout, float* workspace);
int a2s_conv3x3_stat_blocks(int B, int T, int F, int Cin);
/* Data gradient of a convolution (flip form: g = dy (*) w') whose output is the gradient wrt relu(bn(yl)) of the layer below, with that
 * layer's BatchNorm-backward statistics (sum g', sum g' xhat; g' = g where bn(yl) > 0) accumulated in the epilogue into stat_partial
 * [a2s_conv3x3_stat_blocks(B,T,F,Cin)][Cout][2] -- pass it to a2s_bn_bwd_from_partial instead of running the statistics pass of
 * a2s_bn_bwd.  Cin / Cout are those of THIS launch (Cin = channels of dy, Cout = channels of g and yl). */
int a2s_conv3x3_dgrad_bnstats(void* stream, const float* dy, const float* w, float* g, const float* yl, const float* yl_mean, const float* yl_invstd,
                              const float* yl_scale, const float* yl_shift, float* stat_partial, int B, int T, int F, int Cin, int Cout,
                              float* workspace);
int a2s_bn_bwd_from_partial(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale, const float* shift,
                            float* dgamma, float* dbeta, float* dx /* may be NULL */, const float* partial, int nblocks, float* c12, long rows, int C, int F);
/* The same two calls for the TWO-TERM fp16 convolution path (csrc/a2s_conv.hip conv3x3_split<.., 2>: every fp32 operand as two fp16
 * terms, three products instead of the six of the three-term bf16 split).  fp16 has no exponent range to spare for gradients, so the
 * kernel that WRITES a gradient tensor also reduces max |dx| into a device scalar (dx_absmax, 1 float) and the data-gradient convolution
 * that READS it scales its operand by the exact power of two that brings that maximum to 2^12 (dy_absmax; NULL = three-term path). */
int a2s_conv3x3_dgrad_bnstats_scaled(void* stream, const float* dy, const float* w, float* g, const float* yl, const float* yl_mean, const float* yl_invstd,
                                     const float* yl_scale, const float* yl_shift, float* stat_partial, int B, int T, int F, int Cin, int Cout,
                                     float* workspace, const float* dy_absmax);
/* Round 4: the same, also writing g_absmax_out[Cout] = max |g| per channel of the gradient it writes (the launch zeroes it): the range from which
 * a2s_conv3x3_wgrad_bn_ranged bounds the BatchNorm backward of the layer below. */
int a2s_conv3x3_dgrad_bnstats_ranged(void* stream, const float* dy, const float* w, float* g, const float* yl, const float* yl_mean, const float* yl_invstd,
                                     const float* yl_scale, const float* yl_shift, float* stat_partial, int B, int T, int F, int Cin, int Cout,
                                     float* workspace, const float* dy_absmax, float* g_absmax_out);
int a2s_bn_bwd_from_partial_amax(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale, const float* shift,
                                 float* dgamma, float* dbeta, float* dx, const float* partial, int nblocks, float* c12, long rows, int C, int F,
                                 float* dx_absmax);
size_t a2s_conv3x3_workspace_floats(int Cin);   /* scratch for the packed weight image (0 for Cin = 1) */
/* BatchNorm2d/1d statistics -> affine (models.py:499-505): reduces the partials in fixed order (double),
 * updates running stats (momentum, unbiased var) and num_batches_tracked when training, emits mean/invstd
 * (for backward) and scale/shift with y = x*scale+shift.  training=0: uses the running statistics. */
int a2s_bn_finalize(void* stream, const float* partial, int nblocks, int C, double count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* mean, float* invstd, float* scale, float* shift, float eps, float momentum, int training);
int a2s_bn_relu_apply(void* stream, const float* x, float* y, const float* scale, const float* shift, long n, int C, int F);
int a2s_col_stats(void* stream, const float* x, float* partial, long rows, int C, int rows_per_block);
int a2s_bn1d_relu_dropout(void* stream, const float* x, float* y, const float* scale, const float* shift,
                          const uint8_t* keep_mask, float inv_keep, long n, int C);

/* ---- GRU (nn.GRU, gate packing [r;z;n]; models.py:63-67,:107-111,:117-120,:353-356) */
int a2s_gru_gates_fwd(void* stream, const float* gi, long ldgi, const float* gh, long ldgh, const float* hprev, long ldhp,
                      float* hout, long ldho, float* hout2, long ldho2, float* save, int R, int H);
/* one direction of one encoder layer over T steps, h0 = 0 (Encoder.forward, models.py:77) */
int a2s_gru_seq_fwd(void* stream, const float* gi_all, long gi_bstride, long gi_tstride, const float* w_hh,
                    const float* b_hh, float* out, long out_bstride, long out_tstride, float* hbuf, float* gh,
                    float* save, float* hn, int B, int T, int H, int reverse, float* workspace, size_t workspace_bytes);
/* workspace (optional): split-K scratch for the per-step recurrent GEMM (M = B rows, few output tiles) */

/* ---- additive attention step (AttentionLayer.forward models.py:452-461 + bmm :242,:394), keys hoisted:
 * score_t = v . tanh(K[b,t,:] + q[b,:]); a = softmax_t; ctx = sum_t a_t enc[b,t,:].
 * `keys` of every attention entry point is the KEY IMAGE exp(2 K) (a2s_gemm_f32 with act 3 on K = enc W_e^T; |K| clamped to 43), not K:
 * tanh(k + q) = 1 - 2 / (1 + exp(2k) exp(2q)) then costs one transcendental per (frame, unit).  q is passed as is; the gradient
 * a2s_attn_dk_accum returns is the one with respect to K. */
int a2s_attn_step_fwd(void* stream, const float* keys, const float* enc, const float* q, long ldq, const float* v,
                      float* ctx, long ldctx, float* ctx2, long ldctx2, float* attw, int B, int T, int H,
                      const int* n_done, int n_rows_total, float* workspace);
/* the same step over the R = groups * n_clips rows of a fused-bars decoder call (row = group * n_clips + clip; see a2s_note_dec_args):
 * rows of a clip share its keys / enc; clip_order / clip_rank / row_until may be NULL (identity, never finished), n_active = clips
 * computed (prefix of clip_order), step = decode step compared with row_until.  Skipped rows get ctx = 0, attw = 0. */
int a2s_attn_step_fwd_rows(void* stream, const float* keys, const float* enc, const float* q, long ldq, const float* v, float* ctx,
                           long ldctx, float* ctx2, long ldctx2, float* attw, int R, int T, int H, float* workspace, int n_clips,
                           const int* clip_order, const int* clip_rank, const int* row_until, int n_active, int step);
/* workspace (a2s_attn_workspace_floats floats, shared by forward and backward) selects the split-T kernels: the frames of a
 * clip are spread over several workgroups and merged by a combine kernel -- or, with a2s_debug_set("attn_fused_combine", 1), by the
 * workgroup that finishes last for the clip (measured slower on MI355X, off by default); NULL (or hidden_size != 256) = one workgroup
 * per clip.  The workspace must be ZERO-INITIALISED ONCE after allocation (its head holds the workgroups' arrival counters, which
 * every launch leaves at zero) and must not be shared by launches that may run concurrently. */
size_t a2s_attn_workspace_floats(int B, int T, int H);
/* the same for a fused-bars decoder call (a2s_note_dec_args.n_clips): `groups` bars of n_clips clips in one call */
size_t a2s_attn_workspace_floats_fused(int n_clips, int T, int H, int groups);

int a2s_log_softmax_rows(void* stream, const float* x, long ldx, float* y, long ldy, int* argmax_out, int R, int V);
int a2s_embed_rows(void* stream, const float* table, const long long* ids64, const int* ids32, long id_stride,
                   int const_id, float* out, long ldo, int col0, int R, int E, const uint8_t* keep_mask, float inv_keep);

/* ---- one (bar, staff) note decode: NoteDecoder.decode_notes, models.py:366-420 */
typedef struct a2s_note_dec_args {
    const float* attn_w; const float* attn_b; const float* attn_v;
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;
    const float* out_w; const float* out_b; const float* emb;
    const float* keys; const float* enc;
    float* h; float* x; float* q; float* gates; float* attw; float* o;
    float* gh; float* gi; float* logits;
    float* probs; long probs_bstride;
    const long long* gt; long gt_bstride;
    const uint8_t* tf_flags;          /* HOST array, one entry per step: bit g = teacher-force the rows of group g (bit 0 when not fused) */
    const uint8_t* drop; float inv_keep;
    int* argmax_out; long am_bstride;
    int* eos_seen; long long* lengths; int* n_done;
    int* steps_exec;                  /* device counter: +1 per step that actually decoded (greedy early break) */
    float* attn_ws;                   /* a2s_attn_workspace_floats(R,T,H) [_fused(n_clips,T,H,R/n_clips)] floats or NULL */
    float* gemm_ws; size_t gemm_ws_bytes;   /* split-K scratch for the per-step skinny GEMMs (NULL: no split-K) */
    int* t_base;                      /* device int (graph mode): base step index of the chunk being replayed */
    /* optional, training only (the fused training step, not the drop-in module):
       (1) fused bars -- the R rows are R/n_clips bars ("groups", <= 5) of the same n_clips clips, row = group * n_clips + clip; the
           bars' decoders are independent once the bar-level recurrence is teacher-forced, and rows of one clip share its keys and
           encoder outputs, which the attention kernels then stream once for all of them;
       (2) finished rows -- from step row_until[row] on a row's remaining targets are all <pad>: nothing reaching the loss depends
           on it any more, its attention is skipped (context = 0) and its outputs are left untouched.
       Clips sorted by the step their last row finishes at, latest first, so the clips still running are a prefix: */
    const int* clip_order;            /* device, n_clips ints: clip ids in that order */
    const int* clip_rank;             /* device, n_clips ints: inverse permutation */
    const int* row_until;             /* device, R ints */
    const int* n_active;              /* HOST, `steps` ints: clips with an unfinished row at step t; NULL = none of (1)/(2) */
    int n_clips;                      /* 0 = R (one group) */
    const int* m_active;              /* HOST, `steps` ints or NULL: 1 + the largest clip index (position in the call, not in clip_order)
                                         that still has an unfinished row at step t.  When that is at most half of n_clips the per-step
                                         products run on the leading m_active[t] clips of every fused bar only (round 3: the few
                                         full-length rows of a large call no longer drag every row through ~100 further steps) */
    const int* row_list;              /* device, R ints or NULL: the rows sorted by row_until, latest first (stable) -- the rows still running
                                         at step t are its first n_rows_active[t] entries; the few-row step kernels then cover those only */
    const int* n_rows_active;         /* HOST, `steps` ints (with row_list) */
    int R, T, H, E, V, steps, poll, eos_id;
    int use_graph;                    /* greedy decode (gt NULL, nothing saved for backward): capture `poll` steps into a hipGraph and replay */
    float* step_ws; size_t step_ws_floats;   /* a2s_note_step_workspace_floats(H, E) floats or NULL: scratch of the fused few-row step kernels
                                                (csrc/a2s_step.hip: 4 launches per step instead of 9-13; used when R <= "dec_fused_max_rows", default 192; above that the launch-per-step loop with the mid-size kernels dec_gru_mid / dec_outq_mid / dec_bwd_mid) */
    /* round 4: ONE persistent launch for every step of the call when it covers at most 8 clips (csrc/a2s_dec_persist.hip: one clip per XCD,
       keys / encoder outputs resident in LDS, weights in registers).  Needs the teacher-forcing flags on the device too and a scratch area: */
    const int* tf_flags_dev;          /* device, `steps` ints: the same bits as tf_flags (NULL with tf_flags NULL) */
    float* persist_ws; size_t persist_ws_bytes;   /* a2s_note_decoder_persist_ws_bytes(n_clips, R, steps) bytes, 256-byte aligned, or NULL */
} a2s_note_dec_args;
int a2s_note_decoder_fwd(void* stream, const a2s_note_dec_args* args, int* steps_done);
/* Round 6: the two NoteDecoders of a segment (/root/reference/models.py:261-275: decode_notes of the upper and of the lower staff over the same
 * encoder_outputs) issued by ONE host loop on their two streams; while both staves run a step, the step's attention sweep is one launch that reads
 * the encoder outputs once for both (csrc/a2s_seq.hip: attn_fwd_split256_pair).  pair_order / pair_rank: device, n_clips ints -- the clips sorted by
 * the step the last row of EITHER staff finishes at (latest first) and the inverse permutation; pair_n_active: HOST, max(steps) ints.  Training calls
 * with the finished-row bookkeeping only; anything else runs the two calls one after the other (a2s_note_decoder_fwd twice). */
int a2s_note_decoder_fwd_pair(void* stream_upper, void* stream_lower, const a2s_note_dec_args* upper, const a2s_note_dec_args* lower,
                              const int* pair_order, const int* pair_rank, const int* pair_n_active, int* steps_done_upper, int* steps_done_lower);
size_t a2s_note_step_workspace_floats(int H, int E);
/* scratch of the persistent path (0: that many clips are not supported) */
size_t a2s_note_decoder_persist_ws_bytes(int n_clips, int R, int steps);
size_t a2s_note_decoder_bwd_persist_ws_bytes(int n_clips);

/* ---- packed staff-embedding bi-GRU final states (get_staff_token_from_{gt,probs}, models.py:164-189).
 * gru_w: 8 device pointers {w_ih,w_hh,b_ih,b_hh} forward then reverse. */
int a2s_staff_emb_fwd(void* stream, const float* note_emb, const float* const* gru_w, const long long* ids64,
                      const int* ids32, long id_bstride, const long long* lengths, long len_stride, float* out,
                      long ldo, int col0, float* hsave, int R, int maxlen, int E, int S);

/* ======================================================================================= backward
 * Gradients follow torch.autograd on the reference graph.  All "+=" outputs accumulate (beta = 1 semantics). */

/* dx = g - exp(y) * rowsum(g) for y = log_softmax(x).  Row r of g/y at base + (r/inner)*outer_stride + (r%inner)*V;
 * time_major=1 writes dx row (r%inner)*n_outer + r/inner (step-major, to line up with the saved step buffers). */
int a2s_log_softmax_bwd_rows(void* stream, const float* g, const float* y, long outer_stride, int inner, float* dx,
                             int R, int V, int n_outer, int time_major);
/* GRU cell backward from the saved [r|z|n|gh_n]; dh = dh_a + dh_b (dh_b may be NULL); hprev NULL = zeros. */
int a2s_gru_gates_bwd(void* stream, const float* dh_a, long lda, const float* dh_b, long ldb, const float* save,
                      const float* hprev, long ldhp, float* dgi, long ldgi, float* dgh, long ldgh, float* dgh2, long ldgh2,
                      float* dhprev, long lddp, int R, int H);
/* one attention step backward: dq, ds (T per row) and the summed dctx (see csrc/a2s_bwd.hip) */
int a2s_attn_step_bwd(void* stream, const float* keys, const float* enc, const float* q, long ldq, const float* v,
                      const float* attw, const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b,
                      long lddb, float* dctx_out, long lddo, float* dq, long lddq, float* ds_out, int B, int T, int H, float* workspace);
int a2s_attn_step_bwd_rows(void* stream, const float* keys, const float* enc, const float* q, long ldq, const float* v, const float* attw,
                           const float* ctx, long ldctx, const float* dctx_a, long ldda, const float* dctx_b, long lddb, float* dctx_out,
                           long lddo, float* dq, long lddq, float* ds_out, int R, int T, int H, float* workspace, int n_clips,
                           const int* clip_order, const int* clip_rank, const int* row_until, int n_active, int step);
/* deferred key gradient of S steps: dK += ..., dv partials [B*ceil(T/16)][H] (reduce with a2s_col_sum) */
int a2s_attn_dk_accum(void* stream, const float* keys, const float* q_all, const float* ds_all, const float* v, float* dK,
                      float* dv_partial, int B, int T, int S, int H, const int* row_until, int groups);
/* q_all / ds_all rows: (step, group, clip) with `groups` fused bars per step (1 = plain); row_until (optional, groups*B ints):
 * the ds rows of (group, clip) are zero from step row_until[group*B + clip] on and are not read */
int a2s_attn_dk_blocks(int B, int T);
/* out[c] = alpha * sum_r x[r*ld+c] + beta*out[c]; with a workspace (>= 2*C floats, ideally 1024*C) long matrices are reduced in two
 * stages over many workgroups (fixed partition: deterministic). */
int a2s_col_sum(void* stream, const float* x, long ld, float* out, long rows, int C, float alpha, float beta, float* workspace, size_t workspace_floats);
int a2s_embed_scatter_add(void* stream, float* table_grad, const long long* ids64, const int* ids32, long id_stride,
                          int const_id, const float* g, long ldg, int col0, int R, int E, const uint8_t* keep_mask, float inv_keep);
int a2s_ew_act_bwd(void* stream, const float* g, const float* y, float* dx, long n, int act);

typedef struct a2s_note_dec_bwd_args {
    const float* attn_w; const float* attn_v; const float* w_ih; const float* w_hh;
    const float* keys; const float* enc;
    const float* h; const float* x; const float* q; const float* gates; const float* attw;   /* saved by the forward */
    const float* do_all;                 /* (steps, R, 4H): dlogits_all W_out = [dh | dctx] of the output projection */
    float* dgi_all; float* dgh_all; float* dq_all; float* ds_all; float* dctx_all; float* dx;
    float* dh;                           /* (2, R, 2H) carry; dh[0] = gradient wrt the initial hidden on return */
    float* attn_ws;                      /* as in the forward call */
    float* gemm_ws; size_t gemm_ws_bytes;   /* split-K scratch for the per-step skinny GEMMs (NULL: no split-K) */
    const int* clip_order;               /* the forward's fused-bars / finished-rows description (or NULLs): see a2s_note_dec_args */
    const int* clip_rank;
    const int* row_until;
    const int* n_active;                 /* HOST */
    int n_clips;
    const int* m_active;                 /* HOST or NULL: as in the forward call (the call zero-fills dx first) */
    const int* row_list;                 /* device or NULL, and */
    const int* n_rows_active;            /* HOST: as in the forward call */
    int R, T, H, E, steps;
    float* step_ws; size_t step_ws_floats;   /* as in the forward call (holds the transposed weight copies of the fused backward step) */
    float* persist_ws; size_t persist_ws_bytes;   /* round 4: a2s_note_decoder_bwd_persist_ws_bytes(n_clips) bytes (256-byte aligned) or NULL:
                                                     at most 8 clips run the reverse loop as ONE persistent launch (csrc/a2s_dec_persist.hip) */
    const float* w_ih_full;                  /* unused (reserved) */
} a2s_note_dec_bwd_args;
int a2s_note_decoder_bwd(void* stream, const a2s_note_dec_bwd_args* args);
/* Round 6: the reverse loops of a segment's two NoteDecoders from one host loop on their two streams, the attention sweep of a step as ONE launch
 * for both staves while both step (see a2s_note_decoder_fwd_pair; autograd of /root/reference/models.py:261-275). */
int a2s_note_decoder_bwd_pair(void* stream_upper, void* stream_lower, const a2s_note_dec_bwd_args* upper, const a2s_note_dec_bwd_args* lower,
                              const int* pair_order, const int* pair_rank, const int* pair_n_active);

/* BPTT of one encoder GRU direction (reverse of a2s_gru_seq_fwd) */
int a2s_gru_seq_bwd(void* stream, const float* dout, long do_bstride, long do_tstride, const float* out, long out_bstride,
                    long out_tstride, const float* gates, const float* w_hh, const float* dhn, float* dgi_all, float* dgh_shift,
                    float* dgh_first, float* dhbuf, float* dgh_tmp, int B, int T, int H, int reverse, float* workspace, size_t workspace_bytes);
/* BPTT of the packed staff-embedding bi-GRU; grads: DEVICE array of 8 pointers (same order as gru_w) */
int a2s_staff_emb_bwd(void* stream, const float* note_emb, const float* const* gru_w, float* const* grads, float* note_emb_grad,
                      const long long* ids64, const int* ids32, long id_bstride, const long long* lengths, long len_stride,
                      const float* dout, long lddo, int col0, const float* hsave, int R, int maxlen, int E, int S);

/* BatchNorm(+ReLU, + optional dropout on the (rows,C) layout) backward, training statistics (models.py:525-541):
 * dgamma/dbeta +=, dx written (may alias g; NULL = statistics only: c12 is left for a2s_conv3x3_wgrad_bn, which forms dx itself).
 * rows x C x F elements, channel of element i = (i/F)%C.
 * partial: a2s_bn_bwd_partial_floats() floats of scratch; c12: 2*C floats of scratch. */
int a2s_bn_bwd(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale,
               const float* shift, const uint8_t* keep_mask, float inv_keep, float* dgamma, float* dbeta, float* dx,
               float* partial, float* c12, long rows, int C, int F);
int a2s_bn_bwd_amax(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale,
                    const float* shift, const uint8_t* keep_mask, float inv_keep, float* dgamma, float* dbeta, float* dx,
                    float* partial, float* c12, long rows, int C, int F, float* dx_absmax);
size_t a2s_bn_bwd_partial_floats(long rows, int C, int F);
/* the same in two halves for synchronised BatchNorm: (1) this rank's per-channel {sum g', sum g' xhat} -> sums[2C];
 * [host: all-reduce];  (2) dgamma/dbeta += LOCAL sums, dx from the GLOBAL sums / global element count. */
int a2s_bn_bwd_stats(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale,
                     const float* shift, const uint8_t* keep_mask, float inv_keep, float* partial, float* sums, long rows, int C, int F);
int a2s_bn_bwd_apply(void* stream, const float* g, const float* x, const float* mean, const float* invstd, const float* scale,
                     const float* shift, const uint8_t* keep_mask, float inv_keep, const float* sums_local, const float* sums_global,
                     double count_global, float* dgamma, float* dbeta, float* dx, float* c12, long rows, int C, int F);
/* ... and for statistics the producer of g already reduced into per-block partials (a2s_conv3x3_dgrad_bnstats*, a2s_linear_dgrad_bnstats):
 * (1) partial[nblocks][C][2] -> this rank's sums[2C]; [host: all-reduce]; (2) dgamma / dbeta += LOCAL sums, c12 = GLOBAL sums / global element
 * count -- no pass over (g, x); the input gradient is formed by the fused consumer (a2s_conv3x3_wgrad_bn_ranged).  With these the synchronised
 * BatchNorm of torch.nn.SyncBatchNorm (what SpeechBrain's DDP wrapping gives reference pretrain.py:257) keeps every fused path of the per-rank one. */
int a2s_bn_bwd_sums_from_partial(void* stream, const float* partial, int nblocks, int C, float* sums);
int a2s_bn_bwd_c12_from_sums(void* stream, const float* sums_local, const float* sums_global, double count_global, float* dgamma, float* dbeta,
                             float* c12, int C);
/* conv weight gradient dW += dy (*) relu(x*in_scale+in_shift)  (deterministic two-stage reduction) */
int a2s_conv3x3_wgrad(void* stream, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW,
                      float* workspace, size_t workspace_bytes, int B, int T, int F, int Cin, int Cout);
/* ... with max |dy| (device scalar written by a2s_bn_bwd*_amax) for the two-term fp16 path of the split-operand kernel (NULL: as above) */
int a2s_conv3x3_wgrad_scaled(void* stream, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW,
                             float* workspace, size_t workspace_bytes, int B, int T, int F, int Cin, int Cout, const float* dy_absmax);
/* ... with the RANGE of the activated operand relu(x * in_scale[c] + in_shift[c]) as well (round 3): act_absmax = device scalar from
 * a2s_act_bound (NULL: operand used unscaled, as a2s_conv3x3_wgrad_scaled).  Reference: autograd of nn.Conv2d, models.py:525-534. */
int a2s_conv3x3_wgrad_ranged(void* stream, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dW, float* workspace,
                             size_t workspace_bytes, int B, int T, int F, int Cin, int Cout, const float* dy_absmax, const float* act_absmax);
/* out[0] = max_c (|scale[c]| * absmax[c] + |shift[c]|): hard bound of relu(x * scale[c] + shift[c]) over a tensor whose producer wrote the
 * per-channel max |x| (a2s_conv3x3_ranged); the operand range handed to the kernels that apply BatchNorm + ReLU while staging
 * (a2s_conv3x3_wgrad_ranged, a2s_gemm_f32_affine_scaled).  Nothing comparable in the reference (fp32 throughout). */
int a2s_act_bound(void* stream, const float* scale, const float* shift, const float* absmax, int C, float* out);
size_t a2s_conv3x3_wgrad_workspace_bytes(int Cin, int Cout);
/* The same with the BatchNorm backward of the layer's output folded into the staging of the dy operand: g = gradient wrt
 * relu(bn(y)), y = the layer's pre-BN output, c12 from a2s_bn_bwd(..., dx = NULL); dy = scale*(g' - c1 - xhat*c2) is formed on the fly
 * and also written to dy_out (may be NULL) for the data-gradient convolution that follows -- the separate apply pass disappears. */
int a2s_conv3x3_wgrad_bn(void* stream, const float* g, const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                         const float* c12, float* dy_out, const float* x, const float* in_scale, const float* in_shift, float* dW, float* workspace,
                         size_t workspace_bytes, int B, int T, int F, int Cin, int Cout);
/* Round 4: the same fusion on the row-streaming weight-gradient kernel (csrc/a2s_conv_wrows.hip, two-term fp16 operands), for the shapes
 * a2s_conv3x3_wgrad_bn_ranged_eligible accepts (F % 4 == 0, 20 / 40 channels): its staging waves form dz from (g, y), write it to dy_out
 * (required: the data-gradient convolution reads it) and fold max |dz| into dy_absmax_out (device scalar, zeroed by the call).  The operand
 * scale of dz comes from a BOUND derived on the device from g_absmax[g_absmax_n] (max |g|, one value or one per channel, reduced by the kernel
 * that wrote g: a2s_linear_dgrad_bnstats, a2s_conv3x3_dgrad_bnstats_ranged) and y_absmax[Cout] (max |y_c|, written by the forward
 * convolution: a2s_conv3x3_ranged); act_absmax as in a2s_conv3x3_wgrad_ranged. */
int a2s_conv3x3_wgrad_bn_ranged(void* stream, const float* g, const float* y, const float* mean, const float* invstd, const float* scale, const float* shift,
                                const float* c12, const float* g_absmax, int g_absmax_n, const float* y_absmax, float* dy_out, float* dy_absmax_out, const float* x,
                                const float* in_scale, const float* in_shift, float* dW, float* workspace, size_t workspace_bytes, int B, int T, int F,
                                int Cin, int Cout, const float* act_absmax);
int a2s_conv3x3_wgrad_bn_ranged_eligible(int F, int Cin, int Cout);

/* ---- objective and optimizer of the recipe (pretrain.py:72-88,:125-128; pretrain.yaml:44-54)
 * NLL, mean over targets != ignore_index (pass -1 for "none"): loss_out[0] = loss, loss_out[1] = 1/count;
 * dlogp (zero-filled by the caller, or NULL) receives d(gscale*loss)/dlogp.  partial: 2*nblocks doubles. */
int a2s_nll_loss(void* stream, const float* logp, const long long* target, long rows, int V, long long ignore_index,
                 float* loss_out, float* dlogp, float gscale, double* partial, int nblocks);
/* the gradient alone for `rows` rows, with loss_out[1] = 1/count given by the caller (the count of targets != ignore_index over the WHOLE
 * minibatch is known from the targets): dlogp (zero-filled) receives d(gscale*loss)/dlogp of these rows. */
int a2s_nll_grad(void* stream, float* dlogp, const long long* target, const float* loss_out, float gscale, long rows, int V,
                 long long ignore_index);
/* clip_grad_norm_(max_norm) + Adadelta over one flat buffer; skipped when *loss (NULL: not looked at) or the gradient norm is not finite.
 * ctl[0..2] = {total norm, clip coefficient, applied flag}; partial: nblocks doubles; zero_grad clears the gradients. */
int a2s_clip_adadelta(void* stream, float* params, float* grads, float* square_avg, float* acc_delta, long n, const float* loss,
                      float max_norm, float lr, float rho, float eps, float* ctl, double* partial, int nblocks, int zero_grad);

/* ---- VQT front-end epilogue (utilities.get_VQT, utilities.py:240-254): C (B, rows, 2*bins) = [re | im] of the framed complex GEMM
 * (run with a2s_gemm_f32, A row stride = hop) -> out (B, rows, bins) = dB relative to the clip maximum, floor 1e-5, top_db, /80 + 1.
 * partial: B*64 floats of scratch. */
int a2s_vqt_logmag(void* stream, const float* C, float* out, float* partial, int B, long rows, int bins, float top_db);
/* the same over a response laid out octave by octave: C (B, rows, n_oct * 2 * bins_per_octave), octave o (HIGHEST first) = [re | im] of its
 * bins_per_octave bins -- what one framed GEMM per octave against the (n_fft, 2 * bins_per_octave) bank writes */
int a2s_vqt_logmag_octaves(void* stream, const float* C, float* out, float* partial, int B, long rows, int bins, int bins_per_octave, float top_db);
/* decimation by 2 between the octaves of librosa.vqt's recursion (librosa.resample(y, orig_sr=2, target_sr=1, scale=True); the filter is this
 * build's stand-in for libsoxr, see piano_a2s_amd/vqt.py): out[b][m] = sum_j ypad[b][2 m + j] taps[j], ntaps even; ypad (B, padded_len) with
 * padded_len >= 2 n_out + ntaps (reads beyond are taken as zero), out (B, n_out). */
int a2s_vqt_decimate(void* stream, const float* ypad, long padded_len, const float* taps, int ntaps, float* out, long n_out, int B);

#ifdef __cplusplus
}
#endif
#endif
