// Loss and optimizer kernels of the training step.
//   * NLL (mean over non-ignored targets) forward + gradient wrt the log-probabilities -- reference
//     ASR.compute_objectives, pretrain.py:72-88, torch.nn.NLLLoss(ignore_index=147) per hparams/pretrain.yaml:49-54
//   * gradient-norm clipping (SpeechBrain check_gradients -> clip_grad_norm_(5.0)) fused with
//     Adadelta(lr, rho, eps) -- reference ASR.fit_batch pretrain.py:125-128, hparams/pretrain.yaml:44-47 --
//     over ONE flat parameter / gradient / state buffer, fully on the device (no host sync, skip on non-finite).
#include "a2s_common.h"

// per-block partial: sum of -logp[target] and count of valid targets for rows [r0, r1)
__global__ __launch_bounds__(256) void nll_partial(const float* __restrict__ logp, const long long* __restrict__ target, long rows, int V,
                                                   long long ignore_index, double* __restrict__ partial) {
    __shared__ double rs[256], rc[256];
    double s = 0.0, c = 0.0;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const long long t = target[r];
        if (t != ignore_index) { s -= (double)logp[r * V + t]; c += 1.0; }
    }
    rs[threadIdx.x] = s; rc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rc[threadIdx.x] += rc[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = rs[0]; partial[2 * blockIdx.x + 1] = rc[0]; }
}

// loss_out[0] = sum / count ; loss_out[1] = 1 / count   (fixed-order reduction of the block partials)
__global__ void nll_finalize(const double* __restrict__ partial, int nblocks, float* __restrict__ loss_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0, c = 0.0;
    for (int i = 0; i < nblocks; ++i) { s += partial[2 * i]; c += partial[2 * i + 1]; }
    loss_out[0] = (float)(s / c);
    loss_out[1] = (float)(1.0 / c);
}

// dlogp[r, target[r]] = -gscale * inv_count for valid rows (dlogp must be zero-filled by the caller)
__global__ void nll_grad(float* __restrict__ dlogp, const long long* __restrict__ target, const float* __restrict__ loss_out, float gscale,
                         long rows, int V, long long ignore_index) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const long long t = target[r];
    if (t != ignore_index) dlogp[r * V + t] = -gscale * loss_out[1];
}

int a2s_nll_loss_impl(hipStream_t st, const float* logp, const long long* target, long rows, int V, long long ignore_index,
                      float* loss_out /* 2 floats */, float* dlogp /* zero-filled or null */, float gscale, double* partial, int nblocks) {
    A2S_REQUIRE(logp && target && loss_out && partial && nblocks > 0, "nll_loss: null tensor");
    hipLaunchKernelGGL(nll_partial, dim3(nblocks), dim3(256), 0, st, logp, target, rows, V, ignore_index, partial);
    A2S_CHECK_LAUNCH("nll_partial");
    hipLaunchKernelGGL(nll_finalize, dim3(1), dim3(64), 0, st, partial, nblocks, loss_out);
    A2S_CHECK_LAUNCH("nll_finalize");
    if (dlogp) {
        hipLaunchKernelGGL(nll_grad, dim3(a2s_cdiv(rows, 256)), dim3(256), 0, st, dlogp, target, loss_out, gscale, rows, V, ignore_index);
        A2S_CHECK_LAUNCH("nll_grad");
    }
    return A2S_OK;
}

// gradient only, 1/count supplied by the caller (loss_out[1]): lets a part of the rows back-propagate before the rest of the minibatch
// has even been decoded -- the denominator of the mean is a function of the targets alone
int a2s_nll_grad_impl(hipStream_t st, float* dlogp, const long long* target, const float* loss_out, float gscale, long rows, int V, long long ignore_index) {
    A2S_REQUIRE(dlogp && target && loss_out, "nll_grad: null tensor");
    if (rows <= 0) return A2S_OK;
    hipLaunchKernelGGL(nll_grad, dim3(a2s_cdiv(rows, 256)), dim3(256), 0, st, dlogp, target, loss_out, gscale, rows, V, ignore_index);
    A2S_CHECK_LAUNCH("nll_grad");
    return A2S_OK;
}

// ------------------------------------------------------------------------------------------- clip + Adadelta
__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ g, long n, double* __restrict__ partial) {
    __shared__ double rs[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const double v = g[i]; s += v * v; }
    rs[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) rs[threadIdx.x] += rs[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = rs[0];
}

// ctl[0] = total grad norm, ctl[1] = clip coefficient (<= 1), ctl[2] = 1 if the step is applied else 0
__global__ void clip_finalize(const double* __restrict__ partial, int nblocks, const float* __restrict__ loss, float max_norm,
                              float* __restrict__ ctl) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += partial[i];
    const float total = (float)sqrt(s);
    float coef = max_norm / (total + 1e-6f);          // torch.nn.utils.clip_grad_norm_
    if (coef > 1.f) coef = 1.f;
    // check_gradients: non-finite loss -> skip the step.  A non-finite gradient norm skips it too (the reference would apply
    // coef = NaN and destroy the parameters; under data parallelism one rank's NaN gradients reach every rank through the all-reduce)
    const bool finite_loss = loss ? isfinite(*loss) : true;
    ctl[0] = total; ctl[1] = coef; ctl[2] = (finite_loss && isfinite(total)) ? 1.f : 0.f;
}

// torch.optim.Adadelta (weight_decay = 0) on the clipped gradient; also clears the gradient (zero_grad)
__global__ void adadelta_step(float* __restrict__ p, float* __restrict__ g, float* __restrict__ sq, float* __restrict__ acc,
                              const float* __restrict__ ctl, float lr, float rho, float eps, long n, int zero_grad) {
    const long stride = (long)gridDim.x * blockDim.x;
    const bool apply = ctl[2] != 0.f;
    const float coef = ctl[1];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (apply) {
            const float gi = g[i] * coef;
            const float s = rho * sq[i] + (1.f - rho) * gi * gi;
            const float d = sqrtf(acc[i] + eps) / sqrtf(s + eps) * gi;
            sq[i] = s;
            acc[i] = rho * acc[i] + (1.f - rho) * d * d;
            p[i] -= lr * d;
        }
        if (zero_grad) g[i] = 0.f;
    }
}

int a2s_clip_adadelta_impl(hipStream_t st, float* params, float* grads, float* square_avg, float* acc_delta, long n, const float* loss,
                           float max_norm, float lr, float rho, float eps, float* ctl /* 3 floats */, double* partial, int nblocks, int zero_grad) {
    A2S_REQUIRE(params && grads && square_avg && acc_delta && ctl && partial && nblocks > 0, "clip_adadelta: null tensor");
    hipLaunchKernelGGL(sumsq_partial, dim3(nblocks), dim3(256), 0, st, grads, n, partial);
    A2S_CHECK_LAUNCH("sumsq_partial");
    hipLaunchKernelGGL(clip_finalize, dim3(1), dim3(64), 0, st, partial, nblocks, loss, max_norm, ctl);
    A2S_CHECK_LAUNCH("clip_finalize");
    hipLaunchKernelGGL(adadelta_step, dim3(min((long)2048, (n + 255) / 256)), dim3(256), 0, st, params, grads, square_avg, acc_delta, ctl, lr, rho, eps, n, zero_grad);
    A2S_CHECK_LAUNCH("adadelta_step");
    return A2S_OK;
}
