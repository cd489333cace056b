// Optimiser step of the reference's training loop (training/conformer_pipeline.py:424-429 AdamW(lr 5e-4, betas (0.9, 0.98),
// weight_decay 0.01), :509 NaN/Inf step-skip, :514 clip_grad_norm_(5.0)) on ONE flat fp32 parameter / gradient buffer.
// Three launches per step and no host synchronisation:
//   sumsq        : ||g||^2 in fp64 (wave shuffles + one atomic per workgroup)
//   adamw_prepare: 1 thread: unscale, global-norm clip coefficient, skip decision, bias corrections, step counter
//   adamw_apply  : streaming update of p, m, v  (HBM-bound: reads g, p, m, v, writes p, m, v = 28 B/param)
#include "sfm_common.h"

// ctl layout (doubles): [0] step count  [1] sum of squares (input)  [2] flag (>0 -> skip; input)
//                       [3] gscale (output)  [4] skip (output)  [5] bc1  [6] bc2  [7] grad norm after unscale (output)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ out,
                                                    double* __restrict__ ws) {
  __shared__ double red[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double v = g[i];
    s += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ws) ws[blockIdx.x] = red[0] + red[1] + red[2] + red[3];      // folded in workgroup order: the same norm on every run
    else atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
  }
}

__global__ void adamw_prepare_kernel(double* __restrict__ ctl, double inv_scale, double max_norm, double beta1, double beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double norm = sqrt(ctl[1]) * inv_scale;
  const bool bad = !(norm == norm) || norm > 1.7e308 || ctl[2] > 0.0;
  double coef = 1.0;
  if (max_norm > 0.0 && norm > max_norm) coef = max_norm / (norm + 1e-6);     // torch.nn.utils.clip_grad_norm_
  ctl[7] = norm;
  ctl[4] = bad ? 1.0 : 0.0;
  ctl[3] = inv_scale * coef;
  if (!bad) {
    const double step = ctl[0] + 1.0;
    ctl[0] = step;
    ctl[5] = 1.0 - pow(beta1, step);
    ctl[6] = 1.0 - pow(beta2, step);
  }
  ctl[1] = 0.0;                                                                // ready for the next step's sumsq
  ctl[2] = 0.0;
}

// The reference trains under fp16 autocast with torch.amp.GradScaler (training/conformer_pipeline.py:442, 504, 512-517): the loss is
// multiplied by a scale S before backward, unscale_ divides the gradients by S and looks for Inf / NaN, step is skipped when one is
// found, update() halves S after such a step and doubles it after `growth_interval` clean ones.  Here all of that is this one
// thread: ls[0] = S (fp32, the tensor the host multiplies the loss with - read by that multiply BEFORE this kernel runs, in stream
// order), ls[1] = clean steps since the last change, ls[2] = steps skipped for Inf / NaN gradients (total), ls[3] = steps skipped
// for a non-finite loss (total).  A non-finite LOSS skips the whole iteration in the reference (:509 `continue`): no update of S.
__global__ void adamw_prepare_scaled_kernel(double* __restrict__ ctl, float* __restrict__ ls, double inv_world, double max_norm,
                                            double beta1, double beta2, float growth, float backoff, float interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double S = (double)ls[0];
  const double inv_scale = inv_world / S;
  const double norm = sqrt(ctl[1]) * inv_scale;
  const bool bad_grad = !(norm == norm) || norm > 1.7e308;
  const bool bad_loss = ctl[2] > 0.0;
  const bool bad = bad_grad || bad_loss;
  double coef = 1.0;
  if (max_norm > 0.0 && norm > max_norm) coef = max_norm / (norm + 1e-6);
  ctl[7] = norm;
  ctl[4] = bad ? 1.0 : 0.0;
  ctl[3] = inv_scale * coef;
  if (!bad) {
    const double step = ctl[0] + 1.0;
    ctl[0] = step;
    ctl[5] = 1.0 - pow(beta1, step);
    ctl[6] = 1.0 - pow(beta2, step);
  }
  if (bad_loss) {
    ls[3] += 1.f;
  } else if (bad_grad) {                                                       // GradScaler.update(): found_inf
    ls[0] = fmaxf(ls[0] * backoff, 1.17549435e-38f);
    ls[1] = 0.f;
    ls[2] += 1.f;
  } else {
    const float t = ls[1] + 1.f;
    if (t >= interval) {
      const float grown = ls[0] * growth;
      ls[0] = (grown == grown && grown < 3.0e38f) ? grown : ls[0];
      ls[1] = 0.f;
    } else {
      ls[1] = t;
    }
  }
  ctl[1] = 0.0;
  ctl[2] = 0.0;
}

// spans / touched (optional): spans[k] = first element of parameter k in the flat buffers (n_params + 1 entries, ascending),
// touched[k] > 0 when some rank's backward pass produced a gradient for parameter k in this step.  Elements of parameters
// with touched == 0 are left alone - p, m AND v - as torch.optim.AdamW leaves a parameter whose grad is None; the decision is
// made on the device from a mask that was summed over the ranks with the gradients, so every replica takes the same one.
__global__ __launch_bounds__(256) void adamw_apply_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, long long n, const double* __restrict__ ctl,
                                                          float lr, float beta1, float beta2, float eps, float wd,
                                                          int write_back_grad, const long long* __restrict__ spans,
                                                          const float* __restrict__ touched, int n_params) {
  if (ctl[4] > 0.0) return;                                                    // skipped step: nothing changes
  const float gs = (float)ctl[3];
  const float step_size = lr / (float)ctl[5];
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(ctl[6]));
  const float decay = 1.0f - lr * wd;
  for (long long base = (long long)blockIdx.x * 256; base < n; base += (long long)gridDim.x * 256) {
    const long long i = base + threadIdx.x;
    if (spans) {
      // parameter of the chunk's first element (the same binary search in every lane: broadcast loads), then at most a few
      // steps forward for the lanes behind a parameter boundary inside the chunk
      int lo = 0, hi = n_params - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (spans[mid] <= base) lo = mid; else hi = mid - 1;
      }
      int k = lo;
      while (k + 1 < n_params && spans[k + 1] <= i) ++k;
      if (i < n && touched[k] <= 0.f) continue;
    }
    if (i >= n) continue;
    const float gi = g[i] * gs;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = p[i] * decay - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
    if (write_back_grad) g[i] = gi;                                            // leave the unscaled, clipped gradient in place
  }
}

// ws (optional, >= 2048 doubles): one partial per workgroup folded in workgroup order (reduce.hip) instead of an f64 atomic each
extern "C" int sfm_sumsq(const float* g, long long n, double* out, double* ws, void* stream) {
  if (!g || !out) return SFM_ERR_ARG;
  if (n <= 0) return SFM_ERR_SHAPE;
  long long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  SFM_LAUNCH(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g, n, out, ws);
  return ws ? sfm_fold_partials_f64(ws, out, 1, 1, 1, (int)nb, 1, stream) : SFM_OK;
}

static int adamw_step_impl(float* p, float* g, float* m, float* v, long long n, double* ctl, float lr, float beta1,
                           float beta2, float eps, float wd, float inv_scale, float max_norm, int write_back_grad,
                           const long long* spans, const float* touched, int n_params, void* stream, float* loss_scale = nullptr,
                           float growth = 2.f, float backoff = 0.5f, int interval = 2000) {
  if (!p || !g || !m || !v || !ctl) return SFM_ERR_ARG;
  if (n <= 0) return SFM_ERR_SHAPE;
  if ((spans != nullptr) != (touched != nullptr) || (spans && n_params <= 0)) return SFM_ERR_ARG;
  if (loss_scale) {
    if (!(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f) || interval < 1) return SFM_ERR_ARG;
    SFM_LAUNCH(adamw_prepare_scaled_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, loss_scale, (double)inv_scale,
               (double)max_norm, (double)beta1, (double)beta2, growth, backoff, (float)interval);
  } else {
    SFM_LAUNCH(adamw_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, (double)inv_scale, (double)max_norm,
               (double)beta1, (double)beta2);
  }
  long long nb = (n + 255) / 256;
  if (nb > 8192) nb = 8192;
  SFM_LAUNCH(adamw_apply_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, ctl, lr, beta1, beta2,
             eps, wd, write_back_grad, spans, touched, n_params);
  return SFM_OK;
}

extern "C" int sfm_adamw_step(float* p, float* g, float* m, float* v, long long n, double* ctl, float lr, float beta1,
                              float beta2, float eps, float wd, float inv_scale, float max_norm, int write_back_grad,
                              void* stream) {
  return adamw_step_impl(p, g, m, v, n, ctl, lr, beta1, beta2, eps, wd, inv_scale, max_norm, write_back_grad, nullptr, nullptr, 0,
                         stream);
}

extern "C" int sfm_adamw_step_masked(float* p, float* g, float* m, float* v, long long n, double* ctl, float lr, float beta1,
                                     float beta2, float eps, float wd, float inv_scale, float max_norm, int write_back_grad,
                                     const long long* spans, const float* touched, int n_params, void* stream) {
  if (!spans || !touched) return SFM_ERR_ARG;
  return adamw_step_impl(p, g, m, v, n, ctl, lr, beta1, beta2, eps, wd, inv_scale, max_norm, write_back_grad, spans, touched,
                         n_params, stream);
}

// AdamW step under the reference's AMP recipe (training/conformer_pipeline.py:504, 512-517: scaler.scale(loss).backward();
// scaler.unscale_; clip_grad_norm_; scaler.step; scaler.update) with the scale kept ON THE DEVICE: loss_scale = 4 floats
// {S, clean steps since the last change of S, steps skipped for Inf / NaN gradients, steps skipped for a non-finite loss}.
// The gradients in g are S times (and, data parallel, world times: inv_world = 1 / world) the true ones.
extern "C" int sfm_adamw_step_scaled(float* p, float* g, float* m, float* v, long long n, double* ctl, float lr, float beta1,
                                     float beta2, float eps, float wd, float inv_world, float max_norm, int write_back_grad,
                                     const long long* spans, const float* touched, int n_params, float* loss_scale,
                                     float growth_factor, float backoff_factor, int growth_interval, void* stream) {
  if (!loss_scale) return SFM_ERR_ARG;
  return adamw_step_impl(p, g, m, v, n, ctl, lr, beta1, beta2, eps, wd, inv_world, max_norm, write_back_grad, spans, touched,
                         n_params, stream, loss_scale, growth_factor, backoff_factor, growth_interval);
}
