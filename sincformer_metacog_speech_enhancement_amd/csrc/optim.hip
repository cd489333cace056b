// Optimiser step of the reference's training loop (training/conformer_pipeline.py:424-429 AdamW(lr 5e-4, betas (0.9, 0.98),
// weight_decay 0.01), :509 NaN/Inf step-skip, :514 clip_grad_norm_(5.0)) on ONE flat fp32 parameter / gradient buffer.
// Three launches per step and no host synchronisation:
//   sumsq        : ||g||^2 in fp64 (wave shuffles + one atomic per workgroup)
//   adamw_prepare: 1 thread: unscale, global-norm clip coefficient, skip decision, bias corrections, step counter
//   adamw_apply  : streaming update of p, m, v  (HBM-bound: reads g, p, m, v, writes p, m, v = 28 B/param)
#include "sfm_common.h"

// ctl layout (doubles): [0] step count  [1] sum of squares (input)  [2] flag (>0 -> skip; input)
//                       [3] gscale (output)  [4] skip (output)  [5] bc1  [6] bc2  [7] grad norm after unscale (output)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ out) {
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
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
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

extern "C" int sfm_sumsq(const float* g, long long n, double* out, void* stream) {
  if (!g || !out) return SFM_ERR_ARG;
  if (n <= 0) return SFM_ERR_SHAPE;
  long long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  SFM_LAUNCH(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g, n, out);
  return SFM_OK;
}

static int adamw_step_impl(float* p, float* g, float* m, float* v, long long n, double* ctl, float lr, float beta1,
                           float beta2, float eps, float wd, float inv_scale, float max_norm, int write_back_grad,
                           const long long* spans, const float* touched, int n_params, void* stream) {
  if (!p || !g || !m || !v || !ctl) return SFM_ERR_ARG;
  if (n <= 0) return SFM_ERR_SHAPE;
  if ((spans != nullptr) != (touched != nullptr) || (spans && n_params <= 0)) return SFM_ERR_ARG;
  SFM_LAUNCH(adamw_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctl, (double)inv_scale, (double)max_norm,
             (double)beta1, (double)beta2);
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
