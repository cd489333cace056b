// the two small modules behind the path's outputs.
//   MetacognitiveArbitrationAgent (agents/maa.py:70-135): per time step of the PerceptionAgent's sigma,
//     norm = (sigma - running_mean) / (sqrt(running_var) + 1e-8) -> Linear(1,64) ReLU Linear(64,64) ReLU Linear(64,4)
//     -> logits, softmax, argmax, confidence = sigmoid(-norm); train(): EMA of the batch mean / unbiased variance first.
//     One thread per time step, the 4 548 weights in LDS (every lane reads the same address: broadcast), the 64 hidden
//     units of layer 1 in registers, layer 2 and 3 folded into one loop.  Backward: the same recomputation, dsigma and the
//     16-bit activations / pre-activation gradients whose products are the weight gradients (three TN GEMMs on the host side).
//   VectorQuantizer (models/vq.py:54-96): nearest of M <= 16 scalar centroids (first minimum, like torch.argmin),
//     sum of squared distances for the commitment + codebook loss; backward = straight-through + the two loss terms.
#include "sfm_common.h"

#define MAA_H 64
#define MAA_C 4
#define MAA_NPARAM (MAA_H + MAA_H + MAA_H * MAA_H + MAA_H + MAA_C * MAA_H + MAA_C)
#define MAA_W1 0
#define MAA_B1 (MAA_H)
#define MAA_W2 (2 * MAA_H)
#define MAA_B2 (2 * MAA_H + MAA_H * MAA_H)
#define MAA_W3 (3 * MAA_H + MAA_H * MAA_H)
#define MAA_B3 (3 * MAA_H + MAA_H * MAA_H + MAA_C * MAA_H)

__global__ __launch_bounds__(256) void maa_stats_kernel(const float* __restrict__ sigma, long long n, double* __restrict__ acc) {
  __shared__ double red[4][2];
  double s = 0.0, q = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double v = sigma[i];
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], red[0][0] + red[1][0] + red[2][0] + red[3][0]);
    atomicAdd(&acc[1], red[0][1] + red[1][1] + red[2][1] + red[3][1]);
  }
}

// stats = (running_mean, running_var); agents/maa.py:126-135 (torch.var: unbiased)
__global__ void maa_update_kernel(double* __restrict__ acc, long long n, float* __restrict__ stats, long long* __restrict__ num_updates,
                                  float momentum) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double mean = acc[0] / (double)n;
  const double var = n > 1 ? (acc[1] - acc[0] * acc[0] / (double)n) / (double)(n - 1) : 0.0 / 0.0;
  stats[0] = (1.0f - momentum) * stats[0] + momentum * (float)mean;
  stats[1] = (1.0f - momentum) * stats[1] + momentum * (float)var;
  num_updates[0] += 1;
  acc[0] = 0.0;
  acc[1] = 0.0;
}

__device__ __forceinline__ void maa_load_params(const float* __restrict__ params, float* P) {
  for (int i = threadIdx.x; i < MAA_NPARAM; i += 256) P[i] = params[i];
  __syncthreads();
}

__global__ __launch_bounds__(256) void maa_forward_kernel(const float* __restrict__ sigma, const float* __restrict__ stats,
                                                          const float* __restrict__ params, float* __restrict__ logits,
                                                          float* __restrict__ probs, long long* __restrict__ decisions,
                                                          float* __restrict__ confidence, long long n) {
  __shared__ __attribute__((aligned(16))) float P[MAA_NPARAM];
  maa_load_params(params, P);
  const float mean = stats[0], inv = 1.0f / (sqrtf(stats[1]) + 1e-8f);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const float x = (sigma[e] - mean) * inv;
    float h1[MAA_H];
#pragma unroll
    for (int j = 0; j < MAA_H; ++j) h1[j] = fmaxf(P[MAA_W1 + j] * x + P[MAA_B1 + j], 0.f);
    float lg[MAA_C];
#pragma unroll
    for (int c = 0; c < MAA_C; ++c) lg[c] = P[MAA_B3 + c];
    for (int i = 0; i < MAA_H; ++i) {
      float z = P[MAA_B2 + i];
      const f32x4* w = reinterpret_cast<const f32x4*>(&P[MAA_W2 + i * MAA_H]);
#pragma unroll
      for (int j4 = 0; j4 < MAA_H / 4; ++j4) {
        const f32x4 ww = w[j4];
        z += ww[0] * h1[4 * j4] + ww[1] * h1[4 * j4 + 1] + ww[2] * h1[4 * j4 + 2] + ww[3] * h1[4 * j4 + 3];
      }
      const float h2 = fmaxf(z, 0.f);
#pragma unroll
      for (int c = 0; c < MAA_C; ++c) lg[c] += P[MAA_W3 + c * MAA_H + i] * h2;
    }
    float mx = lg[0];
    int arg = 0;
#pragma unroll
    for (int c = 1; c < MAA_C; ++c)
      if (lg[c] > mx) { mx = lg[c]; arg = c; }
    float ex[MAA_C], den = 0.f;
#pragma unroll
    for (int c = 0; c < MAA_C; ++c) { ex[c] = __expf(lg[c] - mx); den += ex[c]; }
    const float r = 1.0f / den;
    *reinterpret_cast<f32x4*>(logits + e * MAA_C) = f32x4{lg[0], lg[1], lg[2], lg[3]};
    *reinterpret_cast<f32x4*>(probs + e * MAA_C) = f32x4{ex[0] * r, ex[1] * r, ex[2] * r, ex[3] * r};
    decisions[e] = arg;
    confidence[e] = 1.0f / (1.0f + __expf(x));
  }
}

template <class T>
__device__ __forceinline__ void maa_store_row(u16* __restrict__ dst, long long e, const float (&v)[MAA_H]) {
#pragma unroll
  for (int c = 0; c < MAA_H / 8; ++c) {
    u32x4 pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) pk[k] = pack2<T>(v[8 * c + 2 * k], v[8 * c + 2 * k + 1]);
    *reinterpret_cast<u32x4*>(dst + e * MAA_H + 8 * c) = pk;
  }
}

// g_logits / g_probs [N,4], g_conf [N]: incoming gradients (any may be null).  Writes dsigma [N] and the 16-bit operands of the
// weight-gradient GEMMs: H1, H2, dZ1, dZ2 [N,64], GL [N,8] (total logit gradient, cols 4..7 zero), XN [N,8] (col 0 = norm).
template <class T>
__global__ __launch_bounds__(256) void maa_backward_kernel(const float* __restrict__ sigma, const float* __restrict__ stats,
                                                           const float* __restrict__ params, const float* __restrict__ g_logits,
                                                           const float* __restrict__ g_probs, const float* __restrict__ g_conf,
                                                           float* __restrict__ dsigma, u16* __restrict__ H1, u16* __restrict__ H2,
                                                           u16* __restrict__ dZ1, u16* __restrict__ dZ2, u16* __restrict__ GL,
                                                           u16* __restrict__ XN, long long n) {
  __shared__ __attribute__((aligned(16))) float P[MAA_NPARAM];
  maa_load_params(params, P);
  const float mean = stats[0], inv = 1.0f / (sqrtf(stats[1]) + 1e-8f);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const float x = (sigma[e] - mean) * inv;
    float h1[MAA_H], h2[MAA_H];
#pragma unroll
    for (int j = 0; j < MAA_H; ++j) h1[j] = fmaxf(P[MAA_W1 + j] * x + P[MAA_B1 + j], 0.f);
    float lg[MAA_C];
#pragma unroll
    for (int c = 0; c < MAA_C; ++c) lg[c] = P[MAA_B3 + c];
    for (int i = 0; i < MAA_H; ++i) {
      float z = P[MAA_B2 + i];
      const f32x4* w = reinterpret_cast<const f32x4*>(&P[MAA_W2 + i * MAA_H]);
#pragma unroll
      for (int j4 = 0; j4 < MAA_H / 4; ++j4) {
        const f32x4 ww = w[j4];
        z += ww[0] * h1[4 * j4] + ww[1] * h1[4 * j4 + 1] + ww[2] * h1[4 * j4 + 2] + ww[3] * h1[4 * j4 + 3];
      }
      h2[i] = fmaxf(z, 0.f);
#pragma unroll
      for (int c = 0; c < MAA_C; ++c) lg[c] += P[MAA_W3 + c * MAA_H + i] * h2[i];
    }
    // total gradient on the logits: direct + through the softmax
    float gl[MAA_C] = {0.f, 0.f, 0.f, 0.f};
    if (g_logits) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(g_logits + e * MAA_C);
      gl[0] = g[0]; gl[1] = g[1]; gl[2] = g[2]; gl[3] = g[3];
    }
    if (g_probs) {
      float mx = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
      float p[MAA_C], den = 0.f;
#pragma unroll
      for (int c = 0; c < MAA_C; ++c) { p[c] = __expf(lg[c] - mx); den += p[c]; }
      const f32x4 g = *reinterpret_cast<const f32x4*>(g_probs + e * MAA_C);
      const float gp[MAA_C] = {g[0], g[1], g[2], g[3]};
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < MAA_C; ++c) { p[c] /= den; dot += gp[c] * p[c]; }
#pragma unroll
      for (int c = 0; c < MAA_C; ++c) gl[c] += p[c] * (gp[c] - dot);
    }
    maa_store_row<T>(H1, e, h1);
    maa_store_row<T>(H2, e, h2);
    float dh1[MAA_H];
#pragma unroll
    for (int j = 0; j < MAA_H; ++j) dh1[j] = 0.f;
    for (int i = 0; i < MAA_H; ++i) {                      // dz2 overwrites h2 (already stored)
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < MAA_C; ++c) d += P[MAA_W3 + c * MAA_H + i] * gl[c];
      d = h2[i] > 0.f ? d : 0.f;
      h2[i] = d;
      const f32x4* w = reinterpret_cast<const f32x4*>(&P[MAA_W2 + i * MAA_H]);
#pragma unroll
      for (int j4 = 0; j4 < MAA_H / 4; ++j4) {
        const f32x4 ww = w[j4];
        dh1[4 * j4] += ww[0] * d; dh1[4 * j4 + 1] += ww[1] * d; dh1[4 * j4 + 2] += ww[2] * d; dh1[4 * j4 + 3] += ww[3] * d;
      }
    }
    maa_store_row<T>(dZ2, e, h2);
    float dn = 0.f;
#pragma unroll
    for (int j = 0; j < MAA_H; ++j) {
      dh1[j] = h1[j] > 0.f ? dh1[j] : 0.f;
      dn += P[MAA_W1 + j] * dh1[j];
    }
    maa_store_row<T>(dZ1, e, dh1);
    if (g_conf) {
      const float c = 1.0f / (1.0f + __expf(x));
      dn -= g_conf[e] * c * (1.0f - c);
    }
    dsigma[e] = dn * inv;
    *reinterpret_cast<u32x4*>(GL + e * 8) = u32x4{pack2<T>(gl[0], gl[1]), pack2<T>(gl[2], gl[3]), 0u, 0u};
    *reinterpret_cast<u32x4*>(XN + e * 8) = u32x4{pack2<T>(x, 0.f), 0u, 0u, 0u};
  }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_forward_kernel(const float* __restrict__ x, const float* __restrict__ cent, int M,
                                                         float* __restrict__ q, long long* __restrict__ idx,
                                                         double* __restrict__ acc, long long n) {
  __shared__ double red[4];
  float c[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) c[k] = k < M ? cent[k] : 0.f;
  double s = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const float v = x[e];
    float best = (v - c[0]) * (v - c[0]);
    int arg = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float d = (v - c[k]) * (v - c[k]);
      if (k < M && d < best) { best = d; arg = k; }
    }
    float qv = c[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) qv = (arg == k) ? c[k] : qv;
    q[e] = v + (qv - v);                               // models/vq.py:91 "x + (quantized - x).detach()", rounding included
    idx[e] = arg;
    const float diff = v - qv;
    s += (double)diff * (double)diff;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}

// dx = g_q + g_loss * beta * 2 (x - q) / n;  dcent[k] += g_loss * 2 (q - x) / n over the elements assigned to k
__global__ __launch_bounds__(256) void vq_backward_kernel(const float* __restrict__ x, const long long* __restrict__ idx,
                                                          const float* __restrict__ cent, int M, const float* __restrict__ g_q,
                                                          const float* __restrict__ g_loss, float beta, float* __restrict__ dx,
                                                          float* __restrict__ dcent, long long n) {
  __shared__ float red[4][16];
  float c[16], dc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { c[k] = k < M ? cent[k] : 0.f; dc[k] = 0.f; }
  const float gl = g_loss ? g_loss[0] * 2.0f / (float)n : 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int a = (int)idx[e];
    float qv = c[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) qv = (a == k) ? c[k] : qv;
    const float diff = x[e] - qv;
    dx[e] = (g_q ? g_q[e] : 0.f) + gl * beta * diff;
#pragma unroll
    for (int k = 0; k < 16; ++k) dc[k] += (a == k) ? -gl * diff : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float v = dc[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < M) atomicAdd(&dcent[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static unsigned routing_blocks(long long n) {
  long long nb = (n + 255) / 256;
  return (unsigned)(nb > 2048 ? 2048 : nb);
}

extern "C" int sfm_maa_update_stats(const float* sigma, long long n, double* acc, float* stats, long long* num_updates,
                                    float momentum, void* stream) {
  if (!sigma || !acc || !stats || !num_updates) return SFM_ERR_ARG;
  if (n <= 0) return SFM_ERR_SHAPE;
  SFM_LAUNCH(maa_stats_kernel, dim3(routing_blocks(n)), dim3(256), 0, (hipStream_t)stream, sigma, n, acc);
  SFM_LAUNCH(maa_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, n, stats, num_updates, momentum);
  return SFM_OK;
}

extern "C" int sfm_maa_forward(const float* sigma, const float* stats, const float* params, float* logits, float* probs,
                               long long* decisions, float* confidence, long long n, void* stream) {
  if (!sigma || !stats || !params || !logits || !probs || !decisions || !confidence) return SFM_ERR_ARG;
  if (n <= 0) return SFM_ERR_SHAPE;
  SFM_LAUNCH(maa_forward_kernel, dim3(routing_blocks(n)), dim3(256), 0, (hipStream_t)stream, sigma, stats, params, logits, probs,
             decisions, confidence, n);
  return SFM_OK;
}

extern "C" int sfm_maa_backward(const float* sigma, const float* stats, const float* params, const float* g_logits,
                                const float* g_probs, const float* g_conf, float* dsigma, void* H1, void* H2, void* dZ1, void* dZ2,
                                void* GL, void* XN, long long n, int dtype, void* stream) {
  if (!sigma || !stats || !params || !dsigma || !H1 || !H2 || !dZ1 || !dZ2 || !GL || !XN) return SFM_ERR_ARG;
  if (n <= 0) return SFM_ERR_SHAPE;
  if (dtype == SFM_DT_F16)
    SFM_LAUNCH((maa_backward_kernel<F16>), dim3(routing_blocks(n)), dim3(256), 0, (hipStream_t)stream, sigma, stats, params, g_logits,
               g_probs, g_conf, dsigma, (u16*)H1, (u16*)H2, (u16*)dZ1, (u16*)dZ2, (u16*)GL, (u16*)XN, n);
  else
    SFM_LAUNCH((maa_backward_kernel<BF16>), dim3(routing_blocks(n)), dim3(256), 0, (hipStream_t)stream, sigma, stats, params, g_logits,
               g_probs, g_conf, dsigma, (u16*)H1, (u16*)H2, (u16*)dZ1, (u16*)dZ2, (u16*)GL, (u16*)XN, n);
  return SFM_OK;
}

extern "C" int sfm_vq_forward(const float* x, const float* centroids, int M, float* q, long long* idx, double* acc, long long n,
                              void* stream) {
  if (!x || !centroids || !q || !idx || !acc) return SFM_ERR_ARG;
  if (n <= 0 || M <= 0 || M > 16) return SFM_ERR_SHAPE;
  SFM_LAUNCH(vq_forward_kernel, dim3(routing_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, centroids, M, q, idx, acc, n);
  return SFM_OK;
}

extern "C" int sfm_vq_backward(const float* x, const long long* idx, const float* centroids, int M, const float* g_q,
                               const float* g_loss, float beta, float* dx, float* dcent, long long n, void* stream) {
  if (!x || !idx || !centroids || !dx || !dcent) return SFM_ERR_ARG;
  if (n <= 0 || M <= 0 || M > 16) return SFM_ERR_SHAPE;
  SFM_LAUNCH(vq_backward_kernel, dim3(routing_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, idx, centroids, M, g_q, g_loss, beta,
             dx, dcent, n);
  return SFM_OK;
}
