// Backward of the PerceptionAgent's normalisation nodes (agents/perception.py:121-129, 157, 192-206, 233-246):
//   out = act( GN(x1) [+ GN(x2)] ),   GN(x)[b,l,c] = (x - mean[b,g]) rstd[b,g] gamma[c] + beta[c],  act = identity | GELU(erf)
// channels-last [B, L, C]; x1 / x2 are the raw conv outputs the forward kept (16-bit or fp32).  Two streaming passes:
//   gn_bwd_reduce : dp = dout * act'(p);  S[b][c] = { sum_l dp, sum_l dp xhat1, sum_l dp xhat2 }   (fp32 atomics)
//                   from S the host forms dgamma / dbeta (sums over b) and, per (b, group), the two correction sums
//   gn_bwd_apply  : dx_i = a_i dp - b_i - xhat_i c_i   with the per-(b, c) coefficients a = rstd gamma,
//                   b = rstd/N sum_g(dp gamma), c = rstd/N sum_g(dp gamma xhat)   (dp is recomputed, not stored)
// Per-(b, c) tables (fp32 [B, C]): sc / sh = the forward's scale and shift (to recompute p), mu / rs = group mean and
// rstd broadcast to channels (xhat = (x - mu) rs).
#include "sfm_common.h"

struct GnIn {
  const void* x;          // raw input [B, L, C]
  const float* sc;        // [B, C]
  const float* sh;
  const float* mu;
  const float* rs;
  const float* ca;        // apply pass: a, b, c coefficients [B, C]
  const float* cb;
  const float* cc;
  void* dx;               // apply pass output [B, L, C]
  int x_f32, dx_f32;
};

template <class T>
__device__ __forceinline__ float gn_ld(const void* p, long long e, int f32) {
  return f32 ? reinterpret_cast<const float*>(p)[e] : T::to_f32(reinterpret_cast<const u16*>(p)[e]);
}

__device__ __forceinline__ float gelu_grad(float z) {
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
  return cdf + z * 0.39894228040143267794f * __expf(-0.5f * z * z);
}

template <class T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const void* __restrict__ dout, int dout_f32, GnIn i1, GnIn i2,
                                                            int two, float* __restrict__ S, int L, int C, int act,
                                                            int rows_per_block) {
  __shared__ float red[256][3];
  const int b = blockIdx.y;
  const int nc = C < 256 ? C : 256;                    // channels covered per pass of the block
  const int rl = 256 / nc;                             // row lanes
  const int tc = threadIdx.x % nc, tr = threadIdx.x / nc;
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  for (int c = tc; c < C; c += nc) {
    const long long bc = (long long)b * C + c;
    const float sc1 = i1.sc[bc], sh1 = i1.sh[bc], mu1 = i1.mu[bc], rs1 = i1.rs[bc];
    float sc2 = 0.f, sh2 = 0.f, mu2 = 0.f, rs2 = 0.f;
    if (two) { sc2 = i2.sc[bc]; sh2 = i2.sh[bc]; mu2 = i2.mu[bc]; rs2 = i2.rs[bc]; }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int l = l0 + tr; l < l1; l += rl) {
      const long long e = ((long long)b * L + l) * C + c;
      const float x1 = gn_ld<T>(i1.x, e, i1.x_f32);
      float p = x1 * sc1 + sh1, x2 = 0.f;
      if (two) { x2 = gn_ld<T>(i2.x, e, i2.x_f32); p += x2 * sc2 + sh2; }
      float dp = gn_ld<T>(dout, e, dout_f32);
      if (act) dp *= gelu_grad(p);
      s0 += dp;
      s1 += dp * (x1 - mu1) * rs1;
      if (two) s2 += dp * (x2 - mu2) * rs2;
    }
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1; red[threadIdx.x][2] = s2;
    __syncthreads();
    if (tr == 0) {
      for (int r = 1; r < rl; ++r) { s0 += red[r * nc + tc][0]; s1 += red[r * nc + tc][1]; s2 += red[r * nc + tc][2]; }
      atomicAdd(&S[bc * 3 + 0], s0);
      atomicAdd(&S[bc * 3 + 1], s1);
      if (two) atomicAdd(&S[bc * 3 + 2], s2);
    }
    __syncthreads();
  }
}

template <class T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const void* __restrict__ dout, int dout_f32, GnIn i1, GnIn i2,
                                                           int two, int L, int C, int act, long long total) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long long b = e / ((long long)L * C);
    const long long bc = b * C + c;
    const float x1 = gn_ld<T>(i1.x, e, i1.x_f32);
    float p = x1 * i1.sc[bc] + i1.sh[bc], x2 = 0.f;
    if (two) { x2 = gn_ld<T>(i2.x, e, i2.x_f32); p += x2 * i2.sc[bc] + i2.sh[bc]; }
    float dp = gn_ld<T>(dout, e, dout_f32);
    if (act) dp *= gelu_grad(p);
    const float d1 = i1.ca[bc] * dp - i1.cb[bc] - (x1 - i1.mu[bc]) * i1.rs[bc] * i1.cc[bc];
    if (i1.dx_f32) reinterpret_cast<float*>(i1.dx)[e] = d1;
    else reinterpret_cast<u16*>(i1.dx)[e] = T::from_f32(d1);
    if (two) {
      const float d2 = i2.ca[bc] * dp - i2.cb[bc] - (x2 - i2.mu[bc]) * i2.rs[bc] * i2.cc[bc];
      if (i2.dx_f32) reinterpret_cast<float*>(i2.dx)[e] = d2;
      else reinterpret_cast<u16*>(i2.dx)[e] = T::from_f32(d2);
    }
  }
}

static GnIn gn_in(const void* x, int x_f32, const float* tab, long long BC, void* dx, int dx_f32, const float* coef) {
  GnIn g;
  g.x = x; g.x_f32 = x_f32;
  g.sc = tab; g.sh = tab ? tab + BC : nullptr; g.mu = tab ? tab + 2 * BC : nullptr; g.rs = tab ? tab + 3 * BC : nullptr;
  g.ca = coef; g.cb = coef ? coef + BC : nullptr; g.cc = coef ? coef + 2 * BC : nullptr;
  g.dx = dx; g.dx_f32 = dx_f32;
  return g;
}

// tab1 / tab2: [4][B][C] fp32 = (scale, shift, mean, rstd);  S: [B][C][3] fp32, zero-filled by the caller
extern "C" int sfm_gn_bwd_reduce(const void* dout, int dout_f32, const void* x1, int x1_f32, const float* tab1,
                                 const void* x2, int x2_f32, const float* tab2, float* S, int B, int L, int C, int act,
                                 int dtype, void* stream) {
  if (!dout || !x1 || !tab1 || !S || ((x2 == nullptr) != (tab2 == nullptr))) return SFM_ERR_ARG;
  if (B <= 0 || L <= 0 || C <= 0 || (C > 256 && C % 256 != 0) || (C < 256 && 256 % C != 0)) return SFM_ERR_SHAPE;
  const long long BC = (long long)B * C;
  const GnIn a = gn_in(x1, x1_f32, tab1, BC, nullptr, 0, nullptr), c = gn_in(x2, x2_f32, tab2, BC, nullptr, 0, nullptr);
  const int rpb = 256;
  dim3 grid((L + rpb - 1) / rpb, B), block(256);
  if (dtype == SFM_DT_F16)
    SFM_LAUNCH((gn_bwd_reduce_kernel<F16>), grid, block, 0, (hipStream_t)stream, dout, dout_f32, a, c, x2 ? 1 : 0, S, L, C, act, rpb);
  else
    SFM_LAUNCH((gn_bwd_reduce_kernel<BF16>), grid, block, 0, (hipStream_t)stream, dout, dout_f32, a, c, x2 ? 1 : 0, S, L, C, act, rpb);
  return SFM_OK;
}

// coef1 / coef2: [3][B][C] fp32 = (a, b, c);  dx1 / dx2: [B, L, C] 16-bit or fp32
extern "C" int sfm_gn_bwd_apply(const void* dout, int dout_f32, const void* x1, int x1_f32, const float* tab1,
                                const float* coef1, void* dx1, int dx1_f32, const void* x2, int x2_f32, const float* tab2,
                                const float* coef2, void* dx2, int dx2_f32, int B, int L, int C, int act, int dtype,
                                void* stream) {
  if (!dout || !x1 || !tab1 || !coef1 || !dx1) return SFM_ERR_ARG;
  if (x2 && (!tab2 || !coef2 || !dx2)) return SFM_ERR_ARG;
  if (B <= 0 || L <= 0 || C <= 0) return SFM_ERR_SHAPE;
  const long long BC = (long long)B * C, total = (long long)B * L * C;
  const GnIn a = gn_in(x1, x1_f32, tab1, BC, dx1, dx1_f32, coef1), c = gn_in(x2, x2_f32, tab2, BC, dx2, dx2_f32, coef2);
  long long nb = (total + 255) / 256;
  if (nb > 32768) nb = 32768;
  if (dtype == SFM_DT_F16)
    SFM_LAUNCH((gn_bwd_apply_kernel<F16>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dout, dout_f32, a, c, x2 ? 1 : 0, L, C, act, total);
  else
    SFM_LAUNCH((gn_bwd_apply_kernel<BF16>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dout, dout_f32, a, c, x2 ? 1 : 0, L, C, act, total);
  return SFM_OK;
}
