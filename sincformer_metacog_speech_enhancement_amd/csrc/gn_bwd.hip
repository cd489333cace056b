// Backward of the PerceptionAgent's normalisation nodes (agents/perception.py:121-129, 157, 192-206, 233-246):
//   out = act( GN(x1) [+ GN(x2)] ),   GN(x)[b,l,c] = (x - mean[b,g]) rstd[b,g] gamma[c] + beta[c],  act = identity | GELU(erf)
// channels-last [B, L, C]; x1 / x2 are the raw conv outputs the forward kept (16-bit or fp32).  Two streaming passes and
// a tiny one in between, all HBM-bound (16-byte loads: a thread owns 8 neighbouring channels and walks down the rows):
//   gn_bwd_reduce : dp = dout * act'(p);  S[b][k][c] = { sum_l dp, sum_l dp xhat1, sum_l dp xhat2 }
//   gn_bwd_coefs  : per (b, c): a = rstd gamma, b = rstd/N sum_g(dp gamma), c = rstd/N sum_g(dp gamma xhat);
//                   dbeta / dgamma = sums of S over b
//   gn_bwd_apply  : dx_i = a_i dp - b_i - xhat_i c_i          (dp is recomputed, not stored)
// sc / sh [B, C]: the forward's per-(utterance, channel) scale and shift (to recompute p); mean / rstd [B, G].
#include "sfm_common.h"

struct GnIn {
  const void* x;          // raw input [B, L, C]
  const float* sc;        // [B, C]
  const float* sh;
  const float* mean;      // [B, G]
  const float* rstd;
  const float* coef;      // apply pass: [3][B][C] = a, b, c
  void* dx;               // apply pass output [B, L, C]
  int x_f32, dx_f32;
};

template <class T>
__device__ __forceinline__ void gn_ld8(const void* p, long long e, int f32, float (&v)[8]) {
  if (f32) {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + e);
    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + e + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(p) + e);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = T::to_f32((u16)(w[j] & 0xffffu));
      v[2 * j + 1] = T::to_f32((u16)(w[j] >> 16));
    }
  }
}

template <class T>
__device__ __forceinline__ void gn_st8(void* p, long long e, int f32, const float (&v)[8]) {
  if (f32) {
    float* o = reinterpret_cast<float*>(p) + e;
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4 q;
    q.x = T::pack(v[0], v[1]); q.y = T::pack(v[2], v[3]); q.z = T::pack(v[4], v[5]); q.w = T::pack(v[6], v[7]);
    *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p) + e) = q;
  }
}

// d/dz of z Phi(z) = Phi(z) + z phi(z) with the polynomial CDF of sfm_common.h (|error| < 5.7e-5, under the 16-bit spacing of the
// d it scales): ONE transcendental (the exponential of phi) instead of the two of the Abramowitz-Stegun form (reciprocal +
// exponential) - the reduce pass is VALU-bound (tools/gn_bwd_bench.py; profiles/README.md, round 3).
__device__ __forceinline__ float gelu_grad(float z) {
  const float ex = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170368f);          // exp(-z^2 / 2)
  return fmaf(z * 0.39894228040143267794f, ex, normal_cdf_poly(z));
}

struct GnRegs {
  float sc[8], sh[8], mu[8], rs[8];
};

__device__ __forceinline__ void gn_load_regs(const GnIn& in, long long b, int c0, int C, int G, GnRegs& r) {
  const int cg = C / G;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    r.sc[j] = in.sc[b * C + c0 + j];
    r.sh[j] = in.sh[b * C + c0 + j];
    r.mu[j] = in.mean[b * G + (c0 + j) / cg];
    r.rs[j] = in.rstd[b * G + (c0 + j) / cg];
  }
}

// grid (row tiles, B); block 256 = (C/8 channel vectors) x (2048/C row lanes); C in {64, 128, 256, 512, 1024, 2048}
template <class T, int TWO, int ACT>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const void* __restrict__ dout, int dout_f32, GnIn i1, GnIn i2,
                                                            float* __restrict__ S, int L, int C, int G, int rows_per_block,
                                                            float* __restrict__ ws) {
  extern __shared__ float red[];                       // [rl][3][C]
  const long long b = blockIdx.y;
  const int nv = C >> 3, rl = 256 / nv;
  const int tv = threadIdx.x % nv, tr = threadIdx.x / nv, c0 = tv * 8;
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  GnRegs r1, r2;
  gn_load_regs(i1, b, c0, C, G, r1);
  if (TWO) gn_load_regs(i2, b, c0, C, G, r2);
  float s0[8] = {}, s1[8] = {}, s2[8] = {};
  for (int l = l0 + tr; l < l1; l += rl) {
    const long long e = (b * L + l) * C + c0;
    float x1[8], x2[8], dp[8];
    gn_ld8<T>(i1.x, e, i1.x_f32, x1);
    if (TWO) gn_ld8<T>(i2.x, e, i2.x_f32, x2);
    gn_ld8<T>(dout, e, dout_f32, dp);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ACT) {
        float p = x1[j] * r1.sc[j] + r1.sh[j];
        if (TWO) p += x2[j] * r2.sc[j] + r2.sh[j];
        dp[j] *= gelu_grad(p);
      }
      const float d = dp[j];
      s0[j] += d;
      s1[j] += d * (x1[j] - r1.mu[j]);                 // x rstd once, below
      if (TWO) s2[j] += d * (x2[j] - r2.mu[j]);
    }
  }
  float* mine = red + (long long)tr * 3 * C;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mine[c0 + j] = s0[j];
    mine[C + c0 + j] = s1[j] * r1.rs[j];
    mine[2 * C + c0 + j] = TWO ? s2[j] * r2.rs[j] : 0.f;
  }
  __syncthreads();
  const int nk = TWO ? 3 * C : 2 * C;
  if (ws) {                                            // partial [row tile][B][3][C], folded over the row tiles in tile order
    float* mine_ws = ws + ((long long)blockIdx.x * gridDim.y + b) * 3 * C;
    for (int i = threadIdx.x; i < 3 * C; i += 256) {
      float a = 0.f;
      if (i < nk)
        for (int r = 0; r < rl; ++r) a += red[(long long)r * 3 * C + i];
      mine_ws[i] = a;
    }
    return;
  }
  for (int i = threadIdx.x; i < nk; i += 256) {
    float a = 0.f;
    for (int r = 0; r < rl; ++r) a += red[(long long)r * 3 * C + i];
    atomicAdd(&S[b * 3 * C + i], a);
  }
}

// grid B, block 256: coefficient tables.  (The parameter gradients dparam [3][C] = dbeta, dgamma1, dgamma2 = sum_b S[b] are taken
// afterwards by the ordered fold of reduce.hip - no atomics, the same bits on every run.)
__global__ __launch_bounds__(256) void gn_bwd_coefs_kernel(const float* __restrict__ S, const float* __restrict__ gamma1,
                                                           const float* __restrict__ rstd1, const float* __restrict__ gamma2,
                                                           const float* __restrict__ rstd2, float* __restrict__ coef1,
                                                           float* __restrict__ coef2, int B, int C, int G, float inv_n) {
  __shared__ float gs[4][256];                         // per group: A1, B1, A2, B2
  const long long b = blockIdx.x;
  const int cg = C / G;
  const float* Sb = S + b * 3 * C;
  for (int g = threadIdx.x; g < G; g += 256) {
    float a1 = 0.f, b1 = 0.f, a2 = 0.f, b2 = 0.f;
    for (int c = g * cg; c < (g + 1) * cg; ++c) {
      a1 += Sb[c] * gamma1[c];
      b1 += Sb[C + c] * gamma1[c];
      if (gamma2) { a2 += Sb[c] * gamma2[c]; b2 += Sb[2 * C + c] * gamma2[c]; }
    }
    gs[0][g] = a1; gs[1][g] = b1; gs[2][g] = a2; gs[3][g] = b2;
  }
  __syncthreads();
  const long long BC = (long long)B * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cg;
    const float r1 = rstd1[b * G + g];
    coef1[b * C + c] = r1 * gamma1[c];
    coef1[BC + b * C + c] = r1 * gs[0][g] * inv_n;
    coef1[2 * BC + b * C + c] = r1 * gs[1][g] * inv_n;
    if (gamma2) {
      const float r2 = rstd2[b * G + g];
      coef2[b * C + c] = r2 * gamma2[c];
      coef2[BC + b * C + c] = r2 * gs[2][g] * inv_n;
      coef2[2 * BC + b * C + c] = r2 * gs[3][g] * inv_n;
    }
  }
}

template <class T, int TWO, int ACT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const void* __restrict__ dout, int dout_f32, GnIn i1, GnIn i2,
                                                           int B, int L, int C, int G, int rows_per_block) {
  const long long b = blockIdx.y;
  const int nv = C >> 3, rl = 256 / nv;
  const int tv = threadIdx.x % nv, tr = threadIdx.x / nv, c0 = tv * 8;
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  const long long BC = (long long)B * C;
  GnRegs r1, r2;
  float a1[8], b1[8], k1[8], a2[8], b2[8], k2[8];
  gn_load_regs(i1, b, c0, C, G, r1);
  if (TWO) gn_load_regs(i2, b, c0, C, G, r2);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // d1 = a d - b - (x - mu) rstd k  =  a d - (b - mu rstd k) - x (rstd k)
    a1[j] = i1.coef[b * C + c0 + j];
    k1[j] = i1.coef[2 * BC + b * C + c0 + j] * r1.rs[j];
    b1[j] = i1.coef[BC + b * C + c0 + j] - r1.mu[j] * k1[j];
    if (TWO) {
      a2[j] = i2.coef[b * C + c0 + j];
      k2[j] = i2.coef[2 * BC + b * C + c0 + j] * r2.rs[j];
      b2[j] = i2.coef[BC + b * C + c0 + j] - r2.mu[j] * k2[j];
    }
  }
  // two rows per trip: 10 % more bytes in flight pays here (8.2 -> 7.6 ms over the nine nodes); the reduce pass LOSES with it (5.9 -> 6.2)
#pragma unroll 2
  for (int l = l0 + tr; l < l1; l += rl) {
    const long long e = (b * L + l) * C + c0;
    float x1[8], x2[8], dp[8], d1[8], d2[8];
    gn_ld8<T>(i1.x, e, i1.x_f32, x1);
    if (TWO) gn_ld8<T>(i2.x, e, i2.x_f32, x2);
    gn_ld8<T>(dout, e, dout_f32, dp);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float d = dp[j];
      if (ACT) {
        float p = x1[j] * r1.sc[j] + r1.sh[j];
        if (TWO) p += x2[j] * r2.sc[j] + r2.sh[j];
        d *= gelu_grad(p);
      }
      d1[j] = fmaf(-x1[j], k1[j], fmaf(a1[j], d, -b1[j]));
      if (TWO) d2[j] = fmaf(-x2[j], k2[j], fmaf(a2[j], d, -b2[j]));
    }
    gn_st8<T>(i1.dx, e, i1.dx_f32, d1);
    if (TWO) gn_st8<T>(i2.dx, e, i2.dx_f32, d2);
  }
}

static GnIn gn_in(const void* x, int x_f32, const float* sc, const float* sh, const float* mean, const float* rstd,
                  const float* coef, void* dx, int dx_f32) {
  GnIn g;
  g.x = x; g.x_f32 = x_f32; g.sc = sc; g.sh = sh; g.mean = mean; g.rstd = rstd; g.coef = coef; g.dx = dx; g.dx_f32 = dx_f32;
  return g;
}

static bool gn_shape_ok(int B, int L, int C, int G) {
  if (B <= 0 || L <= 0 || C < 64 || C > 2048 || (C & (C - 1)) || G <= 0 || G > 256 || C % G || B > 65535) return false;
  return true;
}

static int gn_rows_per_block(int C) { return 8 * (2048 / C) > 256 ? 8 * (2048 / C) : 256; }

// S: [B][3][C] fp32, zero-filled by the caller
extern "C" int sfm_gn_bwd_reduce(const void* dout, int dout_f32, const void* x1, int x1_f32, const float* sc1, const float* sh1,
                                 const float* mean1, const float* rstd1, const void* x2, int x2_f32, const float* sc2,
                                 const float* sh2, const float* mean2, const float* rstd2, float* S, int B, int L, int C, int G,
                                 int act, int dtype, float* ws, void* stream) {
  if (!dout || !x1 || !sc1 || !sh1 || !mean1 || !rstd1 || !S) return SFM_ERR_ARG;
  if (ws && (((uintptr_t)ws) % 16) != 0) return SFM_ERR_ARG;
  if (x2 && (!sc2 || !sh2 || !mean2 || !rstd2)) return SFM_ERR_ARG;
  if (!gn_shape_ok(B, L, C, G)) return SFM_ERR_SHAPE;
  const GnIn a = gn_in(x1, x1_f32, sc1, sh1, mean1, rstd1, nullptr, nullptr, 0);
  const GnIn c = gn_in(x2, x2_f32, sc2, sh2, mean2, rstd2, nullptr, nullptr, 0);
  const int rpb = gn_rows_per_block(C);
  dim3 grid((L + rpb - 1) / rpb, B), block(256);
  const size_t lds = (size_t)(2048 / C) * 3 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define GNR_GO(T, TWO, ACT) SFM_LAUNCH((gn_bwd_reduce_kernel<T, TWO, ACT>), grid, block, lds, st, dout, dout_f32, a, c, S, L, C, G, rpb, ws)
#define GNR_T(T) do { if (x2) { if (act) GNR_GO(T, 1, 1); else GNR_GO(T, 1, 0); } else { if (act) GNR_GO(T, 0, 1); else GNR_GO(T, 0, 0); } } while (0)
  if (dtype == SFM_DT_F16) GNR_T(F16); else GNR_T(BF16);
#undef GNR_T
#undef GNR_GO
  return ws ? sfm_fold_partials(ws, S, B, 3 * C, 3 * C, (int)grid.x, 1, stream) : SFM_OK;
}

// floats of the optional workspace of sfm_gn_bwd_reduce (one partial [B][3][C] per row tile, folded in tile order; NULL = atomics)
extern "C" long long sfm_gn_bwd_reduce_ws_floats(int B, int L, int C) {
  if (B <= 0 || L <= 0 || C < 64 || C > 2048) return 0;
  const int rpb = gn_rows_per_block(C);
  return (long long)((L + rpb - 1) / rpb + 1) * B * 3 * C;
}

// coef1 / coef2: [3][B][C] fp32 (a, b, c);  dparam: [3][C] fp32 = dbeta, dgamma1, dgamma2, zero-filled by the caller (+=).
// S [B][3][C] is DESTROYED: after the coefficient tables it is the scratch of the ordered fold over b that produces dparam.
extern "C" int sfm_gn_bwd_coefs(float* S, const float* gamma1, const float* rstd1, const float* gamma2,
                                const float* rstd2, float* coef1, float* coef2, float* dparam, int B, int L, int C, int G,
                                void* stream) {
  if (!S || !gamma1 || !rstd1 || !coef1 || !dparam) return SFM_ERR_ARG;
  if (gamma2 && (!rstd2 || !coef2)) return SFM_ERR_ARG;
  if (!gn_shape_ok(B, L, C, G)) return SFM_ERR_SHAPE;
  const float inv_n = 1.0f / ((float)L * (float)(C / G));
  SFM_LAUNCH(gn_bwd_coefs_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, S, gamma1, rstd1, gamma2, rstd2, coef1, coef2,
             B, C, G, inv_n);
  return sfm_fold_partials(S, dparam, 1, 3 * C, 3 * C, B, 1, stream);      // (third block: zeros when there is no second input)
}

// dx1 / dx2: [B, L, C] 16-bit or fp32
extern "C" int sfm_gn_bwd_apply(const void* dout, int dout_f32, const void* x1, int x1_f32, const float* sc1, const float* sh1,
                                const float* mean1, const float* rstd1, const float* coef1, void* dx1, int dx1_f32,
                                const void* x2, int x2_f32, const float* sc2, const float* sh2, const float* mean2,
                                const float* rstd2, const float* coef2, void* dx2, int dx2_f32, int B, int L, int C, int G,
                                int act, int dtype, void* stream) {
  if (!dout || !x1 || !sc1 || !sh1 || !mean1 || !rstd1 || !coef1 || !dx1) return SFM_ERR_ARG;
  if (x2 && (!sc2 || !sh2 || !mean2 || !rstd2 || !coef2 || !dx2)) return SFM_ERR_ARG;
  if (!gn_shape_ok(B, L, C, G)) return SFM_ERR_SHAPE;
  const GnIn a = gn_in(x1, x1_f32, sc1, sh1, mean1, rstd1, coef1, dx1, dx1_f32);
  const GnIn c = gn_in(x2, x2_f32, sc2, sh2, mean2, rstd2, coef2, dx2, dx2_f32);
  const int rpb = gn_rows_per_block(C);
  dim3 grid((L + rpb - 1) / rpb, B), block(256);
  hipStream_t st = (hipStream_t)stream;
#define GNA_GO(T, TWO, ACT) SFM_LAUNCH((gn_bwd_apply_kernel<T, TWO, ACT>), grid, block, 0, st, dout, dout_f32, a, c, B, L, C, G, rpb)
#define GNA_T(T) do { if (x2) { if (act) GNA_GO(T, 1, 1); else GNA_GO(T, 1, 0); } else { if (act) GNA_GO(T, 0, 1); else GNA_GO(T, 0, 0); } } while (0)
  if (dtype == SFM_DT_F16) GNA_T(F16); else GNA_T(BF16);
#undef GNA_T
#undef GNA_GO
  return SFM_OK;
}
