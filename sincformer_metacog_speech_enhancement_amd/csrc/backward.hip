// Backward / training-mode kernels of the ConformerBlock (models/conformer.py:28-151) that are not GEMMs:
// LayerNorm backward, activation (Swish / GLU) backward with counter-based dropout, BatchNorm1d
// training statistics and backward, depthwise-conv weight gradient.  All HBM-bound streaming kernels.
#include "sfm_common.h"

// ---------------------------------------------------------------------------
// LayerNorm backward: dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//                     dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy          (D <= 512)
// ---------------------------------------------------------------------------
// DYF: format of dy - 0 fp32, 1 bf16, 2 fp16 (the 16-bit result of the GEMM that produced it: half the bytes of that stream,
// written and read).  NH: 256-column halves a row spans (1 for D <= 256, 2 up to 512).  R: rows a wave carries per trip, all
// their loads (x, dy, dres) issued before the first wave reduction - one row per trip left the four dependent DPP reductions
// with nothing in flight behind them (2.6 TB/s on the c3t rows; see profiles/README.md, round 3).
template <bool VEC, int DYF, int NH, int R>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const void* __restrict__ dyv, const float* __restrict__ dres,
                                                            float* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int M, int D, int ldx, int ldy, int ld,
                                                            float eps, u16* __restrict__ nxt, float nalpha, float np_,
                                                            uint32_t nseed, int nfmt, float* __restrict__ ws) {
  // nxt (optional): the 16-bit operand the NEXT backward node starts from, nalpha * dropout(dx; np_, nseed) as [M, D] contiguous
  // rows (counter m * D + d, the one sfm_ew_train mode 4 uses) - that node's own pass over dx (read 4 B + write 2 B per
  // element, 24 launches per training step) is folded into this store.  nfmt: 1 bf16, 2 fp16.
  constexpr int NI = 4 * NH;
  const float nik = (np_ > 0.f) ? 1.0f / (1.0f - np_) : 1.0f;
  const uint32_t nthr = sfm_keep_threshold(np_);
  __shared__ float red[4][2][256 * NH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // VEC: lane owns columns 4*lane..4*lane+3 of each 256-column half (16-byte loads); else column lane + 64*i
#define LN_COL(i) (VEC ? (((i) >> 2) * 256 + 4 * lane + ((i) & 3)) : (lane + 64 * (i)))
  float pg[NI], pb[NI], gm[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    pg[i] = 0.f; pb[i] = 0.f;
    const int d = LN_COL(i);
    gm[i] = (d < D) ? gamma[d] : 0.f;
  }
  const float inv_d = 1.f / (float)D;
  for (int row0 = (blockIdx.x * 4 + wave) * R; row0 < M; row0 += gridDim.x * 4 * R) {
    float v[R][NI], g[R][NI], dr[R][NI];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      const bool live = row < M;
      const float* xr = x + (long long)row * ldx;
      const float* gr = reinterpret_cast<const float*>(dyv) + (long long)row * ldy;          // DYF == 0
      const u16* gh = reinterpret_cast<const u16*>(dyv) + (long long)row * ldy;               // DYF != 0
      const float* rr = dres + (long long)row * ld;
      if (VEC) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int d = h * 256 + 4 * lane;
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f}, e = {0.f, 0.f, 0.f, 0.f};
          if (live && d < D) {
            a = *reinterpret_cast<const f32x4*>(xr + d);
            if (DYF == 0) c = *reinterpret_cast<const f32x4*>(gr + d);
            else {
              const u32x2 w = *reinterpret_cast<const u32x2*>(gh + d);
              if (DYF == 1) c = f32x4{BF16::to_f32((u16)(w[0] & 0xffffu)), BF16::to_f32((u16)(w[0] >> 16)), BF16::to_f32((u16)(w[1] & 0xffffu)), BF16::to_f32((u16)(w[1] >> 16))};
              else c = f32x4{F16::to_f32((u16)(w[0] & 0xffffu)), F16::to_f32((u16)(w[0] >> 16)), F16::to_f32((u16)(w[1] & 0xffffu)), F16::to_f32((u16)(w[1] >> 16))};
            }
            if (dres) e = *reinterpret_cast<const f32x4*>(rr + d);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[r][4 * h + j] = a[j]; g[r][4 * h + j] = c[j]; dr[r][4 * h + j] = e[j]; }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int d = LN_COL(i);
          const bool ok = live && d < D;
          v[r][i] = ok ? xr[d] : 0.f;
          g[r][i] = ok ? (DYF == 0 ? gr[d] : DYF == 1 ? BF16::to_f32(gh[d]) : F16::to_f32(gh[d])) : 0.f;
          dr[r][i] = (ok && dres) ? rr[d] : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) s += v[r][i];
      const float mean = wave_sum_dpp(s) * inv_d;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int d = LN_COL(i);
        const float c = (d < D) ? v[r][i] - mean : 0.f;
        v[r][i] = c;
        q += c * c;
      }
      const float rstd = rsqrtf(wave_sum_dpp(q) * inv_d + eps);
      float a = 0.f, bsum = 0.f;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const float xh = v[r][i] * rstd;
        v[r][i] = xh;
        pg[i] += g[r][i] * xh;
        pb[i] += g[r][i];
        const float gg = g[r][i] * gm[i];
        g[r][i] = gg;
        a += gg;
        bsum += gg * xh;
      }
      a = wave_sum_dpp(a) * inv_d;
      bsum = wave_sum_dpp(bsum) * inv_d;
      if (row < M) {
        if (VEC) {
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            const int d = h * 256 + 4 * lane;
            if (d < D) {
              f32x4 o;
#pragma unroll
              for (int j = 0; j < 4; ++j) o[j] = rstd * (g[r][4 * h + j] - a - v[r][4 * h + j] * bsum) + dr[r][4 * h + j];
              *reinterpret_cast<f32x4*>(dx + (long long)row * ld + d) = o;
              if (nxt) {
                const unsigned long long e0 = (unsigned long long)row * D + d;      // d % 4 == 0: the four share one 8-element group
                float q[4];
                if (np_ > 0.f) {
                  const uint32_t gh = sfm_hash(nseed, e0 >> 3);
#pragma unroll
                  for (int j = 0; j < 4; ++j) q[j] = nalpha * o[j] * sfm_keep_from_group(gh, (uint32_t)(e0 & 7ull) + j, nthr, nik);
                } else {
#pragma unroll
                  for (int j = 0; j < 4; ++j) q[j] = nalpha * o[j];
                }
                u32x2 pk;
                if (nfmt == 2) { pk[0] = F16::pack(q[0], q[1]); pk[1] = F16::pack(q[2], q[3]); }
                else { pk[0] = BF16::pack(q[0], q[1]); pk[1] = BF16::pack(q[2], q[3]); }
                *reinterpret_cast<u32x2*>(nxt + e0) = pk;
              }
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const int d = LN_COL(i);
            if (d < D) {
              const float o = rstd * (g[r][i] - a - v[r][i] * bsum) + dr[r][i];
              dx[(long long)row * ld + d] = o;
              if (nxt) {
                const unsigned long long e = (unsigned long long)row * D + d;
                const float q = nalpha * o * (np_ > 0.f ? sfm_keep_scale(nseed, e, np_, nik) : 1.0f);
                nxt[e] = (nfmt == 2) ? F16::from_f32(q) : BF16::from_f32(q);
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    red[wave][0][LN_COL(i)] = pg[i];
    red[wave][1][LN_COL(i)] = pb[i];
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    const float sg = red[0][0][d] + red[1][0][d] + red[2][0][d] + red[3][0][d];
    const float sb = red[0][1][d] + red[1][1][d] + red[2][1][d] + red[3][1][d];
    if (ws) {                                                // this workgroup's partial row [dgamma | dbeta]: folded in block order
      ws[(long long)blockIdx.x * 2 * D + d] = sg;
      ws[(long long)blockIdx.x * 2 * D + D + d] = sb;
    } else {
      atomicAdd(&dgamma[d], sg);
      atomicAdd(&dbeta[d], sb);
    }
  }
}
#undef LN_COL

// dy_16: 0 = dy fp32, 1 = dy in the 16-bit format `dtype`; ldy = row stride of dy (elements), ld = row stride of dres and dx
static int layernorm_bwd_go(const float* x, const float* gamma, const void* dy, int dy_16, const float* dres, float* dx,
                            float* dgamma, float* dbeta, int M, int D, int ldx, int ldy, int ld, float eps, int dtype,
                            void* next16, float next_alpha, float next_p, unsigned int next_seed, float* ws, void* stream) {
  if (!x || !gamma || !dy || !dx || !dgamma || !dbeta) return SFM_ERR_ARG;
  if (ws && (((uintptr_t)ws) % 16) != 0) return SFM_ERR_ARG;
  if (M <= 0 || D <= 0 || D > 512) return SFM_ERR_SHAPE;
  if (next16 && (next_p < 0.f || next_p >= 1.f || ((uintptr_t)next16 % 8) != 0)) return SFM_ERR_SHAPE;
  u16* nxt = (u16*)next16;
  const int nfmt = dtype == SFM_DT_F16 ? 2 : 1;
  constexpr int LNB_R = 2;
  int nb = (M + 4 * LNB_R - 1) / (4 * LNB_R);
  if (nb > 1024) nb = 1024;                                  // 1024 x 512 contended atomics at the end: measured best of 512..4096
  const int dyb = dy_16 ? 2 : 4;
  const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (ld % 4 == 0) && (ldy % 4 == 0) &&
                   ((((uintptr_t)x | (uintptr_t)dres | (uintptr_t)dx) % 16) == 0) && (((uintptr_t)dy) % (4 * dyb)) == 0;
  const int f = dy_16 ? (dtype == SFM_DT_F16 ? 2 : 1) : 0;
  hipStream_t st = (hipStream_t)stream;
#define LNB_GO2(V, F, H) SFM_LAUNCH((layernorm_bwd_kernel<V, F, H, LNB_R>), dim3(nb), dim3(256), 0, st, x, gamma, dy, dres, dx, dgamma, dbeta, M, D, ldx, ldy, ld, eps, nxt, next_alpha, next_p, next_seed, nfmt, ws)
#define LNB_GO(V, F) do { if (D <= 256) LNB_GO2(V, F, 1); else LNB_GO2(V, F, 2); } while (0)
  if (vec) { if (f == 0) LNB_GO(true, 0); else if (f == 1) LNB_GO(true, 1); else LNB_GO(true, 2); }
  else { if (f == 0) LNB_GO(false, 0); else if (f == 1) LNB_GO(false, 1); else LNB_GO(false, 2); }
#undef LNB_GO
#undef LNB_GO2
  return ws ? sfm_fold_partials2(ws, dgamma, dbeta, D, 2 * D, nb, 1, stream) : SFM_OK;
}

// floats of the optional workspace `ws` of the three entry points below: with it, dgamma / dbeta are accumulated from one partial
// row per workgroup folded in a fixed order (bit-reproducible); without it (NULL), with fp32 atomics
extern "C" long long sfm_layernorm_bwd_ws_floats(int M, int D) {
  long long nb = ((long long)M + 7) / 8;
  if (nb > 1024) nb = 1024;
  return nb * 2 * (D > 0 ? D : 0);
}

extern "C" int sfm_layernorm_bwd_ex(const float* x, const float* gamma, const void* dy, int dy_16, const float* dres, float* dx,
                                    float* dgamma, float* dbeta, int M, int D, int ldx, int ldy, int ld, float eps, int dtype,
                                    float* ws, void* stream) {
  return layernorm_bwd_go(x, gamma, dy, dy_16, dres, dx, dgamma, dbeta, M, D, ldx, ldy, ld, eps, dtype, nullptr, 1.f, 0.f, 0u, ws,
                          stream);
}

// the same, also writing next16 [M, D] (16-bit format `dtype`, contiguous) = next_alpha * dropout(dx; next_p, next_seed) with the
// counters of sfm_ew_train mode 4 (m * D + d): the operand the next backward node of a residual chain starts from
extern "C" int sfm_layernorm_bwd_next(const float* x, const float* gamma, const void* dy, int dy_16, const float* dres, float* dx,
                                      float* dgamma, float* dbeta, int M, int D, int ldx, int ldy, int ld, float eps, int dtype,
                                      void* next16, float next_alpha, float next_p, unsigned int next_seed, float* ws,
                                      void* stream) {
  if (!next16) return SFM_ERR_ARG;
  return layernorm_bwd_go(x, gamma, dy, dy_16, dres, dx, dgamma, dbeta, M, D, ldx, ldy, ld, eps, dtype, next16, next_alpha, next_p,
                          next_seed, ws, stream);
}

extern "C" int sfm_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* dres, float* dx,
                                 float* dgamma, float* dbeta, int M, int D, int ldx, int ld, float eps, float* ws, void* stream) {
  return sfm_layernorm_bwd_ex(x, gamma, dy, 0, dres, dx, dgamma, dbeta, M, D, ldx, ld, ld, eps, SFM_DT_BF16, ws, stream);
}

// ---------------------------------------------------------------------------
// Element-wise forward/backward helpers (training mode)
//   mode 0  SWISH_FWD : out16 = drop(swish(z))                       z 16-bit [M, N]
//   mode 1  SWISH_BWD : out16 = g * drop * swish'(z)                 g fp32 or 16-bit
//   mode 2  GLU_FWD   : out16 = a * sigmoid(b)                       z = [a | b] 16-bit [M, 2N]
//   mode 3  GLU_BWD   : out16[M, 2N] = [ g*sigmoid(b) | g*a*sigmoid(b)*(1-sigmoid(b)) ]
//   mode 4  SCALE_DROP: out(fp32|16) = [z fp32 +] alpha * g * drop   (residual-branch dropout, fwd and bwd)
//   mode 5  GELU_FWD  : out(fp32|16) = gelu(z)                         z fp32 [M, N], exact erf
//   mode 6  GELU_BWD  : out(fp32|16) = g * gelu'(z)                    z fp32, g fp32 or 16-bit
//   mode 7  CPEA_BWD  : out fp32 = g * f'(.) from the OUTPUTS z = y fp32 [M, N] of the EPI_CPEA epilogue: columns < N/2 are
//                       sigmoid outputs (f' = y (1 - y)), the rest alpha * tanh outputs (f' = alpha (1 - (y / alpha)^2))
// `drop` = counter-based keep/(1-p) with element index m*N+n (p == 0 -> 1).
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void ew_train_kernel(const void* __restrict__ z, const void* __restrict__ g, void* __restrict__ out,
                                                       long long M, int N, int mode, int g_f32, int out_f32, float alpha,
                                                       float p, uint32_t seed) {
  const float inv_keep = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
  const long long total = M * N;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long m = e / N;
    const int n = (int)(e - m * N);
    const float dr = (p > 0.f) ? sfm_keep_scale(seed, (unsigned long long)e, p, inv_keep) : 1.0f;
    float gv = 1.0f;
    if (g) gv = g_f32 ? reinterpret_cast<const float*>(g)[e] : T::to_f32(reinterpret_cast<const u16*>(g)[e]);
    if (mode == 0 || mode == 1) {
      const float zv = T::to_f32(reinterpret_cast<const u16*>(z)[e]);
      const float sg = sigmoid_f(zv);
      const float r = (mode == 0) ? zv * sg * dr : gv * dr * sg * (1.0f + zv * (1.0f - sg));
      reinterpret_cast<u16*>(out)[e] = T::from_f32(r);
    } else if (mode == 2 || mode == 3) {
      const u16* zr = reinterpret_cast<const u16*>(z) + m * (2LL * N);
      const float a = T::to_f32(zr[n]), b = T::to_f32(zr[N + n]);
      const float sg = sigmoid_f(b);
      if (mode == 2) reinterpret_cast<u16*>(out)[e] = T::from_f32(a * sg);
      else {
        u16* orow = reinterpret_cast<u16*>(out) + m * (2LL * N);
        orow[n] = T::from_f32(gv * sg);
        orow[N + n] = T::from_f32(gv * a * sg * (1.0f - sg));
      }
    } else if (mode == 5 || mode == 6) {
      // exact-erf GELU on an fp32 operand (fusion MLP / mask heads of agents/msa.py:42-71): forward, or g * gelu'(z)
      const float zv = reinterpret_cast<const float*>(z)[e];
      const float cdf = 0.5f * (1.0f + erff(zv * 0.70710678118654752440f));
      const float r = (mode == 5) ? zv * cdf : gv * (cdf + zv * 0.39894228040143267794f * __expf(-0.5f * zv * zv));
      if (out_f32) reinterpret_cast<float*>(out)[e] = r;
      else reinterpret_cast<u16*>(out)[e] = T::from_f32(r);
    } else if (mode == 7) {
      const float y = reinterpret_cast<const float*>(z)[e];
      const float t = y / alpha;
      reinterpret_cast<float*>(out)[e] = gv * ((n < N / 2) ? y * (1.0f - y) : alpha * (1.0f - t * t));
    } else {
      float r = alpha * gv * dr;
      if (z) r += reinterpret_cast<const float*>(z)[e];          // mode 4: optional fp32 residual
      if (out_f32) reinterpret_cast<float*>(out)[e] = r;
      else reinterpret_cast<u16*>(out)[e] = T::from_f32(r);
    }
  }
}

// 8 consecutive elements per thread (N % 8 == 0, 16-byte aligned rows): 16-byte loads/stores of the 16-bit operands,
// 2 x 16 bytes of the fp32 ones.  Same arithmetic and the same dropout counters as the scalar kernel above.
template <class T, int MODE>
__global__ __launch_bounds__(256) void ew_train_vec_kernel(const void* __restrict__ z, const void* __restrict__ g,
                                                           void* __restrict__ out, long long M, int N, int g_f32, int out_f32,
                                                           float alpha, float p, uint32_t seed) {
  const float inv_keep = (p > 0.f) ? 1.0f / (1.0f - p) : 1.0f;
  const long long chunks = M * N / 8;
  const int cpr = N / 8;                                          // chunks per row
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (long long)gridDim.x * 256) {
    const long long e0 = c * 8;
    float gv[8], dr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      gv[i] = 1.0f;
      dr[i] = 1.0f;
    }
    if (p > 0.f && MODE != 2 && MODE != 3 && MODE != 5 && MODE != 6) sfm_keep_scale8(seed, (unsigned long long)e0, p, inv_keep, dr);
    if (g) {
      if (g_f32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g) + e0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g) + e0 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { gv[i] = a[i]; gv[4 + i] = b[i]; }
      } else {
        const u32x4 a = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(g) + e0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { gv[2 * i] = T::to_f32((u16)(a[i] & 0xffffu)); gv[2 * i + 1] = T::to_f32((u16)(a[i] >> 16)); }
      }
    }
    if (MODE == 0 || MODE == 1) {
      const u32x4 zz = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(z) + e0);
      float r[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float zv = T::to_f32((u16)((i & 1) ? (zz[i >> 1] >> 16) : (zz[i >> 1] & 0xffffu)));
        const float sg = sigmoid_f(zv);
        r[i] = (MODE == 0) ? zv * sg * dr[i] : gv[i] * dr[i] * sg * (1.0f + zv * (1.0f - sg));
      }
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (uint32_t)T::from_f32(r[2 * i]) | ((uint32_t)T::from_f32(r[2 * i + 1]) << 16);
      *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(out) + e0) = o;
    } else if (MODE == 2 || MODE == 3) {
      const long long m = c / cpr;
      const int n0 = (int)(c - m * cpr) * 8;
      const u16* zr = reinterpret_cast<const u16*>(z) + m * (2LL * N);
      const u32x4 za = *reinterpret_cast<const u32x4*>(zr + n0);
      const u32x4 zb = *reinterpret_cast<const u32x4*>(zr + N + n0);
      float r0[8], r1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float a = T::to_f32((u16)((i & 1) ? (za[i >> 1] >> 16) : (za[i >> 1] & 0xffffu)));
        const float b = T::to_f32((u16)((i & 1) ? (zb[i >> 1] >> 16) : (zb[i >> 1] & 0xffffu)));
        const float sg = sigmoid_f(b);
        if (MODE == 2) r0[i] = a * sg;
        else { r0[i] = gv[i] * sg; r1[i] = gv[i] * a * sg * (1.0f - sg); }
      }
      u32x4 o0, o1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o0[i] = (uint32_t)T::from_f32(r0[2 * i]) | ((uint32_t)T::from_f32(r0[2 * i + 1]) << 16);
        if (MODE == 3) o1[i] = (uint32_t)T::from_f32(r1[2 * i]) | ((uint32_t)T::from_f32(r1[2 * i + 1]) << 16);
      }
      if (MODE == 2) *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(out) + e0) = o0;
      else {
        u16* orow = reinterpret_cast<u16*>(out) + m * (2LL * N);
        *reinterpret_cast<u32x4*>(orow + n0) = o0;
        *reinterpret_cast<u32x4*>(orow + N + n0) = o1;
      }
    } else {
      float r[8];
      if (MODE == 5 || MODE == 6) {                                 // exact-erf GELU on an fp32 operand: forward, or g * gelu'(z)
        const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(z) + e0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(z) + e0 + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float zv = i < 4 ? a[i & 3] : b[i & 3];
          const float cdf = 0.5f * (1.0f + erff(zv * 0.70710678118654752440f));
          r[i] = (MODE == 5) ? zv * cdf : gv[i] * (cdf + zv * 0.39894228040143267794f * __expf(-0.5f * zv * zv));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = alpha * gv[i] * dr[i];
        if (z) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(z) + e0);
          const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(z) + e0 + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { r[i] += a[i]; r[4 + i] += b[i]; }
        }
      }
      if (out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + e0) = f32x4{r[0], r[1], r[2], r[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + e0 + 4) = f32x4{r[4], r[5], r[6], r[7]};
      } else {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (uint32_t)T::from_f32(r[2 * i]) | ((uint32_t)T::from_f32(r[2 * i + 1]) << 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(out) + e0) = o;
      }
    }
  }
}

template <class T>
static int ew_train_launch(const void* z, const void* g, void* out, long long M, int N, int mode, int g_f32, int out_f32,
                           float alpha, float p, unsigned int seed, hipStream_t st) {
  const bool aligned = (mode <= 6) && (N % 8 == 0) && (((uintptr_t)z | (uintptr_t)g | (uintptr_t)out) % 16 == 0);
  if (!aligned) {
    long long nb = (M * N + 255) / 256;
    if (nb > 16384) nb = 16384;
    SFM_LAUNCH((ew_train_kernel<T>), dim3((unsigned)nb), dim3(256), 0, st, z, g, out, M, N, mode, g_f32, out_f32, alpha, p, seed);
    return SFM_OK;
  }
  long long nb = (M * N / 8 + 255) / 256;
  if (nb > 16384) nb = 16384;
  const dim3 grid((unsigned)nb), block(256);
  switch (mode) {
    case 0: SFM_LAUNCH((ew_train_vec_kernel<T, 0>), grid, block, 0, st, z, g, out, M, N, g_f32, out_f32, alpha, p, seed); break;
    case 1: SFM_LAUNCH((ew_train_vec_kernel<T, 1>), grid, block, 0, st, z, g, out, M, N, g_f32, out_f32, alpha, p, seed); break;
    case 2: SFM_LAUNCH((ew_train_vec_kernel<T, 2>), grid, block, 0, st, z, g, out, M, N, g_f32, out_f32, alpha, p, seed); break;
    case 3: SFM_LAUNCH((ew_train_vec_kernel<T, 3>), grid, block, 0, st, z, g, out, M, N, g_f32, out_f32, alpha, p, seed); break;
    case 5: SFM_LAUNCH((ew_train_vec_kernel<T, 5>), grid, block, 0, st, z, g, out, M, N, g_f32, out_f32, alpha, p, seed); break;
    case 6: SFM_LAUNCH((ew_train_vec_kernel<T, 6>), grid, block, 0, st, z, g, out, M, N, g_f32, out_f32, alpha, p, seed); break;
    default: SFM_LAUNCH((ew_train_vec_kernel<T, 4>), grid, block, 0, st, z, g, out, M, N, g_f32, out_f32, alpha, p, seed); break;
  }
  return SFM_OK;
}

extern "C" int sfm_ew_train(const void* z, const void* g, void* out, long long M, int N, int mode, int g_f32, int out_f32,
                            float alpha, float p, unsigned int seed, int dtype, void* stream) {
  if (!out || ((mode <= 3 || mode >= 5) && !z) || ((mode == 1 || mode == 3 || mode == 4 || mode == 6 || mode == 7) && !g))
    return SFM_ERR_ARG;
  if (M <= 0 || N <= 0 || mode < 0 || mode > 7 || p < 0.f || p >= 1.f) return SFM_ERR_SHAPE;
  if (mode == 7 && (!g_f32 || !out_f32)) return SFM_ERR_SHAPE;
  if (dtype == SFM_DT_F16) return ew_train_launch<F16>(z, g, out, M, N, mode, g_f32, out_f32, alpha, p, seed, (hipStream_t)stream);
  return ew_train_launch<BF16>(z, g, out, M, N, mode, g_f32, out_f32, alpha, p, seed, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// BatchNorm1d training (models/conformer.py:95,118): statistics over all B*T rows per channel.
//   col_stats: S[c] = { sum y, sum y^2 }  or, with aux (backward), { sum dy, sum dy * xhat }
//   (accumulated with atomics; S must be zero-filled; y fp32 [M, C])
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ y, const float* __restrict__ aux,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        float* __restrict__ S, int M, int C, int rows_per_block,
                                                        float* __restrict__ ws) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f;
  if (!aux) {
    for (int m = m0; m < m1; ++m) {
      const float v = y[(long long)m * C + c];
      s0 += v;
      s1 += v * v;
    }
  } else {                                               // y = dy, aux = pre-normalisation activations
    const float mu = mean[c], rs = rstd[c];
    for (int m = m0; m < m1; ++m) {
      const float d = y[(long long)m * C + c];
      s0 += d;
      s1 += d * (aux[(long long)m * C + c] - mu) * rs;
    }
  }
  if (ws) {
    ws[(long long)blockIdx.y * 2 * C + 2 * c] = s0;
    ws[(long long)blockIdx.y * 2 * C + 2 * c + 1] = s1;
  } else {
    atomicAdd(&S[2 * c], s0);
    atomicAdd(&S[2 * c + 1], s1);
  }
}

// Vector form (C % 4 == 0, C / 4 divides 256): a thread owns 4 neighbouring columns (16-byte loads) and one of 1024 / C row lanes
// of a 128-row block; the row lanes meet in LDS, one atomic pair per column and block.  (The scalar kernel above walks a column with
// 4-byte loads, 256 dependent trips: 1.5 TB/s on [205 056, 256].)
__global__ __launch_bounds__(256) void col_stats_vec_kernel(const float* __restrict__ y, const float* __restrict__ aux,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ S, int M, int C, int rows_per_block,
                                                            float* __restrict__ ws) {
  __shared__ float red[2048];                            // [row lane][2][C]
  const int nv = C >> 2, rl = 256 / nv;
  const int tv = threadIdx.x % nv, tr = threadIdx.x / nv, c0 = tv * 4;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  if (!aux) {
#pragma unroll 4
    for (int m = m0 + tr; m < m1; m += rl) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(y + (long long)m * C + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] += v[j]; s1[j] = fmaf(v[j], v[j], s1[j]); }
    }
  } else {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0);
#pragma unroll 4
    for (int m = m0 + tr; m < m1; m += rl) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(y + (long long)m * C + c0);
      const f32x4 x = *reinterpret_cast<const f32x4*>(aux + (long long)m * C + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] += d[j]; s1[j] = fmaf(d[j], x[j] - mu[j], s1[j]); }
    }
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[j] *= rs[j];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[tr * 2 * C + c0 + j] = s0[j];
    red[tr * 2 * C + C + c0 + j] = s1[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float a = 0.f;
    for (int r = 0; r < rl; ++r) a += red[r * 2 * C + i];
    const int k = i / C, c = i - k * C;
    if (ws) ws[(long long)blockIdx.x * 2 * C + 2 * c + k] = a;
    else atomicAdd(&S[2 * c + k], a);
  }
}

// floats of the optional workspace of sfm_col_stats and of pass 0 of sfm_bn_swish_bwd (one partial row [C][2] per row block,
// folded in block order: bit-reproducible statistics; NULL = fp32 atomics)
extern "C" long long sfm_col_stats_ws_floats(int M, int C) { return (long long)((M + 127) / 128 + 1) * 2 * (C > 0 ? C : 0); }

extern "C" int sfm_col_stats(const float* y, const float* aux, const float* mean, const float* rstd, float* S, int M, int C,
                             float* ws, void* stream) {
  if (!y || !S || (aux && (!mean || !rstd))) return SFM_ERR_ARG;
  if (M <= 0 || C <= 0) return SFM_ERR_SHAPE;
  if (ws && (((uintptr_t)ws) % 16) != 0) return SFM_ERR_ARG;
  if (C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 &&
      (((uintptr_t)y | (uintptr_t)aux | (uintptr_t)mean | (uintptr_t)rstd) % 16) == 0) {
    const int rpbv = 128;
    SFM_LAUNCH(col_stats_vec_kernel, dim3((M + rpbv - 1) / rpbv), dim3(256), 0, (hipStream_t)stream, y, aux, mean, rstd, S, M, C, rpbv,
               ws);
    return ws ? sfm_fold_partials(ws, S, 1, 2 * C, 2 * C, (M + rpbv - 1) / rpbv, 1, stream) : SFM_OK;
  }
  const int rpb = 256;
  SFM_LAUNCH(col_stats_kernel, dim3((C + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, y, aux, mean,
             rstd, S, M, C, rpb, ws);
  return ws ? sfm_fold_partials(ws, S, 1, 2 * C, 2 * C, (M + rpb - 1) / rpb, 1, stream) : SFM_OK;
}

// Gradient fan-in of a tensor whose first Cb columns also fed a second consumer: out[m, c] = a[m, c] + (c < Cb ? b[m, c] : 0)
// (fp32, row strides lda / ldb / ldo; training: the pooled latents feed the mask-synthesis fusion whole and the BiLSTM by their
// real half).  One pass instead of a zero-fill, a strided copy and an add.
__global__ __launch_bounds__(256) void add_cols_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                       long long M, int C, int Cb, long long lda, long long ldb, long long ldo) {
  const int c4 = C >> 2;
  const long long total = M * c4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long m = e / c4;
    const int c = (int)(e - m * c4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a + m * lda + c);
    if (c < Cb) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(b + m * ldb + c);
      v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
    }
    *reinterpret_cast<f32x4*>(out + m * ldo + c) = v;
  }
}

extern "C" int sfm_add_cols(const float* a, const float* b, float* out, long long M, int C, int Cb, long long lda, long long ldb,
                            long long ldo, void* stream) {
  if (!a || !b || !out) return SFM_ERR_ARG;
  if (M <= 0 || C <= 0 || Cb < 0 || Cb > C || (C % 4) || (Cb % 4) || (lda % 4) || (ldb % 4) || (ldo % 4) || lda < C || ldb < Cb ||
      ldo < C || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16) != 0)
    return SFM_ERR_SHAPE;
  long long nb = (M * (C >> 2) + 255) / 256;
  if (nb > 16384) nb = 16384;
  SFM_LAUNCH(add_cols_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a, b, out, M, C, Cb, lda, ldb, ldo);
  return SFM_OK;
}

// BiLSTM weight gradient operand: the previous output of each chain as 16-bit rows,
//   out[b, t, 0:H) = h[b, t - 1, 0:H) (forward chain, 0 at t = 0),  out[b, t, H:2H) = h[b, t + 1, H:2H) (reverse chain, 0 at t = T - 1)
// h fp32 [B, T, 2H] -> out [B*T, 2H] in one pass (was: zero-fill + two strided copies + convert).
template <class T>
__global__ __launch_bounds__(256) void lstm_hprev16_kernel(const float* __restrict__ h, u16* __restrict__ out, int Tn, int H,
                                                           long long total4) {
  const int W = 2 * H, w4 = W >> 2;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long long)gridDim.x * 256) {
    const long long row = e / w4;
    const int c = (int)(e - row * w4) * 4;
    const int t = (int)(row % Tn);
    const long long src = (c < H) ? row - 1 : row + 1;
    const bool ok = (c < H) ? (t > 0) : (t < Tn - 1);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f32x4*>(h + src * W + c);
    u32x2 pk = {T::pack(v[0], v[1]), T::pack(v[2], v[3])};
    *reinterpret_cast<u32x2*>(out + row * W + c) = pk;
  }
}

extern "C" int sfm_lstm_hprev16(const float* h, void* out, int B, int Tn, int H, int dtype, void* stream) {
  if (!h || !out) return SFM_ERR_ARG;
  if (B <= 0 || Tn <= 0 || H <= 0 || (H % 4) || (((uintptr_t)h) % 16) != 0 || (((uintptr_t)out) % 8) != 0) return SFM_ERR_SHAPE;
  const long long total4 = (long long)B * Tn * (2 * H / 4);
  long long nb = (total4 + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (dtype == SFM_DT_F16) SFM_LAUNCH((lstm_hprev16_kernel<F16>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, h, (u16*)out, Tn, H, total4);
  else SFM_LAUNCH((lstm_hprev16_kernel<BF16>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, h, (u16*)out, Tn, H, total4);
  return SFM_OK;
}

// BatchNorm1d training statistics finalised on the device (models/conformer.py ConvolutionModule's nn.BatchNorm1d, train() mode):
// S[c] = {sum y, sum y^2} over the M rows ->  mean, rstd = 1/sqrt(biased var + eps), the folded affine
// sc = gamma rstd, sh = beta - mean sc, and the running statistics (unbiased variance, momentum) updated in place.
// eval_mode: mean / var come from run_mean / run_var and nothing is updated.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ S, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ sc, float* __restrict__ sh,
                                                          int C, float M, float eps, float momentum, int eval_mode) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float mu, var;
  if (eval_mode) {
    mu = run_mean[c];
    var = run_var[c];
  } else {
    mu = S[2 * c] / M;
    var = fmaxf(S[2 * c + 1] / M - mu * mu, 0.0f);
    if (run_mean) {
      run_mean[c] = run_mean[c] * (1.0f - momentum) + momentum * mu;
      run_var[c] = run_var[c] * (1.0f - momentum) + momentum * (var * (M / fmaxf(M - 1.0f, 1.0f)));
    }
  }
  const float rs = rsqrtf(var + eps);
  mean[c] = mu;
  rstd[c] = rs;
  const float a = gamma[c] * rs;
  sc[c] = a;
  sh[c] = beta[c] - mu * a;
}

extern "C" int sfm_bn_finalize(const float* S, const float* gamma, const float* beta, float* run_mean, float* run_var, float* mean,
                               float* rstd, float* sc, float* sh, int C, long long M, float eps, float momentum, int eval_mode,
                               void* stream) {
  if (!gamma || !beta || !mean || !rstd || !sc || !sh) return SFM_ERR_ARG;
  if (eval_mode ? (!run_mean || !run_var) : (!S || ((run_mean == nullptr) != (run_var == nullptr)))) return SFM_ERR_ARG;
  if (C <= 0 || M <= 0) return SFM_ERR_SHAPE;
  SFM_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, S, gamma, beta, run_mean, run_var, mean,
             rstd, sc, sh, C, (float)M, eps, momentum, eval_mode);
  return SFM_OK;
}

// BatchNorm backward applied through the following Swish:  given g = dL/d(swish out), y (pre-BN), batch mean/rstd,
// gamma/beta:  t = (y-mu)*rs*gamma+beta ; dt = g * swish'(t) ;
//   pass 0 (stats): S[c] = { sum dt, sum dt * xhat }       pass 1 (apply): dy = gamma*rs*(dt - S0/M - xhat*S1/M)
template <class T>
__global__ __launch_bounds__(256) void bn_swish_bwd_kernel(const void* __restrict__ g, const float* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ S, float* __restrict__ dy, int M, int C,
                                                           int g_f32, int pass, int rows_per_block, float* __restrict__ ws) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
  const float invM = 1.0f / (float)M;
  float s0 = 0.f, s1 = 0.f;
  const float a0 = pass ? S[2 * c] * invM : 0.f, a1 = pass ? S[2 * c + 1] * invM : 0.f;
  for (int m = m0; m < m1; ++m) {
    const long long e = (long long)m * C + c;
    const float xh = (y[e] - mu) * rs;
    const float t = xh * ga + be;
    const float sg = sigmoid_f(t);
    const float gv = g_f32 ? reinterpret_cast<const float*>(g)[e] : T::to_f32(reinterpret_cast<const u16*>(g)[e]);
    const float dt = gv * sg * (1.0f + t * (1.0f - sg));
    if (!pass) { s0 += dt; s1 += dt * xh; }
    else dy[e] = ga * rs * (dt - a0 - xh * a1);
  }
  if (!pass) {
    if (ws) {
      ws[(long long)blockIdx.y * 2 * C + 2 * c] = s0;
      ws[(long long)blockIdx.y * 2 * C + 2 * c + 1] = s1;
    } else {
      atomicAdd(&S[2 * c], s0);
      atomicAdd(&S[2 * c + 1], s1);
    }
  }
}

// Vector form of the two passes (same thread map as col_stats_vec_kernel: 4 columns x one row lane of a 128-row block)
template <class T, int PASS>
__global__ __launch_bounds__(256) void bn_swish_bwd_vec_kernel(const void* __restrict__ g, const float* __restrict__ y,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ S, float* __restrict__ dy, int M, int C,
                                                               int g_f32, int rows_per_block, float* __restrict__ ws) {
  __shared__ float red[PASS ? 1 : 2048];
  const int nv = C >> 2, rl = 256 / nv;
  const int tv = threadIdx.x % nv, tr = threadIdx.x / nv, c0 = tv * 4;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c0), rs = *reinterpret_cast<const f32x4*>(rstd + c0);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
  const float invM = 1.0f / (float)M;
  float a0[4], a1[4], s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a0[j] = PASS ? S[2 * (c0 + j)] * invM : 0.f;
    a1[j] = PASS ? S[2 * (c0 + j) + 1] * invM : 0.f;
  }
#pragma unroll 2
  for (int m = m0 + tr; m < m1; m += rl) {
    const long long e = (long long)m * C + c0;
    const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e);
    f32x4 gv;
    if (g_f32) gv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g) + e);
    else {
      const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const u16*>(g) + e);
      gv = f32x4{T::to_f32((u16)(w[0] & 0xffffu)), T::to_f32((u16)(w[0] >> 16)), T::to_f32((u16)(w[1] & 0xffffu)), T::to_f32((u16)(w[1] >> 16))};
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (yv[j] - mu[j]) * rs[j];
      const float t = fmaf(xh, ga[j], be[j]);
      const float sg = sigmoid_f(t);
      const float dt = gv[j] * sg * (1.0f + t * (1.0f - sg));
      if (!PASS) { s0[j] += dt; s1[j] = fmaf(dt, xh, s1[j]); }
      else o[j] = ga[j] * rs[j] * (dt - a0[j] - xh * a1[j]);
    }
    if (PASS) *reinterpret_cast<f32x4*>(dy + e) = o;
  }
  if (!PASS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[tr * 2 * C + c0 + j] = s0[j];
      red[tr * 2 * C + C + c0 + j] = s1[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
      float a = 0.f;
      for (int r = 0; r < rl; ++r) a += red[r * 2 * C + i];
      const int k = i / C, c = i - k * C;
      if (ws) ws[(long long)blockIdx.x * 2 * C + 2 * c + k] = a;
      else atomicAdd(&S[2 * c + k], a);
    }
  }
}

extern "C" int sfm_bn_swish_bwd(const void* g, const float* y, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, float* S, float* dy, int M, int C, int g_f32, int pass, int dtype,
                                float* ws, void* stream) {
  if (!g || !y || !mean || !rstd || !gamma || !beta || !S || (pass && !dy)) return SFM_ERR_ARG;
  if (M <= 0 || C <= 0) return SFM_ERR_SHAPE;
  if (pass) ws = nullptr;                                    // (only pass 0 reduces)
  if (ws && (((uintptr_t)ws) % 16) != 0) return SFM_ERR_ARG;
  if (C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 &&
      (((uintptr_t)y | (uintptr_t)mean | (uintptr_t)rstd | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)dy) % 16) == 0 &&
      ((uintptr_t)g % (g_f32 ? 16 : 8)) == 0) {
    const int rpbv = 128;
    dim3 gridv((M + rpbv - 1) / rpbv), blockv(256);
    hipStream_t st = (hipStream_t)stream;
#define BNV_GO(T, P) SFM_LAUNCH((bn_swish_bwd_vec_kernel<T, P>), gridv, blockv, 0, st, g, y, mean, rstd, gamma, beta, S, dy, M, C, g_f32, rpbv, ws)
    if (dtype == SFM_DT_F16) { if (pass) BNV_GO(F16, 1); else BNV_GO(F16, 0); }
    else { if (pass) BNV_GO(BF16, 1); else BNV_GO(BF16, 0); }
#undef BNV_GO
    return ws ? sfm_fold_partials(ws, S, 1, 2 * C, 2 * C, (int)gridv.x, 1, stream) : SFM_OK;
  }
  const int rpb = 256;
  dim3 grid((C + 255) / 256, (M + rpb - 1) / rpb), block(256);
  if (dtype == SFM_DT_F16)
    SFM_LAUNCH((bn_swish_bwd_kernel<F16>), grid, block, 0, (hipStream_t)stream, g, y, mean, rstd, gamma, beta, S, dy, M, C,
               g_f32, pass, rpb, ws);
  else
    SFM_LAUNCH((bn_swish_bwd_kernel<BF16>), grid, block, 0, (hipStream_t)stream, g, y, mean, rstd, gamma, beta, S, dy, M, C,
               g_f32, pass, rpb, ws);
  return ws ? sfm_fold_partials(ws, S, 1, 2 * C, 2 * C, (int)grid.y, 1, stream) : SFM_OK;
}

// ---------------------------------------------------------------------------
// Depthwise conv weight gradient: dW[c][k] += sum_{b,t} dY[b,t,c] * X[b, t+k-pad, c]   (X 16-bit, dY fp32)
// one thread per channel, KS accumulators, a block covers a span of frames of one utterance
// ---------------------------------------------------------------------------
template <class T, int KS>
__global__ __launch_bounds__(128) void dwconv_wgrad_kernel(const u16* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ part, int Tlen, int C, int span) {
  // two adjacent channels per thread (one 4-byte load of x, one 8-byte load of dy per frame); the KS-frame window of x
  // lives in registers and is indexed with compile-time (j + k) % KS, so sliding it costs no moves
  const int c = (blockIdx.x * 128 + threadIdx.x) * 2;
  if (c >= C) return;
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * span, t1 = min(Tlen, t0 + span);
  constexpr int pad = (KS - 1) / 2;
  const u16* xb = x + (long long)b * Tlen * C + c;
  const float* gb = dy + (long long)b * Tlen * C + c;
  float acc0[KS], acc1[KS], w0[KS], w1[KS];
  auto load_x = [&](int tt, float& a, float& bb) {
    if (tt >= 0 && tt < Tlen) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(xb + (long long)tt * C);
      a = T::to_f32((u16)(v & 0xffffu));
      bb = T::to_f32((u16)(v >> 16));
    } else {
      a = 0.f;
      bb = 0.f;
    }
  };
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    acc0[k] = 0.f;
    acc1[k] = 0.f;
    load_x(t0 + k - pad, w0[k], w1[k]);
  }
  float sb0 = 0.f, sb1 = 0.f;
  auto load_g = [&](int t, float& a, float& bb) {
    a = 0.f;
    bb = 0.f;
    if (t < t1) {
      const f32x2 g = *reinterpret_cast<const f32x2*>(gb + (long long)t * C);
      a = g[0];
      bb = g[1];
    }
  };
  // software pipeline of depth 1: the loads of frame t+1 are issued before the 2 x KS FMAs of frame t; the
  // sched_barrier keeps the scheduler from hoisting a whole chunk's loads (that spilled the register window)
  float gn0, gn1, xn0, xn1;
  load_g(t0, gn0, gn1);
  load_x(t0 + pad + 1, xn0, xn1);
  for (int tb = t0; tb < t1; tb += KS) {
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int t = tb + j;
      const float g0 = gn0, g1 = gn1, xa = xn0, xb = xn1;
      load_g(t + 1, gn0, gn1);
      load_x(t + pad + 2, xn0, xn1);
      sb0 += g0;
      sb1 += g1;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        acc0[k] += g0 * w0[(j + k) % KS];
        acc1[k] += g1 * w1[(j + k) % KS];
      }
      w0[j] = xa;                                            // slot j held x[t - pad]: no longer needed
      w1[j] = xb;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // partial sums of this (utterance, span): every workgroup adding into the same C*KS addresses serialises the
  // memory-side atomics (MI355X_MICROARCH.md, "Global float atomics": one row shared by all = 14x slower), so the
  // partials go to a scratch [nparts][KS + 1][C] (coalesced over channels) and dwconv_wgrad_reduce sums them
  float* pp = part + ((long long)(blockIdx.z * gridDim.y + blockIdx.y) * (KS + 1)) * C + c;
#pragma unroll
  for (int k = 0; k < KS; ++k) *reinterpret_cast<f32x2*>(pp + (long long)k * C) = f32x2{acc0[k], acc1[k]};
  *reinterpret_cast<f32x2*>(pp + (long long)KS * C) = f32x2{sb0, sb1};
}

// KS = 31, C % 256 == 0 (the Conformer's depthwise conv): thread = one channel of a 256-channel block; the dy rows and the incoming
// x rows of a 31-frame chunk go through LDS, the NEXT chunk's 12 x 16-byte loads per thread are in flight while the current
// one is multiplied (the kernel above prefetches one frame = 12 bytes per thread ahead and waits out a full memory round trip per
// frame: 0.33 ms per launch at [205 056, 256] against 0.05 ms of FMA issue; tools/train_profile.py).  Same partial layout.
template <class T>
__global__ __launch_bounds__(256) void dwconv_wgrad31_lds_kernel(const u16* __restrict__ x, const float* __restrict__ dy,
                                                                 float* __restrict__ part, int Tlen, int C, int span) {
  constexpr int KS = 31, pad = 15;
  __shared__ __attribute__((aligned(16))) float gL[KS][256];
  __shared__ __attribute__((aligned(16))) u16 xL[KS][256];
  const int tid = threadIdx.x;
  const int cb = blockIdx.x * 256, c = cb + tid;
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * span, t1 = min(Tlen, t0 + span);
  const u16* xb = x + (long long)b * Tlen * C + cb;
  const float* gb = dy + (long long)b * Tlen * C + cb;
  float acc[KS], w[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    acc[k] = 0.f;
    const int tt = t0 + k - pad;
    w[k] = (tt >= 0 && tt < Tlen) ? T::to_f32(xb[(long long)tt * C + tid]) : 0.f;
  }
  float sb = 0.f;
  f32x4 gr[8];
  u32x4 xr[4];
  auto fetch = [&](int tb) {       // dy rows tb .. tb + 30 (zero from t1 on), x rows tb + pad + 1 .. tb + pad + 31 (zero outside the utterance)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int p = tid + 256 * q, row = p >> 6, col = (p & 63) * 4, t = tb + row;
      gr[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < KS && t < t1) gr[q] = *reinterpret_cast<const f32x4*>(gb + (long long)t * C + col);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = tid + 256 * q, row = p >> 5, col = (p & 31) * 8, t = tb + pad + 1 + row;
      xr[q] = u32x4{0u, 0u, 0u, 0u};
      if (row < KS && t >= 0 && t < Tlen) xr[q] = *reinterpret_cast<const u32x4*>(xb + (long long)t * C + col);
    }
  };
  fetch(t0);
  for (int tb = t0; tb < t1; tb += KS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int p = tid + 256 * q, row = p >> 6, col = (p & 63) * 4;
      if (row < KS) *reinterpret_cast<f32x4*>(&gL[row][col]) = gr[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = tid + 256 * q, row = p >> 5, col = (p & 31) * 8;
      if (row < KS) *reinterpret_cast<u32x4*>(&xL[row][col]) = xr[q];
    }
    __syncthreads();
    if (tb + KS < t1) fetch(tb + KS);
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const float g = gL[j][tid];
      const float xa = T::to_f32(xL[j][tid]);
      sb += g;
#pragma unroll
      for (int k = 0; k < KS; ++k) acc[k] = fmaf(g, w[(j + k) % KS], acc[k]);
      w[j] = xa;                                             // slot j held x[t - pad]: no longer needed
    }
    __syncthreads();
  }
  float* pp = part + ((long long)(blockIdx.z * gridDim.y + blockIdx.y) * (KS + 1)) * C + c;
#pragma unroll
  for (int k = 0; k < KS; ++k) pp[(long long)k * C] = acc[k];
  pp[(long long)KS * C] = sb;
}

// dw[c][k] += sum_parts part[p][k][c] ; db[c] += sum_parts part[p][KS][c], in a FIXED order (bit-reproducible): chunks of
// `parts_per_block` consecutive partials (partial p at part + p * stride) are summed by one thread per element; final == 0: the chunk
// sum replaces the chunk's first partial (in place), final == 1 (one chunk left): it is added to dw / db
__global__ __launch_bounds__(256) void dwconv_wgrad_reduce_kernel(float* __restrict__ part, float* __restrict__ dw,
                                                                  float* __restrict__ db, int nparts, int C, int KS,
                                                                  int parts_per_block, long long stride, int final) {
  const int e = blockIdx.x * 256 + threadIdx.x;                 // e = k * C + c
  if (e >= (KS + 1) * C) return;
  const int p0 = blockIdx.y * parts_per_block, p1 = min(nparts, p0 + parts_per_block);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = p0;
  for (; p + 4 <= p1; p += 4) {
    s0 += part[(long long)(p + 0) * stride + e];
    s1 += part[(long long)(p + 1) * stride + e];
    s2 += part[(long long)(p + 2) * stride + e];
    s3 += part[(long long)(p + 3) * stride + e];
  }
  for (; p < p1; ++p) s0 += part[(long long)p * stride + e];
  const float s = (s0 + s1) + (s2 + s3);
  if (!final) {
    part[(long long)p0 * stride + e] = s;
    return;
  }
  const int k = e / C, c = e - k * C;
  if (k < KS) dw[c * KS + k] += s;
  else db[c] += s;
}

extern "C" long long sfm_dwconv_wgrad_scratch_floats(int B, int T, int C, int KS) {
  int nspan = (2048 + B - 1) / B;
  if (nspan > (T + 31) / 32) nspan = (T + 31) / 32;
  if (nspan < 1) nspan = 1;
  const int span = (T + nspan - 1) / nspan;
  const long long parts = (long long)B * ((T + span - 1) / span);
  return parts * (KS + 1) * C;
}

extern "C" int sfm_dwconv_wgrad(const void* x, const float* dy, float* dw, float* db, float* scratch, int B, int T, int C,
                                int KS, int dtype, void* stream) {
  if (!x || !dy || !dw || !db || !scratch) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0 || C <= 0 || (C & 1) || (KS != 31 && KS != 7)) return SFM_ERR_SHAPE;
  int nspan = (2048 + B - 1) / B;                              // enough workgroups to fill the chip
  if (nspan > (T + 31) / 32) nspan = (T + 31) / 32;
  if (nspan < 1) nspan = 1;
  const int span = (T + nspan - 1) / nspan;
  const int ny = (T + span - 1) / span;
  dim3 grid((C + 255) / 256, ny, B), block(128);
  hipStream_t st = (hipStream_t)stream;
  if (KS == 31 && C % 256 == 0 && (((uintptr_t)x | (uintptr_t)dy) % 16) == 0) {
    if (dtype == SFM_DT_F16) SFM_LAUNCH((dwconv_wgrad31_lds_kernel<F16>), grid, dim3(256), 0, st, (const u16*)x, dy, scratch, T, C, span);
    else SFM_LAUNCH((dwconv_wgrad31_lds_kernel<BF16>), grid, dim3(256), 0, st, (const u16*)x, dy, scratch, T, C, span);
  } else {
#define GO(TT, KK) SFM_LAUNCH((dwconv_wgrad_kernel<TT, KK>), grid, block, 0, st, (const u16*)x, dy, scratch, T, C, span)
    if (dtype == SFM_DT_F16) { if (KS == 31) GO(F16, 31); else GO(F16, 7); }
    else { if (KS == 31) GO(BF16, 31); else GO(BF16, 7); }
#undef GO
  }
  int nparts = B * ny;
  const int ppb = 32;
  long long stride = (long long)(KS + 1) * C;
  while (true) {
    const int final = nparts <= ppb;
    dim3 g2(((KS + 1) * C + 255) / 256, final ? 1 : (nparts + ppb - 1) / ppb);
    SFM_LAUNCH(dwconv_wgrad_reduce_kernel, g2, dim3(256), 0, st, scratch, dw, db, nparts, C, KS, ppb, stride, final);
    if (final) break;
    nparts = (int)g2.y;
    stride *= ppb;
  }
  return SFM_OK;
}
