// Self-attention core softmax(Q K^T / sqrt(hd)) V  (models/conformer.py:69 ->
// nn.MultiheadAttention, no mask, no positional term).
//
// attn_fwd_hd64: flash-style, never materialises [T, T].  Block = 4 waves =
// 128 query rows of one (batch, head); each wave owns 32 query rows.  Per
// 64-key tile (K, V staged in LDS, zero-filled past T):
//   S^T = K Q^T   (32x32x16 MFMA; A = K rows from LDS, B = Q held in registers)
//         -> each lane holds 16 keys of ONE query row (col = lane&31), so the
//            row max / row sum are in-lane + one exchange with lane^32
//   online softmax in fp32, exp2 domain.  Two extra MFMAs per key tile take VALU work off the
//   softmax: an augmented k-step (K side [1,1,0..], Q side [-m_hi,-m_lo,0..]) makes the matrix
//   core deliver s*c - m_run directly (the scale c is folded into Q), and a ones-row on the V^T
//   side accumulates the row sums l; per score only max / exp2 / convert remain on the VALU.
//   O^T += V^T P^T: the S^T accumulator, converted to 16-bit, IS the B operand
//            (k order 16s + 8(j>>2) + 4h + (j&3)); the matching A operand V^T
//            comes from ds_read_b64_tr_b16 on the row-major V tile.
// Epilogue: O^T / l -> LDS transpose -> 16-byte coalesced row stores.
//
// attn_fwd_generic: small-shape path (any head_dim <= 256, e.g. the reference
// test config d_model 64 / 4 heads = 16), one wave per query row, fp32 VALU.
#include "sfm_common.h"

#define KS_ROW 72    // u16 elements: 144-byte K rows  (ds_read_b128 conflict-free)
#define VS_ROW 96    // u16 elements: 192-byte V rows  (4 rows x 64 B tile the 256-B bank row for tr reads)
#define OS_ROW 72
#define DEFER_THR 4.0f

// same counter-based keep function as attention_bwd.hip / backward.hip
__device__ __forceinline__ float attn_keep_fwd(uint32_t seed, unsigned long long idx, float p, float inv_keep) {
  uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ (uint32_t)(idx >> 32) * 0x85EBCA77u ^ seed;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return ((x >> 8) * (1.0f / 16777216.0f) >= p) ? inv_keep : 0.f;
}

template <class T>
__global__ __launch_bounds__(256) void attn_fwd_hd64_kernel(const u16* __restrict__ qkv, u16* __restrict__ out,
                                                            int Tlen, int ldqkv, int ldo, int koff, int voff,
                                                            long long qkv_batch_stride, long long o_batch_stride,
                                                            float scale_log2e, int nqt, int nheads,
                                                            float* __restrict__ lse_out, float p_drop, uint32_t seed) {
  constexpr int KV_BUF = 64 * KS_ROW + 64 * VS_ROW;
  __shared__ __attribute__((aligned(16))) u16 smem[2 * KV_BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
  // run of (batch, head, q-tile) ids: the q-tiles that share one (batch, head)'s K/V then share an L2.
  int id = blockIdx.x;
  {
    const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int qt = id % nqt;
  const int h = (id / nqt) % nheads, b = id / (nqt * nheads);
  const int q0 = qt * 128 + wave * 32;
  const int hl = lane >> 5, l31 = lane & 31;
  const u16* base = qkv + (long long)b * qkv_batch_stride + h * 64;

  // Q fragments: B operand, col = query (lane&31), k = d = ks*16 + 8*hl + j
  u32x4 qf[4];
  {
    const int q = q0 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q < Tlen) v = *reinterpret_cast<const u32x4*>(base + (long long)q * ldqkv + ks * 16 + hl * 8);
      qf[ks] = v;
    }
  }

  if (scale_log2e != 1.0f) {          // fold softmax scale * log2(e) into Q (callers may pre-fold it into W_q)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t w = qf[ks][e];
        qf[ks][e] = pack2<T>(T::to_f32((u16)(w & 0xffffu)) * scale_log2e, T::to_f32((u16)(w >> 16)) * scale_log2e);
      }
  }
  const uint32_t one16 = T::from_f32(1.0f);
  const uint32_t ones2 = one16 | (one16 << 16);
  const u32x4 kaug = {hl == 0 ? ones2 : 0u, 0u, 0u, 0u};            // K side of the augmented k-step
  const u32x4 vones = {ones2, ones2, ones2, ones2};                 // V^T side: a row of ones -> row sums
  u32x4 qaug = {0u, 0u, 0u, 0u};                                    // Q side: (-m_hi, -m_lo, 0, ...)

  f32x16 o[2], lacc;
#pragma unroll
  for (int dj = 0; dj < 2; ++dj)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dj][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  float m_run = 0.f;                                                // value currently subtracted by the MFMA

  // staging coordinates: 64 rows x 8 chunks(16 B) for K and for V; 2 chunks each per thread
  int srow[2], scol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int c = tid + 256 * i;
    srow[i] = c >> 3;
    scol[i] = (c & 7) * 8;
  }
  u32x4 rk[2], rv[2];
  const int ntiles = (Tlen + 63) / 64;

  auto load_kv = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = kt * 64 + srow[i];
      u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
      if (key < Tlen) {
        const u16* rowp = base + (long long)key * ldqkv + scol[i];
        a = *reinterpret_cast<const u32x4*>(rowp + koff);
        c = *reinterpret_cast<const u32x4*>(rowp + voff);
      }
      rk[i] = a;
      rv[i] = c;
    }
  };

  auto store_kv = [&](int buf) {
    u16* Kd = smem + buf * KV_BUF;
    u16* Vd = Kd + 64 * KS_ROW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(&Kd[srow[i] * KS_ROW + scol[i]]) = rk[i];
      *reinterpret_cast<u32x4*>(&Vd[srow[i] * VS_ROW + scol[i]]) = rv[i];
    }
  };

  load_kv(0);
  store_kv(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const u16* Ks = smem + (kt & 1) * KV_BUF;
    const u16* Vs = Ks + 64 * KS_ROW;
    if (kt + 1 < ntiles) load_kv(kt + 1);             // global -> registers, lands under this tile's math

    // ---- S^T = K Q^T : two 32-key sub-tiles ----
    f32x16 s[2];
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      u32x4 kf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        kf[ks] = *reinterpret_cast<const u32x4*>(&Ks[(kj * 32 + l31) * KS_ROW + ks * 16 + hl * 8]);
      __builtin_amdgcn_s_setprio(1);
      s[kj] = T::mfma(kaug, qaug, zero);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[kj] = T::mfma(kf[ks], qf[ks], s[kj]);
      __builtin_amdgcn_s_setprio(0);
    }
    // ---- online softmax (row = query = lane&31; keys spread over regs and lane halves) ----
    // s already equals score*c - m_run.  The running max is raised only when some row grew by more than
    // DEFER_THR (log2 units) - a rare, wave-uniform branch that rescales O, l and this tile's scores;
    // otherwise P = exp2(s) directly (values up to 2^DEFER_THR).
    const int kbase = kt * 64;
    if (kbase + 64 > Tlen) {                            // wave-uniform: only the last, partial tile masks keys
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + kj * 32 + mfma_row(r, lane) >= Tlen) s[kj][r] = -3.0e38f;
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (kt == 0 || __any(mx > DEFER_THR)) {
      // new subtracted value = m_run + mx, split in two 16-bit terms so the MFMA subtracts it to ~2^-17
      const float want = m_run + mx;
      const float hi = T::to_f32(T::from_f32(want));
      const float lo = T::to_f32(T::from_f32(want - hi));
      const float m_new = hi + lo;
      const float delta = m_new - m_run;
      if (kt > 0) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dj][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
      }
#pragma unroll
      for (int kj = 0; kj < 2; ++kj)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kj][r] -= delta;
      m_run = m_new;
      qaug[0] = (hl == 0) ? pack2<T>(-hi, -lo) : 0u;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[0][r] = __builtin_amdgcn_exp2f(s[0][r]);
      s[1][r] = __builtin_amdgcn_exp2f(s[1][r]);
    }

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 pf;
        pf[0] = pack2<T>(s[kj][8 * s2 + 0], s[kj][8 * s2 + 1]);
        pf[1] = pack2<T>(s[kj][8 * s2 + 2], s[kj][8 * s2 + 3]);
        pf[2] = pack2<T>(s[kj][8 * s2 + 4], s[kj][8 * s2 + 5]);
        pf[3] = pack2<T>(s[kj][8 * s2 + 6], s[kj][8 * s2 + 7]);
        lacc = T::mfma(vones, pf, lacc);                  // row sums of the (rounded) P, all 32 rows equal
        if (p_drop > 0.f) {                               // attention dropout (training): O uses keep/(1-p) * P, l does not
          const float ik = 1.0f / (1.0f - p_drop);
          const unsigned long long rowbase = (((unsigned long long)b * nheads + h) * Tlen + (q0 + l31 < Tlen ? q0 + l31 : 0)) * Tlen;
          float pd[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int key = kbase + kj * 32 + mfma_row(8 * s2 + e, lane);
            pd[e] = s[kj][8 * s2 + e] * attn_keep_fwd(seed, rowbase + key, p_drop, ik);
          }
          pf[0] = pack2<T>(pd[0], pd[1]);
          pf[1] = pack2<T>(pd[2], pd[3]);
          pf[2] = pack2<T>(pd[4], pd[5]);
          pf[3] = pack2<T>(pd[6], pd[7]);
        }
        // transposed V reads: 16-lane group g -> d block (g&1)*16, lane half = g>>1;
        // lane 4q+p of the group addresses row q, cols 4p..4p+3 and receives column (lane&15)
        const int g16 = lane >> 4, i16 = lane & 15;
        const int qq = i16 >> 2, pp = i16 & 3;
        const int keyrow = kj * 32 + s2 * 16 + 4 * (g16 >> 1) + qq;
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
          const int dcol = dj * 32 + (g16 & 1) * 16 + 4 * pp;
          s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(&Vs[keyrow * VS_ROW + dcol]));
          s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(&Vs[(keyrow + 8) * VS_ROW + dcol]));
          u32x2 a0 = __builtin_bit_cast(u32x2, v0);
          u32x2 a1 = __builtin_bit_cast(u32x2, v1);
          u32x4 vf = {a0[0], a0[1], a1[0], a1[1]};
          o[dj] = T::mfma(vf, pf, o[dj]);
        }
      }
    }
    if (kt + 1 < ntiles) store_kv((kt + 1) & 1);      // the other buffer: its readers finished before the last barrier
    __syncthreads();
  }

  // ---- epilogue: normalise, transpose through LDS, coalesced stores ----
  const float inv = 1.0f / lacc[0];
  if (lse_out && hl == 0 && q0 + l31 < Tlen)             // log2-domain log-sum-exp of the scaled scores (for the backward)
    lse_out[((long long)b * nheads + h) * Tlen + q0 + l31] = m_run + __builtin_amdgcn_logf(lacc[0]);
  u16* Os = smem + wave * (32 * OS_ROW);
#pragma unroll
  for (int dj = 0; dj < 2; ++dj)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      int d0 = dj * 32 + 8 * rq + 4 * hl;
      u32x2 w;
      w[0] = pack2<T>(o[dj][4 * rq + 0] * inv, o[dj][4 * rq + 1] * inv);
      w[1] = pack2<T>(o[dj][4 * rq + 2] * inv, o[dj][4 * rq + 3] * inv);
      *reinterpret_cast<u32x2*>(&Os[l31 * OS_ROW + d0]) = w;
    }
  __syncthreads();
  u16* ob = out + (long long)b * o_batch_stride + h * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = lane + 64 * i;
    int row = c >> 3, ch = (c & 7) * 8;
    int q = q0 + row;
    if (q < Tlen) {
      u32x4 v = *reinterpret_cast<const u32x4*>(&Os[row * OS_ROW + ch]);
      *reinterpret_cast<u32x4*>(ob + (long long)q * ldo + ch) = v;
    }
  }
}

// ---------------------------------------------------------------------------
// generic small-shape attention: one wave per query row, lanes over head_dim
// (<= 256 => up to 4 elements per lane), two passes over the keys in fp32.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void attn_fwd_generic_kernel(const u16* __restrict__ qkv, u16* __restrict__ out,
                                                               int Tlen, int hd, int ldqkv, int ldo, int koff,
                                                               int voff, long long qkv_batch_stride,
                                                               long long o_batch_stride, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  const int h = blockIdx.y, b = blockIdx.z;
  if (q >= Tlen) return;
  const u16* base = qkv + (long long)b * qkv_batch_stride + h * hd;
  float qv[4], acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int d = lane + 64 * i;
    qv[i] = (d < hd) ? T::to_f32(base[(long long)q * ldqkv + d]) * scale : 0.f;
    acc[i] = 0.f;
  }
  float m = -1e30f, l = 0.f;
  for (int key = 0; key < Tlen; ++key) {
    const u16* kr = base + (long long)key * ldqkv;
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int d = lane + 64 * i;
      if (d < hd) part += qv[i] * T::to_f32(kr[koff + d]);
    }
    float sc = wave_sum(part);
    float mn = fmaxf(m, sc);
    float al = expf(m - mn), pv = expf(sc - mn);
    l = l * al + pv;
    m = mn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int d = lane + 64 * i;
      if (d < hd) acc[i] = acc[i] * al + pv * T::to_f32(kr[voff + d]);
    }
  }
  u16* ob = out + (long long)b * o_batch_stride + h * hd;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int d = lane + 64 * i;
    if (d < hd) ob[(long long)q * ldo + d] = T::from_f32(acc[i] / l);
  }
}

// qkv: [B, T, ldqkv] 16-bit with q at column h*hd, k at koff + h*hd, v at voff + h*hd.
extern "C" int sfm_attention_fwd_train(const void* qkv, void* out, float* lse, int B, int T, int H, int hd, int ldqkv,
                                       int ldo, int koff, int voff, long long qkv_batch_stride,
                                       long long o_batch_stride, float scale, float p_drop, unsigned int seed,
                                       int dtype, void* stream);

extern "C" int sfm_attention_fwd(const void* qkv, void* out, int B, int T, int H, int hd, int ldqkv, int ldo,
                                 int koff, int voff, long long qkv_batch_stride, long long o_batch_stride,
                                 float scale, int dtype, void* stream) {
  return sfm_attention_fwd_train(qkv, out, nullptr, B, T, H, hd, ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride,
                                 scale, 0.f, 0u, dtype, stream);
}

// training-mode forward: also writes lse [B,H,T] (log2 domain) and applies attention dropout (head_dim 64 only
// when lse != NULL or p_drop > 0)
extern "C" int sfm_attention_fwd_train(const void* qkv, void* out, float* lse, int B, int T, int H, int hd, int ldqkv,
                                       int ldo, int koff, int voff, long long qkv_batch_stride,
                                       long long o_batch_stride, float scale, float p_drop, unsigned int seed,
                                       int dtype, void* stream) {
  if (!qkv || !out) return SFM_ERR_ARG;
  if ((lse || p_drop > 0.f) && hd != 64) return SFM_ERR_SHAPE;
  if (p_drop < 0.f || p_drop >= 1.f) return SFM_ERR_SHAPE;
  if ((lse || p_drop > 0.f) && (qkv_batch_stride != (long long)T * ldqkv)) return SFM_ERR_SHAPE;
  if (B <= 0 || T <= 0 || H <= 0 || hd <= 0 || hd > 256) return SFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (hd == 64 && (ldqkv % 8) == 0 && (ldo % 8) == 0 && (koff % 8) == 0 && (voff % 8) == 0 &&
      (qkv_batch_stride % 8) == 0 && (o_batch_stride % 8) == 0) {
    const int nqt = (T + 127) / 128;
    dim3 grid(nqt * H * B), block(256);
    // scale <= 0: Q already carries softmax_scale * log2(e) (folded into W_q by the caller)
    float sl2 = (scale > 0.f) ? scale * 1.44269504088896340736f : 1.0f;
    if (dtype == SFM_DT_F16)
      SFM_LAUNCH((attn_fwd_hd64_kernel<F16>), grid, block, 0, st, (const u16*)qkv, (u16*)out, T, ldqkv, ldo,
                         koff, voff, qkv_batch_stride, o_batch_stride, sl2, nqt, H, lse, p_drop, seed);
    else
      SFM_LAUNCH((attn_fwd_hd64_kernel<BF16>), grid, block, 0, st, (const u16*)qkv, (u16*)out, T, ldqkv, ldo,
                         koff, voff, qkv_batch_stride, o_batch_stride, sl2, nqt, H, lse, p_drop, seed);
  } else {
    dim3 grid((T + 3) / 4, H, B), block(256);
    if (scale <= 0.f) scale = 0.69314718055994530942f;      // pre-scaled Q carries log2(e): exp(x ln2) = 2^x
    if (dtype == SFM_DT_F16)
      SFM_LAUNCH((attn_fwd_generic_kernel<F16>), grid, block, 0, st, (const u16*)qkv, (u16*)out, T, hd, ldqkv,
                         ldo, koff, voff, qkv_batch_stride, o_batch_stride, scale);
    else
      SFM_LAUNCH((attn_fwd_generic_kernel<BF16>), grid, block, 0, st, (const u16*)qkv, (u16*)out, T, hd,
                         ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride, scale);
  }
  SFM_CHECK_LAUNCH();
  return SFM_OK;
}
