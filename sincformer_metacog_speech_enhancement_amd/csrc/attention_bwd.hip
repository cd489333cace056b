// Backward of the attention core (training path of models/conformer.py:69).  Recomputes P from Q, K and
// the forward's per-row log-sum-exp instead of storing [T, T]; two kernels, no atomics:
//   attn_bwd_dq  : workgroup = 128 queries (4 waves x 32) of one (b, h), loop over 64-key tiles
//   attn_bwd_dkv : workgroup = 128 keys    (4 waves x 32) of one (b, h), loop over 64-query tiles
// Conventions follow attention.hip: q is PRE-SCALED (softmax_scale*log2(e) folded into W_q), so
// s2 = q'.k is in log2 units, P = exp2(s2 - LSE2), dS2 = ln2 * P * (dP - delta), delta = rowsum(dO*O).
// Attention dropout (nn.MultiheadAttention(dropout=p), training) uses the same counter-based keep
// function as the forward: index ((b*H + h)*T + q)*T + key.
// Every "transposed" operand (K^T, Q^T, dO^T) comes from ds_read_b64_tr_b16 on a row-major LDS tile
// with 192-byte rows; row-read operands use 144-byte rows (both conflict-free, see attention.hip).
#include "sfm_common.h"

#define RS_ROW 72     // u16 per row, row-read tiles (144 B)
#define TS_ROW 96     // u16 per row, transposed-read tiles (192 B)

// Attention dropout: keep(row, key) = finalise(rowhash(row) + key * golden), rowhash = the full counter hash of the probability
// row (b, h, q), computed once per row; per score one multiply-add, one xorshift-multiply round and an integer compare
// (~9 VALU instructions instead of ~16).  The same function in attention.hip (forward) and attention_bwd.hip.
__device__ __forceinline__ uint32_t attn_row_hash(uint32_t seed, unsigned long long row) {
  uint32_t x = (uint32_t)row * 0x9E3779B1u ^ (uint32_t)(row >> 32) * 0x85EBCA77u ^ seed;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t attn_keep_threshold(float p) { return (uint32_t)ceilf(p * 16777216.0f); }
__device__ __forceinline__ float attn_keep_rk(uint32_t rowh, int key, uint32_t thr24, float inv_keep) {
  uint32_t x = rowh + (uint32_t)key * 0x9E3779B1u;
  x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return ((x >> 8) >= thr24) ? inv_keep : 0.f;
}

__device__ __forceinline__ u32x4 tr_frag(const u16* tile, int row0, int col0, int lane) {
  // 8 contraction rows (row0 + 4*(lane>>5) + {0..3}, +8) of column col0 + (lane & 31), as an MFMA A fragment
  const int g16 = lane >> 4, i16 = lane & 15;
  const u16* base = tile + (row0 + 4 * (g16 >> 1) + (i16 >> 2)) * TS_ROW + col0 + (g16 & 1) * 16 + 4 * (i16 & 3);
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * TS_ROW));
  const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
  return u32x4{a0[0], a0[1], a1[0], a1[1]};
}

// delta[b,h,q] = sum_d dO[q, h*64+d] * O[q, h*64+d]
template <class T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const u16* __restrict__ dO, const u16* __restrict__ O,
                                                         float* __restrict__ delta, int Tlen, int H, int ldo,
                                                         long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (b, h, q)
  if (i >= total) return;
  const int q = (int)(i % Tlen);
  const int h = (int)((i / Tlen) % H);
  const long long b = i / ((long long)Tlen * H);
  const u16* a = dO + (b * Tlen + q) * ldo + h * 64;
  const u16* c = O + (b * Tlen + q) * ldo + h * 64;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < 64; d += 8) {
    const u32x4 x = *reinterpret_cast<const u32x4*>(a + d), y = *reinterpret_cast<const u32x4*>(c + d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s += T::to_f32((u16)(x[e] & 0xffffu)) * T::to_f32((u16)(y[e] & 0xffffu));
      s += T::to_f32((u16)(x[e] >> 16)) * T::to_f32((u16)(y[e] >> 16));
    }
  }
  delta[i] = s;
}

// ----------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const u16* __restrict__ qkv, const u16* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          u16* __restrict__ dqkv, int Tlen, int H, int ldqkv, int ldo,
                                                          int koff, int voff, int nqt, float p_drop, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) u16 smem[64 * RS_ROW * 2 + 64 * TS_ROW];
  u16* Kr = smem;                       // K rows     (S^T  = K Q^T)
  u16* Vr = smem + 64 * RS_ROW;         // V rows     (dP^T = V dO^T)
  u16* Kt = smem + 2 * 64 * RS_ROW;     // K again, transposed-read layout (dQ^T += K^T dS^T)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hl = lane >> 5;
  const int id = blockIdx.x;
  const int qt = id % nqt, h = (id / nqt) % H, b = id / (nqt * H);
  const int q0 = qt * 128 + wave * 32;
  const int q = q0 + l31;
  const u16* base = qkv + (long long)b * Tlen * ldqkv + h * 64;
  const float inv_keep = (p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;

  u32x4 qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
    if (q < Tlen) {
      a = *reinterpret_cast<const u32x4*>(base + (long long)q * ldqkv + ks * 16 + hl * 8);
      c = *reinterpret_cast<const u32x4*>(dO + ((long long)b * Tlen + q) * ldo + h * 64 + ks * 16 + hl * 8);
    }
    qf[ks] = a;
    dof[ks] = c;
  }
  const long long rowid = ((long long)b * H + h) * Tlen + (q < Tlen ? q : 0);
  const float lse_q = (q < Tlen) ? lse[rowid] : 1e30f;
  const float dl_q = (q < Tlen) ? delta[rowid] : 0.f;
  const uint32_t drop_rowh = attn_row_hash(seed, (unsigned long long)rowid);   // the lane's probability row is fixed
  const uint32_t drop_thr = attn_keep_threshold(p_drop);

  f32x16 dq[2];
#pragma unroll
  for (int dj = 0; dj < 2; ++dj)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dj][r] = 0.f;

  int srow[2], scol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    srow[i] = c >> 3;
    scol[i] = (c & 7) * 8;
  }
  const int ntiles = (Tlen + 63) / 64;
  u32x4 rk[2], rv[2];                                 // next key tile, global -> registers under the current tile's math
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = kt * 64 + srow[i];
      u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
      if (key < Tlen) {
        const u16* rp = base + (long long)key * ldqkv + scol[i];
        a = *reinterpret_cast<const u32x4*>(rp + koff);
        c = *reinterpret_cast<const u32x4*>(rp + voff);
      }
      rk[i] = a;
      rv[i] = c;
    }
  };
  load_tile(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(&Kr[srow[i] * RS_ROW + scol[i]]) = rk[i];
      *reinterpret_cast<u32x4*>(&Kt[srow[i] * TS_ROW + scol[i]]) = rk[i];
      *reinterpret_cast<u32x4*>(&Vr[srow[i] * RS_ROW + scol[i]]) = rv[i];
    }
    __syncthreads();
    if (kt + 1 < ntiles) load_tile(kt + 1);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[2], dp[2];
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
      s[kj] = zero;
      dp[kj] = zero;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(&Kr[(kj * 32 + l31) * RS_ROW + ks * 16 + hl * 8]);
        const u32x4 vf = *reinterpret_cast<const u32x4*>(&Vr[(kj * 32 + l31) * RS_ROW + ks * 16 + hl * 8]);
        s[kj] = T::mfma(kf, qf[ks], s[kj]);
        dp[kj] = T::mfma(vf, dof[ks], dp[kj]);
      }
    }
    // dS^T (rows = keys in registers, column = this lane's query)
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 64 + kj * 32 + mfma_row(r, lane);
        float pv = (key < Tlen) ? __builtin_amdgcn_exp2f(s[kj][r] - lse_q) : 0.f;
        float dpv = dp[kj][r];
        if (p_drop > 0.f) dpv *= attn_keep_rk(drop_rowh, key, drop_thr, inv_keep);
        s[kj][r] = 0.69314718056f * pv * (dpv - dl_q);
      }
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 df;
        df[0] = pack2<T>(s[kj][8 * s2 + 0], s[kj][8 * s2 + 1]);
        df[1] = pack2<T>(s[kj][8 * s2 + 2], s[kj][8 * s2 + 3]);
        df[2] = pack2<T>(s[kj][8 * s2 + 4], s[kj][8 * s2 + 5]);
        df[3] = pack2<T>(s[kj][8 * s2 + 6], s[kj][8 * s2 + 7]);
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) dq[dj] = T::mfma(tr_frag(Kt, kj * 32 + s2 * 16, dj * 32, lane), df, dq[dj]);
      }
  }
  // dQ^T -> LDS transpose -> rows
  __syncthreads();
  u16* Os = smem + wave * (32 * RS_ROW);
#pragma unroll
  for (int dj = 0; dj < 2; ++dj)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      u32x2 w;
      w[0] = pack2<T>(dq[dj][4 * rq + 0], dq[dj][4 * rq + 1]);
      w[1] = pack2<T>(dq[dj][4 * rq + 2], dq[dj][4 * rq + 3]);
      *reinterpret_cast<u32x2*>(&Os[l31 * RS_ROW + dj * 32 + 8 * rq + 4 * hl]) = w;
    }
  __syncthreads();
  u16* ob = dqkv + (long long)b * Tlen * ldqkv + h * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    const int row = c >> 3, ch = (c & 7) * 8;
    if (q0 + row < Tlen)
      *reinterpret_cast<u32x4*>(ob + (long long)(q0 + row) * ldqkv + ch) = *reinterpret_cast<const u32x4*>(&Os[row * RS_ROW + ch]);
  }
}

// ----------------------------------------------------------------------------------------------------
template <class T, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const u16* __restrict__ qkv, const u16* __restrict__ dO,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           u16* __restrict__ dqkv, int Tlen, int H, int ldqkv, int ldo,
                                                           int koff, int voff, int nkt, float p_drop, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) u16 smem[2 * 64 * RS_ROW + 2 * 64 * TS_ROW];
  __shared__ float sl[64], sd[64];
  __shared__ uint32_t srh[64];                         // row hashes of the staged query tile (attention dropout)
  u16* Qr = smem;                               // Q rows   (S  = Q K^T)
  u16* Dr = smem + 64 * RS_ROW;                 // dO rows  (dP = dO V^T)
  u16* Qt = smem + 2 * 64 * RS_ROW;             // Q, transposed-read layout  (dK^T += Q^T dS)
  u16* Dt = Qt + 64 * TS_ROW;                   // dO, transposed-read layout (dV^T += dO^T P~)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hl = lane >> 5;
  const int id = blockIdx.x;
  const int ktile = id % nkt, h = (id / nkt) % H, b = id / (nkt * H);
  const int key0 = ktile * 128 + wave * 32;
  const int key = key0 + l31;
  const u16* base = qkv + (long long)b * Tlen * ldqkv + h * 64;
  const float inv_keep = (p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;

  u32x4 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
    if (key < Tlen) {
      const u16* rp = base + (long long)key * ldqkv + ks * 16 + hl * 8;
      a = *reinterpret_cast<const u32x4*>(rp + koff);
      c = *reinterpret_cast<const u32x4*>(rp + voff);
    }
    kf[ks] = a;
    vf[ks] = c;
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dj = 0; dj < 2; ++dj)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[dj][r] = 0.f; dv[dj][r] = 0.f; }

  int srow[2], scol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 256 * i;
    srow[i] = c >> 3;
    scol[i] = (c & 7) * 8;
  }
  const long long bh = (long long)b * H + h;
  const int nqtiles = (Tlen + 63) / 64;
  u32x4 rq[2], rd[2];                                 // next query tile, global -> registers under the current tile's math
  float rl = 1e30f, rdl = 0.f;
  uint32_t rrh = 0u;
  const uint32_t drop_thr = DROP ? attn_keep_threshold(p_drop) : 0u;
  auto load_tile = [&](int qt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = qt * 64 + srow[i];
      u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
      if (q < Tlen) {
        a = *reinterpret_cast<const u32x4*>(base + (long long)q * ldqkv + scol[i]);
        c = *reinterpret_cast<const u32x4*>(dO + ((long long)b * Tlen + q) * ldo + h * 64 + scol[i]);
      }
      rq[i] = a;
      rd[i] = c;
    }
    if (tid < 64) {
      const int q = qt * 64 + tid;
      rl = (q < Tlen) ? lse[bh * Tlen + q] : 1e30f;             // rows past T get P = 0
      rdl = (q < Tlen) ? delta[bh * Tlen + q] : 0.f;
      if (DROP) rrh = attn_row_hash(seed, (unsigned long long)(bh * Tlen + (q < Tlen ? q : 0)));
    }
  };
  load_tile(0);
  for (int qt = 0; qt < nqtiles; ++qt) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(&Qr[srow[i] * RS_ROW + scol[i]]) = rq[i];
      *reinterpret_cast<u32x4*>(&Qt[srow[i] * TS_ROW + scol[i]]) = rq[i];
      *reinterpret_cast<u32x4*>(&Dr[srow[i] * RS_ROW + scol[i]]) = rd[i];
      *reinterpret_cast<u32x4*>(&Dt[srow[i] * TS_ROW + scol[i]]) = rd[i];
    }
    if (tid < 64) {
      sl[tid] = rl;
      sd[tid] = rdl;
      if (DROP) srh[tid] = rrh;
    }
    __syncthreads();
    if (qt + 1 < nqtiles) load_tile(qt + 1);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // one 32-query sub-tile at a time: S, dP (32 accumulator registers live instead of 64)
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      f32x16 s = zero, dp = zero;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 qfr = *reinterpret_cast<const u32x4*>(&Qr[(qi * 32 + l31) * RS_ROW + ks * 16 + hl * 8]);
        const u32x4 dfr = *reinterpret_cast<const u32x4*>(&Dr[(qi * 32 + l31) * RS_ROW + ks * 16 + hl * 8]);
        s = T::mfma(qfr, kf[ks], s);
        dp = T::mfma(dfr, vf[ks], dp);
      }
      // rows = queries (registers), column = this lane's key
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = qi * 32 + mfma_row(r, lane);
        const float pv = __builtin_amdgcn_exp2f(s[r] - sl[ql]);
        float keep = 1.0f;
        if (DROP) keep = attn_keep_rk(srh[ql], key, drop_thr, inv_keep);
        const float dsv = 0.69314718056f * pv * (dp[r] * keep - sd[ql]);
        s[r] = pv * keep;                // P~ (dropped, rescaled) for dV
        dp[r] = dsv;                     // dS2 for dK
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 pf, df;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pf[e] = pack2<T>(s[8 * s2 + 2 * e], s[8 * s2 + 2 * e + 1]);
          df[e] = pack2<T>(dp[8 * s2 + 2 * e], dp[8 * s2 + 2 * e + 1]);
        }
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
          dv[dj] = T::mfma(tr_frag(Dt, qi * 32 + s2 * 16, dj * 32, lane), pf, dv[dj]);
          dk[dj] = T::mfma(tr_frag(Qt, qi * 32 + s2 * 16, dj * 32, lane), df, dk[dj]);
        }
      }
    }
  }
  // dK^T, dV^T -> LDS transpose -> rows
  __syncthreads();
  u16* Os = smem + wave * (32 * RS_ROW);
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x16& src = which ? dv[dj] : dk[dj];
        u32x2 w;
        w[0] = pack2<T>(src[4 * rq + 0], src[4 * rq + 1]);
        w[1] = pack2<T>(src[4 * rq + 2], src[4 * rq + 3]);
        *reinterpret_cast<u32x2*>(&Os[l31 * RS_ROW + dj * 32 + 8 * rq + 4 * hl]) = w;
      }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    u16* ob = dqkv + (long long)b * Tlen * ldqkv + h * 64 + (which ? voff : koff);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      const int row = c >> 3, ch = (c & 7) * 8;
      if (key0 + row < Tlen)
        *reinterpret_cast<u32x4*>(ob + (long long)(key0 + row) * ldqkv + ch) = *reinterpret_cast<const u32x4*>(&Os[row * RS_ROW + ch]);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
  }
}

// qkv [B,T,ldqkv] (q' | k | v, q' pre-scaled), O / dO [B,T,ldo] 16-bit, lse [B,H,T] fp32 (log2 domain, from
// sfm_attention_fwd_train), dqkv [B,T,ldqkv] 16-bit out, delta = workspace [B,H,T] fp32.  head_dim 64 only.
// ---------------------------------------------------------------------------
// generic small-shape backward (any head_dim <= 256, e.g. the reference's test config d_model 64 / 4 heads = 16):
// one wave per query row, lanes over head_dim, fp32 VALU; dQ is written directly, dK / dV are accumulated with
// float atomics into a zero-filled fp32 scratch [B*T, 2*H*hd] that the caller converts afterwards.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void attn_bwd_generic_kernel(const u16* __restrict__ qkv, const u16* __restrict__ O,
                                                               const u16* __restrict__ dO, const float* __restrict__ lse,
                                                               u16* __restrict__ dqkv, float* __restrict__ dkv32, int Tlen,
                                                               int hd, int ldqkv, int ldo, int koff, int voff,
                                                               float p_drop, uint32_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  const int h = blockIdx.y, b = blockIdx.z, nheads = gridDim.y;
  if (q >= Tlen) return;
  const int D = nheads * hd;
  const u16* base = qkv + (long long)b * Tlen * ldqkv + h * hd;
  const long long orow = ((long long)b * Tlen + q) * ldo + h * hd;
  float qv[4], dov[4], dq[4];
  float part = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = lane + 64 * i;
    qv[i] = (d < hd) ? T::to_f32(base[(long long)q * ldqkv + d]) : 0.f;
    dov[i] = (d < hd) ? T::to_f32(dO[orow + d]) : 0.f;
    const float ov = (d < hd) ? T::to_f32(O[orow + d]) : 0.f;
    part += dov[i] * ov;
    dq[i] = 0.f;
  }
  const float delta = wave_sum(part);
  const float lse_q = lse[((long long)b * nheads + h) * Tlen + q];
  const float inv_keep = (p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;
  const uint32_t drop_rowh = attn_row_hash(seed, ((unsigned long long)b * nheads + h) * Tlen + q);
  const uint32_t drop_thr = attn_keep_threshold(p_drop);
  for (int key = 0; key < Tlen; ++key) {
    const u16* kr = base + (long long)key * ldqkv;
    float kv[4], vv[4];
    float ps = 0.f, pd = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = lane + 64 * i;
      kv[i] = (d < hd) ? T::to_f32(kr[koff + d]) : 0.f;
      vv[i] = (d < hd) ? T::to_f32(kr[voff + d]) : 0.f;
      ps += qv[i] * kv[i];
      pd += dov[i] * vv[i];
    }
    const float s2 = wave_sum(ps), dp = wave_sum(pd);
    const float pv = exp2f(s2 - lse_q);
    const float keep = (p_drop > 0.f) ? attn_keep_rk(drop_rowh, key, drop_thr, inv_keep) : 1.0f;
    const float ds = 0.69314718056f * pv * (dp * keep - delta);
    float* dkr = dkv32 + ((long long)b * Tlen + key) * (2LL * D) + h * hd;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = lane + 64 * i;
      if (d < hd) {
        dq[i] += ds * kv[i];
        atomicAdd(&dkr[d], ds * qv[i]);
        atomicAdd(&dkr[D + d], pv * keep * dov[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = lane + 64 * i;
    if (d < hd) dqkv[((long long)b * Tlen + q) * ldqkv + h * hd + d] = T::from_f32(dq[i]);
  }
}

// scratch -> 16-bit dK | dV columns of dqkv
template <class T>
__global__ __launch_bounds__(256) void attn_bwd_pack_kv_kernel(const float* __restrict__ dkv32, u16* __restrict__ dqkv,
                                                               long long M, int D, int ldqkv, int koff) {
  const long long total = M * 2LL * D;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long m = e / (2LL * D);
    const int c = (int)(e - m * 2LL * D);
    dqkv[m * ldqkv + koff + c] = T::from_f32(dkv32[e]);
  }
}

extern "C" int sfm_attention_bwd_generic(const void* qkv, const void* O, const void* dO, const float* lse, float* dkv32,
                                         void* dqkv, int B, int T, int H, int hd, int ldqkv, int ldo, int koff, int voff,
                                         float p_drop, unsigned int seed, int dtype, void* stream) {
  if (!qkv || !O || !dO || !lse || !dkv32 || !dqkv) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0 || H <= 0 || hd <= 0 || hd > 256 || voff != koff + H * hd) return SFM_ERR_SHAPE;
  if (p_drop < 0.f || p_drop >= 1.f) return SFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((T + 3) / 4, H, B), block(256);
  const long long M = (long long)B * T;
  long long nb = (M * 2 * H * hd + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (dtype == SFM_DT_F16) {
    SFM_LAUNCH((attn_bwd_generic_kernel<F16>), grid, block, 0, st, (const u16*)qkv, (const u16*)O, (const u16*)dO, lse,
               (u16*)dqkv, dkv32, T, hd, ldqkv, ldo, koff, voff, p_drop, seed);
    SFM_LAUNCH((attn_bwd_pack_kv_kernel<F16>), dim3((unsigned)nb), block, 0, st, dkv32, (u16*)dqkv, M, H * hd, ldqkv, koff);
  } else {
    SFM_LAUNCH((attn_bwd_generic_kernel<BF16>), grid, block, 0, st, (const u16*)qkv, (const u16*)O, (const u16*)dO, lse,
               (u16*)dqkv, dkv32, T, hd, ldqkv, ldo, koff, voff, p_drop, seed);
    SFM_LAUNCH((attn_bwd_pack_kv_kernel<BF16>), dim3((unsigned)nb), block, 0, st, dkv32, (u16*)dqkv, M, H * hd, ldqkv, koff);
  }
  return SFM_OK;
}

extern "C" int sfm_attention_bwd(const void* qkv, const void* O, const void* dO, const float* lse, float* delta, void* dqkv,
                                 int B, int T, int H, int hd, int ldqkv, int ldo, int koff, int voff, float p_drop,
                                 unsigned int seed, int dtype, void* stream) {
  if (!qkv || !O || !dO || !lse || !delta || !dqkv) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0 || H <= 0 || hd != 64 || (ldqkv % 8) || (ldo % 8) || (koff % 8) || (voff % 8) || p_drop < 0.f ||
      p_drop >= 1.f)
    return SFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)B * H * T;
  const int nq = (T + 127) / 128;
  dim3 g1((unsigned)((total + 255) / 256)), g2(nq * H * B), blk(256);
  if (dtype == SFM_DT_F16) {
    SFM_LAUNCH((attn_delta_kernel<F16>), g1, blk, 0, st, (const u16*)dO, (const u16*)O, delta, T, H, ldo, total);
    SFM_LAUNCH((attn_bwd_dq_kernel<F16>), g2, blk, 0, st, (const u16*)qkv, (const u16*)dO, lse, delta, (u16*)dqkv, T, H, ldqkv,
               ldo, koff, voff, nq, p_drop, seed);
    if (p_drop > 0.f)
      SFM_LAUNCH((attn_bwd_dkv_kernel<F16, true>), g2, blk, 0, st, (const u16*)qkv, (const u16*)dO, lse, delta, (u16*)dqkv, T,
                 H, ldqkv, ldo, koff, voff, nq, p_drop, seed);
    else
      SFM_LAUNCH((attn_bwd_dkv_kernel<F16, false>), g2, blk, 0, st, (const u16*)qkv, (const u16*)dO, lse, delta, (u16*)dqkv,
                 T, H, ldqkv, ldo, koff, voff, nq, p_drop, seed);
  } else {
    SFM_LAUNCH((attn_delta_kernel<BF16>), g1, blk, 0, st, (const u16*)dO, (const u16*)O, delta, T, H, ldo, total);
    SFM_LAUNCH((attn_bwd_dq_kernel<BF16>), g2, blk, 0, st, (const u16*)qkv, (const u16*)dO, lse, delta, (u16*)dqkv, T, H,
               ldqkv, ldo, koff, voff, nq, p_drop, seed);
    if (p_drop > 0.f)
      SFM_LAUNCH((attn_bwd_dkv_kernel<BF16, true>), g2, blk, 0, st, (const u16*)qkv, (const u16*)dO, lse, delta, (u16*)dqkv, T,
                 H, ldqkv, ldo, koff, voff, nq, p_drop, seed);
    else
      SFM_LAUNCH((attn_bwd_dkv_kernel<BF16, false>), g2, blk, 0, st, (const u16*)qkv, (const u16*)dO, lse, delta, (u16*)dqkv,
                 T, H, ldqkv, ldo, koff, voff, nq, p_drop, seed);
  }
  return SFM_OK;
}
