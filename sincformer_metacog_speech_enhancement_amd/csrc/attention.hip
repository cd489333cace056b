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

// Row maxima: this file is compiled with -fno-honor-nans (build.py), so fmaxf() on MFMA results is a plain v_max_f32 /
// v_max3_f32; with NaNs honoured hipcc canonicalises every operand first (one extra v_max_f32 x, x per score pair, +3 %
// kernel time).  The scores cannot be NaN here (finite operands, -BIG instead of -inf).  (An inline-asm v_max3 would also
// avoid the canonicalisation but hides the MFMA-result hazard from the compiler's nop insertion: wrong values now and then.)
__device__ __forceinline__ float sfm_max2_raw(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float sfm_max3_raw(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// O is written in the operands' format T, or (out_other) in the other 16-bit format: the attention core may run in bf16 behind
// fp16 projections (precision policy, ops.STAGES)
template <class T>
__device__ __forceinline__ uint32_t pack2_o(float lo, float hi, bool other) {
  if (T::id == SFM_DT_BF16) return other ? F16::pack(lo, hi) : BF16::pack(lo, hi);
  return other ? BF16::pack(lo, hi) : F16::pack(lo, hi);
}

template <class T, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_hd64_kernel(const u16* __restrict__ qkv, u16* __restrict__ out,
                                                            int Tlen, int ldqkv, int ldo, int koff, int voff,
                                                            long long qkv_batch_stride, long long o_batch_stride,
                                                            float scale_log2e, int nqt, int nheads,
                                                            float* __restrict__ lse_out, float p_drop, uint32_t seed,
                                                            int out_other) {
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
  // augmented k-step: K side (1, 1, pad, 0...) x Q side (-m_hi, -m_lo, -BIG, 0...): the matrix core subtracts the running
  // max AND pushes the scores of padding keys (>= Tlen, last tile only) to -BIG.  (A VALU mask under a wave-uniform
  // `if (last tile)` was if-converted by the compiler into ~110 compare / select instructions on EVERY tile.)
  const uint32_t negbig = (uint32_t)T::from_f32(T::id == SFM_DT_F16 ? -60000.0f : -3.0e38f);
  const u32x4 vones = {ones2, ones2, ones2, ones2};                 // V^T side: a row of ones -> row sums
  u32x4 qaug = {0u, hl == 0 ? negbig : 0u, 0u, 0u};                 // Q side: (-m_hi, -m_lo, -BIG, 0, ...)

  f32x16 o[2], lacc;
#pragma unroll
  for (int dj = 0; dj < 2; ++dj)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dj][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
  float m_run = 0.f;                                                // value currently subtracted by the MFMA
  // attention dropout (training): the lane's probability row is fixed, so its row hash is computed once
  const uint32_t drop_rowh = DROP ? attn_row_hash(seed, ((unsigned long long)b * nheads + h) * Tlen + (q0 + l31 < Tlen ? q0 + l31 : 0)) : 0u;
  const uint32_t drop_thr = DROP ? attn_keep_threshold(p_drop) : 0u;
  const float drop_ik = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;

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
      const bool pad_ = kt * 64 + kj * 32 + l31 >= Tlen;
      const u32x4 kaug = {hl == 0 ? ones2 : 0u, (hl == 0 && pad_) ? one16 : 0u, 0u, 0u};
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
    float mx = sfm_max2_raw(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = sfm_max3_raw(mx, s[0][r], s[1][r]);
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
        if (DROP) {                                       // attention dropout (training): O uses keep/(1-p) * P, l does not
          float pd[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int key = kbase + kj * 32 + mfma_row(8 * s2 + e, lane);
            pd[e] = s[kj][8 * s2 + e] * attn_keep_rk(drop_rowh, key, drop_thr, drop_ik);
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
      w[0] = pack2_o<T>(o[dj][4 * rq + 0] * inv, o[dj][4 * rq + 1] * inv, out_other != 0);
      w[1] = pack2_o<T>(o[dj][4 * rq + 2] * inv, o[dj][4 * rq + 3] * inv, out_other != 0);
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
// v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of the second: fed two copies of x it
// leaves [x.lo, x.lo] and [x.hi, x.hi].  The s_nop covers the 2 wait states a VALU write of an operand needs before
// the swap reads it.  (Scalars in and out: this compiler reads element 0 for __builtin_bit_cast(float, vec[i]).)
__device__ __forceinline__ void xhalf_swap(float v, float& lo, float& hi) {
  uint32_t a = __builtin_bit_cast(uint32_t, v), c = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, c);
}
__device__ __forceinline__ float xhalf_max(float v) {
  float lo, hi;
  xhalf_swap(v, lo, hi);
  return fmaxf(lo, hi);
}
__device__ __forceinline__ float xhalf_sum(float v) {
  float lo, hi;
  xhalf_swap(v, lo, hi);
  return lo + hi;
}

// ---- ablation hooks of the ring kernel (diagnostic builds only: -DSFM_ABL=n; results are then WRONG on purpose) ----
#ifndef SFM_ABL
#define SFM_ABL 0
#endif
#if SFM_ABL == 1
#define SFM_ABL_EXP(x) (x)
#else
#define SFM_ABL_EXP(x) __builtin_amdgcn_exp2f(x)
#endif
#if SFM_ABL == 13
#define SFM_ABL_ROWMAX(sn) ([&]() { float m_ = fmaxf(sn[0], sn[1]); _Pragma("unroll") for (int r = 2; r < 16; ++r) m_ = fmaxf(m_, sn[r]); return m_; }())
#else
#define SFM_ABL_ROWMAX(sn) ([&]() { float m_ = sfm_max2_raw(sn[0], sn[1]); _Pragma("unroll") for (int r = 2; r < 16; r += 2) m_ = sfm_max3_raw(m_, sn[r], sn[r + 1]); return m_; }())
#endif
#if SFM_ABL == 5
#define SFM_ABL_LACC(x)
#else
#define SFM_ABL_LACC(x) x
#endif
#if SFM_ABL == 4
#define SFM_ABL_SCHED
#else
#define SFM_ABL_SCHED                                                                                                  \
  _Pragma("unroll") for (int g_ = 0; g_ < 11; ++g_) {                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                                 \
  }
#endif
#if SFM_ABL == 10
#define SFM_ABL_STORE_AUX 17
#else
#define SFM_ABL_STORE_AUX 0
#endif
#if SFM_ABL == 11
#define SFM_ABL_STAGGER() do { if (wave >= 4) __builtin_amdgcn_s_sleep(6); } while (0)
#else
#define SFM_ABL_STAGGER() do { } while (0)
#endif
#if SFM_ABL == 3
#define SFM_ABL_SYNC()
#else
#define SFM_ABL_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#endif

typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;

template <class T>
__global__ __launch_bounds__(512, 2) void attn_fwd_hd64r_kernel(const u16* __restrict__ qkv, u16* __restrict__ out, int Tlen,
                                                                int ldqkv, int ldo, int koff, int voff,
                                                                long long qkv_batch_stride, long long o_batch_stride,
                                                                float scale_log2e, int nqt, int nheads, int n_items,
                                                                float* __restrict__ lse_out, int out_other) {
  constexpr int SLOT = 16384;                                       // one key tile: K 64 x 128 B, then V 64 x 128 B
  constexpr int GT = 3;                                             // key tiles per group (ring = 2 groups)
  constexpr int QBASE = 2 * GT * SLOT;                              // Q prefetch region: 8 waves x 64 rows x 128 B
  extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane >> 5, l31 = lane & 31;
  const int nkt = (Tlen + 63) >> 6, ngrp = (nkt + GT - 1) / GT, nsteps = 2 * nkt;
  const int rec_bytes = Tlen * ldqkv * 2;                          // keys / queries >= Tlen are out of range: read as zero
  const int orec_bytes = Tlen * ldo * 2;

  // ---- LDS-DMA lane constants: an instruction moves 8 rows x 128 B; this wave owns rows 8*wave .. 8*wave+7 of every tile ----
  const int prow = wave * 8 + (lane >> 3);
  const int kconst = prow * ldqkv * 2 + koff * 2 + (((lane & 7) ^ ((prow >> 1) & 7)) << 4);
  const int vconst = prow * ldqkv * 2 + voff * 2 + (((lane & 7) ^ (((prow >> 1) & 1) << 2)) << 4);
  auto issue_group = [&](int item, int g, int half) {              // K/V tiles GT g .. GT g + GT-1 of `item` -> ring half `half`
    const int bh = item / nqt;
    const int h = bh % nheads, b = bh / nheads;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (long long)b * qkv_batch_stride), 0, rec_bytes, 0x00020000);
#pragma unroll
    for (int tl = 0; tl < GT; ++tl) {
      const int kt = g * GT + tl;
      if (kt < nkt) {
        const int off = kt * 64 * ldqkv * 2 + h * 128;
        unsigned char* dst = rsm + (half * GT + tl) * SLOT + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (attn_lds_ptr_t)dst, 16, kconst + off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (attn_lds_ptr_t)(dst + 8192), 16, vconst + off, 0, 0, 0);
      }
    }
  };

  // Q rows of `item` for this wave (64 rows x 128 B, same chunk swizzle as a K tile) -> the wave's 8 KB of the Q region
  auto issue_q = [&](int item) {
    const int qt = item % nqt, bh = item / nqt;
    const int h = bh % nheads, b = bh / nheads;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (long long)b * qkv_batch_stride), 0, rec_bytes, 0x00020000);
    const int off = (qt * 512 + wave * 64) * ldqkv * 2 + h * 128;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // LDS chunk position lane&7 of row 8j + lane/8 holds logical chunk (lane&7) ^ ((4j + lane/16) & 7)
      const int c = ((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) << 4;
      const int voff = off + (8 * j + (lane >> 3)) * ldqkv * 2 + c;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (attn_lds_ptr_t)(rsm + QBASE + wave * 8192 + j * 1024), 16, voff, 0, 0, 0);
    }
  };

  // ---- fragment read lane constants ----
  int klane[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) klane[ks] = l31 * 128 + (((2 * ks + hl) ^ ((l31 >> 1) & 7)) << 4);
  const int g16 = lane >> 4, i16 = lane & 15;
  const int vrow0 = 4 * (g16 >> 1) + (i16 >> 2);
  int vlane[2];
#pragma unroll
  for (int dj = 0; dj < 2; ++dj) {
    const int chunk = dj * 4 + (g16 & 1) * 2 + ((i16 & 3) >> 1);
    vlane[dj] = 8192 + vrow0 * 128 + ((chunk ^ (((vrow0 >> 1) & 1) << 2)) << 4) + (i16 & 1) * 8;
  }

  const uint32_t one16 = T::from_f32(1.0f);
  const uint32_t ones2 = one16 | (one16 << 16);
  const uint32_t negbig = (uint32_t)T::from_f32(T::id == SFM_DT_F16 ? -60000.0f : -3.0e38f);
  const u32x4 vones = {ones2, ones2, ones2, ones2};

  u32x4 kf[4];
  u32x2 vt[2][2][2];                                                // V^T fragment halves [s2][dj][first / second 4 keys]
  auto load_kf = [&](int sbase) {                                  // K rows of one 32-key step (sbase: byte offset of its rows)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const u32x4*>(rsm + sbase + klane[ks]);
  };
  // The transposed V reads are issued as inline asm: behind an LDS-DMA the compiler guards every ds_read_b64_tr_b16 builtin
  // with `s_waitcnt vmcnt(0)` (it cannot prove the read does not alias the DMA's destination), which would make every step
  // wait for the next group's K/V prefetch and for the previous item's O stores.  The ring protocol (vmcnt + barrier at the
  // group boundary) is what orders reads against fills; `vt_wait` is the lgkmcnt wait the compiler no longer inserts, tied
  // to the fragment registers so that their consumers stay behind it.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(attn_lds_ptr_t)rsm;
  auto load_vf = [&](int sbase) {
#pragma unroll
    for (int dj = 0; dj < 2; ++dj) {
      const uint32_t a = lds0 + sbase + vlane[dj];
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[0][dj][0]) : "v"(a) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(vt[0][dj][1]) : "v"(a) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vt[1][dj][0]) : "v"(a) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(vt[1][dj][1]) : "v"(a) : "memory");
    }
  };
  auto vt_wait = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]), "+v"(vt[1][0][0]),
                   "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1])
                 :
                 : "memory");
  };
#define SFM_VF(S2, DJ) (u32x4{vt[S2][DJ][0][0], vt[S2][DJ][0][1], vt[S2][DJ][1][0], vt[S2][DJ][1][1]})

  u32x4 qf[2][4];
  uint32_t qaug[2];
  float m_run[2];
  f32x16 o[2][2], lacc[2], s[2];

#define SFM_ATTN_RITEM(U, STEP, HAS_PREV, FIRST, WAITV)                                                                    \
  {                                                                                                                   \
    constexpr int V_ = 1 - (U);                                                                                       \
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};              \
    const bool pad_ = (STEP) * 32 + l31 >= Tlen;                                                                      \
    const u32x4 ka = {hl == 0 ? ones2 : 0u, (hl == 0 && pad_) ? one16 : 0u, 0u, 0u};                                  \
    const u32x4 qa = {qaug[U], hl == 0 ? negbig : 0u, 0u, 0u};                                                        \
    f32x16 sn = T::mfma(ka, qa, zero);                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) sn = T::mfma(kf[ks], qf[U][ks], sn);                             \
    if (WAITV) vt_wait();                                                                                             \
    if (HAS_PREV) {                                                                                                   \
      u32x4 pf[2];                                                                                                    \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                              \
        float e_[8];                                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) e_[r] = SFM_ABL_EXP(s[V_][8 * s2 + r]);                         \
        pf[s2][0] = pack2<T>(e_[0], e_[1]);                                                                           \
        pf[s2][1] = pack2<T>(e_[2], e_[3]);                                                                           \
        pf[s2][2] = pack2<T>(e_[4], e_[5]);                                                                           \
        pf[s2][3] = pack2<T>(e_[6], e_[7]);                                                                           \
      }                                                                                                               \
      _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                              \
        SFM_ABL_LACC(lacc[V_] = T::mfma(vones, pf[s2], lacc[V_]);)                                                    \
        o[V_][0] = T::mfma(SFM_VF(s2, 0), pf[s2], o[V_][0]);                                                              \
        o[V_][1] = T::mfma(SFM_VF(s2, 1), pf[s2], o[V_][1]);                                                              \
      }                                                                                                               \
    }                                                                                                                 \
    float mx = SFM_ABL_ROWMAX(sn);                                                                                    \
    mx = xhalf_max(mx);                                                                                               \
    SFM_ABL_SCHED                                                                                                     \
    if ((FIRST) || __any(mx > DEFER_THR)) {                                                                           \
      const float want = m_run[U] + mx;                                                                               \
      const float hi = T::to_f32(T::from_f32(want));                                                                  \
      const float lo = T::to_f32(T::from_f32(want - hi));                                                             \
      const float m_new = hi + lo;                                                                                    \
      const float delta = m_new - m_run[U];                                                                           \
      if (!(FIRST)) {                                                                                                 \
        const float alpha = __builtin_amdgcn_exp2f(-delta);                                                           \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                              \
          o[U][0][r] *= alpha;                                                                                        \
          o[U][1][r] *= alpha;                                                                                        \
          lacc[U][r] *= alpha;                                                                                        \
        }                                                                                                             \
      }                                                                                                               \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) sn[r] -= delta;                                                  \
      m_run[U] = m_new;                                                                                               \
      qaug[U] = (hl == 0) ? pack2<T>(-hi, -lo) : 0u;                                                                  \
    }                                                                                                                 \
    s[U] = sn;                                                                                                        \
  }

  // ---- O of the finished item, held in registers until the next item's first barrier has been passed; then transposed
  //      through the wave's own 8 KB of the Q region (free between the Q fragment reads and the next Q prefetch) so that
  //      every store instruction writes 8 whole 128-byte rows (per-lane row-strided stores cost ~600 cycles of issue each) ----
  uint32_t ow[2][2][4][2];                                          // [sub-block][dj][rq][2 dwords] = 4 consecutive d, 16-bit
  int st_item = -1;
  auto store_o = [&]() {
    if (st_item < 0) return;
    const int qt = st_item % nqt, bh = st_item / nqt;
    const int h = bh % nheads, b = bh / nheads;
    auto ors = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long long)b * o_batch_stride), 0, orec_bytes, 0x00020000);
    unsigned char* ob = rsm + QBASE + wave * 8192;
    // the lane id is made opaque here so that the two dozen per-lane addresses below are computed HERE, once per item:
    // hoisted to kernel entry they live across the whole tile loop, get spilled, and every reload (scratch = vector memory)
    // brings a vmcnt wait into the store sequence
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int l31o = ln & 31, hlo = ln >> 5;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 32 * u + l31o;
#pragma unroll
      for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
          *reinterpret_cast<u32x2*>(ob + row * 128 + (((dj * 4 + rq) ^ ((row >> 1) & 7)) << 4) + 8 * hlo) =
              u32x2{ow[u][dj][rq][0], ow[u][dj][rq][1]};
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                             // lgkmcnt(0): the wave's own image is complete
    __builtin_amdgcn_wave_barrier();
    const int qbase = qt * 512 + wave * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = ln + 64 * i;
      const int row = c >> 3, ch = c & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ob + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, ors, (qbase + row) * ldo * 2 + h * 128 + ch * 16, 0, SFM_ABL_STORE_AUX);
    }
    st_item = -1;
  };

#if SFM_ABL == 9
  // diagnostic build: lse_out is a stamp buffer, 8 x uint64 per workgroup: start, end, end of items 0..5 (100 MHz ticks)
  unsigned long long* dbg = reinterpret_cast<unsigned long long*>(lse_out) + (size_t)blockIdx.x * 8;
  if (tid == 0) dbg[0] = __builtin_amdgcn_s_memrealtime();
  int dbg_k = 0;
  lse_out = nullptr;
#endif
#if SFM_ABL != 12
  // the second-dispatched half of the workgroup loses the VALU arbitration against its SIMD partner on every segment
  // (priority, then age): one static s_setprio for that half (+1.5 %; `wave` is wave-uniform by readfirstlane)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  int gcount = 0;                                                   // groups consumed so far by this workgroup (ring parity)
  if ((int)blockIdx.x < n_items) {
    issue_q(blockIdx.x);
    issue_group(blockIdx.x, 0, 0);
  }
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int qt = item % nqt, bh = item / nqt;
    const int h = bh % nheads, b = bh / nheads;
    const int q0 = qt * 512 + wave * 64;
    for (int g = 0; g < ngrp; ++g) {
      // ---- group boundary: this wave's pieces of group g have landed (vmcnt), everyone's have and everyone is done with
      //      group g-1 (barrier): refill that half with the next group of this item or group 0 of the next item ----
      SFM_ABL_SYNC();
      SFM_ABL_STAGGER();
      const int half = gcount & 1;
      if (g + 1 < ngrp) issue_group(item, g + 1, half ^ 1);
      else if (item + (int)gridDim.x < n_items) issue_group(item + gridDim.x, 0, half ^ 1);
      ++gcount;
      if (g == 0) {
        // ---- Q fragments (B operand: col = query, k = d) from the prefetched LDS rows (rows >= Tlen were zero-filled; the
        //      vmcnt wait above covered this wave's own Q pieces, issued an item ago) ----
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            qf[u][ks] = *reinterpret_cast<const u32x4*>(rsm + QBASE + wave * 8192 + u * 4096 + klane[ks]);
        if (scale_log2e != 1.0f) {                                 // callers normally fold the scale into W_q (scale_log2e == 1)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const uint32_t w = qf[u][ks][e];
                qf[u][ks][e] = pack2<T>(T::to_f32((u16)(w & 0xffffu)) * scale_log2e, T::to_f32((u16)(w >> 16)) * scale_log2e);
              }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // Q fragments are in registers: the region may be reused
        store_o();                                                  // the previous item's O: drains under this item's math
#pragma unroll
        for (int u = 0; u < 2; ++u) {                              // (after the stores: their registers are free again)
          qaug[u] = 0u;
          m_run[u] = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            o[u][0][r] = 0.f;
            o[u][1][r] = 0.f;
            lacc[u][r] = 0.f;
            s[u][r] = 0.f;
          }
        }
      }
      // the next item's Q rows -> this wave's own region (its last LDS accesses, the O read-back, have completed)
      if (g == (ngrp > 1 ? 1 : 0) && item + (int)gridDim.x < n_items) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_q(item + gridDim.x);
      }
      const int step_end = min(nsteps, (g + 1) * 2 * GT);
      int step = g * 2 * GT;
      if (g == 0) {
        // ---- first step of the item, peeled: the running maxima are initialised, A(0) has no predecessor ----
        const int sb = (half * GT) * SLOT;
        load_kf(sb);
        SFM_ATTN_RITEM(0, 0, false, true, false)
        load_vf(sb);
        SFM_ATTN_RITEM(1, 0, true, true, true)
        step = 1;
      }
      for (; step < step_end; ++step) {
        const int sb = (half * GT + ((step >> 1) - g * GT)) * SLOT + (step & 1) * 4096;
        load_kf(sb);
        SFM_ATTN_RITEM(0, step, true, false, false)
        load_vf(sb);
        SFM_ATTN_RITEM(1, step, true, false, true)
      }
    }
    // ---- drain: the last item B(last step) still has to be exponentiated and multiplied into O ----
    {
      u32x4 pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float e_[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) e_[r] = __builtin_amdgcn_exp2f(s[1][8 * s2 + r]);
        pf[s2][0] = pack2<T>(e_[0], e_[1]);
        pf[s2][1] = pack2<T>(e_[2], e_[3]);
        pf[s2][2] = pack2<T>(e_[4], e_[5]);
        pf[s2][3] = pack2<T>(e_[6], e_[7]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        lacc[1] = T::mfma(vones, pf[s2], lacc[1]);
        o[1][0] = T::mfma(SFM_VF(s2, 0), pf[s2], o[1][0]);
        o[1][1] = T::mfma(SFM_VF(s2, 1), pf[s2], o[1][1]);
      }
    }
    // ---- normalise and pack; the stores themselves are issued after the next barrier (store_o) ----
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float l = lacc[u][0];
      const float inv = 1.0f / l;
      const int q = q0 + 32 * u + l31;
      if (lse_out && hl == 0 && q < Tlen)
        lse_out[((long long)b * nheads + h) * Tlen + q] = m_run[u] + __builtin_amdgcn_logf(l);
#pragma unroll
      for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          ow[u][dj][rq][0] = pack2_o<T>(o[u][dj][4 * rq + 0] * inv, o[u][dj][4 * rq + 1] * inv, out_other != 0);
          ow[u][dj][rq][1] = pack2_o<T>(o[u][dj][4 * rq + 2] * inv, o[u][dj][4 * rq + 3] * inv, out_other != 0);
        }
    }
    st_item = item;
#if SFM_ABL == 9
    if (tid == 0 && dbg_k < 6) dbg[2 + dbg_k] = __builtin_amdgcn_s_memrealtime();
    ++dbg_k;
#endif
  }
#undef SFM_ATTN_RITEM
#undef SFM_VF
  store_o();
#if SFM_ABL == 9
  if (tid == 0) dbg[1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------
// generic small-shape attention: one wave per query row, lanes over head_dim
// (<= 256 => up to 4 elements per lane), two passes over the keys in fp32.
// ---------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void attn_fwd_generic_kernel(const u16* __restrict__ qkv, u16* __restrict__ out,
                                                               int Tlen, int hd, int ldqkv, int ldo, int koff,
                                                               int voff, long long qkv_batch_stride,
                                                               long long o_batch_stride, float scale,
                                                               float* __restrict__ lse_out, float p_drop, uint32_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  const int h = blockIdx.y, b = blockIdx.z;
  const int nheads = gridDim.y;
  if (q >= Tlen) return;
  const float inv_keep = (p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;
  const uint32_t drop_rowh = attn_row_hash(seed, ((unsigned long long)b * nheads + h) * Tlen + q);
  const uint32_t drop_thr = attn_keep_threshold(p_drop);
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
    const float pd = (p_drop > 0.f) ? pv * attn_keep_rk(drop_rowh, key, drop_thr, inv_keep) : pv;   // O only, not l
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int d = lane + 64 * i;
      if (d < hd) acc[i] = acc[i] * al + pd * T::to_f32(kr[voff + d]);
    }
  }
  if (lse_out && lane == 0)                        // log2-domain log-sum-exp of the scaled scores, as the MFMA kernels write it
    lse_out[((long long)b * nheads + h) * Tlen + q] = (m + logf(l)) * 1.44269504088896340736f;
  u16* ob = out + (long long)b * o_batch_stride + h * hd;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int d = lane + 64 * i;
    if (d < hd) ob[(long long)q * ldo + d] = T::from_f32(acc[i] / l);
  }
}

// kernel selection for the head_dim 64 no-dropout path, the `variant` argument of sfm_attention_fwd_ex (a per-call argument: the
// library keeps no state): 0 = by shape (below), 1 = always 32 query rows per wave (attn_fwd_hd64), 3 = the persistent ring
// kernel of round 2 (attn_fwd_hd64r; 2 is accepted as 3), 4 / 5 = the pipelined persistent kernel attn_fwd_hd64p
// (attention_pipe.hip) with one 8-wave / two 4-wave workgroups per CU
int sfm_attn_pipe_launch(const void* qkv, void* out, float* lse, int B, int T, int H, int ldqkv, int ldo, int koff, int voff,
                         long long qkv_batch_stride, long long o_batch_stride, float sl2, int dtype, int out_other, int nw,
                         hipStream_t st);

// qkv: [B, T, ldqkv] 16-bit with q at column h*hd, k at koff + h*hd, v at voff + h*hd.
static int attention_fwd_impl(const void* qkv, void* out, float* lse, int B, int T, int H, int hd, int ldqkv, int ldo, int koff,
                              int voff, long long qkv_batch_stride, long long o_batch_stride, float scale, float p_drop,
                              unsigned int seed, int dtype, int out_dtype, int variant, void* stream) {
  if (!qkv || !out) return SFM_ERR_ARG;
  if (variant < 0 || variant > 6) return SFM_ERR_ARG;
  const int sfm_attn_variant = (variant == 2) ? 3 : variant;
  if (p_drop < 0.f || p_drop >= 1.f) return SFM_ERR_SHAPE;
  if ((lse || p_drop > 0.f) && (qkv_batch_stride != (long long)T * ldqkv)) return SFM_ERR_SHAPE;
  if (B <= 0 || T <= 0 || H <= 0 || hd <= 0 || hd > 256) return SFM_ERR_SHAPE;
  if ((dtype != SFM_DT_BF16 && dtype != SFM_DT_F16) || (out_dtype != SFM_DT_BF16 && out_dtype != SFM_DT_F16)) return SFM_ERR_ARG;
  const int out_other = (out_dtype != dtype) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  if (hd == 64 && (ldqkv % 8) == 0 && (ldo % 8) == 0 && (koff % 8) == 0 && (voff % 8) == 0 &&
      (qkv_batch_stride % 8) == 0 && (o_batch_stride % 8) == 0) {
    // scale <= 0: Q already carries softmax_scale * log2(e) (folded into W_q by the caller)
    float sl2 = (scale > 0.f) ? scale * 1.44269504088896340736f : 1.0f;
    // pipelined persistent kernel (attention_pipe.hip): by measured shape window (tools/attn_ab.py, profiles/README.md) - query
    // tiles of 512 rows (one 8-wave workgroup per CU) when at least 78 % of the tile rows are real (T 400..512, 800..1024) and
    // for every T >= 1024 (its step loop is 15-20 % faster than the 32-rows-per-wave kernel's: 1.0 PFLOP/s at T 6001); tiles
    // of 256 rows (two 4-wave workgroups per CU) for T 231..256; the 32-rows-per-wave kernel otherwise and for short launches
    const long long bytes_q = (long long)T * ldqkv * 2, bytes_o = (long long)T * ldo * 2;
    const int nqt5 = (T + 511) / 512;
    const bool enough = (long long)B * H * nqt5 >= 128;
    const bool pipe8_auto = enough && (T >= 1024 || 100 * T >= 78 * 512 * nqt5);
    const bool pipe4_auto = enough && !pipe8_auto && T <= 256 && 100 * T >= 90 * 256;
    const bool ring_auto = pipe8_auto || pipe4_auto;
    if (p_drop == 0.f && bytes_q < (1LL << 31) && bytes_o < (1LL << 31) &&
        (sfm_attn_variant >= 4 || (sfm_attn_variant == 0 && ring_auto)))
      return sfm_attn_pipe_launch(qkv, out, lse, B, T, H, ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride, sl2, dtype,
                                  out_other, sfm_attn_variant == 6 ? 44 : ((sfm_attn_variant == 5 || (sfm_attn_variant == 0 && pipe4_auto)) ? 4 : 8), st);
    if (p_drop == 0.f && bytes_q < (1LL << 31) && bytes_o < (1LL << 31) && sfm_attn_variant == 3) {
      const int n_items = nqt5 * H * B;
      // (CU count and the dynamic-LDS attribute are per device: caches keyed by hipGetDevice())
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SFM_ERR_LAUNCH;
      static int ncus[64] = {0};
      if (ncus[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SFM_ERR_LAUNCH;
        ncus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      }
      const int ncu = ncus[dev];
      constexpr int lds = 6 * 16384 + 65536;                        // K/V ring + Q prefetch region = 160 KB
      static bool attr_set[64][2] = {{false, false}};
      const int ti = dtype == SFM_DT_F16 ? 1 : 0;
      if (!attr_set[dev][ti]) {
        const void* fn = ti ? (const void*)attn_fwd_hd64r_kernel<F16> : (const void*)attn_fwd_hd64r_kernel<BF16>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return SFM_ERR_LAUNCH;
        attr_set[dev][ti] = true;
      }
      dim3 gridr(n_items < ncu ? n_items : ncu), blockr(512);
      if (dtype == SFM_DT_F16)
        SFM_LAUNCH((attn_fwd_hd64r_kernel<F16>), gridr, blockr, lds, st, (const u16*)qkv, (u16*)out, T, ldqkv, ldo, koff, voff,
                   qkv_batch_stride, o_batch_stride, sl2, nqt5, H, n_items, lse, out_other);
      else
        SFM_LAUNCH((attn_fwd_hd64r_kernel<BF16>), gridr, blockr, lds, st, (const u16*)qkv, (u16*)out, T, ldqkv, ldo, koff, voff,
                   qkv_batch_stride, o_batch_stride, sl2, nqt5, H, n_items, lse, out_other);
      return SFM_OK;
    }
    const int nqt = (T + 127) / 128;
    dim3 grid(nqt * H * B), block(256);
#define ATTN_GO(TT, DD)                                                                                            \
  SFM_LAUNCH((attn_fwd_hd64_kernel<TT, DD>), grid, block, 0, st, (const u16*)qkv, (u16*)out, T, ldqkv, ldo, koff, voff, \
             qkv_batch_stride, o_batch_stride, sl2, nqt, H, lse, p_drop, seed, out_other)
    if (dtype == SFM_DT_F16) { if (p_drop > 0.f) ATTN_GO(F16, true); else ATTN_GO(F16, false); }
    else { if (p_drop > 0.f) ATTN_GO(BF16, true); else ATTN_GO(BF16, false); }
#undef ATTN_GO
  } else {
    if (out_other) return SFM_ERR_SHAPE;                  // the small-shape kernel writes the operands' format only
    dim3 grid((T + 3) / 4, H, B), block(256);
    if (scale <= 0.f) scale = 0.69314718055994530942f;      // pre-scaled Q carries log2(e): exp(x ln2) = 2^x
    if (dtype == SFM_DT_F16)
      SFM_LAUNCH((attn_fwd_generic_kernel<F16>), grid, block, 0, st, (const u16*)qkv, (u16*)out, T, hd, ldqkv,
                         ldo, koff, voff, qkv_batch_stride, o_batch_stride, scale, lse, p_drop, seed);
    else
      SFM_LAUNCH((attn_fwd_generic_kernel<BF16>), grid, block, 0, st, (const u16*)qkv, (u16*)out, T, hd,
                         ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride, scale, lse, p_drop, seed);
  }
  SFM_CHECK_LAUNCH();
  return SFM_OK;
}

extern "C" int sfm_attention_fwd(const void* qkv, void* out, int B, int T, int H, int hd, int ldqkv, int ldo,
                                 int koff, int voff, long long qkv_batch_stride, long long o_batch_stride,
                                 float scale, int dtype, void* stream) {
  return attention_fwd_impl(qkv, out, nullptr, B, T, H, hd, ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride,
                            scale, 0.f, 0u, dtype, dtype, 0, stream);
}

// sfm_attention_fwd with the result written in `out_dtype` (SFM_DT_BF16 / SFM_DT_F16), which may differ from the operands' `dtype`,
// and with the kernel chosen by `variant` (0 = by shape)
extern "C" int sfm_attention_fwd_ex(const void* qkv, void* out, int B, int T, int H, int hd, int ldqkv, int ldo,
                                    int koff, int voff, long long qkv_batch_stride, long long o_batch_stride,
                                    float scale, int dtype, int out_dtype, int variant, void* stream) {
  return attention_fwd_impl(qkv, out, nullptr, B, T, H, hd, ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride,
                            scale, 0.f, 0u, dtype, out_dtype, variant, stream);
}

// training-mode forward: also writes lse [B,H,T] (log2 domain) and applies attention dropout
extern "C" int sfm_attention_fwd_train(const void* qkv, void* out, float* lse, int B, int T, int H, int hd, int ldqkv,
                                       int ldo, int koff, int voff, long long qkv_batch_stride,
                                       long long o_batch_stride, float scale, float p_drop, unsigned int seed,
                                       int dtype, void* stream) {
  return attention_fwd_impl(qkv, out, lse, B, T, H, hd, ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride, scale,
                            p_drop, seed, dtype, dtype, 0, stream);
}
