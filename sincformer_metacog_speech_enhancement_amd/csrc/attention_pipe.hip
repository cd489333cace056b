// attn_fwd_hd64p: the head_dim-64 attention core (models/conformer.py:66-71 -> nn.MultiheadAttention's softmax(QK^T/sqrt(hd)) V)
// as a persistent, three-stage software pipeline.  One 512-thread workgroup per CU walks (batch, head, 512-query tile) items;
// K/V by LDS-DMA into a ring of 3 groups x 2 key tiles, one barrier per group; the next item's Q rows prefetched; O transposed
// through LDS and stored as whole rows.  Everything a wave does between two groups - the refill of the ring (the slot it
// refills was consumed TWO groups ago, so no barrier is needed first), and at the end of an item the next Q fragments, the
// pipeline drain, normalisation, the O stores and the re-initialisation - comes BEFORE the barrier: the wave of a SIMD that
// finishes its steps first (static priority) does this work under its partner's MFMA steps instead of idling at the barrier
// and then doing it in lockstep with the partner (in-kernel stamps: 18 % of a wave's time was such lockstep work).
//
//   * a wave owns 64 query rows = two 32-row sub-blocks U = 0, 1 and walks the keys in 32-key steps; one "ritem" per
//     (step, sub-block).  Ritem r issues, in ONE basic block,
//        S^T(r)   = K Q^T - m           4 x v_mfma_f32_32x32x16 + one augmented k-step (K side [1, 1, pad, 0..], Q side
//                                       [-m_hi, -m_lo, -BIG, 0..]): the matrix core subtracts the running maximum and pushes
//                                       padding keys (>= T) to -BIG.  (The alternative - -m as the C operand of the first
//                                       MFMA - needs a 16-register block per sub-block; the kernel has no registers to spare.)
//        P(r-1)   = exp2(S^T(r-1))      16 v_exp_f32 + 8 packing converts on the VALU, other sub-block
//        O^T(r-2) += V^T P^T(r-2)       4 x v_mfma_f32_32x32x16 + the row sums l += 1^T P as 2 x v_mfma_f32_16x16x32 against
//                                       a lane-patterned ones operand (each lane gets the sum of ITS query over both lane
//                                       halves: 4 accumulator registers instead of 16, half the matrix-pipe time)
//     so every MFMA has independent VALU work beside it and nothing in the block waits for a result produced in it.
//   * overflow check instead of a row maximum per block: P is packed 16-bit and >= 0, so "some P >= 2" is bit 14 of either
//     half (top exponent bit in bf16 and in fp16): OR of the 8 packed registers, one AND, one compare.  Only then - a rare,
//     wave-uniform branch, also forced on the first block of a sub-block - the true row maxima are computed, m is raised to
//     max + 3 (log2 units: P <= 1/8 afterwards, so the branch fires when a score exceeds the running maximum by 4), O and l
//     are rescaled and P(r-1) is recomputed.  P(r-2) entered O at the old scale before the rescale: consistent.
//   * steps made of padding keys only (T mod 64 in 1..32) are skipped.
//   * K fragments of step s+1 and V^T fragments of step s are read from LDS under the second ritem of step s (inline asm,
//     counted lgkmcnt): no LDS latency in front of an MFMA chain.
#include "sfm_common.h"
#include <stdlib.h>

#ifndef SFM_ATTNP_ABL
#define SFM_ATTNP_ABL 0
#endif
typedef __attribute__((address_space(3))) void* attnp_lds_ptr_t;

// timing experiments (results wrong): SFM_ATTNP_ABL 9 = the 16 exponentials of a ritem replaced by plain VALU adds (what do the
// quarter-rate transcendentals cost?), 10 = no LDS fragment reads inside the step loop (stale K / V^T fragments), 11 = both
#if SFM_ATTNP_ABL == 9 || SFM_ATTNP_ABL == 11
#define ATTNP_EXP2(x) __builtin_amdgcn_fmed3f((x), 0.0f, 0.25f)      /* one plain VALU op, values stay below the rescale threshold */
#else
#define ATTNP_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif

__device__ __forceinline__ float attnp_xhalf_max(float v) {
  // v_permlane32_swap exchanges lanes 32-63 of its first operand with lanes 0-31 of the second (s_nop: VALU write -> swap)
  uint32_t a = __builtin_bit_cast(uint32_t, v), c = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, c));
}

template <class T>
__device__ __forceinline__ uint32_t attnp_pack2_o(float lo, float hi, bool other) {
  if (T::id == SFM_DT_BF16) return other ? F16::pack(lo, hi) : BF16::pack(lo, hi);
  return other ? BF16::pack(lo, hi) : F16::pack(lo, hi);
}

#define ATTNP_HEADROOM 3.0f

// diagnostic build (-DSFM_ATTNP_STAMPS, tools/attnp_stamps.py): lse_out becomes a stamp buffer, 8 x uint64 per wave:
// kernel start, kernel end, and the sums of s_memtime differences over [wait + barrier], [group-0 prologue], [step loops],
// [drain + normalise].  No stamp executes in the real kernel.
#ifdef SFM_ATTNP_STAMPS
__device__ __forceinline__ unsigned long long attnp_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define ATTNP_T(...) __VA_ARGS__
#else
#define ATTNP_T(...)
#endif

// NW = waves per workgroup (8: one workgroup per CU, items of 512 query rows, ring of 3 x 2 key tiles; 4: TWO workgroups per
// CU, items of 256 query rows, ring of 3 x 1 key tile, 80 KB of LDS each - the two waves of a SIMD then belong to different
// workgroups and never wait for each other at a barrier; K/V are streamed twice per 512 query rows instead of once)
// SB = 32-row sub-blocks per wave: 2 (64 query rows per wave; NW 8 or 4) or 4 (128 rows per wave, NW 4: ONE wave per SIMD with the whole
// 512-register file - round 4: a lone wave of the 2-sub-block form sustains 83 % of what two waves per SIMD do together (0.125
// against 0.104 ms with half the waves), i.e. the step loop is bound by each wave's own issue / dependency chain, not by a shared
// pipe; four independent sub-block chains in one wave give that chain twice the slack, every K / V^T fragment read serves twice
// the rows, and nothing is arbitrated between SIMD partners)
template <class T, int NW, int SB>
__device__ __forceinline__ void attnp_body(const u16* __restrict__ qkv, u16* __restrict__ out, int Tlen, int ldqkv, int ldo, int koff,
                                           int voff, long long qkv_batch_stride, long long o_batch_stride, float scale_log2e,
                                           int nqt, int nheads, int n_items, float* __restrict__ lse_out, int out_other) {
  constexpr int SLOT = 16384;                                       // one key tile: K 64 x 128 B, then V 64 x 128 B
  constexpr int GT = NW * SB / 8;                                   // key tiles per group
  constexpr int NSLOT = 3;                                          // group slots of the ring
  constexpr int RW = 32 * SB;                                       // query rows of a wave
  constexpr int QBASE = NSLOT * GT * SLOT;                          // Q prefetch region: NW waves x RW rows x 128 B
  constexpr int QROWS = RW * NW;                                    // query rows of an item
  constexpr int QWB = RW * 128;                                     // bytes of a wave's part of the Q region
  constexpr int PPG = GT * 2 * (64 / NW / 8);                       // LDS-DMA pieces a wave issues for a whole group
  constexpr int RPW = 64 / NW;                                      // rows of a key tile this wave fetches (8 per DMA instruction)
  extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if defined(SFM_ATTNP_STAGGER)
  // timing experiment: the second resident workgroup of a CU starts SFM_ATTNP_STAGGER x 1024 cycles late, so that the two
  // workgroups' item epilogues do not coincide
  if (NW == 4 && blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < SFM_ATTNP_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
#endif
  const int hl = lane >> 5, l31 = lane & 31;
  const int nkt = (Tlen + 63) >> 6, ngrp = (nkt + GT - 1) / GT, nsteps = (Tlen + 31) >> 5;
  const int rec_bytes = Tlen * ldqkv * 2;                          // keys / queries >= Tlen are out of range: read as zero
  const int orec_bytes = Tlen * ldo * 2;

  // ---- LDS-DMA lane constants: an instruction moves 8 rows x 128 B; this wave owns rows 8*wave .. 8*wave+7 of every tile ----
  // (K chunk swizzle = (row >> 1) & 7: period 16 rows, so each of the wave's 8-row pieces has its own lane constant; the V
  //  swizzle ((row >> 1) & 1) << 2 has period 4: one constant, the piece enters through the row offset)
  const int prow = wave * RPW + (lane >> 3);
  int kconst[RPW / 8];
#pragma unroll
  for (int pc = 0; pc < RPW / 8; ++pc)
    kconst[pc] = (prow + 8 * pc) * ldqkv * 2 + koff * 2 + (((lane & 7) ^ (((prow + 8 * pc) >> 1) & 7)) << 4);
  const int vconst = prow * ldqkv * 2 + voff * 2 + (((lane & 7) ^ (((prow >> 1) & 1) << 2)) << 4);
  // (issuing these pieces one tile per step from inside the step loop instead of as a burst behind the barrier was tried:
  //  6 % slower - a piece issued between ritems costs the in-order wave more than the same piece in a burst)
  auto issue_group = [&](int item, int g, int half) -> int {       // K/V tiles GT g .. GT g + GT-1 of `item` -> ring slot `half`; returns the DMA pieces issued
    int np = 0;
    const int bh = item / nqt;
    const int h = bh % nheads, b = bh / nheads;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (long long)b * qkv_batch_stride), 0, rec_bytes, 0x00020000);
#pragma unroll
    for (int tl = 0; tl < GT; ++tl) {
      const int kt = g * GT + tl;
      if (kt < nkt) {
        const int off = kt * 64 * ldqkv * 2 + h * 128;
        unsigned char* dst = rsm + (half * GT + tl) * SLOT + wave * (RPW * 128);
#pragma unroll
        for (int pc = 0; pc < RPW / 8; ++pc) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (attnp_lds_ptr_t)(dst + pc * 1024), 16, kconst[pc] + off, 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (attnp_lds_ptr_t)(dst + pc * 1024 + 8192), 16, vconst + off + pc * 8 * ldqkv * 2, 0, 0, 0);
        }
        np += 2 * (RPW / 8);
      }
    }
    return np;
  };
  // Q rows of `item` for this wave (64 rows x 128 B, same chunk swizzle as a K tile) -> the wave's 8 KB of the Q region
  auto issue_q = [&](int item) {
    const int qt = item % nqt, bh = item / nqt;
    const int h = bh % nheads, b = bh / nheads;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (long long)b * qkv_batch_stride), 0, rec_bytes, 0x00020000);
    const int off = (qt * QROWS + wave * RW) * ldqkv * 2 + h * 128;
#pragma unroll
    for (int j = 0; j < RW / 8; ++j) {
      const int c = ((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) << 4;
      const int vo = off + (8 * j + (lane >> 3)) * ldqkv * 2 + c;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (attnp_lds_ptr_t)(rsm + QBASE + wave * QWB + j * 1024), 16, vo, 0, 0, 0);
    }
  };

  // ---- fragment read lane constants (byte offsets inside a 32-key step: K rows at +0, V rows at +8192) ----
  const uint32_t lds0 = (uint32_t)(uintptr_t)(attnp_lds_ptr_t)rsm;
  uint32_t klane[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) klane[ks] = lds0 + l31 * 128 + (((2 * ks + hl) ^ ((l31 >> 1) & 7)) << 4);
  const int g16 = lane >> 4, i16 = lane & 15;
  const int vrow0 = 4 * (g16 >> 1) + (i16 >> 2);
  uint32_t vlane[2];
#pragma unroll
  for (int dj = 0; dj < 2; ++dj) {
    const int chunk = dj * 4 + (g16 & 1) * 2 + ((i16 & 3) >> 1);
    vlane[dj] = lds0 + 8192 + vrow0 * 128 + ((chunk ^ (((vrow0 >> 1) & 1) << 2)) << 4) + (i16 & 1) * 8;
  }

  // ones operand of the row-sum MFMA (16x16x32, A side): row i of the result sums the lane groups G with G&1 == (i>>2)&1, so
  // that lane l (result rows 4*(l>>4)..+3, column l&15) receives the sum over both lane halves of query l&31
  const uint32_t one16 = T::from_f32(1.0f);
  const uint32_t ones2 = (((lane >> 4) & 1) == ((lane >> 2) & 1)) ? (one16 | (one16 << 16)) : 0u;
  const u32x4 onesA = {ones2, ones2, ones2, ones2};
  const uint32_t ones2k = one16 | (one16 << 16);                    // augmented k-step, K side
  const uint32_t negbig = (uint32_t)T::from_f32(T::id == SFM_DT_F16 ? -60000.0f : -3.0e38f);

  // All LDS fragment reads are inline asm with hand-counted lgkmcnt waits: behind an LDS-DMA the compiler guards its own LDS
  // reads with vmcnt(0) (it cannot prove they do not alias the DMA destination) and it sinks early reads down to their use.
  // The ring protocol (vmcnt + barrier at the group boundary) is what orders reads against fills.
  u32x4 kf[4];
  u32x2 vt[2][2][2];                                                // V^T fragment halves [s2][dj][first / second 4 keys]
  auto load_kf = [&](int sbase) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(kf[ks]) : "v"(klane[ks] + sbase) : "memory");
  };
  auto load_vf = [&](int sbase) {
#pragma unroll
    for (int dj = 0; dj < 2; ++dj) {
      const uint32_t a = vlane[dj] + sbase;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[0][dj][0]) : "v"(a) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(vt[0][dj][1]) : "v"(a) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(vt[1][dj][0]) : "v"(a) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:3072" : "=v"(vt[1][dj][1]) : "v"(a) : "memory");
    }
  };
#define ATTNP_KF_WAIT(N)                                                                                               \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]) : : "memory")
#define ATTNP_VT_WAIT()                                                                                                \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
               : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]), "+v"(vt[1][0][0]),        \
                 "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1])                                               \
               :                                                                                                       \
               : "memory")
#define ATTNP_VF(S2, DJ) (u32x4{vt[S2][DJ][0][0], vt[S2][DJ][0][1], vt[S2][DJ][1][0], vt[S2][DJ][1][1]})

  u32x4 qf[SB][4];
  float m_run[SB];
  // augmented k-step operands, kept as whole 4-register tuples (assembling them per ritem costs two moves and a hazard nop):
  // Q side (-m_hi, -m_lo | -BIG, 0 | 0...) per sub-block, [0] rewritten by a rescale; K side (1, 1 | pad, 0 | 0...), [1]
  // rewritten per step (nonzero only in a last step that contains padding keys)
  u32x4 qa[SB], ka;
  f32x16 o[SB][2], s[SB];
  f32x4 lacc[SB];
  u32x4 pf[SB][2];                                                   // packed P of the pending block of each sub-block [U][s2]

  // P = exp2(S) of sub-block X, packed; FLAG |= "some P >= 2"
#define ATTNP_EXP_PACK(X, FLAGW)                                                                                       \
  {                                                                                                                    \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                                 \
      float e_[8];                                                                                                     \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) e_[r] = ATTNP_EXP2(s[X][8 * s2 + r]);                              \
      pf[X][s2][0] = pack2<T>(e_[0], e_[1]);                                                                           \
      pf[X][s2][1] = pack2<T>(e_[2], e_[3]);                                                                           \
      pf[X][s2][2] = pack2<T>(e_[4], e_[5]);                                                                           \
      pf[X][s2][3] = pack2<T>(e_[6], e_[7]);                                                                           \
    }                                                                                                                  \
    FLAGW = (pf[X][0][0] | pf[X][0][1] | pf[X][0][2]) | (pf[X][0][3] | pf[X][1][0] | pf[X][1][1]) |                    \
            (pf[X][1][2] | pf[X][1][3]);                                                                               \
  }
  // O^T(U) += V^T P^T(U), l(U) += 1^T P(U).  SB == 4 (one wave per SIMD, 512 registers): the O accumulators (128 registers)
  // and the row sums live in the ACCUMULATOR half of the register file - the only instructions that touch them are these MFMAs
  // (inline asm with "a" constraints: the file is built with -amdgpu-mfma-vgpr-form for the S chain, whose results the VALU
  // reads), the rare rescale and the item epilogue; with them in architectural VGPRs the kernel needed 452 and spilled 92.
#define ATTNP_MFMA_O(ACC, A_, B_)                                                                                      \
  {                                                                                                                    \
    if constexpr (SB == 4) {                                                                                           \
      const u32x4 a_ = (A_);                                                                                           \
      if constexpr (T::id == SFM_DT_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a_), "v"(B_)); \
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a_), "v"(B_));                      \
    } else {                                                                                                           \
      ACC = T::mfma((A_), (B_), ACC);                                                                                  \
    }                                                                                                                  \
  }
#define ATTNP_MFMA_L(ACC, B_)                                                                                          \
  {                                                                                                                    \
    if constexpr (SB == 4) {                                                                                           \
      if constexpr (T::id == SFM_DT_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(onesA), "v"(B_)); \
      else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(onesA), "v"(B_));                   \
    } else {                                                                                                           \
      ACC = T::mfma16(onesA, (B_), ACC);                                                                               \
    }                                                                                                                  \
  }
#define ATTNP_PV(U)                                                                                                    \
  {                                                                                                                    \
    ATTNP_MFMA_L(lacc[U], pf[U][0])                                                                                    \
    ATTNP_MFMA_O(o[U][0], ATTNP_VF(0, 0), pf[U][0])                                                                    \
    ATTNP_MFMA_O(o[U][1], ATTNP_VF(0, 1), pf[U][0])                                                                    \
    ATTNP_MFMA_L(lacc[U], pf[U][1])                                                                                    \
    ATTNP_MFMA_O(o[U][0], ATTNP_VF(1, 0), pf[U][1])                                                                    \
    ATTNP_MFMA_O(o[U][1], ATTNP_VF(1, 1), pf[U][1])                                                                    \
  }
  // (-m as the C operand of the first S MFMA instead of the augmented k-step was tried in round 3: one MFMA fewer per block, but a
  //  16-register block per sub-block -> 19 spills, 517 against 672 TFLOP/s)
#define ATTNP_SET_NEGM(X, HI, LO, MNEW) qa[X][0] = (hl == 0) ? pack2<T>(-(HI), -(LO)) : 0u;
  // rare path: raise the running maximum of sub-block X from the block whose scores are in s[X], rescale O and l, redo P
#define ATTNP_RESCALE(X, FORCE)                                                                                        \
  {                                                                                                                    \
    float mx = fmaxf(s[X][0], s[X][1]);                                                                                \
    _Pragma("unroll") for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[X][r]), s[X][r + 1]);                     \
    mx = attnp_xhalf_max(mx);                                                                                          \
    /* new subtracted value, split in two 16-bit terms so that the MFMA subtracts it to ~2^-17 */                      \
    const float want_ = m_run[X] + ((FORCE) ? mx + ATTNP_HEADROOM : fmaxf(mx + ATTNP_HEADROOM, 0.f));                  \
    const float hi_ = T::to_f32(T::from_f32(want_));                                                                   \
    const float lo_ = T::to_f32(T::from_f32(want_ - hi_));                                                             \
    const float m_new_ = hi_ + lo_;                                                                                    \
    const float delta = m_new_ - m_run[X];                                                                             \
    const float alpha = (FORCE) ? 0.f : __builtin_amdgcn_exp2f(-delta);                                                \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                   \
      o[X][0][r] *= alpha;                                                                                             \
      o[X][1][r] *= alpha;                                                                                             \
    }                                                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) lacc[X][r] *= alpha;                                                 \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) s[X][r] -= delta;                                                   \
    m_run[X] = m_new_;                                                                                                 \
    ATTNP_SET_NEGM(X, hi_, lo_, m_new_)                                                                                \
    uint32_t fl_;                                                                                                      \
    ATTNP_EXP_PACK(X, fl_)                                                                                             \
    (void)fl_;                                                                                                         \
  }

  // One ritem: S chain of (STEP, U); exponentials of the previous ritem's block (sub-block 1-U); PV of sub-block U's pending
  // block.  Every MFMA opens a scheduling region of its own (sched_barrier between them) that holds the VALU work and AT MOST ONE
  // LDS fragment read placed beside it: the fragment reads used to sit in two bursts (4 x ds_read_b128, then 8 x ds_read_b64_tr_b16
  // behind the step's last MFMA) and cost 15 % of the kernel (ablation "no fragment reads": 0.0853 against 0.0987 ms); a register
  // is re-read as soon as the MFMA that consumed it has issued:
  //   ritem 0 (AFIRST): V^T second half (keys 16..31 of the PREVIOUS step, base VHSB) under its S chain
  //   ritem 1 (KPRE / VLO): the NEXT step's K fragments (base KSB) under its S chain, this step's V^T first half (base VSB) under
  //                         the second half of its PV chain
  // counted waits: lgkmcnt(4) leaves the four youngest reads in flight.
// "+v" on the destination: it IS the loop-carried fragment register (with "=v" the allocator picks fresh registers and rotates
// them back with six v_mov_b64 per step).  TIE_: the accumulator of the MFMA that has just consumed the register's old contents -
// passed through the asm so that the read can neither be hoisted above that MFMA (it would need a copy of the old fragment) nor
// sink below the next one
#define ATTNP_RD_K(KS_, SB_, TIE_) asm volatile("ds_read_b128 %0, %2" : "+v"(kf[KS_]), "+v"(TIE_) : "v"(klane[KS_] + (SB_)) : "memory")
#define ATTNP_RD_V(S2_, DJ_, H_, SB_, TIE_)                                                                            \
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3" : "+v"(vt[S2_][DJ_][H_]), "+v"(TIE_) : "v"(vlane[DJ_] + (SB_)), "n"(2048 * (S2_) + 1024 * (H_)) : "memory")
// the same with the tie in the accumulator file (SB == 4: O and the row sums live there; a "+v" tie would make the compiler copy
// the accumulator out and back around the asm - right behind the MFMA that is still writing it)
#define ATTNP_RD_V_A(S2_, DJ_, H_, SB_, TIE_)                                                                          \
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3" : "+v"(vt[S2_][DJ_][H_]), "+a"(TIE_) : "v"(vlane[DJ_] + (SB_)), "n"(2048 * (S2_) + 1024 * (H_)) : "memory")
#define ATTNP_RD_V_O(S2_, DJ_, H_, SB_, TIE_)                                                                          \
  { if constexpr (SB == 4) ATTNP_RD_V_A(S2_, DJ_, H_, SB_, TIE_); else ATTNP_RD_V(S2_, DJ_, H_, SB_, TIE_); }
#define ATTNP_VT_WAIT_LO()                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]) : : "memory")
#define ATTNP_VT_WAIT_HI()                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[1][0][0]), "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1]) : : "memory")
#define ATTNP_SB() __builtin_amdgcn_sched_barrier(0);
#define ATTNP_SG(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#if SFM_ATTNP_ABL == 8
#define ATTNP_AUG(U) zero
#else
#define ATTNP_AUG(U) T::mfma(ka, qa[U], zero)
#endif
#if SFM_ATTNP_ABL == 10 || SFM_ATTNP_ABL == 11
#define ATTNP_IFRD(C) false
#else
#define ATTNP_IFRD(C) (C)
#endif
#define ATTNP_RITEM(U, STEP, AFIRST, KPRE, KSB, VLO, VSB, VHSB, FORCE)                                                 \
  {                                                                                                                    \
    constexpr int V_ = ((U) + SB - 1) % SB;                                                                            \
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};              \
    float e_[16];                                                                                                      \
    ATTNP_PRIO(U)                                                                                                      \
    ATTNP_SB()                                                                                                         \
    /* ---- the S chain: augmented k-step, then the four d-slices; 12 exponentials and 4 converts beside it ---- */    \
    f32x16 sn = ATTNP_AUG(U);                                                                                          \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) e_[r] = ATTNP_EXP2(s[V_][r]);                                        \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x400, 3)                                                                              \
    ATTNP_SB()                                                                                                         \
    if (AFIRST) ATTNP_KF_WAIT(4);                                                                                      \
    sn = T::mfma(kf[0], qf[U][0], sn);                                                                                 \
    if (ATTNP_IFRD(KPRE)) ATTNP_RD_K(0, KSB, sn);                                                                          \
    if (ATTNP_IFRD(AFIRST)) ATTNP_RD_V(1, 0, 0, VHSB, sn);                                                                 \
    _Pragma("unroll") for (int r = 3; r < 6; ++r) e_[r] = ATTNP_EXP2(s[V_][r]);                                        \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x400, 3)                                                                              \
    ATTNP_SB()                                                                                                         \
    if (SFM_ATTNP_ABL != 13) sn = T::mfma(kf[1], qf[U][1], sn);                                                        \
    if (ATTNP_IFRD(KPRE)) ATTNP_RD_K(1, KSB, sn);                                                                          \
    if (ATTNP_IFRD(AFIRST)) ATTNP_RD_V(1, 0, 1, VHSB, sn);                                                                 \
    _Pragma("unroll") for (int r = 6; r < 9; ++r) e_[r] = ATTNP_EXP2(s[V_][r]);                                        \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x400, 3)                                                                              \
    ATTNP_SB()                                                                                                         \
    if (SFM_ATTNP_ABL != 13) sn = T::mfma(kf[2], qf[U][2], sn);                                                        \
    if (ATTNP_IFRD(KPRE)) ATTNP_RD_K(2, KSB, sn);                                                                          \
    if (ATTNP_IFRD(AFIRST)) ATTNP_RD_V(1, 1, 0, VHSB, sn);                                                                 \
    _Pragma("unroll") for (int r = 9; r < 12; ++r) e_[r] = ATTNP_EXP2(s[V_][r]);                                       \
    pf[V_][0][0] = pack2<T>(e_[0], e_[1]);                                                                             \
    pf[V_][0][1] = pack2<T>(e_[2], e_[3]);                                                                             \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x400, 3) ATTNP_SG(0x002, 2)                                                           \
    ATTNP_SB()                                                                                                         \
    if (SFM_ATTNP_ABL != 13) sn = T::mfma(kf[3], qf[U][3], sn);                                                        \
    if (ATTNP_IFRD(KPRE)) ATTNP_RD_K(3, KSB, sn);                                                                          \
    if (ATTNP_IFRD(AFIRST)) ATTNP_RD_V(1, 1, 1, VHSB, sn);                                                                 \
    pf[V_][0][2] = pack2<T>(e_[4], e_[5]);                                                                             \
    pf[V_][0][3] = pack2<T>(e_[6], e_[7]);                                                                             \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x002, 2)                                                                              \
    /* order at the IR level too (MFMAs have no side effects: without a data tie the PV chain may be emitted first and   \
       the sched_barrier then freezes that order): the PV operands pass through an empty asm that also takes sn */        \
    asm volatile("" : "+v"(sn), "+v"(pf[U][0]), "+v"(pf[U][1]));                                                       \
    ATTNP_SB()                                                                                                         \
    /* ---- PV of the pending block, the remaining exponentials, the overflow test ---- */                             \
    if (AFIRST) ATTNP_VT_WAIT_LO();                                                                                    \
    ATTNP_MFMA_L(lacc[U], pf[U][0])                                                                                    \
    e_[12] = ATTNP_EXP2(s[V_][12]);                                                                                    \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x400, 1)                                                                              \
    ATTNP_SB()                                                                                                         \
    ATTNP_MFMA_O(o[U][0], ATTNP_VF(0, 0), pf[U][0])                                                                    \
    _Pragma("unroll") for (int r = 13; r < 16; ++r) e_[r] = ATTNP_EXP2(s[V_][r]);                                      \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x400, 3)                                                                              \
    ATTNP_SB()                                                                                                         \
    ATTNP_MFMA_O(o[U][1], ATTNP_VF(0, 1), pf[U][0])                                                                    \
    pf[V_][1][0] = pack2<T>(e_[8], e_[9]);                                                                             \
    pf[V_][1][1] = pack2<T>(e_[10], e_[11]);                                                                           \
    pf[V_][1][2] = pack2<T>(e_[12], e_[13]);                                                                           \
    pf[V_][1][3] = pack2<T>(e_[14], e_[15]);                                                                           \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x002, 4)                                                                              \
    ATTNP_SB()                                                                                                         \
    if (AFIRST) ATTNP_VT_WAIT_HI();                                                                                    \
    ATTNP_MFMA_L(lacc[U], pf[U][1])                                                                                    \
    if (ATTNP_IFRD(VLO)) { ATTNP_RD_V_O(0, 0, 0, VSB, lacc[U]) ATTNP_RD_V_O(0, 0, 1, VSB, lacc[U]) }                                       \
    uint32_t flag_ = (pf[V_][0][0] | pf[V_][0][1] | pf[V_][0][2]) | (pf[V_][0][3] | pf[V_][1][0] | pf[V_][1][1]);      \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x002, 2)                                                                              \
    ATTNP_SB()                                                                                                         \
    ATTNP_MFMA_O(o[U][0], ATTNP_VF(1, 0), pf[U][1])                                                                    \
    if (ATTNP_IFRD(VLO)) { ATTNP_RD_V_O(0, 1, 0, VSB, o[U][0]) ATTNP_RD_V_O(0, 1, 1, VSB, o[U][0]) }                                       \
    flag_ |= pf[V_][1][2] | pf[V_][1][3];                                                                              \
    ATTNP_SG(0x008, 1) ATTNP_SG(0x002, 1)                                                                              \
    ATTNP_SB()                                                                                                         \
    ATTNP_MFMA_O(o[U][1], ATTNP_VF(1, 1), pf[U][1])                                                                    \
    ATTNP_SB()                                                                                                         \
    s[U] = sn;                                                                                                         \
    /* the forced case enters through the same data-dependent test (a short-circuit on FORCE lets the compiler sink the   \
       exponentials out of this block, behind the branch) */                                                           \
    if (__any((((SFM_ATTNP_ABL == 8) ? (flag_ & 0u) : flag_) | ((FORCE) ? 0x4000u : 0u)) & 0x40004000u) != 0u)          \
      ATTNP_RESCALE(V_, FORCE)                                                                                         \
  }

#if SFM_ATTNP_ABL == 2
  // experiment: the two waves of a SIMD take turns at priority 1, one ritem each
#define ATTNP_PRIO(U) if ((wave >= 4) == ((U) == 1)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#else
#define ATTNP_PRIO(U)
#endif

  // ---- O of the finished item: transposed through the wave's own 8 KB of the Q region (the next item's Q fragments have
  //      been read out of it) so that every store instruction writes 8 whole 128-byte rows ----
  uint32_t ow[SB][2][4][2];                                         // [sub-block][dj][rq][2 dwords] = 4 consecutive d, 16-bit
  auto store_o = [&](int st_item) {
    const int qt = st_item % nqt, bh = st_item / nqt;
    const int h = bh % nheads, b = bh / nheads;
    auto ors = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long long)b * o_batch_stride), 0, orec_bytes, 0x00020000);
    unsigned char* ob = rsm + QBASE + wave * QWB;
    // the lane id is made opaque so that the per-lane addresses below are computed HERE, once per item (hoisted to kernel
    // entry they live across the whole tile loop and get spilled)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int l31o = ln & 31, hlo = ln >> 5;
#pragma unroll
    for (int u = 0; u < SB; ++u) {
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
    const int qbase = qt * QROWS + wave * RW;
    // all row reads first (one LDS round trip; the compiler otherwise pairs read, read, store, store: four round trips)
    u32x4 rv[RW / 8];
#pragma unroll
    for (int i = 0; i < RW / 8; ++i) {
      const int c = ln + 64 * i;
      const int row = c >> 3, ch = c & 7;
      rv[i] = *reinterpret_cast<const u32x4*>(ob + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RW / 8; ++i) {
      const int c = ln + 64 * i;
      const int row = c >> 3, ch = c & 7;
      __builtin_amdgcn_raw_buffer_store_b128(rv[i], ors, (qbase + row) * ldo * 2 + h * 128 + ch * 16, 0, 0);
    }
  };
  // Q fragments (B operand: col = query, k = d) of the item whose rows are in the wave's Q region (rows >= Tlen zero-filled)
  auto load_qf = [&]() {
#pragma unroll
    for (int u = 0; u < SB; ++u)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        qf[u][ks] = *reinterpret_cast<const u32x4*>(rsm + QBASE + wave * QWB + u * 4096 + (klane[ks] - lds0));
    if (scale_log2e != 1.0f) {                                     // callers normally fold the scale into W_q (scale_log2e == 1)
#pragma unroll
      for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t w = qf[u][ks][e];
            qf[u][ks][e] = pack2<T>(T::to_f32((u16)(w & 0xffffu)) * scale_log2e, T::to_f32((u16)(w >> 16)) * scale_log2e);
          }
    }
    // lgkmcnt(0) as a BUILTIN: the compiler must know that its Q loads have landed (behind an asm wait they stay "pending"
    // in its model and it then waits in every ritem).  The Q region may be reused from here on.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
  };
  // pipeline start: nothing pending -> P = 0 (s = -1e30 exponentiates to 0), V^T fragments zero (0 x stale data)
  auto init_state = [&]() {
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      m_run[u] = 0.f;
      qa[u] = u32x4{0u, hl == 0 ? negbig : 0u, 0u, 0u};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[u][0][r] = 0.f;
        o[u][1][r] = 0.f;
        s[u][r] = -1.0e30f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) lacc[u][r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) pf[u][s2] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        vt[s2][dj][0] = u32x2{0u, 0u};
        vt[s2][dj][1] = u32x2{0u, 0u};
      }
  };

  // the second-dispatched half of the workgroup loses the VALU arbitration against its SIMD partner (priority, then age)
#if SFM_ATTNP_ABL != 1 && SFM_ATTNP_ABL != 2
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  // ring bookkeeping: `cslot` = slot of the group being consumed; the prefetch cursor (p_item, p_g, pslot) names the next
  // group to fetch; it runs two groups ahead of the consumer
  // XCD-aware item order: workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of virtual ids so
  // that the query tiles of one (batch, head) - consecutive items - share an L2 (placement is a speed matter only)
  const int nwg = gridDim.x;
  const int vid = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  int cslot = 0, pslot = 0;
  int p_item = vid, p_g = 0;
  auto issue_next_group = [&]() -> int {
    if (p_item >= n_items) return 0;
    const int np = issue_group(p_item, p_g, pslot);
    pslot = (pslot == NSLOT - 1) ? 0 : pslot + 1;
    if (++p_g == ngrp) { p_g = 0; p_item += gridDim.x; }
    return np;
  };
  int vprev = -1;                                                   // LDS base of the step whose V^T second half has not been read yet
  bool qf_ready = false;                                            // Q fragments of the coming item already in registers
  int inflight = 0;                                                 // vector-memory operations issued since the pieces the next barrier waits for
  bool first_q_pending = false;
  if (vid < n_items) {
    // cold start: the first MFMA needs Q and the first group only - the second group's pieces stay in flight across the
    // first barrier (every CU fetches at once here: 96 KB instead of 128 KB before the first instruction of work)
    issue_q(vid);
    const int n0 = issue_next_group();
    inflight = issue_next_group();
    first_q_pending = (n0 + inflight == 2 * PPG);                   // both groups whole: the counted wait below is exact
  }
  const int pad_step = (Tlen & 31) ? nsteps - 1 : -1;               // the only step that can contain padding keys
  ka = u32x4{hl == 0 ? ones2k : 0u, 0u, 0u, 0u};
  ATTNP_T(unsigned long long t_bar = 0, t_pre = 0, t_steps = 0, t_post = 0, t_last = 0, t_q = 0, t_so = 0, t_dr = 0; const unsigned long long t_start = attnp_stamp();
          unsigned long long* dbg = reinterpret_cast<unsigned long long*>(lse_out) + ((size_t)blockIdx.x * NW + wave) * 8; lse_out = nullptr;)

  // ---- group boundary: this wave's pieces of the group to consume have landed - counted vmcnt: the `inflight` youngest
  //      operations (this wave's refill of another slot, Q prefetch, O stores; all issued after those pieces) may stay in
  //      flight - then everyone's have (barrier; its lgkmcnt(0) also covers the V^T reads of the last step, issued before).
  //      Then the K fragments of the group's first step. ----
#define ATTNP_BOUNDARY()                                                                                               \
  ATTNP_T(const unsigned long long tb0_ = attnp_stamp();)                                                              \
  if (inflight >= 32) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");                                     \
  else if (inflight >= 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");                                \
  else if (inflight >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");                                \
  else if (inflight >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");                                \
  else if (inflight >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                  \
  else if (inflight >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                  \
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                     \
  __builtin_amdgcn_s_barrier();            /* raw: __syncthreads() would drain vmcnt to 0 (refill and O stores in flight) */ \
  asm volatile("" ::: "memory");                                                                                       \
  inflight = 0;                                                                                                        \
  ATTNP_T(const unsigned long long tb1_ = attnp_stamp(); t_bar += tb1_ - tb0_;)                                        \
  load_kf(cslot * GT * SLOT);
  // the steps of group `g` (the K prefetch of a step beyond the group reads stale ring / Q-region bytes that are never
  // used: the next group's first fragments are read after its barrier)
#define ATTNP_STEPS()                                                                                                  \
  ATTNP_KF_WAIT(0);                                                                                                    \
  ATTNP_T(const unsigned long long ts0_ = attnp_stamp();)                                                              \
  {                                                                                                                    \
    const int step_end = min(nsteps, (g + 1) * 2 * GT);                                                                \
    for (int step = g * 2 * GT; step < step_end; ++step) {                                                             \
      const int sbk = cslot * GT * SLOT + ((step >> 1) - g * GT) * SLOT + (step & 1) * 4096;   /* K rows; V at +8192 */  \
      const int sbn = cslot * GT * SLOT + (((step + 1) >> 1) - g * GT) * SLOT + ((step + 1) & 1) * 4096;               \
      /* second half of the previous step's V^T (first step of an item: nothing is pending - P = 0 -, any finite data will   \
         do: this step's own rows) */                                                                                  \
      const int vhs = (vprev >= 0) ? vprev : sbk;                                                                      \
      ka[1] = (step == pad_step && hl == 0 && step * 32 + l31 >= Tlen) ? one16 : 0u;                                   \
      /* FORCE = the block whose exponentials this ritem takes is the FIRST block of its sub-block: (step 0, U - 1), or     \
         (step 0, SB - 1) in ritem (1, 0) */                                                                           \
      if constexpr (SB == 2) {                                                                                         \
        ATTNP_RITEM(0, step, true, false, 0, false, 0, vhs, step == 1)                                                 \
        ATTNP_RITEM(1, step, false, true, sbn, true, sbk, 0, step == 0)                                                \
      } else {                                                                                                         \
        ATTNP_RITEM(0, step, true, false, 0, false, 0, vhs, step == 1)                                                 \
        ATTNP_RITEM(1, step, false, false, 0, false, 0, 0, step == 0)                                                  \
        ATTNP_RITEM(2, step, false, false, 0, false, 0, 0, step == 0)                                                  \
        ATTNP_RITEM(3, step, false, true, sbn, true, sbk, 0, step == 0)                                                \
      }                                                                                                                \
      vprev = sbk;                                                                                                     \
    }                                                                                                                  \
  }                                                                                                                    \
  ATTNP_T(t_last = attnp_stamp(); t_steps += t_last - ts0_;)

  for (int item = vid; item < n_items; item += gridDim.x) {
    const int qt = item % nqt, bh = item / nqt;
    const int h = bh % nheads, b = bh / nheads;
    const int q0 = qt * QROWS + wave * RW;
    const bool has_next = item + (int)gridDim.x < n_items;
    if (!qf_ready) {
      // first item of the workgroup (or single-group items, below): this wave's own Q pieces have landed -> fragments
      if (first_q_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPG) : "memory");      // the youngest = the two K/V groups
      else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); inflight = 0; }
      first_q_pending = false;
      load_qf();
      init_state();
    }
    vprev = -1;
    for (int g = 0; g < ngrp; ++g) {
      ATTNP_BOUNDARY()
      ATTNP_STEPS()
      // ---- between groups, BEFORE the barrier: refill the slot that was consumed two groups ago; after the first group
      //      of an item also fetch the next item's Q rows (the wave's Q region is free: the O read-back finished long ago) ----
      if (g == 0 && ngrp > 1 && has_next) {
        issue_q(item + gridDim.x);
        inflight += RW / 8;
      }
      if (g + 1 < ngrp) inflight += issue_next_group();
      cslot = (cslot == NSLOT - 1) ? 0 : cslot + 1;
    }
    // ---- end of the item ----
    qf_ready = false;
    if (ngrp > 1 && has_next) {
      // the next item's Q fragments: its rows were fetched a group or more ago (operations issued after them: >= one refill)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");             // (a refill is 2 or 4 pieces)
#pragma unroll
      for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          qf[u][ks] = *reinterpret_cast<const u32x4*>(rsm + QBASE + wave * QWB + u * 4096 + (klane[ks] - lds0));
      qf_ready = true;
    }
    // ---- drain: exponentials of the last block of the last sub-block, then the pending PV products of every sub-block ----
    {
      uint32_t flag_;
      ATTNP_EXP_PACK(SB - 1, flag_)
      const bool force1 = nsteps == 1;
      if (__any(((flag_ | (force1 ? 0x4000u : 0u)) & 0x40004000u) != 0u)) ATTNP_RESCALE(SB - 1, force1)
      if (ATTNP_IFRD(true)) {                                       // second half of the last step's V^T
        ATTNP_RD_V_O(1, 0, 0, vprev, lacc[0]) ATTNP_RD_V_O(1, 0, 1, vprev, lacc[0]) ATTNP_RD_V_O(1, 1, 0, vprev, lacc[0]) ATTNP_RD_V_O(1, 1, 1, vprev, lacc[0])
      }
      ATTNP_VT_WAIT();
      ATTNP_PV(0)
      ATTNP_PV(1)
      if constexpr (SB == 4) {
        ATTNP_PV(2)
        ATTNP_PV(3)
      }
    }
    ATTNP_T(t_dr += attnp_stamp() - t_last;)
    // ---- normalise and pack; the stores themselves are issued after the next barrier (store_o) ----
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const float l = lacc[u][0];
      const float inv = __builtin_amdgcn_rcpf(l);                  // 1 ulp; O is rounded to 16 bits
      const int q = q0 + 32 * u + l31;
      if (lse_out && hl == 0 && q < Tlen)
        lse_out[((long long)b * nheads + h) * Tlen + q] = m_run[u] + __builtin_amdgcn_logf(l);
#pragma unroll
      for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[u][dj][r] *= inv;
    }
    // ONE branch on the output format around the 32 packing converts (a select per convert compiles to a branch per convert)
#define ATTNP_PACK_O(PK)                                                                                               \
  _Pragma("unroll") for (int u = 0; u < SB; ++u)                                                                       \
  _Pragma("unroll") for (int dj = 0; dj < 2; ++dj)                                                                     \
  _Pragma("unroll") for (int rq = 0; rq < 4; ++rq) {                                                                   \
    ow[u][dj][rq][0] = PK(o[u][dj][4 * rq + 0], o[u][dj][4 * rq + 1]);                                                 \
    ow[u][dj][rq][1] = PK(o[u][dj][4 * rq + 2], o[u][dj][4 * rq + 3]);                                                 \
  }
    if ((out_other != 0) == (T::id == SFM_DT_BF16)) { ATTNP_PACK_O(F16::pack) } else { ATTNP_PACK_O(BF16::pack) }
    ATTNP_T(const unsigned long long tso_ = attnp_stamp();)
    if (qf_ready) {
      if (scale_log2e != 1.0f) {
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t w = qf[u][ks][e];
              qf[u][ks][e] = pack2<T>(T::to_f32((u16)(w & 0xffffu)) * scale_log2e, T::to_f32((u16)(w >> 16)) * scale_log2e);
            }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);                           // (builtin: see load_qf) the Q region is free now
      asm volatile("" ::: "memory");
    }
    store_o(item);                                                  // RW / 8 stores, left in flight across the next barrier
    inflight += RW / 8;
    ATTNP_T(t_so += attnp_stamp() - tso_;)
    if (has_next) {
      if (ngrp == 1) {                                              // single-group items: no room for the Q prefetch earlier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_q(item + gridDim.x);
        inflight = 0;                                               // (conservative: the next boundary waits for everything)
      }
      if (qf_ready) init_state();
    }
    {
      const int np = issue_next_group();                            // the refill that belongs to the item's last group
      inflight = (ngrp == 1) ? 0 : inflight + np;
    }
    ATTNP_T(t_post += attnp_stamp() - t_last;)
  }
  ATTNP_T(if (lane == 0) { dbg[0] = t_start; dbg[1] = attnp_stamp(); dbg[2] = t_bar; dbg[3] = t_pre; dbg[4] = t_steps; dbg[5] = t_post; dbg[6] = (t_q << 32) | t_so; dbg[7] = t_dr; })
}

// (the launch bounds cannot depend on a template parameter with this hipcc: thin kernels around the body)
#define ATTNP_KARGS const u16* __restrict__ qkv, u16* __restrict__ out, int Tlen, int ldqkv, int ldo, int koff, int voff,            \
                    long long qkv_batch_stride, long long o_batch_stride, float scale_log2e, int nqt, int nheads, int n_items,   \
                    float* __restrict__ lse_out, int out_other
#define ATTNP_KPASS qkv, out, Tlen, ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride, scale_log2e, nqt, nheads, n_items, lse_out, out_other
template <class T>
__global__ __launch_bounds__(512, 2) void attn_fwd_hd64p8_kernel(ATTNP_KARGS) { attnp_body<T, 8, 2>(ATTNP_KPASS); }
template <class T>
__global__ __launch_bounds__(256, 2) void attn_fwd_hd64p4_kernel(ATTNP_KARGS) { attnp_body<T, 4, 2>(ATTNP_KPASS); }
// one wave per SIMD, 128 query rows per wave (4 sub-blocks), the whole register file
template <class T>
__global__ __launch_bounds__(256, 1) void attn_fwd_hd64q4_kernel(ATTNP_KARGS) { attnp_body<T, 4, 4>(ATTNP_KPASS); }

// launch (called by attention.hip's dispatcher): head_dim 64, no dropout, operands below 2 GiB per batch element.
// (NW, SB) = (8, 2): one 512-thread workgroup per CU; (4, 2): two 256-thread workgroups per CU; (4, 4): one 256-thread workgroup per CU
template <class T, int NW, int SB>
static int attnp_launch(const void* qkv, void* out, float* lse, int B, int Tlen, int H, int ldqkv, int ldo, int koff, int voff,
                        long long qkv_batch_stride, long long o_batch_stride, float sl2, int out_other, hipStream_t st) {
  constexpr int QROWS = 32 * SB * NW;
  const int nqt = (Tlen + QROWS - 1) / QROWS;
  const int n_items = nqt * H * B;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SFM_ERR_LAUNCH;
  static int ncu[64] = {0};
  static bool attr_set[64] = {false};
  if (ncu[dev] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return SFM_ERR_LAUNCH;
    ncu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  constexpr int lds = 3 * (NW * SB / 8) * 16384 + NW * SB * 4096;   // K/V ring + Q prefetch region: 160 KB (8, 2) / 80 KB (4, 2) / 160 KB (4, 4)
  const void* fn = SB == 4 ? (const void*)attn_fwd_hd64q4_kernel<T>
                           : (NW == 8 ? (const void*)attn_fwd_hd64p8_kernel<T> : (const void*)attn_fwd_hd64p4_kernel<T>);
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return SFM_ERR_LAUNCH;
    attr_set[dev] = true;
  }
  constexpr int wgs_per_cu = (SB == 4) ? 1 : 8 / NW;
  // (diagnostic: SFM_ATTNP_WGS_PER_CU=1 with the (4, 2) form = one wave per SIMD, what a lone wave's step loop sustains)
  static const int wgs_env = getenv("SFM_ATTNP_WGS_PER_CU") ? atoi(getenv("SFM_ATTNP_WGS_PER_CU")) : 0;
  const int resident = ncu[dev] * ((wgs_env > 0 && wgs_env <= wgs_per_cu) ? wgs_env : wgs_per_cu);
  dim3 gridr(n_items < resident ? n_items : resident), blockr(64 * NW);
#define ATTNP_LAUNCH(K) SFM_LAUNCH((K<T>), gridr, blockr, lds, st, (const u16*)qkv, (u16*)out, Tlen, ldqkv, ldo, koff, voff,          \
                                   qkv_batch_stride, o_batch_stride, sl2, nqt, H, n_items, lse, out_other)
  if (SB == 4) ATTNP_LAUNCH(attn_fwd_hd64q4_kernel);
  else if (NW == 8) ATTNP_LAUNCH(attn_fwd_hd64p8_kernel);
  else ATTNP_LAUNCH(attn_fwd_hd64p4_kernel);
#undef ATTNP_LAUNCH
  return SFM_OK;
}

// nw: 8 = (8 waves, 2 sub-blocks), 4 = (4, 2), 44 = (4, 4)
int sfm_attn_pipe_launch(const void* qkv, void* out, float* lse, int B, int T, int H, int ldqkv, int ldo, int koff, int voff,
                         long long qkv_batch_stride, long long o_batch_stride, float sl2, int dtype, int out_other, int nw,
                         hipStream_t st) {
#define ATTNP_GO(TT, NW_, SB_) return attnp_launch<TT, NW_, SB_>(qkv, out, lse, B, T, H, ldqkv, ldo, koff, voff, qkv_batch_stride, o_batch_stride, sl2, out_other, st)
  if (dtype == SFM_DT_F16) { if (nw == 44) ATTNP_GO(F16, 4, 4); else if (nw == 4) ATTNP_GO(F16, 4, 2); else ATTNP_GO(F16, 8, 2); }
  else { if (nw == 44) ATTNP_GO(BF16, 4, 4); else if (nw == 4) ATTNP_GO(BF16, 4, 2); else ATTNP_GO(BF16, 8, 2); }
#undef ATTNP_GO
}
