#include "gemm16_epi.h"

#ifndef SFM_CONVP_GELU_H2
#define SFM_CONVP_GELU_H2 1
#endif
#ifndef SFM_CONVP_MIX
#define SFM_CONVP_MIX 1                 // fp16 operands: GroupNorm affine by v_fma_mix* on the packed inputs (A/B: -DSFM_CONVP_MIX=0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// conv16p: the PerceptionAgent's Conv1d layers (agents/perception.py:192-206, 167-171) with the GroupNorm + GELU of their
// INPUT applied while the operand is staged, so the normalised activation never exists in HBM:
//     y[b, l, n] = bias[n] + sum_{t, c} x[b, l s - p + t, c] W[n, (t, c)],   x = GELU( sc1[b,c] r1 + sh1[b,c]  [+ sc2[b,c] r2 + sh2[b,c]] )
// r1 (r2) = the RAW output of the producing conv(s), sc / sh = the finalised GroupNorm scale / shift of that output
// (sfm_gn_finalize); the two-input form is the residual block's  GELU(GN(main) + GN(skip))  (agents/perception.py:129).
// What changes against sfm_gemm16's implicit GEMM:
//   * the input rows of a 128-row output tile ("patch": 127 s + k rows x 64 channels) are fetched ONCE, transformed in
//     registers (unpack, scale / shift, exact-erf GELU by Abramowitz-Stegun 7.1.26, |err| < 1.5e-7, pack) and written to LDS;
//     the k taps then read the same patch at shifted rows (stride 2: even / odd input rows live in two planes, so that the
//     rows of consecutive outputs stay consecutive in LDS and the usual chunk swizzle keeps ds_read_b128 conflict-free).
//     The implicit im2col of sfm_gemm16 re-reads every input row k / s times from L2 and cannot transform in flight
//     (LDS-DMA), which is why a separate gn_apply pass (read + write of every activation) used to sit between two convs;
//   * more than 64 input channels: 64-channel slabs, one patch at a time, accumulators kept across slabs;
//   * the 1x1 stride-2 skip conv of a residual block reads the centre tap of the same patch: SKIP adds its k-tiles and a
//     second accumulator / output (N = 128 layers); N = 256 runs as two 128-column passes over the same patch (NPASS);
//   * weights stream L2 -> LDS by LDS-DMA in 128 x 64 tiles through a 2-stage ring, one barrier per k-tile, as in gemm16w.
// LDS: patch 16.6-33.8 KB + ring 32 KB: two workgroups per CU, so one's staging (VALU) overlaps the other's MFMA loop.
// Epilogue = gemm16_epilogue_strips (bias, GroupNorm partials of the raw output, 16-byte row stores).
// ---------------------------------------------------------------------------------------------------------------
struct ConvPParams {
  Gemm2Params g;                                       // main conv: W [N][KS * Cin] tap-major, bias, out, gn_partial, ...
  Gemm2Params gs;                                      // skip conv (SKIP): W [N][Cin], bias, out, gn_partial
  const u16* x1; const float* sc1; const float* sh1;
  const u16* x2; const float* sc2; const float* sh2;
  int Lin, Cin, pad, nMt;
  long long x_batch_stride;
};

__device__ __forceinline__ float gelu_as(float z) {    // z Phi(z), erf by A&S 7.1.26 (the backward in gn_bwd.hip uses the same)
  const float ax = fabsf(z) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);     // 1 ulp: the result is rounded to 16 bits afterwards
  const float ex = __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * ex;
  return z * (0.5f + 0.5f * copysignf(erfa, z));
}

// The same arithmetic on TWO fp16 values per instruction (v_pk_fma_f16 is a real 2 x, unlike v_pk_fma_f32): 21 instructions per
// pair instead of ~21 per value.  Used when the stage's operand format is fp16 (the default policy): the result is rounded to
// fp16 anyway; the fp16 intermediates cost ~1 ulp (|err| <= 6e-4 relative on GELU, 2.5e-4 |z| absolute near 0, where
// 1 - poly * ex cancels).  z itself (scale / shift of the GroupNorm) is still formed in fp32 from the fp16 inputs: a packed
// x * a + d would cancel |mean / std| ulps.
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t gelu_as_h2p(uint32_t zb);
__device__ __forceinline__ uint32_t gelu_as_h2(float z0, float z1) {
  const h2_t z = {(_Float16)z0, (_Float16)z1};
  return gelu_as_h2p(__builtin_bit_cast(uint32_t, z));
}
// z pair already packed (the GroupNorm affine of the fp16 inputs formed by v_fma_mixlo / mixhi_f16: fp32 arithmetic, ONE rounding
// to fp16, no unpack / pack instructions)
__device__ __forceinline__ uint32_t gelu_as_h2p(uint32_t zb) {
  const h2_t z = __builtin_bit_cast(h2_t, zb);
  const h2_t az = __builtin_bit_cast(h2_t, zb & 0x7fff7fffu);
  const h2_t u = az * (h2_t)(_Float16)(0.3275911f * 0.70710678118654752440f) + (h2_t)(_Float16)1.0f;
  h2_t t;
  t[0] = __builtin_amdgcn_rcph(u[0]);
  t[1] = __builtin_amdgcn_rcph(u[1]);
  const h2_t ex = __builtin_elementwise_exp2(z * z * (h2_t)(_Float16)(-0.72134752044448170368f));
  h2_t poly = t * (_Float16)1.061405429f + (_Float16)(-1.453152027f);
  poly = poly * t + (_Float16)1.421413741f;
  poly = poly * t + (_Float16)(-0.284496736f);
  poly = poly * t + (_Float16)0.254829592f;
  poly = poly * t;
  const h2_t erfa = (h2_t)(_Float16)1.0f - poly * ex;
  const h2_t sg = __builtin_bit_cast(h2_t, (__builtin_bit_cast(uint32_t, erfa) & 0x7fff7fffu) | (zb & 0x80008000u));
  return __builtin_bit_cast(uint32_t, z * (sg * (_Float16)0.5f + (_Float16)0.5f));
}

// Epilogue of one 64 x 64 wave tile.  The accumulators hold the TRANSPOSED tile (weights were the A operand of the MFMAs):
// lane (m = lane & 31, h = lane >> 5) of tile (n-block j, m-block i) holds, in register r, output row m, channel
// n = (r & 3) + 8 (r >> 2) + 4 h: four consecutive channels per register quad.  So
//   * the GroupNorm partial sums (per 8 / 16 / 32-channel group over the tile's 64 rows) are in-lane sums over register
//     quads followed by DPP reductions over the lanes: no LDS;
//   * the 16-bit result of a quad is one 8-byte LDS write into a row-major [64][64] 16-bit image (144-byte rows), ONE
//     write -> read round trip for the whole tile, then eight 16-byte coalesced row stores.
// (A first version passed 8 x 8-row fp32 strips through LDS like gemm16's epilogue: 8 round trips, 9 k cycles per tile - as much
// as staging the patch and the MFMA loop together.  gemm16_epilogue_strips itself is not used here: it carries every
// activation / residual / dropout mode of sfm_gemm16 inline, and two instances of it made instruction fetch the bound.)
// Round 4 (the two epilogues of a b0.c1 + skip tile were 2 000 instructions, a third of the tile's cycles): FULL = every row of the
// wave's 64 x 64 tile is inside the utterance (a wave-uniform choice made by the caller): no row predicates in the statistics;
// the 16-bit converts are hoisted behind ONE branch on the output format (a select per convert computed both formats and chose:
// 2 x 64 converts + 128 v_cndmask per tile); the row stores are buffer stores (32-bit offsets against a per-utterance descriptor
// whose range check drops the rows beyond Lout: no 64-bit address arithmetic, no exec-mask branches).
template <class T, bool FULL>
__device__ __forceinline__ void convp_epilogue_impl(const Gemm2Params& g, f32x16 (&acc)[2][2], unsigned char* img, const float* bias_s,
                                                    int lane, int b, int colb, int row_base) {
  constexpr int ROWB = 144;                            // image row: 64 x 16-bit + 16 bytes of padding (fp32 mode: two passes)
  const int l31 = lane & 31, hl = lane >> 5;
  // ---- bias, statistics ----
  float ps[2][4], pq[2][4];                            // per (n-block j, 8-channel quad-pair g): sums over this lane's rows
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n0 = colb + j * 32 + 8 * gq + 4 * hl;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + n0);     // LDS copy made at kernel start (zeros without a bias)
      float s_ = 0.f, q_ = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool rowok = FULL || (row_base + i * 32 + l31 < g.Lout);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[i][j][4 * gq + e] + bv[e];
          acc[i][j][4 * gq + e] = v;
          if (rowok) {
            s_ += v;
            q_ = __builtin_fmaf(v, v, q_);
          }
        }
      }
      ps[j][gq] = s_;
      pq[j][gq] = q_;
    }
  if (g.gn_partial) {
    // lanes = rows (and the two channel halves of a quad pair): the 16 per-lane sums (8 chunks of 8 channels x {sum, sum of
    // squares}) go through ONE transposing reduction; lane v < 16 then holds the tile total of (stat = v >> 3, j = (v >> 2) & 1,
    // gq = v & 3).  Groups wider than 8 channels add the neighbouring lanes' chunks (gq pairs: xor 1, quads: xor 2).
    const int cpg = g.gn_group >> 3;                   // 8-channel chunks per group: 1, 2 or 4
    const int ngroups = g.N / g.gn_group;
    float v16[16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        v16[j * 4 + gq] = ps[j][gq];
        v16[8 + j * 4 + gq] = pq[j][gq];
      }
    float tot = wave_sum16_transpose(v16, lane);
    if (cpg >= 2) tot += dpp_perm<0xB1>(tot);
    if (cpg == 4) tot += dpp_perm<0x4E>(tot);
    const int gq_ = lane & 3, j_ = (lane >> 2) & 1, stat = (lane >> 3) & 1;
    if (lane < 16 && (gq_ & (cpg - 1)) == 0 && (row_base >> 6) < g.gn_slots) {
      const long long sl = ((long long)b * g.gn_slots + (row_base >> 6)) * ngroups + (colb + j_ * 32 + 8 * gq_) / g.gn_group;
      g.gn_partial[sl * 2 + stat] = tot;
    }
  }
  // ---- store: 16-bit through one LDS image; fp32 (the latent heads' consumers) as two 32-row halves through the same image ----
  const int osz = g.out_f32 == 1 ? 4 : 2;
  // rows >= Lout fall outside the descriptor's range and are dropped by the hardware
  auto ors = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<unsigned char*>(g.out) + (long long)b * g.o_batch_stride * osz), 0,
                                               g.Lout * g.ldo * osz, 0x00020000);
  if (g.out_f32 != 1) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#define CONVP_PACK_IMG(PK)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                        \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                        \
  _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                                   \
    u32x2 w;                                                                                                           \
    w[0] = PK(acc[i][j][4 * gq + 0], acc[i][j][4 * gq + 1]);                                                           \
    w[1] = PK(acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]);                                                           \
    *reinterpret_cast<u32x2*>(img + (i * 32 + l31) * ROWB + (j * 32 + 8 * gq + 4 * hl) * 2) = w;                       \
  }
    if ((g.out_f32 == 2) == (T::id == SFM_DT_BF16)) { CONVP_PACK_IMG(F16::pack) } else { CONVP_PACK_IMG(BF16::pack) }
#undef CONVP_PACK_IMG
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    u32x4 rv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = lane + 64 * k;
      rv[k] = *reinterpret_cast<const u32x4*>(img + (c >> 3) * ROWB + (c & 7) * 16);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = lane + 64 * k;
      const int row = c >> 3, ch = c & 7;
      __builtin_amdgcn_raw_buffer_store_b128(rv[k], ors, ((row_base + row) * g.ldo + colb + ch * 8) * 2, 0, 0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<f32x4*>(img + l31 * 272 + (j * 32 + 8 * gq + 4 * hl) * 4) =
              f32x4{acc[i][j][4 * gq + 0], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = lane + 64 * k;
        const int row = c >> 4, ch = c & 15;           // 32 rows x 16 chunks of 4 floats
        const u32x4 v = *reinterpret_cast<const u32x4*>(img + row * 272 + ch * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, ors, ((row_base + i * 32 + row) * g.ldo + colb + ch * 4) * 4, 0, 0);
      }
    }
  }
}

template <class T>
__device__ __forceinline__ void convp_epilogue(const Gemm2Params& g, f32x16 (&acc)[2][2], unsigned char* img, const float* bias_s,
                                               int lane, int b, int colb, int row_base) {
  if (row_base + 64 <= g.Lout) convp_epilogue_impl<T, true>(g, acc, img, bias_s, lane, b, colb, row_base);       // wave-uniform
  else convp_epilogue_impl<T, false>(g, acc, img, bias_s, lane, b, colb, row_base);
}

#ifdef SFM_CONVP_STAMPS
// diagnostic build only: s_memtime stamps per workgroup (start, patch staged, k-loop done, end) -> sfm_conv16p_read_stamps
__device__ unsigned long long sfm_convp_stamps[8 * 32768];
#define SFM_STAMP(i) do { if (tid == 0 && blockIdx.x < 32768) sfm_convp_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int sfm_conv16p_read_stamps(void* host, int nblocks) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(sfm_convp_stamps), (size_t)nblocks * 64, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#else
#define SFM_STAMP(i) do { } while (0)
#endif

template <class T, int KS, int STRIDE, int NPASS, bool SKIP, bool TWO_IN>
__global__ __launch_bounds__(256, 2) void conv16p_kernel(ConvPParams p) {
  constexpr int R = 127 * STRIDE + KS;                 // input rows of a 128-row output tile
  constexpr int PLANES = (STRIDE == 2 && KS > 1) ? 2 : 1;
  constexpr int PR = (((STRIDE == 2) ? (R + 1) / 2 : R) + 7) / 8 * 8;               // rows per plane (whole 1-KB DMA pieces)
  constexpr int PATCH = PLANES * PR * 128;             // bytes
  constexpr int WT = 128 * 128;                        // one weight tile: 128 output channels x 64 k
  constexpr int TPS = NPASS * KS + (SKIP ? 1 : 0);     // weight tiles per 64-channel slab
  static_assert(!SKIP || NPASS == 1, "the fused skip conv needs N = 128");
  static_assert(PATCH + 2 * WT >= 4 * 9216, "epilogue images fit in the patch + ring area");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem + PATCH;                  // 2 weight tiles; TWO_IN: also the landing area of the second raw input
  constexpr int RING = (TWO_IN && PATCH > 2 * WT) ? PATCH : 2 * WT;
  float* bias_s = reinterpret_cast<float*>(smem + PATCH + RING);     // [256 main | 128 skip]: a global load in the epilogue
                                                                     // would put a memory round trip in front of the statistics

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hl = lane >> 5;

  for (int i = tid; i < NPASS * 128; i += 256) bias_s[i] = p.g.bias ? p.g.bias[i] : 0.f;
  if (SKIP && tid < 128) bias_s[256 + tid] = p.gs.bias ? p.gs.bias[tid] : 0.f;
  int id = blockIdx.x;                                 // XCD-aware order: neighbouring tiles (shared halo rows) on one XCD
  {
    const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int mtile = id % p.nMt, b = id / p.nMt;
  const int l0 = mtile * 128;
  const int nslab = p.Cin >> 6;
  const int Q = nslab * TPS;                           // weight tiles of this output tile

  auto w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.g.W, 0, p.g.w_records, 0x00020000);
  auto ws_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(SKIP ? p.gs.W : p.g.W), 0, SKIP ? p.gs.w_records : p.g.w_records, 0x00020000);
  // LDS-DMA lane coordinates of a weight tile (8 rows x 128 B per instruction, 4 instructions per wave)
  int b_row[4], b_swz[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b_row[i] = (wave * 4 + i) * 8 + (lane >> 3);
    b_swz[i] = ((lane & 7) ^ ((b_row[i] >> 1) & 7)) * 8;
  }
  // weight tile q of the sequence: slab = q / TPS; within the slab: (pass, tap) main tiles, then the skip tile
  auto issue_w = [&](int q) {
    const int slab = q / TPS, j = q - slab * TPS;
    unsigned char* dst = ring + (q & 1) * WT;
    if (SKIP && j == TPS - 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ws_rs, (lds_ptr_t)(dst + (wave * 4 + i) * 1024), 16,
                                                 (b_row[i] * p.gs.Kpad + slab * 64 + b_swz[i]) * 2, 0, 0, 0);
    } else {
      const int np = j / KS, t = j - np * KS;
      const int koff = t * p.Cin + slab * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(dst + (wave * 4 + i) * 1024), 16,
                                                 ((np * 128 + b_row[i]) * p.g.Kpad + koff + b_swz[i]) * 2, 0, 0, 0);
    }
  };

  f32x16 acc[NPASS][2][2], accs[2][2];
#pragma unroll
  for (int n = 0; n < NPASS; ++n)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accs[i][j][r] = 0.f;

  int fb_off[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wn * 64 + j * 32 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) fb_off[j][s] = row * 128 + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4);
  }

  // ---- patch staging.  (1) the RAW rows of the slab go global -> LDS by LDS-DMA, all at once (33 KB in flight per workgroup:
  //      with register staging three dependent global round trips per slab left ~1 TB/s of read stream; rows outside
  //      [0, Lin) and LDS rows beyond the patch are out of the descriptor's range and arrive as zeros); the second input of a
  //      residual block lands in the weight ring, which is idle until the transform is done.  (2) every thread transforms
  //      its 16-byte chunks IN PLACE (scale / shift [+ second input] -> GELU -> 16 bit); zero-filled padding rows stay zero,
  //      as the conv's zero padding of x requires.  The chunk swizzle is applied on the DMA's source address, and
  //      ((row >> 1) & 7) of a thread's rows (row = tid / 8 + 32 i) is constant, so a thread owns the same 8 channels in every
  //      row: their scale / shift sit in registers. ----
  constexpr int JSTEP = (PLANES == 1 && STRIDE == 2) ? 2 : 1;        // k = 1, stride 2: only the even input rows are read
  constexpr int LROWS = PLANES * PR;                                 // LDS rows of the patch
  constexpr int NPIECE = (LROWS + 7) / 8;                            // 1-KB DMA pieces
  auto x1_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x1 + (long long)b * p.x_batch_stride), 0, p.Lin * p.Cin * 2, 0x00020000);
  auto x2_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((TWO_IN ? p.x2 : p.x1) + (long long)b * p.x_batch_stride), 0,
                                                 p.Lin * p.Cin * 2, 0x00020000);
  // input position of LDS row lr (or -1): plane / index de-interleaving for stride 2
  auto row_pos = [&](int lr) {
    const int plane = (PLANES == 2) ? (lr >= PR ? 1 : 0) : 0;
    const int idx = lr - plane * PR;
    const int j = (PLANES == 2) ? 2 * idx + plane : idx * JSTEP;
    const int pos = l0 * STRIDE - p.pad + j;
    return (lr < LROWS && j < R && pos >= 0 && pos < p.Lin) ? pos : -1;
  };
  const int sp_ = tid & 7, srow = tid >> 3;
  const int sc_ = sp_ ^ ((srow >> 1) & 7);                           // logical chunk (8 channels) this thread owns in every row
  auto stage_patch = [&](int slab) {
    for (int pc = wave; pc < NPIECE; pc += 4) {
      const int lr = pc * 8 + (lane >> 3);
      const int pos = row_pos(lr);
      const int c = (lane & 7) ^ ((lr >> 1) & 7);
      const int voff = pos >= 0 ? (pos * p.Cin + slab * 64 + c * 8) * 2 : -1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x1_rs, (lds_ptr_t)(smem + pc * 1024), 16, voff, 0, 0, 0);
      if (TWO_IN) __builtin_amdgcn_raw_ptr_buffer_load_lds(x2_rs, (lds_ptr_t)(ring + pc * 1024), 16, voff, 0, 0, 0);
    }
    const int ch0 = slab * 64 + sc_ * 8;
    float a1[8], d1[8], a2[8];
    {
      const f32x4* ps = reinterpret_cast<const f32x4*>(p.sc1 + (long long)b * p.Cin + ch0);
      const f32x4* ph = reinterpret_cast<const f32x4*>(p.sh1 + (long long)b * p.Cin + ch0);
      const f32x4 s0 = ps[0], s1 = ps[1], h0 = ph[0], h1 = ph[1];
#pragma unroll
      for (int e = 0; e < 4; ++e) { a1[e] = s0[e]; a1[4 + e] = s1[e]; d1[e] = h0[e]; d1[4 + e] = h1[e]; }
      if (TWO_IN) {
        const f32x4* qs = reinterpret_cast<const f32x4*>(p.sc2 + (long long)b * p.Cin + ch0);
        const f32x4* qh = reinterpret_cast<const f32x4*>(p.sh2 + (long long)b * p.Cin + ch0);
        const f32x4 t0 = qs[0], t1 = qs[1], g0 = qh[0], g1 = qh[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a2[e] = t0[e]; a2[4 + e] = t1[e]; d1[e] += g0[e]; d1[4 + e] += g1[e]; }
      }
    }
    if (slab == 0) SFM_STAMP(4);
    wait_vmcnt<0>();                                   // this wave's pieces have landed ...
    __syncthreads();                                   // ... everyone's have
    if (slab == 0) SFM_STAMP(5);
#pragma unroll 2
    for (int lr = srow; lr < LROWS; lr += 32) {
      if (row_pos(lr) < 0) continue;                   // zero-filled row: stays the conv's zero padding
      unsigned char* q1 = smem + lr * 128 + sp_ * 16;
      const u32x4 v1 = *reinterpret_cast<const u32x4*>(q1);
      u32x4 v2 = {0u, 0u, 0u, 0u};
      if (TWO_IN) v2 = *reinterpret_cast<const u32x4*>(ring + lr * 128 + sp_ * 16);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#if SFM_CONVP_GELU_H2 && SFM_CONVP_MIX
        if (T::id == SFM_DT_F16) {
          // z = x1 * a1 [+ x2 * a2] + d straight from the packed fp16 inputs (mixed-precision FMA: fp16 source halves, fp32
          // coefficients and arithmetic) into a packed fp16 pair: 2 (4) instructions per pair instead of 4 (7)
          uint32_t zpk;
          float t0 = d1[2 * e], t1 = d1[2 * e + 1];
          if (TWO_IN) {
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(v2[e]), "v"(a2[2 * e]), "v"(d1[2 * e]));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(v2[e]), "v"(a2[2 * e + 1]), "v"(d1[2 * e + 1]));
          }
          asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
              "v_fma_mixhi_f16 %0, %1, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
              : "=&v"(zpk) : "v"(v1[e]), "v"(a1[2 * e]), "v"(t0), "v"(a1[2 * e + 1]), "v"(t1));
          o[e] = gelu_as_h2p(zpk);
          continue;
        }
#endif
        float z0 = T::to_f32((u16)(v1[e] & 0xffffu)) * a1[2 * e] + d1[2 * e];
        float z1 = T::to_f32((u16)(v1[e] >> 16)) * a1[2 * e + 1] + d1[2 * e + 1];
        if (TWO_IN) {
          z0 += T::to_f32((u16)(v2[e] & 0xffffu)) * a2[2 * e];
          z1 += T::to_f32((u16)(v2[e] >> 16)) * a2[2 * e + 1];
        }
#if SFM_CONVP_GELU_H2
        if (T::id == SFM_DT_F16) o[e] = gelu_as_h2(z0, z1);
        else
#endif
          o[e] = pack2<T>(gelu_as(z0), gelu_as(z1));
      }
      *reinterpret_cast<u32x4*>(q1) = o;
    }
  };

  // ---- one weight tile against the patch rows of tap `t` ----
  auto mma_tile = [&](int q, int t, f32x16 (&a)[2][2]) {
    const unsigned char* wb = ring + (q & 1) * WT;
    int arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rw = wm * 64 + i * 32 + l31;
      arow[i] = (PLANES == 2) ? (t & 1) * PR + rw + (t >> 1) : ((STRIDE == 2) ? rw : rw + t);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = *reinterpret_cast<const u32x4*>(smem + arow[i] * 128 + (((2 * s + hl) ^ ((arow[i] >> 1) & 7)) << 4));
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u32x4*>(wb + fb_off[j][s]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) a[i][j] = T::mfma(fb[j], fa[i], a[i][j]);      // transposed tile: rows of C^T = channels
    }
  };

  SFM_STAMP(0);
#ifdef SFM_CONVP_STAMPS
  if (tid == 0 && blockIdx.x < 32768)                  // which CU / thread-group slot this workgroup ran on (HW_ID, XCC_ID)
    sfm_convp_stamps[blockIdx.x * 8 + 6] = (unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |
                                           ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
#endif
  if (!TWO_IN) issue_w(0);
  for (int slab = 0; slab < nslab; ++slab) {
    if (slab > 0) __syncthreads();                     // every wave is done reading the previous slab's patch (and weight tiles)
    stage_patch(slab);
    __syncthreads();                                   // patch complete
    if (slab == 0) SFM_STAMP(1);
    const int q0 = slab * TPS;
    if (TWO_IN) issue_w(q0);                           // the ring held the second raw input until now
#pragma unroll
    for (int j = 0; j < TPS; ++j) {
      const int q = q0 + j;
      wait_ring<0>();                                  // this wave's share of weight tile q has landed, its reads of tile q-1 have returned
      __builtin_amdgcn_s_barrier();                    // ... everyone's have: tile q is complete and tile q-1's slot may be refilled
      // Inside a slab the next tile's refill is UNCONDITIONAL (q + 1 < Q always holds there): behind a conditional refill the
      // compiler no longer knows which LDS-DMA operations are pending at the join and guards the tile's first fragment reads
      // with vmcnt waits of its own - i.e. waits for the refill it has just issued (seen in the ISA of round 3's race hunt).
      // The slab's last tile issues the next slab's first tile AFTER its own reads instead: the patch staging that follows is
      // far longer than a weight tile's latency.
      if (j + 1 < TPS) issue_w(q + 1);
      if (SKIP && j == TPS - 1) {
        mma_tile(q, p.pad, accs);                      // 1x1 stride-2 conv = the centre tap of the same patch
      } else {
        const int np = j / KS, t = j - np * KS;
        if (np == 0) mma_tile(q, t, acc[0]);
        else mma_tile(q, t, acc[NPASS - 1]);
      }
      if (j + 1 == TPS && !TWO_IN && q + 1 < Q) issue_w(q + 1);
    }
  }
  __syncthreads();                                     // the ring becomes the epilogue strips
  SFM_STAMP(2);
  unsigned char* img = smem + wave * 9216;             // 64 x 144 B per wave, in the (now free) patch + ring area
#pragma unroll
  for (int n = 0; n < NPASS; ++n) convp_epilogue<T>(p.g, acc[n], img, bias_s, lane, b, n * 128 + wn * 64, l0 + wm * 64);
  if (SKIP) convp_epilogue<T>(p.gs, accs, img, bias_s + 256, lane, b, wn * 64, l0 + wm * 64);
  SFM_STAMP(3);
}

template <class T, int KS, int STRIDE, int NPASS, bool SKIP, bool TWO_IN>
static int launch_convp(const ConvPParams& p, hipStream_t stream) {
  constexpr int R = 127 * STRIDE + KS;
  constexpr int PLANES = (STRIDE == 2 && KS > 1) ? 2 : 1;
  constexpr int PR = (((STRIDE == 2) ? (R + 1) / 2 : R) + 7) / 8 * 8;
  constexpr int patch = PLANES * PR * 128;
  constexpr int lds = patch + ((TWO_IN && patch > 2 * 128 * 128) ? patch : 2 * 128 * 128) + 384 * 4;   // + the bias copy
  static bool attr_set_dev[64] = {false};        // hipFuncSetAttribute is per device
  int attr_dev_ = 0;
  if (hipGetDevice(&attr_dev_) != hipSuccess || attr_dev_ < 0 || attr_dev_ >= 64) return SFM_ERR_LAUNCH;
  bool& attr_set = attr_set_dev[attr_dev_];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv16p_kernel<T, KS, STRIDE, NPASS, SKIP, TWO_IN>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return SFM_ERR_LAUNCH;
    attr_set = true;
  }
  SFM_LAUNCH((conv16p_kernel<T, KS, STRIDE, NPASS, SKIP, TWO_IN>), dim3(p.nMt * p.g.B), dim3(256), lds, stream, p);
  return SFM_OK;
}

static void convp_fill(Gemm2Params& g, const void* W, const float* bias, void* out, float* gn_partial, int B, int Lout, int N,
                       int Kpad, int out_f32, int gn_group) {
  g = Gemm2Params{};
  g.W = (const u16*)W; g.bias = bias; g.out = out; g.gn_partial = gn_partial;
  g.B = B; g.Lout = Lout; g.N = N; g.Npad = N; g.K = Kpad; g.Kpad = Kpad; g.ldo = N; g.o_batch_stride = (long long)Lout * N;
  g.alpha = 1.0f; g.epi = EPI_NONE; g.out_f32 = out_f32; g.gn_group = gn_group; g.nsplit = 0;
  g.w_records = (int)((long long)N * Kpad * 2);
  g.gn_slots = 2 * ((Lout + 127) / 128);
  const int osz = out_f32 == 1 ? 4 : 2;
  g.vec_ok = ((((uintptr_t)out) % 16) == 0 && ((N * osz) % 16) == 0) ? 1 : 0;
  g.p_drop = 0.f; g.seed = 0u; g.aux = nullptr; g.out2 = nullptr; g.resid = nullptr;
}

// Conv1d(Cin -> N, ksize, stride, zero padding `pad`) on x = GELU(sc1 * x1 + sh1 [+ sc2 * x2 + sh2]) (see conv16p above).
//   x1, x2 [B, Lin, Cin] 16-bit channels-last raw conv outputs; sc / sh [B, Cin] fp32 (sfm_gn_finalize)
//   W [N][ksize * Cin] 16-bit tap-major (the layout sfm_gemm16 takes), bias [N] fp32; out [B, Lout, N] (out_f32 as sfm_gemm16);
//   gn_partial [B][2 ceil(Lout / 128)][N / gn_group][2] or NULL
//   Ws / bias_s / out_s / gn_partial_s: optional fused Conv1d(Cin -> N, 1, stride 2) on the same x (N = 128, ksize 7 only)
// Supported (the PerceptionAgent's layers): Cin % 64 == 0 and (ksize, stride, pad, N, inputs) in {(7, 2, 3, 128 + fused skip, 1 | 2),
// (7, 2, 3, 256, 2), (3, 1, 1, 128 | 256, 1), (5, 2, 2, 256, 2), (1, 2, 0, 256, 2)}; anything else returns SFM_ERR_SHAPE.
extern "C" int sfm_conv16p(const void* x1, const float* sc1, const float* sh1, const void* x2, const float* sc2, const float* sh2,
                           const void* W, const float* bias, void* out, float* gn_partial, const void* Ws, const float* bias_s,
                           void* out_s, float* gn_partial_s, int B, int Lin, int Cin, int N, int ksize, int stride, int pad,
                           int out_f32, int gn_group, int dtype, void* stream) {
  if (!x1 || !sc1 || !sh1 || !W || !out) return SFM_ERR_ARG;
  if ((x2 != nullptr) != (sc2 != nullptr) || (x2 != nullptr) != (sh2 != nullptr)) return SFM_ERR_ARG;
  if (B <= 0 || Lin <= 0 || Cin <= 0 || (Cin % 64) != 0 || (N != 128 && N != 256)) return SFM_ERR_SHAPE;
  if (out_f32 < 0 || out_f32 > 2) return SFM_ERR_SHAPE;
  if (gn_partial && gn_group != 8 && gn_group != 16 && gn_group != 32) return SFM_ERR_SHAPE;
  if ((long long)Lin * Cin * 2 >= (1LL << 31) || (long long)N * ksize * Cin * 2 >= (1LL << 31)) return SFM_ERR_SHAPE;
  if ((long long)((Lin + 2 * pad - ksize) / stride + 1) * N * 4 >= (1LL << 31)) return SFM_ERR_SHAPE;       // 32-bit store offsets per utterance
  const int Lout = (Lin + 2 * pad - ksize) / stride + 1;
  if (Lout <= 0) return SFM_ERR_SHAPE;
  const bool skip = Ws != nullptr;
  if (skip && (!out_s || N != 128 || ksize != 7 || stride != 2)) return SFM_ERR_SHAPE;
  if ((((uintptr_t)out) % 16) != 0 || (out_s && (((uintptr_t)out_s) % 16) != 0)) return SFM_ERR_SHAPE;   // 16-byte row stores
  ConvPParams p{};
  convp_fill(p.g, W, bias, out, gn_partial, B, Lout, N, ksize * Cin, out_f32, gn_group);
  if (skip) convp_fill(p.gs, Ws, bias_s, out_s, gn_partial_s, B, Lout, N, Cin, out_f32, gn_group);
  p.x1 = (const u16*)x1; p.sc1 = sc1; p.sh1 = sh1; p.x2 = (const u16*)x2; p.sc2 = sc2; p.sh2 = sh2;
  p.Lin = Lin; p.Cin = Cin; p.pad = pad; p.nMt = (Lout + 127) / 128; p.x_batch_stride = (long long)Lin * Cin;
  hipStream_t st = (hipStream_t)stream;
  const bool two = x2 != nullptr;
  const int key = ksize * 100 + stride * 10 + pad;
  // the layer shapes of the PerceptionAgent (agents/perception.py:160-171 with encoder_channels 256); anything else: SFM_ERR_SHAPE
#define CONVP_GO(TT)                                                                                       \
  if (key == 723 && N == 128 && skip) return two ? launch_convp<TT, 7, 2, 1, true, true>(p, st) : launch_convp<TT, 7, 2, 1, true, false>(p, st); \
  if (key == 723 && N == 256 && !skip && two) return launch_convp<TT, 7, 2, 2, false, true>(p, st);        \
  if (key == 311 && N == 128 && !two) return launch_convp<TT, 3, 1, 1, false, false>(p, st);                \
  if (key == 311 && N == 256 && !two) return launch_convp<TT, 3, 1, 2, false, false>(p, st);                \
  if (key == 522 && N == 256 && two) return launch_convp<TT, 5, 2, 2, false, true>(p, st);                  \
  if (key == 120 && N == 256 && two) return launch_convp<TT, 1, 2, 2, false, true>(p, st);
  if (dtype == SFM_DT_BF16) { CONVP_GO(BF16) }
  if (dtype == SFM_DT_F16) { CONVP_GO(F16) }
#undef CONVP_GO
  return SFM_ERR_SHAPE;
}

