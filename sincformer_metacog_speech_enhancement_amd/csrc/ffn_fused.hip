// Fused Conformer feed-forward module (models/conformer.py:41-49, eval):
//     out = x + alpha * ( W2 * swish(W1 * LayerNorm(x) + b1) + b2 )          D = 256, FF % 64 == 0
// One workgroup (8 waves) owns 128 rows.  x is read from HBM once, normalised in registers and kept
// as the 16-bit A operand [128 x 256] in LDS for the whole kernel; the FF-wide hidden activation never
// leaves the CU: per 64-unit chunk  S1 = H W1c^T + b1 (MFMA, accumulator started from the bias) -> swish -> 16-bit U in
// LDS -> acc2 += U W2c^T (MFMA).  The GEMM1 A operand of a wave (its 32 rows of LN(x), all 256 k) stays in registers for
// the whole kernel; weight chunks (W1c + W2c = 64 KB) stream from L2 by LDS-DMA into a 2-stage ring.
// Schedule (round 2; stamps and ablations in profiles/README.md): the Swish of chunk c shares an instruction stream with the
// GEMM1 MFMAs of chunk c+1, the weight refills are issued from inside the MFMA streams (a burst of 8 pieces after a barrier
// cost every wave ~500 issue cycles per chunk), two raw barriers per chunk with counted vmcnt waits; the epilogue works in
// the prologue's row layout (no second read of x from a far cache line, LayerNorm of the next sub-layer by DPP adds).
// HBM traffic: x in (1 KB/row, read twice: the residual re-read comes from L2 / Infinity Cache) + out (1 KB/row) instead of
// ~8 KB/row for LN + two GEMM launches.
#include "sfm_common.h"

#ifndef SFM_FFN_ABL
#define SFM_FFN_ABL 0                   // diagnostic builds only (tools/variant_lib.sh): 1 no weight refills, 2 no Swish (results wrong)
#endif
#define FF_D 256
#define FF_BM 128
#define FF_CH 64                        // hidden units per chunk

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void ff_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ff_frag_read(u32x4& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void ff_frag_wait(u32x4& frag) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N) : "memory");
}
__device__ __forceinline__ float ff_lane_bcast(float v, int r) {       // lane r's value in every lane (r static: v_readlane)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r));
}
__device__ __forceinline__ void ff_barrier() {
  // LDS writes of this wave must have completed before other waves pass the barrier; the LDS-DMA
  // queue (vmcnt) is deliberately NOT drained here - that is what the counted waits are for.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#ifdef SFM_FFN_STAMPS
// diagnostic build only (tools/variant_lib.sh): s_memtime stamps per workgroup -> sfm_ffn_read_stamps
__device__ unsigned long long sfm_ffn_stamps[8 * 4096];
#define FF_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) sfm_ffn_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int sfm_ffn_read_stamps(void* host, int nblocks) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(sfm_ffn_stamps), (size_t)nblocks * 64, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#else
#define FF_STAMP(i) do { } while (0)
#endif

template <class T>
__global__ __launch_bounds__(512) void ffn_fused_kernel(const float* __restrict__ x, const float* __restrict__ lnw,
                                                        const float* __restrict__ lnb, const u16* __restrict__ W1,
                                                        const float* __restrict__ b1, const u16* __restrict__ W2,
                                                        const float* __restrict__ b2, float* __restrict__ out, int M,
                                                        int FF, float alpha, float eps, int w1_bytes, int w2_bytes,
                                                        const float* __restrict__ ln2w, const float* __restrict__ ln2b,
                                                        void* __restrict__ ln_out, int ln_out_f32) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS map: two weight stages, each [W1 chunk 64 x 512 B | W2 chunk 256 x 128 B] = 64 KB, then U [128][128 B].
  // During the prologue the first 64 KB (the two W1 slots are NOT yet in use) hold LN(x) as [128][512 B].
  constexpr int STG = FF_CH * 512 + FF_D * 128;
  unsigned char* Us = smem + 2 * STG;                     // chunk c at c ^ ((row >> 1) & 7)
  float* b1s = reinterpret_cast<float*>(Us + FF_BM * 128);  // [FF] first-layer bias (no global loads in the loop)
  unsigned char* Hs = smem;                               // prologue only; chunk c at c ^ (row & 15)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hl = lane >> 5;
  const int m0 = blockIdx.x * FF_BM;
  FF_STAMP(0);
  auto w1_rs = __builtin_amdgcn_make_buffer_rsrc((void*)W1, 0, w1_bytes, 0x00020000);
  auto w2_rs = __builtin_amdgcn_make_buffer_rsrc((void*)W2, 0, w2_bytes, 0x00020000);

  // LDS-DMA pieces (issue_group below): W1 chunk = 64 rows x 512 B (4 x 1 KB per wave, 2 rows each, chunk c at
  // c ^ (row & 15)); W2 chunk = 256 rows x 128 B (4 x 1 KB per wave, 8 rows each, chunk c at c ^ ((row >> 1) & 7))
  // ---- LayerNorm prologue: wave w normalises rows 16w .. 16w+15, 4 consecutive columns per lane.
  //      All 16 row loads are issued before the first use; row statistics by DPP adds (no LDS round trips).
  {
    for (int i = tid; i < FF; i += 512) b1s[i] = b1[i];
    const f32x4 gw = *reinterpret_cast<const f32x4*>(lnw + lane * 4);
    const f32x4 gb = *reinterpret_cast<const f32x4*>(lnb + lane * 4);
    f32x4 xv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = m0 + wave * 16 + r;
      m = m < M ? m : M - 1;                                  // clamp: tail rows are computed but never stored
      xv[r] = *reinterpret_cast<const f32x4*>(x + (long long)m * FF_D + lane * 4);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {                           // row by row as the loads return (the phase is load-latency bound:
      const int row = wave * 16 + r;                         // one transposing reduction over all 16 rows was slower here,
      const f32x4 v = xv[r];                                 // it cannot start before the last row has arrived)
      const float mean = wave_sum_dpp(v[0] + v[1] + v[2] + v[3]) * (1.0f / FF_D);
      const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
      const float rstd = rsqrtf(wave_sum_dpp(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.0f / FF_D) + eps);
      u32x2 pk;
      pk[0] = pack2<T>(d0 * rstd * gw[0] + gb[0], d1 * rstd * gw[1] + gb[1]);
      pk[1] = pack2<T>(d2 * rstd * gw[2] + gb[2], d3 * rstd * gw[3] + gb[3]);
      const int chunk = (lane >> 1) ^ (row & 15);
      *reinterpret_cast<u32x2*>(Hs + row * 512 + chunk * 16 + (lane & 1) * 8) = pk;
    }
  }
  __syncthreads();
  // GEMM1 A operand of this wave (rows 32*(w>>1) .. +32, all 256 k) lives in registers for the whole kernel
  const int r1 = (wave >> 1) * 32 + l31, n1 = (wave & 1) * 32 + l31;
  u32x4 hf[16];
#pragma unroll
  for (int s = 0; s < 16; ++s)
    hf[s] = *reinterpret_cast<const u32x4*>(Hs + r1 * 512 + (((2 * s + hl) ^ (r1 & 15)) << 4));
  __syncthreads();                                         // LN image consumed: the weight ring may overwrite it
  FF_STAMP(1);
  const int nch = FF / FF_CH;
  // Weight pieces: a wave moves 4 x 1 KB of every W1 chunk and 4 x 1 KB of every W2 chunk (ring stage c & 1).  A chunk index
  // past the end is issued all the same: the buffer range check turns it into zeros written to ring bytes nobody reads
  // again, and the counted vmcnt waits below stay the same from the first chunk to the last.
  auto w1_piece = [&](int c, int i) {
    unsigned char* w1s = smem + (c & 1) * STG;
    const int inst = wave * 4 + i;
    const int row = inst * 2 + (lane >> 5);
    const int lc = (lane & 31) ^ (row & 15);
    const int voff = ((c * FF_CH + row) * FF_D + lc * 8) * 2;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rs, (lds_ptr_t)(w1s + inst * 1024), 16, voff, 0, 0, 0);
  };
  auto w2_piece = [&](int c, int i) {
    unsigned char* w2s = smem + (c & 1) * STG + FF_CH * 512;
    const int inst = wave * 4 + i;
    const int row = inst * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);
    const int voff = c < nch ? (row * FF + c * FF_CH + lc * 8) * 2 : w2_bytes;   // W2 rows interleave the chunks: force the miss
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rs, (lds_ptr_t)(w2s + inst * 1024), 16, voff, 0, 0, 0);
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) w1_piece(0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) w2_piece(0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) w1_piece(1, i);

  // wave roles: GEMM1 tile = rows 32*(w>>1), hidden cols 32*(w&1); GEMM2 tile = rows 64*(w>>2), out cols 64*(w&3).
  // A chunk period has two phases with a barrier after each:
  //   X(c): GEMM1 of chunk c+1 (MFMA) in the same instruction stream as the Swish of chunk c (VALU on the previous
  //         GEMM1's accumulator) -> U(c);  plus the refill W2(c+1) (its ring bytes held W2(c-1), read in Y(c-1))
  //   Y(c): GEMM2 of chunk c (MFMA);       plus the refill W1(c+3) (held W1(c+1), read in X(c))
  // so the matrix pipe has work in both phases and the Swish never runs alone.  A wave's pieces in issue order:
  // W1(0) W2(0) W1(1) | W1(2) | X(0): W2(1) | Y(0): W1(3) | X(1): W2(2) | Y(1): W1(4) ... ; the barrier after X(c) needs
  // W2(c) and the one after Y(c) needs W1(c+2): both have exactly two younger groups behind them (vmcnt 8).
  const int wm2 = wave >> 2, wn2 = wave & 3;
  const int row1 = (wave >> 1) * 32 + l31;                  // GEMM1 / Swish: x row of this lane
  // W1 fragment k of ring stage s sits at s * STG + n1 * 512 + (((2k + hl) ^ (n1 & 15)) << 4) = w1_lane + s * STG + (hx4 ^ (k << 5))
  const uint32_t w1_lane = (uint32_t)(uintptr_t)(lds_ptr_t)smem + (uint32_t)(n1 * 512);
  const int hx4 = (hl ^ (n1 & 15)) << 4;
  const int ub = (wave & 1) * 32;                           // first hidden unit (inside the chunk) of this wave
  f32x16 acc2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;

  // GEMM1 (transposed): S1^T[32 units x 32 rows] = W1c[32 units x 256] * H[32 rows x 256]^T + b1 (the accumulator starts
  // from the bias).  W1c is the MFMA A operand, the register-resident H the B operand: a lane then holds 4 CONSECUTIVE hidden
  // units of one row per register quad, so U goes to LDS with 8-byte stores.
  auto s1_init = [&](int c, f32x16& s1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + c * FF_CH + ub + 8 * q + 4 * hl);
#pragma unroll
      for (int e = 0; e < 4; ++e) s1[4 * q + e] = bv[e];
    }
  };
  auto w1_frag = [&](int c, int k) {
    return *reinterpret_cast<const u32x4*>(smem + (c & 1) * STG + n1 * 512 + (((2 * k + hl) ^ (n1 & 15)) << 4));
  };
  // Swish of 4 accumulator registers (units u0..u0+3 of this lane's row) -> 8 bytes of U
  auto swish_quad = [&](const f32x16& s1, int q) {
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = s1[4 * q + e];
#if SFM_FFN_ABL == 2
      y[e] = z;
#else
      y[e] = z * __builtin_amdgcn_rcpf(1.0f + __expf(-z));
#endif
    }
    u32x2 pk;
    pk[0] = pack2<T>(y[0], y[1]);
    pk[1] = pack2<T>(y[2], y[3]);
    const int u0 = ub + 8 * q + 4 * hl;
    const int chunk = (u0 >> 3) ^ ((row1 >> 1) & 7);
    *reinterpret_cast<u32x2*>(Us + row1 * 128 + chunk * 16 + (u0 & 7) * 2) = pk;
  };
  auto gemm2 = [&](int c, bool refill) {
    const unsigned char* W2s = smem + (c & 1) * STG + FF_CH * 512;
    u32x4 fa[4][2], fb[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm2 * 64 + i * 32 + l31;
        fa[k][i] = *reinterpret_cast<const u32x4*>(Us + row * 128 + (((2 * k + hl) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wn2 * 64 + j * 32 + l31;
        fb[k][j] = *reinterpret_cast<const u32x4*>(W2s + row * 128 + (((2 * k + hl) ^ ((row >> 1) & 7)) << 4));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc2[i][j] = T::mfma(fa[k][i], fb[k][j], acc2[i][j]);
          const int m = k * 4 + i * 2 + j;
#if SFM_FFN_ABL != 1
          if (refill && (m & 3) == 1) w1_piece(c + 3, m >> 2);
#endif
        }
  };

  f32x16 s1;                                                // GEMM1 accumulator of the chunk whose Swish comes next
  ff_wait_vmcnt<8>();                                       // W1(0) is in
  ff_barrier();
  FF_STAMP(2);
  s1_init(0, s1);
  {
    u32x4 fw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) fw[k] = w1_frag(0, k);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      s1 = T::mfma(fw[k & 3], hf[k], s1);
      if (k + 4 < 16) fw[k & 3] = w1_frag(0, k + 4);
    }
  }
  ff_wait_vmcnt<0>();                                       // W2(0), W1(1) are in
  ff_barrier();                                             // W1(0) has been read by everyone
#pragma unroll
  for (int i = 0; i < 4; ++i) w1_piece(2, i);
  for (int c = 0; c + 1 < nch; ++c) {
    // ---- X(c) ----
    {
      f32x16 s1n;
      s1_init(c + 1, s1n);
      // W1 fragments by inline-asm reads kept FOUR MFMAs ahead of their use (left to the compiler the reads sink to one
      // MFMA ahead to save registers).  ff_frag_wait is the lgkmcnt wait the compiler no longer inserts (N = this wave's
      // younger fragment reads), tied to the registers.
      const uint32_t fbase = w1_lane + (uint32_t)(((c + 1) & 1) * STG);
      u32x4 fw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) ff_frag_read(fw[k], fbase + (uint32_t)(hx4 ^ (k << 5)));
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k <= 12) ff_frag_wait<3>(fw[k & 3]);
        else if (k == 13) ff_frag_wait<2>(fw[k & 3]);
        else if (k == 14) ff_frag_wait<1>(fw[k & 3]);
        else ff_frag_wait<0>(fw[k & 3]);
        s1n = T::mfma(fw[k & 3], hf[k], s1n);
        if (k + 4 < 16) ff_frag_read(fw[k & 3], fbase + (uint32_t)(hx4 ^ ((k + 4) << 5)));
        if ((k & 3) == 1) swish_quad(s1, k >> 2);
#if SFM_FFN_ABL != 1
        if ((k & 3) == 3) w2_piece(c + 1, k >> 2);
#endif
      }
      s1 = s1n;
    }
    ff_wait_vmcnt<8>();                                     // W2(c) is in
    ff_barrier();                                           // U(c) complete; W1(c+1) and W2(c-1) have been read
    // ---- Y(c): acc2[64 x 64] += U[64 rows x 64] * W2c[64 out cols x 64]^T ----
    gemm2(c, true);
    ff_wait_vmcnt<8>();                                     // W1(c+2) is in
    ff_barrier();                                           // U(c) and W2(c) have been read
  }
  // ---- last chunk: no GEMM1 left, so H's registers take the residual rows of x now (prologue layout: wave w rows 16w..+15,
  //      4 consecutive columns per lane) and the loads fly under the last Swish and GEMM2; no refills either ----
  int ln = lane;                                           // opaque: addresses built from it are computed here, not
  asm volatile("" : "+v"(ln));                             // hoisted above the chunk loop and spilled across it
  f32x4 xv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int m = m0 + wave * 16 + r;
    m = m < M ? m : M - 1;
    xv[r] = *reinterpret_cast<const f32x4*>(x + (long long)m * FF_D + ln * 4);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) swish_quad(s1, q);
  ff_wait_vmcnt<20>();                                      // W2(nch-1) is in (behind it: the zeros of W1(nch+1), the 16 rows)
  ff_barrier();
  gemm2(nch - 1, false);
  ff_wait_vmcnt<0>();                                       // the refills past the last chunk (zeros) have landed in the
  ff_barrier();                                             // ring bytes the epilogue image is about to take

  FF_STAMP(3);
  // ---- epilogue in the prologue's row layout: acc2 -> fp32 image [128][260] -> wave w takes rows 16w..16w+15, a lane 4
  //      consecutive columns: out = x + alpha * (acc2 + b2) leaves as 1-KB row stores, and the LayerNorm of the NEXT sub-layer
  //      (mhsa.layer_norm after ff1, final_norm after ff2; models/conformer.py:66, 151) is a per-wave DPP reduction ----
  constexpr int IW = 260;
  float* img = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        img[(wm2 * 64 + i * 32 + mfma_row(r, lane)) * IW + wn2 * 64 + j * 32 + l31] = acc2[i][j][r];
  ff_barrier();
  FF_STAMP(4);
  const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + ln * 4);
  f32x4 g2 = {0.f, 0.f, 0.f, 0.f}, h2 = {0.f, 0.f, 0.f, 0.f};
  if (ln2w != nullptr) {
    g2 = *reinterpret_cast<const f32x4*>(ln2w + ln * 4);
    h2 = *reinterpret_cast<const f32x4*>(ln2b + ln * 4);
  }
  f32x4 yv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wave * 16 + r;
    const int m = m0 + row;
    const f32x4 a = *reinterpret_cast<const f32x4*>(&img[row * IW + ln * 4]);
#pragma unroll
    for (int e = 0; e < 4; ++e) yv[r][e] = xv[r][e] + alpha * (a[e] + bb[e]);
    if (out && m < M) *reinterpret_cast<f32x4*>(out + (long long)m * FF_D + ln * 4) = yv[r];
  }
  if (ln2w != nullptr) {                                   // block-uniform
    // the 16 row means, then the 16 centred variances, each by ONE transposing reduction (lane l gets the sum of row l & 15;
    // v_readlane hands row r's value back to every lane): ~150 instructions where 32 wave_sum_dpp chains took ~450
    float st[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = (yv[r][0] + yv[r][1]) + (yv[r][2] + yv[r][3]);
    const float mean_l = wave_sum16_transpose(st, lane) * (1.0f / FF_D);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float mean = ff_lane_bcast(mean_l, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) yv[r][e] -= mean;
      st[r] = (yv[r][0] * yv[r][0] + yv[r][1] * yv[r][1]) + (yv[r][2] * yv[r][2] + yv[r][3] * yv[r][3]);
    }
    const float rstd_l = rsqrtf(wave_sum16_transpose(st, lane) * (1.0f / FF_D) + eps);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 16 + r;
      if (m >= M) break;                                   // wave-uniform
      const float rstd = ff_lane_bcast(rstd_l, r);
      const float z0 = yv[r][0] * rstd * g2[0] + h2[0], z1 = yv[r][1] * rstd * g2[1] + h2[1];
      const float z2 = yv[r][2] * rstd * g2[2] + h2[2], z3 = yv[r][3] * rstd * g2[3] + h2[3];
      if (ln_out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ln_out) + (long long)m * FF_D + ln * 4) = f32x4{z0, z1, z2, z3};
      } else {
        u32x2 pk;
        pk[0] = pack2<T>(z0, z1);
        pk[1] = pack2<T>(z2, z3);
        *reinterpret_cast<u32x2*>(reinterpret_cast<u16*>(ln_out) + (long long)m * FF_D + ln * 4) = pk;
      }
    }
  }
  FF_STAMP(5);
}

// x, out [M, 256] fp32 contiguous rows; W1 [FF, 256], W2 [256, FF] 16-bit row-major (nn.Linear layout); fp32 biases.
// sfm_ffn_fused_ln: as sfm_ffn_fused, plus the LayerNorm that follows the module in the block (ln2w / ln2b) applied to
// y in the epilogue: ln_out [M, 256] 16-bit (feeds the next GEMM) or fp32; `out` (y itself) may be NULL when only the
// normalised rows are needed (ff2 -> final_norm).
extern "C" int sfm_ffn_fused_ln(const float* x, const float* lnw, const float* lnb, const void* W1, const float* b1,
                                const void* W2, const float* b2, float* out, int M, int D, int FF, float alpha, float eps,
                                const float* ln2w, const float* ln2b, void* ln_out, int ln_out_f32, int dtype,
                                void* stream) {
  if (!x || !lnw || !lnb || !W1 || !b1 || !W2 || !b2) return SFM_ERR_ARG;
  if (!out && !ln_out) return SFM_ERR_ARG;
  if ((ln2w || ln2b || ln_out) && !(ln2w && ln2b && ln_out)) return SFM_ERR_ARG;
  if (D != FF_D || FF <= 0 || FF % FF_CH != 0 || FF > 2048 || M <= 0) return SFM_ERR_SHAPE;
  const int lds_ring = 2 * (FF_CH * 512 + FF_D * 128) + FF_BM * 128 + FF * 4;
  const int lds_img = 128 * 260 * 4;                        // the fp32 image of the epilogue
  const int lds = lds_ring > lds_img ? lds_ring : lds_img;
  const int wbytes = FF * FF_D * 2;
  dim3 grid((M + FF_BM - 1) / FF_BM), block(512);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SFM_DT_F16) {
    static bool s16_dev[64] = {false};                     // hipFuncSetAttribute is per device
    int d16 = 0;
    if (hipGetDevice(&d16) != hipSuccess || d16 < 0 || d16 >= 64) return SFM_ERR_LAUNCH;
    bool& s16 = s16_dev[d16];
    if (!s16) {
      if (hipFuncSetAttribute((const void*)ffn_fused_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return SFM_ERR_LAUNCH;
      s16 = true;
    }
    SFM_LAUNCH((ffn_fused_kernel<F16>), grid, block, lds, st, x, lnw, lnb, (const u16*)W1, b1, (const u16*)W2, b2, out, M,
               FF, alpha, eps, wbytes, wbytes, ln2w, ln2b, ln_out, ln_out_f32);
  } else {
    static bool sb_dev[64] = {false};
    int db = 0;
    if (hipGetDevice(&db) != hipSuccess || db < 0 || db >= 64) return SFM_ERR_LAUNCH;
    bool& sb = sb_dev[db];
    if (!sb) {
      if (hipFuncSetAttribute((const void*)ffn_fused_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return SFM_ERR_LAUNCH;
      sb = true;
    }
    SFM_LAUNCH((ffn_fused_kernel<BF16>), grid, block, lds, st, x, lnw, lnb, (const u16*)W1, b1, (const u16*)W2, b2, out, M,
               FF, alpha, eps, wbytes, wbytes, ln2w, ln2b, ln_out, ln_out_f32);
  }
  return SFM_OK;
}

extern "C" int sfm_ffn_fused(const float* x, const float* lnw, const float* lnb, const void* W1, const float* b1,
                             const void* W2, const float* b2, float* out, int M, int D, int FF, float alpha, float eps,
                             int dtype, void* stream) {
  if (!out) return SFM_ERR_ARG;
  return sfm_ffn_fused_ln(x, lnw, lnb, W1, b1, W2, b2, out, M, D, FF, alpha, eps, nullptr, nullptr, nullptr, 0, dtype, stream);
}
