// Fused Conformer feed-forward module (models/conformer.py:41-49, eval):
//     out = x + alpha * ( W2 * swish(W1 * LayerNorm(x) + b1) + b2 )          D = 256, FF % 64 == 0
// One workgroup (8 waves) owns 128 rows.  x is read from HBM once, normalised in registers and kept
// as the 16-bit A operand [128 x 256] in LDS for the whole kernel; the FF-wide hidden activation never
// leaves the CU: per 64-unit chunk  S1 = H W1c^T (MFMA) -> +b1, swish -> 16-bit U in LDS ->
// acc2 += U W2c^T (MFMA).  The GEMM1 A operand of a wave (its 32 rows of LN(x), all 256 k) stays in
// registers for the whole kernel; weight chunks (W1c + W2c = 64 KB) stream from L2 by LDS-DMA into a
// 2-stage ring one full chunk ahead; two raw barriers per chunk, counted vmcnt.
// HBM traffic: x in (1 KB/row) + out (1 KB/row) instead of ~8 KB/row for LN + two GEMM launches.
#include "sfm_common.h"

#define FF_D 256
#define FF_BM 128
#define FF_CH 64                        // hidden units per chunk

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void ff_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ff_barrier() {
  // LDS writes of this wave must have completed before other waves pass the barrier; the LDS-DMA
  // queue (vmcnt) is deliberately NOT drained here - that is what the counted waits are for.
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

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
  auto w1_rs = __builtin_amdgcn_make_buffer_rsrc((void*)W1, 0, w1_bytes, 0x00020000);
  auto w2_rs = __builtin_amdgcn_make_buffer_rsrc((void*)W2, 0, w2_bytes, 0x00020000);

  // LDS-DMA pieces: W1 chunk = 64 rows x 512 B (4 x 1 KB per wave, 2 rows each, chunk c at c ^ (row & 15));
  //                 W2 chunk = 256 rows x 128 B (4 x 1 KB per wave, 8 rows each, chunk c at c ^ ((row >> 1) & 7))
  auto issue_w = [&](int c, int stage) {
    unsigned char* w1s = smem + stage * STG;
    unsigned char* w2s = w1s + FF_CH * 512;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int inst = wave * 4 + i;
      const int row = inst * 2 + (lane >> 5);
      const int lc = (lane & 31) ^ (row & 15);
      const int voff = ((c * FF_CH + row) * FF_D + lc * 8) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w1_rs, (lds_ptr_t)(w1s + inst * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int inst = wave * 4 + i;
      const int row = inst * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ ((row >> 1) & 7);
      const int voff = (row * FF + c * FF_CH + lc * 8) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rs, (lds_ptr_t)(w2s + inst * 1024), 16, voff, 0, 0, 0);
    }
  };

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
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 16 + r;
      const f32x4 v = xv[r];
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
  const int nch = FF / FF_CH;
  issue_w(0, 0);
  if (nch > 1) issue_w(1, 1);

  // wave roles: GEMM1 tile = rows 32*(w>>1), hidden cols 32*(w&1); GEMM2 tile = rows 64*(w>>2), out cols 64*(w&3)
  const int wm2 = wave >> 2, wn2 = wave & 3;
  f32x16 acc2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;

  if (nch > 1) ff_wait_vmcnt<8>(); else ff_wait_vmcnt<0>();   // chunk 0 landed (chunk 1 may be in flight)
  ff_barrier();
  for (int c = 0; c < nch; ++c) {
    const unsigned char* W1s = smem + (c & 1) * STG;
    const unsigned char* W2s = W1s + FF_CH * 512;
    // ---- GEMM1 (transposed): S1^T[32 units x 32 rows] = W1c[32 units x 256] * H[32 rows x 256]^T ----
    // (W1c is the MFMA A operand, the register-resident H the B operand: a lane then holds 4 CONSECUTIVE
    //  hidden units of one row per register quad, so U goes to LDS with 8-byte stores)
    f32x16 s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const u32x4 fw = *reinterpret_cast<const u32x4*>(W1s + n1 * 512 + (((2 * s + hl) ^ (n1 & 15)) << 4));
      s1 = T::mfma(fw, hf[s], s1);
    }
    {
      const int row = (wave >> 1) * 32 + l31;                // x row of this lane
      const int ub = (wave & 1) * 32;                        // first hidden unit (inside the chunk) of this wave
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u0 = ub + 8 * q + 4 * hl;                  // units u0 .. u0+3 = registers 4q .. 4q+3
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b1s + c * FF_CH + u0);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = s1[4 * q + e] + bv[e];
          y[e] = z * __frcp_rn(1.0f + __expf(-z));
        }
        u32x2 pk;
        pk[0] = pack2<T>(y[0], y[1]);
        pk[1] = pack2<T>(y[2], y[3]);
        const int chunk = (u0 >> 3) ^ ((row >> 1) & 7);
        *reinterpret_cast<u32x2*>(Us + row * 128 + chunk * 16 + (u0 & 7) * 2) = pk;
      }
    }
    ff_barrier();                                           // U(c) complete (U(c-1) readers finished before the last barrier)
    // ---- GEMM2: acc2[64 x 64] += U[64 rows x 64] * W2c[64 out cols x 64]^T ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm2 * 64 + i * 32 + l31;
        fa[i] = *reinterpret_cast<const u32x4*>(Us + row * 128 + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wn2 * 64 + j * 32 + l31;
        fb[j] = *reinterpret_cast<const u32x4*>(W2s + row * 128 + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = T::mfma(fa[i], fb[j], acc2[i][j]);
    }
    ff_wait_vmcnt<0>();                                      // own pieces of chunk c+1 have landed
    ff_barrier();                                           // stage (c & 1) and U are free; chunk c+1 complete for all
    if (c + 2 < nch) issue_w(c + 2, c & 1);
  }

  // ---- epilogue: acc2 -> per-wave fp32 image [64][68] -> out = x + alpha * (acc2 + b2), 16-byte row stores ----
  float* img = reinterpret_cast<float*>(smem) + wave * (64 * 68);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) img[(i * 32 + mfma_row(r, lane)) * 68 + j * 32 + l31] = acc2[i][j][r];
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  const int c8 = (lane & 7) * 8, rsub = lane >> 3;
  const int ncol = wn2 * 64 + c8;
  const f32x4 bb0 = *reinterpret_cast<const f32x4*>(b2 + ncol);
  const f32x4 bb1 = *reinterpret_cast<const f32x4*>(b2 + ncol + 4);
  float ov[8][8];                                        // this lane's 8 rows x 8 columns of y = x + alpha * ffn(x)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + rsub;
    const int m = m0 + wm2 * 64 + row;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(&img[row * 68 + c8]);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(&img[row * 68 + c8 + 4]);
    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      const float* xp = x + (long long)m * FF_D + ncol;
      x0 = *reinterpret_cast<const f32x4*>(xp);
      x1 = *reinterpret_cast<const f32x4*>(xp + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ov[it][e] = x0[e] + alpha * (a0[e] + bb0[e]);
      ov[it][4 + e] = x1[e] + alpha * (a1[e] + bb1[e]);
    }
    if (out && m < M) {
      float* op = out + (long long)m * FF_D + ncol;
      *reinterpret_cast<f32x4*>(op) = f32x4{ov[it][0], ov[it][1], ov[it][2], ov[it][3]};
      *reinterpret_cast<f32x4*>(op + 4) = f32x4{ov[it][4], ov[it][5], ov[it][6], ov[it][7]};
    }
  }
  if (ln2w == nullptr) return;                           // block-uniform

  // ---- fused LayerNorm of the NEXT sub-layer on y (mhsa.layer_norm after ff1, final_norm after ff2; models/conformer.py:
  // 66, 151): a row's 256 columns live in the 4 waves wn2 = 0..3, so the row statistics cross the waves through LDS ----
  float* red = reinterpret_cast<float*>(smem + 8 * 64 * 68 * 4);   // [128 rows][4] behind the images
  float mean[8], rstd[8];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      float v = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = pass ? (ov[it][e] - mean[it]) : ov[it][e];
        v += pass ? d * d : d;
      }
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      if ((lane & 7) == 0) red[(wm2 * 64 + it * 8 + rsub) * 4 + wn2] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(&red[(wm2 * 64 + it * 8 + rsub) * 4]);
      const float t = (q[0] + q[1] + q[2] + q[3]) * (1.0f / FF_D);
      if (pass == 0) mean[it] = t;
      else rstd[it] = rsqrtf(t + eps);
    }
    __syncthreads();
  }
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln2w + ncol), g1 = *reinterpret_cast<const f32x4*>(ln2w + ncol + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(ln2b + ncol), h1 = *reinterpret_cast<const f32x4*>(ln2b + ncol + 4);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int m = m0 + wm2 * 64 + it * 8 + rsub;
    if (m >= M) continue;
    float y[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] = (ov[it][e] - mean[it]) * rstd[it] * g0[e] + h0[e];
      y[4 + e] = (ov[it][4 + e] - mean[it]) * rstd[it] * g1[e] + h1[e];
    }
    if (ln_out_f32) {
      float* op = reinterpret_cast<float*>(ln_out) + (long long)m * FF_D + ncol;
      *reinterpret_cast<f32x4*>(op) = f32x4{y[0], y[1], y[2], y[3]};
      *reinterpret_cast<f32x4*>(op + 4) = f32x4{y[4], y[5], y[6], y[7]};
    } else {
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = pack2<T>(y[2 * e], y[2 * e + 1]);
      *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(ln_out) + (long long)m * FF_D + ncol) = pk;
    }
  }
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
  const int lds_img = 8 * 64 * 68 * 4 + 128 * 4 * 4;       // images + the row-statistics exchange of the fused LayerNorm
  const int lds = lds_ring > lds_img ? lds_ring : lds_img;
  const int wbytes = FF * FF_D * 2;
  dim3 grid((M + FF_BM - 1) / FF_BM), block(512);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SFM_DT_F16) {
    static bool s16 = false;
    if (!s16) {
      if (hipFuncSetAttribute((const void*)ffn_fused_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return SFM_ERR_LAUNCH;
      s16 = true;
    }
    SFM_LAUNCH((ffn_fused_kernel<F16>), grid, block, lds, st, x, lnw, lnb, (const u16*)W1, b1, (const u16*)W2, b2, out, M,
               FF, alpha, eps, wbytes, wbytes, ln2w, ln2b, ln_out, ln_out_f32);
  } else {
    static bool sb = false;
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
