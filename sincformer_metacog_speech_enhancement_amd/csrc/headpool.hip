// PerceptionAgent latent heads + the time pooling that follows them in the fused path, one launch:
//     raw[b, t, :] = xd[b, t, :256] W^T + bias            (agents/perception.py:183-199: real_proj | imag_proj 1x1 convs, stacked)
//     partial sums of raw, raw^2 per (utterance, row tile, group)  -> GroupNorm statistics of the heads (gn_finalize as before)
//     pooled[b, i, :] = mean over t in [floor(i Tin / Tout), ceil((i + 1) Tin / Tout)) of raw[b, t, :]      (glue G1)
// The GroupNorm that follows the heads has no activation, so it is affine per (utterance, channel) and commutes with the
// average: the caller applies it to `pooled` (sfm_pool_time_affine16 with Tin == Tout).  The full-rate raw tensor
// ([B, Tpa, 512] 16-bit: 670 MB written and 670 MB read back by the pooling pass at B 256 x 512 frames) never exists.
// Kernel = csrc/lin256.hip's scheme (A tile resident in registers, W through a 3-stage LDS-DMA ring, epilogue of a 64-column chunk
// under the next chunk's MFMAs) with row tiles cut at POOLED-FRAME boundaries of one utterance: a workgroup owns `fpt` pooled
// frames = at most 128 input rows.  A chunk's results cross a double-buffered 16-bit LDS image as in lin256 (the rounding the
// un-fused pair applies when it writes the raw tensor); instead of row stores, 16 lanes per pooled frame average the window's
// rows out of the image in fp32.  A row shared by two neighbouring windows at a
// tile boundary is computed by both tiles and counted in the statistics of the first only.  Statistics: fixed summation order
// (wave reduction, then the four row groups of the workgroup in order): bit-reproducible.
#include "sfm_common.h"

#define HP_BM 128
#define HP_K 256
#define HP_STAGE 32768
#define HP_NSTAGE 3
#define HP_IMG 16384

typedef __attribute__((address_space(3))) void* hp_lds_ptr_t;

template <int N>
__device__ __forceinline__ void hp_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void hp_frag_read(u32x4& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void hp_frag_wait(u32x4& frag) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N) : "memory");
}
__device__ __forceinline__ void hp_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <class T>
__global__ __launch_bounds__(512) void headpool_kernel(const u16* __restrict__ A, const u16* __restrict__ W,
                                                       const float* __restrict__ bias, u16* __restrict__ pooled,
                                                       float* __restrict__ gn_partial, int Tin, int Tout, int NW, int lda,
                                                       int ldp, int fpt, int P, int gcols, int w_bytes, int p_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* img0 = smem + HP_NSTAGE * HP_STAGE;
  float* bs = reinterpret_cast<float*>(img0 + 2 * HP_IMG);          // [NW] bias
  float* st = bs + NW;                                              // [4 row groups][NW / gcols groups][2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hl = lane >> 5;
  const int j = blockIdx.x % P, b = blockIdx.x / P;
  const int i0 = j * fpt, i1 = min(i0 + fpt, Tout);
  const int r_lo = (int)(((long long)i0 * Tin) / Tout);             // first input row of the tile
  const int own_end = (j + 1 < P) ? (int)(((long long)i1 * Tin) / Tout) : Tin;   // rows >= own_end belong to the next tile's statistics
  const int nch = NW >> 6;
  const int ngr = NW / gcols;
  auto a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long long)b * Tin * lda), 0, Tin * lda * 2, 0x00020000);
  auto w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);
  auto p_rs = __builtin_amdgcn_make_buffer_rsrc((void*)pooled, 0, p_bytes, 0x00020000);

  // ---- prologue: the A tile (rows r_lo .. r_lo + 127 of utterance b; rows >= Tin are outside the descriptor: zeros) ----
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int inst = wave * 8 + i;
    const int row = inst * 2 + (lane >> 5);
    const int lc = (lane & 31) ^ (row & 15);
    const int r = r_lo + row;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (hp_lds_ptr_t)(smem + inst * 1024), 16,
                                             r < Tin ? (r * lda + lc * 8) * 2 : Tin * lda * 2, 0, 0, 0);
  }
  for (int i = tid; i < NW; i += 512) bs[i] = bias ? bias[i] : 0.f;
  hp_wait_vmcnt<0>();
  __syncthreads();
  const int r1 = (wave >> 1) * 32 + l31;                            // this lane's row of the tile
  u32x4 hf[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) hf[s] = *reinterpret_cast<const u32x4*>(smem + r1 * 512 + (((2 * s + hl) ^ (r1 & 15)) << 4));
  __syncthreads();

  auto w_piece = [&](int c, int stage, int i) {
    const int inst = wave * 4 + i;
    const int row = inst * 2 + (lane >> 5);
    const int lc = (lane & 31) ^ (row & 15);
    const int voff = c < nch ? ((c * 64 + row) * HP_K + lc * 8) * 2 : w_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (hp_lds_ptr_t)(smem + stage * HP_STAGE + inst * 1024), 16, voff, 0, 0, 0);
  };
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) w_piece(c, c, i);

  const int half = wave & 1;
  const int n1 = half * 32 + l31;
  const uint32_t w_lane = (uint32_t)(uintptr_t)(hp_lds_ptr_t)smem + (uint32_t)(n1 * 512);
  const int hx4 = (hl ^ (n1 & 15)) << 4;
  const bool own = (r_lo + r1) < own_end;                           // this lane's row counts in this tile's statistics
  float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};                     // per-lane sums of the chunk in flight: the wave's two groups
  // result of chunk c, register quad q: + bias -> statistics (fp32), 16-bit image c & 1 (row r1, as lin256)
  auto epi_quad = [&](const f32x16& s, int c, int q) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + c * 64 + half * 32 + 8 * q + 4 * hl);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = s[4 * q + e] + bv[e];
    u32x2 pk;
    pk[0] = pack2<T>(y[0], y[1]);
    pk[1] = pack2<T>(y[2], y[3]);
    const int col = half * 32 + 8 * q + 4 * hl;                     // first of 4 consecutive columns inside the 64-column image
    unsigned char* img = img0 + ((c & 1) ? HP_IMG : 0);
    *reinterpret_cast<u32x2*>(img + r1 * 128 + (((col >> 3) ^ ((r1 >> 1) & 7)) << 4) + (col & 7) * 2) = pk;
    if (own) {
      // (gcols = 16: quads 0, 1 = the first group of the wave's 32 columns, quads 2, 3 = the second)
      const int g = (8 * q) / gcols;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gs[g] += y[e];
        gq[g] = __builtin_fmaf(y[e], y[e], gq[g]);
      }
    }
  };
  // after a chunk's four quads: the wave's sums -> st[row group][group of the chunk][2]
  auto stats_out = [&](int c) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float a = wave_sum_dpp(gs[g]), q2 = wave_sum_dpp(gq[g]);
      if (lane == 0) {
        float* d = st + (((wave >> 1) * ngr + (c * 64 + half * 32) / gcols + g) << 1);
        d[0] = a;
        d[1] = q2;
      }
      gs[g] = 0.f;
      gq[g] = 0.f;
    }
  };
  // pooled frame f = tid / 16 of the tile, 16 lanes x 4 columns = the 64 columns of a chunk (all 8 waves take part: 25 frames keep
  // 400 of the 512 threads busy): average the window's rows out of the image
  const int pf = tid >> 4, pc = tid & 15;                            // pc: 8-byte piece (4 columns) of the 128-byte image row
  const int fi = i0 + pf;
  int ws = 0, we = 0;
  if (fi < i1) {
    ws = (int)(((long long)fi * Tin) / Tout) - r_lo;
    we = (int)((((long long)(fi + 1)) * Tin + Tout - 1) / Tout) - r_lo;
  }
  const float winv = we > ws ? 1.0f / (float)(we - ws) : 0.f;
  auto pool_chunk = [&](int c) {
    const unsigned char* img = img0 + ((c & 1) ? HP_IMG : 0);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = ws; r < we; ++r) {
      const u32x2 v = *reinterpret_cast<const u32x2*>(img + r * 128 + (((pc >> 1) ^ ((r >> 1) & 7)) << 4) + (pc & 1) * 8);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t w = v[e];
        a[2 * e] += T::to_f32((u16)(w & 0xffffu));
        a[2 * e + 1] += T::to_f32((u16)(w >> 16));
      }
    }
    u32x2 pk;
    pk[0] = pack2<T>(a[0] * winv, a[1] * winv);
    pk[1] = pack2<T>(a[2] * winv, a[3] * winv);
    // ALWAYS one vector-memory operation per thread (the counted waits depend on it): lanes without a frame store out of range
    const int voff = (fi < i1) ? (((b * Tout + fi) * ldp) + c * 64 + pc * 4) * 2 : p_bytes;
    __builtin_amdgcn_raw_buffer_store_b64(pk, p_rs, voff, 0, 0);
  };

  // ---- chunk 0 alone ----
  hp_wait_vmcnt<8>();
  hp_barrier();
  f32x16 s1;
#pragma unroll
  for (int r = 0; r < 16; ++r) s1[r] = 0.f;
  {
    u32x4 fw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) fw[k] = *reinterpret_cast<const u32x4*>(smem + n1 * 512 + (((2 * k + hl) ^ (n1 & 15)) << 4));
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      s1 = T::mfma(fw[k & 3], hf[k], s1);
      if (k + 4 < 16) fw[k & 3] = *reinterpret_cast<const u32x4*>(smem + n1 * 512 + (((2 * (k + 4) + hl) ^ (n1 & 15)) << 4));
    }
  }
  hp_wait_vmcnt<4>();
  hp_barrier();

  int stage_next = 1, stage_free = 0;
  for (int c = 0; c + 1 < nch; ++c) {
    if (c >= 1) pool_chunk(c - 1);                                  // image (c - 1) & 1 was completed one barrier ago; chunk c goes to the other
    f32x16 s1n;
#pragma unroll
    for (int r = 0; r < 16; ++r) s1n[r] = 0.f;
    const uint32_t fbase = w_lane + (uint32_t)(stage_next * HP_STAGE);
    u32x4 fw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hp_frag_read(fw[k], fbase + (uint32_t)(hx4 ^ (k << 5)));
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k <= 12) hp_frag_wait<3>(fw[k & 3]);
      else if (k == 13) hp_frag_wait<2>(fw[k & 3]);
      else if (k == 14) hp_frag_wait<1>(fw[k & 3]);
      else hp_frag_wait<0>(fw[k & 3]);
      s1n = T::mfma(fw[k & 3], hf[k], s1n);
      if (k + 4 < 16) hp_frag_read(fw[k & 3], fbase + (uint32_t)(hx4 ^ ((k + 4) << 5)));
      if ((k & 3) == 1) epi_quad(s1, c, k >> 2);
      if ((k & 3) == 3) w_piece(c + 3, stage_free, k >> 2);
    }
    stats_out(c);
    s1 = s1n;
    // W(c + 2) has landed: behind it are the 4 pieces of W(c + 3) and (c >= 1) this period's pooled store
    if (c >= 1) hp_wait_vmcnt<5>();
    else hp_wait_vmcnt<4>();
    hp_barrier();
    stage_free = stage_next;
    stage_next = (stage_next == HP_NSTAGE - 1) ? 0 : stage_next + 1;
  }
  if (nch >= 2) pool_chunk(nch - 2);
#pragma unroll
  for (int q = 0; q < 4; ++q) epi_quad(s1, nch - 1, q);
  stats_out(nch - 1);
  hp_wait_vmcnt<0>();
  hp_barrier();
  pool_chunk(nch - 1);
  // statistics of the tile: the four row groups in order
  if (tid < ngr * 2) {
    const float v = ((st[tid] + st[ngr * 2 + tid]) + st[2 * ngr * 2 + tid]) + st[3 * ngr * 2 + tid];
    gn_partial[((long long)b * P + j) * ngr * 2 + tid] = v;
  }
}

// frames per tile for a (Tin, Tout) pair: the largest f <= 32 whose window union never exceeds 128 rows; 0 = not supported
extern "C" int sfm_headpool_frames_per_tile(int Tin, int Tout) {
  if (Tin <= 0 || Tout <= 0 || Tin < Tout) return 0;
  for (int f = 32; f >= 1; --f) {
    bool ok = true;
    for (int i0 = 0; i0 < Tout && ok; i0 += f) {
      const int i1 = i0 + f < Tout ? i0 + f : Tout;
      const long long lo = ((long long)i0 * Tin) / Tout, hi = (((long long)i1) * Tin + Tout - 1) / Tout;
      if (hi - lo > HP_BM) ok = false;
    }
    if (ok) return f;
  }
  return 0;
}

// xd [B, Tin, lda] 16-bit (256 valid columns), W [NW, 256] 16-bit, bias [NW] fp32 or NULL ->
//   pooled [B, Tout, ldp] 16-bit (NW columns): the time-pooled RAW head outputs,
//   gn_partial [B, P, NW / gcols, 2] fp32 with P = ceil(Tout / sfm_headpool_frames_per_tile(Tin, Tout)): sum and sum of squares of the
//   full-rate raw outputs per (utterance, tile, group of gcols channels) - the operand sfm_gn_finalize reduces.
extern "C" int sfm_headpool(const void* xd, const void* W, const float* bias, void* pooled, float* gn_partial, int B, int Tin,
                            int Tout, int NW, int lda, int ldp, int gcols, int dtype, void* stream) {
  if (!xd || !W || !pooled || !gn_partial) return SFM_ERR_ARG;
  if (dtype != SFM_DT_BF16 && dtype != SFM_DT_F16) return SFM_ERR_ARG;
  const int fpt = sfm_headpool_frames_per_tile(Tin, Tout);
  if (B <= 0 || fpt <= 0 || NW <= 0 || (NW % 64) != 0 || NW > 1024 || gcols != 16 || lda < HP_K || (lda % 8) != 0 || (ldp % 4) != 0 ||
      ldp < NW)
    return SFM_ERR_SHAPE;
  const int P = (Tout + fpt - 1) / fpt;
  const long long p_bytes = (long long)B * Tout * ldp * 2, a_b = (long long)Tin * lda * 2;
  if (p_bytes >= (1LL << 31) || a_b >= (1LL << 31) || (long long)B * P > 2147483647LL) return SFM_ERR_SHAPE;
  const int ngr = NW / gcols;
  const int lds = HP_NSTAGE * HP_STAGE + 2 * HP_IMG + NW * 4 + 4 * ngr * 2 * 4;
  const int w_bytes = NW * HP_K * 2;
  dim3 grid(B * P), block(512);
  hipStream_t st = (hipStream_t)stream;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SFM_ERR_LAUNCH;
  static bool attr_set[64][2] = {{false}};
  const int ki = dtype == SFM_DT_F16 ? 1 : 0;
  const void* fn = ki ? (const void*)headpool_kernel<F16> : (const void*)headpool_kernel<BF16>;
  if (!attr_set[dev][ki]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SFM_ERR_LAUNCH;
    attr_set[dev][ki] = true;
  }
  if (ki)
    SFM_LAUNCH((headpool_kernel<F16>), grid, block, lds, st, (const u16*)xd, (const u16*)W, bias, (u16*)pooled, gn_partial, Tin, Tout, NW,
               lda, ldp, fpt, P, gcols, w_bytes, (int)p_bytes);
  else
    SFM_LAUNCH((headpool_kernel<BF16>), grid, block, lds, st, (const u16*)xd, (const u16*)W, bias, (u16*)pooled, gn_partial, Tin, Tout, NW,
               lda, ldp, fpt, P, gcols, w_bytes, (int)p_bytes);
  return SFM_OK;
}
