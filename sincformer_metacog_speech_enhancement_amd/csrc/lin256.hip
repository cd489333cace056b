// Linear layers with K = 256 inputs (models/conformer.py:57-59 Q | K | V projection; :92-96 pointwise_conv1 + GLU):
//     out[M, N] = A[M, 256] W^T + b          16-bit operands, fp32 accumulation, 16-bit result
//     out[M, N/2] = GLU(A W^T + b)           (W rows packed by ops.pack_linear(glu=True): 64-row groups = 32 values | 32 gates)
// A K = 256 GEMM on 128 x 128 tiles spends most of a launch outside its k-loop: 4 k-tiles per workgroup between a cold operand
// fetch and an epilogue (profiles/README.md, "Where the K = 256 linears ... spend their launch").  Here one 8-wave workgroup owns
// 128 rows for ALL N columns (the scheme of ffn_fused.hip's first GEMM): the A tile crosses LDS once and then lives in registers
// as MFMA operand fragments (a wave: its 32 rows x 256 k = 16 fragments), W streams L2 -> LDS in 64-row chunks (32 KB) through a
// 3-stage LDS-DMA ring whose refills are issued from inside the MFMA stream, and the epilogue of chunk c (bias, GLU,
// one rounding: the expressions of sfm_gemm16's epilogue, bit for bit) shares an instruction stream with the MFMAs of chunk c + 1.  Results cross a
// double-buffered LDS image so that they leave as 128-byte row segments.  One barrier per chunk.
// W is the MFMA's A operand (accumulators hold the transposed tile: a lane owns 4 consecutive output columns of one row).
#include "sfm_common.h"
#include "gemm16_epi.h"

#ifndef SFM_L2_ABL
#define SFM_L2_ABL 0                      // timing experiments (results wrong): 1 no row stores, 2 no W refills, 3 no image writes, 4 no MFMAs
#endif
#define L2_BM 128
#define L2_K 256
#define L2_STAGE 32768                    // one W chunk: 64 rows x 512 B
#define L2_NSTAGE 3
#define L2_IMG 16384                      // one result image: 128 rows x 128 B (64 columns of 16 bits)

typedef __attribute__((address_space(3))) void* l2_lds_ptr_t;

template <int N>
__device__ __forceinline__ void l2_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void l2_frag_read(u32x4& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void l2_frag_wait(u32x4& frag) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N) : "memory");
}
__device__ __forceinline__ void l2_barrier() {
  // this wave's LDS writes / reads have completed; the LDS-DMA queue (vmcnt) is NOT drained: counted waits do that
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// GLU: 0 = plain (64 output columns per chunk), 1 = GLU (a chunk's 64 W rows = 32 values | 32 gates -> 32 output columns)
// OTHER: 1 = the result is written in the other 16-bit format (a run-time flag here is a branch per convert inside the MFMA
// stream); 2 = fp32 result, stored straight from the accumulator quads (16 bytes per lane, no LDS image)
// LN: the operand rows are LayerNorm(X32[m, :256]) (models/conformer.py:88, conv.layer_norm in front of pointwise_conv1): one
// workgroup owns whole rows for all N columns, so the rows are normalised ONCE, in the prologue, from the fp32 residual stream
// (the expressions of layernorm256_kernel, pointwise.hip, bit for bit) and the 16-bit normalised tensor never exists in HBM.
struct L2Ln {
  const float* x;                         // [M, ldx] fp32 (NULL: A is the 16-bit operand)
  const float* w;
  const float* b;
  int ldx;
  float eps;
};
template <class T, int GLU, int OTHER, int LN>
__global__ __launch_bounds__(512) void lin256_kernel(const u16* __restrict__ A, const u16* __restrict__ W,
                                                     const float* __restrict__ bias, u16* __restrict__ out, int M, int NW,
                                                     int lda, int ldo, int a_bytes, int w_bytes, int o_bytes, L2Ln ln) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS map: W ring 3 x 32 KB (the first 64 KB hold the A tile [128][512 B] during the prologue), two result images, bias
  unsigned char* img0 = smem + L2_NSTAGE * L2_STAGE;
  float* bs = reinterpret_cast<float*>(img0 + 2 * L2_IMG);          // [NW] bias (packed row order)
  constexpr int CPG = GLU ? 2 : 1;                                  // chunks per 64-column result image

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hl = lane >> 5;
  const int m0 = blockIdx.x * L2_BM;
  const int nch = NW >> 6;
  auto a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  auto w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);
  auto o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, o_bytes, 0x00020000);

  // ---- prologue: the A tile by LDS-DMA (an instruction = 2 rows x 512 B; 16-byte chunk p of row r holds logical chunk
  //      p ^ (r & 15); rows >= M are outside the descriptor's range and arrive as zeros), the bias vector ----
  if constexpr (LN) {
    // wave w normalises rows 16w .. 16w + 15, a lane 4 consecutive columns; all 16 row loads are issued before the first use
    const f32x4 gw = *reinterpret_cast<const f32x4*>(ln.w + 4 * lane), gb = *reinterpret_cast<const f32x4*>(ln.b + 4 * lane);
    f32x4 xv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = m0 + wave * 16 + r;
      m = m < M ? m : M - 1;                                        // clamped: computed, never stored
      xv[r] = *reinterpret_cast<const f32x4*>(ln.x + (long long)m * ln.ldx + 4 * lane);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 16 + r;
      const f32x4 v = xv[r];
      const float mean = wave_sum_dpp((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
      float d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = v[j] - mean;
      const float rstd = rsqrtf(wave_sum_dpp((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f) + ln.eps);
      float y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = d[j] * rstd * gw[j] + gb[j];
      u32x2 pk;
      pk[0] = pack2<T>(y[0], y[1]);
      pk[1] = pack2<T>(y[2], y[3]);
      *reinterpret_cast<u32x2*>(smem + row * 512 + (((lane >> 1) ^ (row & 15)) << 4) + (lane & 1) * 8) = pk;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int inst = wave * 8 + i;
      const int row = inst * 2 + (lane >> 5);
      const int lc = (lane & 31) ^ (row & 15);
      const long long voff = ((long long)(m0 + row) * lda + lc * 8) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (l2_lds_ptr_t)(smem + inst * 1024), 16,
                                               voff < a_bytes ? (int)voff : a_bytes, 0, 0, 0);
    }
  }
  for (int i = tid; i < NW; i += 512) bs[i] = bias ? bias[i] : 0.f;
  l2_wait_vmcnt<0>();
  __syncthreads();
  const int r1 = (wave >> 1) * 32 + l31;                            // this lane's row of the tile
  u32x4 hf[16];                                                     // the wave's 32 rows x 256 k as MFMA B-operand fragments
#pragma unroll
  for (int s = 0; s < 16; ++s) hf[s] = *reinterpret_cast<const u32x4*>(smem + r1 * 512 + (((2 * s + hl) ^ (r1 & 15)) << 4));
  __syncthreads();                                                  // A image consumed: the ring may overwrite it

  // W chunk c -> ring stage c % 3: 64 rows x 512 B, 4 x 1 KB per wave (2 rows each), chunk p of row r = logical p ^ (r & 15).
  // A chunk index past the end is issued all the same (offset out of range -> zeros into ring bytes nobody reads): the counted
  // waits below stay the same from the first chunk to the last.
  auto w_piece = [&](int c, int stage, int i) {
    const int inst = wave * 4 + i;
    const int row = inst * 2 + (lane >> 5);
    const int lc = (lane & 31) ^ (row & 15);
    const int voff = c < nch ? ((c * 64 + row) * L2_K + lc * 8) * 2 : w_bytes;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (l2_lds_ptr_t)(smem + stage * L2_STAGE + inst * 1024), 16, voff, 0, 0, 0);
  };
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) w_piece(c, c, i);

  // this lane's W row inside a chunk (= MFMA A-operand row l31 of the wave's 32 units)
  const int half = wave & 1;
  const int n1 = GLU ? (half * 16 + l31 + ((l31 >= 16) ? 16 : 0)) : (half * 32 + l31);
  const uint32_t w_lane = (uint32_t)(uintptr_t)(l2_lds_ptr_t)smem + (uint32_t)(n1 * 512);
  const int hx4 = (hl ^ (n1 & 15)) << 4;                            // fragment k at w_lane + stage + (hx4 ^ (k << 5))
  // accumulator register 4q + e of this lane = unit 8q + 4hl + e of the wave = chunk row:
  auto unit_row = [&](int q) { return GLU ? (half * 16 + 8 * (q & 1) + 4 * hl + ((q >= 2) ? 32 : 0)) : (half * 32 + 8 * q + 4 * hl); };
  // (accumulators start from zero and the bias is an epilogue addend, as in sfm_gemm16: the two kernels then produce the SAME bits
  //  - same MFMA, same k order, same epilogue expressions -, so an utterance's result does not depend on which of them its batch
  //  size selects: tests/test_fullsize_properties_gpu.py::test_batch_independence_and_permutation)
  auto s_init = [&](int, f32x16& s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
  };
  // result of chunk c, register quad q -> 8 bytes of the image of group c / CPG (row r1; 16-byte chunk j at j ^ ((row >> 1) & 7))
  constexpr bool other = OTHER == 1;
  static_assert(!(GLU && OTHER == 2), "fp32 result: plain epilogue only");
  auto epi_quad = [&](const f32x16& s, int c, int q) {
    float y[4];
    int col;                                                        // first of 4 consecutive columns inside the 64-column image
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + c * 64 + unit_row(q));
    if (GLU) {
      if (q >= 2) return;                                           // the gates ride with their values
      const f32x4 bg = *reinterpret_cast<const f32x4*>(bs + c * 64 + unit_row(q + 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (s[4 * q + e] + bv[e]) * __builtin_amdgcn_rcpf(1.0f + __expf(-(s[4 * (q + 2) + e] + bg[e])));
      col = (c & 1) * 32 + half * 16 + 8 * q + 4 * hl;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = s[4 * q + e] + bv[e];
      col = half * 32 + 8 * q + 4 * hl;
    }
    if constexpr (OTHER == 2) {
      // fp32: the two images hold the two 32-column halves of ONE chunk (image = wave & 1), 128-byte rows of 8 x 4 floats;
      // a direct 16-byte store per lane from here (32 rows x 32 bytes per instruction) measured 286 against 219 us for sfm_gemm16
      unsigned char* img = img0 + (half ? L2_IMG : 0);
      const int ch = 2 * q + hl;                                    // 16-byte chunk = columns 8q + 4hl .. + 3 of the half
      *reinterpret_cast<f32x4*>(img + r1 * 128 + ((ch ^ ((r1 >> 1) & 7)) << 4)) = f32x4{y[0], y[1], y[2], y[3]};
      return;
    }
    u32x2 pk;
    pk[0] = pack2_out<T>(y[0], y[1], other);
    pk[1] = pack2_out<T>(y[2], y[3], other);
    unsigned char* img = img0 + (((c / CPG) & 1) ? L2_IMG : 0);
    *reinterpret_cast<u32x2*>(img + r1 * 128 + (((col >> 3) ^ ((r1 >> 1) & 7)) << 4) + (col & 7) * 2) = pk;
  };
  // image of group g -> HBM: 128 rows x 8 chunks of 16 B, two per thread; a wave stores 8 rows x 128 B per instruction
  const int ocols = GLU ? (NW >> 1) : NW;
  auto store_group = [&](int g) {
    const unsigned char* img = img0 + ((g & 1) ? L2_IMG : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i;
      const int row = idx >> 3, pc = idx & 7;
      const int lc = pc ^ ((row >> 1) & 7);
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + row * 128 + pc * 16);
      const int m = m0 + row, n = g * 64 + lc * 8;
      // buffer stores: ALWAYS two vector-memory operations per thread (the counted waits depend on it); rows / columns outside the
      // result get an out-of-range offset, which the descriptor's range check drops
      const int voff = (m < M && n < ocols) ? (m * ldo + n) * 2 : o_bytes;
      __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, voff, 0, 0);
    }
  };

  // fp32 result: both images of chunk c -> HBM, four 16-byte stores per thread (a wave stores 8 rows x 128 B per instruction)
  auto store_chunk32 = [&](int c) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int idx = tid + 512 * j;
        const int row = idx >> 3, pc = idx & 7;
        const int lc = pc ^ ((row >> 1) & 7);
        const u32x4 v = *reinterpret_cast<const u32x4*>(img0 + (i ? L2_IMG : 0) + row * 128 + pc * 16);
        const int m = m0 + row, n = c * 64 + i * 32 + lc * 4;
        const int voff = (m < M && n < ocols) ? (m * ldo + n) * 4 : o_bytes;
        __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, voff, 0, 0);
      }
  };

  // ---- chunk 0 alone ----
  l2_wait_vmcnt<8>();                                               // W(0) has landed (W(1), W(2) behind it)
  l2_barrier();
  f32x16 s1;
  s_init(0, s1);
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
  l2_wait_vmcnt<4>();                                               // W(1) has landed
  l2_barrier();                                                     // ... for everyone; W(0) has been read by everyone

  // ---- X(c), c = 0 .. nch - 2: MFMAs of chunk c + 1 | epilogue of chunk c | refill W(c + 3) -> the stage W(c) left |
  //      row stores of the image completed one barrier ago ----
  int stage_next = 1;                                               // stage of chunk c + 1
  int stage_free = 0;                                               // stage of chunk c (read in X(c - 1), free since its barrier)
  int stored = 0;                                                   // groups already written to HBM
  for (int c = 0; c + 1 < nch; ++c) {
    bool did_store = false;
    if constexpr (OTHER == 2) {
      if (c >= 1) {                                                 // chunk c - 1: out of the images, then they may be rewritten
        store_chunk32(c - 1);
        l2_barrier();
      }
    }
    if (SFM_L2_ABL != 1 && OTHER != 2 && c >= CPG && (c % CPG) == 0) {   // group c / CPG - 1 was completed in X(c - 1)
      store_group(stored);
      ++stored;
      did_store = true;
    }
    f32x16 s1n;
    s_init(c + 1, s1n);
    const uint32_t fbase = w_lane + (uint32_t)(stage_next * L2_STAGE);
    u32x4 fw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) l2_frag_read(fw[k], fbase + (uint32_t)(hx4 ^ (k << 5)));
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k <= 12) l2_frag_wait<3>(fw[k & 3]);
      else if (k == 13) l2_frag_wait<2>(fw[k & 3]);
      else if (k == 14) l2_frag_wait<1>(fw[k & 3]);
      else l2_frag_wait<0>(fw[k & 3]);
      if (SFM_L2_ABL != 4) s1n = T::mfma(fw[k & 3], hf[k], s1n);
      else asm volatile("" : "+v"(s1n), "+v"(fw[k & 3]));
      if (k + 4 < 16) l2_frag_read(fw[k & 3], fbase + (uint32_t)(hx4 ^ ((k + 4) << 5)));
      if (SFM_L2_ABL != 3 && (k & 3) == 1) epi_quad(s1, c, k >> 2);
      if (SFM_L2_ABL != 2 && (k & 3) == 3) w_piece(c + 3, stage_free, k >> 2);
    }
    s1 = s1n;
    // W(c + 2) has landed: behind it in this wave's queue are the 4 pieces of W(c + 3) and this period's row stores
    if (OTHER == 2) l2_wait_vmcnt<8>();                             // (fp32: 4 row stores per period; none in X(0), where 8 is lenient
                                                                    //  by nothing: W(2) then has only the 4 pieces of W(3) behind it)
    else if (did_store) l2_wait_vmcnt<6>();
    else l2_wait_vmcnt<4>();
    l2_barrier();                                                   // image of chunk c complete; W(c + 1) read by everyone
    stage_free = stage_next;
    stage_next = (stage_next == L2_NSTAGE - 1) ? 0 : stage_next + 1;
  }
  // ---- last chunk: its epilogue, then whatever has not been stored ----
  if constexpr (OTHER == 2) {
    if (nch >= 2) {                                                 // the images still hold chunk nch - 2
      store_chunk32(nch - 2);
      l2_barrier();
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) epi_quad(s1, nch - 1, q);
  l2_wait_vmcnt<0>();                                               // (the zero refills past the last chunk have landed)
  l2_barrier();
  if constexpr (OTHER == 2) {
    store_chunk32(nch - 1);
  } else {
    const int ngroups = (nch + CPG - 1) / CPG;
    for (; stored < ngroups; ++stored) store_group(stored);
  }
}

// A [M, lda] 16-bit rows with 256 valid columns, W [NW, 256] 16-bit row-major (nn.Linear layout; glu != 0: the rows in the
// order of ops.pack_linear(glu=True)), bias [NW] fp32 or NULL, out [M, ldo] 16-bit: NW columns (glu: NW / 2) in the operands'
// format or, when out_dtype differs from dtype, in the other 16-bit format; out_dtype 2 (SFM_DT_F32, plain epilogue only): fp32.
static int lin256_launch(const void* A, const float* X32, int ldx, const float* lnw, const float* lnb, float eps, const void* W,
                         const float* bias, void* out, int M, int NW, int lda, int ldo, int glu, int dtype, int out_dtype,
                         void* stream) {
  const bool has_ln = X32 != nullptr;
  if ((!A && !has_ln) || !W || !out || (has_ln && (!lnw || !lnb))) return SFM_ERR_ARG;
  if ((dtype != SFM_DT_BF16 && dtype != SFM_DT_F16) || (out_dtype != SFM_DT_BF16 && out_dtype != SFM_DT_F16 && out_dtype != 2))
    return SFM_ERR_ARG;
  const bool f32o = out_dtype == 2;
  if (M <= 0 || NW <= 0 || (NW % 64) != 0 || NW > 2048 || (glu && ((NW % 128) != 0 || f32o)) || (ldo % (f32o ? 4 : 8)) != 0 ||
      ldo < (glu ? NW / 2 : NW))
    return SFM_ERR_SHAPE;
  if (has_ln ? (ldx < L2_K || (ldx % 4) != 0 || f32o || (((uintptr_t)X32 | (uintptr_t)lnw | (uintptr_t)lnb) % 16) != 0)
             : (lda < L2_K || (lda % 8) != 0))
    return SFM_ERR_SHAPE;
  const long long a_bytes = has_ln ? 0 : (long long)(M - 1) * lda * 2 + L2_K * 2;
  const long long o_bytes = ((long long)(M - 1) * ldo + (glu ? NW / 2 : NW)) * (f32o ? 4 : 2);
  if (a_bytes >= (1LL << 31) || o_bytes >= (1LL << 31)) return SFM_ERR_SHAPE;   // 32-bit buffer offsets
  const int lds = L2_NSTAGE * L2_STAGE + 2 * L2_IMG + NW * 4;
  const int w_bytes = NW * L2_K * 2;
  const int other = f32o ? 2 : (out_dtype != dtype ? 1 : 0);
  dim3 grid((M + L2_BM - 1) / L2_BM), block(512);
  hipStream_t st = (hipStream_t)stream;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SFM_ERR_LAUNCH;
  static bool attr_set[64][24] = {{false}};
  const int ki = (has_ln ? 12 : 0) + (dtype == SFM_DT_F16 ? 6 : 0) + (glu ? 3 : 0) + other;
#define L2_FN(TT, G, O, N) (const void*)lin256_kernel<TT, G, O, N>
  const void* fns[24] = {L2_FN(BF16, 0, 0, 0), L2_FN(BF16, 0, 1, 0), L2_FN(BF16, 0, 2, 0), L2_FN(BF16, 1, 0, 0), L2_FN(BF16, 1, 1, 0), nullptr,
                         L2_FN(F16, 0, 0, 0),  L2_FN(F16, 0, 1, 0),  L2_FN(F16, 0, 2, 0),  L2_FN(F16, 1, 0, 0),  L2_FN(F16, 1, 1, 0),  nullptr,
                         L2_FN(BF16, 0, 0, 1), L2_FN(BF16, 0, 1, 1), nullptr, L2_FN(BF16, 1, 0, 1), L2_FN(BF16, 1, 1, 1), nullptr,
                         L2_FN(F16, 0, 0, 1),  L2_FN(F16, 0, 1, 1),  nullptr, L2_FN(F16, 1, 0, 1),  L2_FN(F16, 1, 1, 1),  nullptr};
#undef L2_FN
  if (!fns[ki]) return SFM_ERR_SHAPE;
  if (!attr_set[dev][ki]) {
    if (hipFuncSetAttribute(fns[ki], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return SFM_ERR_LAUNCH;
    attr_set[dev][ki] = true;
  }
  const L2Ln ln = {X32, lnw, lnb, ldx, eps};
#define L2_GO(TT, G, O, N) SFM_LAUNCH((lin256_kernel<TT, G, O, N>), grid, block, lds, st, (const u16*)A, (const u16*)W, bias, (u16*)out, \
                                      M, NW, lda, ldo, (int)a_bytes, w_bytes, (int)o_bytes, ln)
  switch (ki) {
    case 0: L2_GO(BF16, 0, 0, 0); break;
    case 1: L2_GO(BF16, 0, 1, 0); break;
    case 2: L2_GO(BF16, 0, 2, 0); break;
    case 3: L2_GO(BF16, 1, 0, 0); break;
    case 4: L2_GO(BF16, 1, 1, 0); break;
    case 6: L2_GO(F16, 0, 0, 0); break;
    case 7: L2_GO(F16, 0, 1, 0); break;
    case 8: L2_GO(F16, 0, 2, 0); break;
    case 9: L2_GO(F16, 1, 0, 0); break;
    case 10: L2_GO(F16, 1, 1, 0); break;
    case 12: L2_GO(BF16, 0, 0, 1); break;
    case 13: L2_GO(BF16, 0, 1, 1); break;
    case 15: L2_GO(BF16, 1, 0, 1); break;
    case 16: L2_GO(BF16, 1, 1, 1); break;
    case 18: L2_GO(F16, 0, 0, 1); break;
    case 19: L2_GO(F16, 0, 1, 1); break;
    case 21: L2_GO(F16, 1, 0, 1); break;
    default: L2_GO(F16, 1, 1, 1); break;
  }
#undef L2_GO
  return SFM_OK;
}

extern "C" int sfm_lin256(const void* A, const void* W, const float* bias, void* out, int M, int NW, int lda, int ldo, int glu,
                          int dtype, int out_dtype, void* stream) {
  return lin256_launch(A, nullptr, 0, nullptr, nullptr, 0.f, W, bias, out, M, NW, lda, ldo, glu, dtype, out_dtype, stream);
}

// the same with LayerNorm(X32[m, :256]; lnw, lnb, eps) as the operand rows (16-bit results only)
extern "C" int sfm_ln_lin256(const float* X32, int ldx, const float* lnw, const float* lnb, float eps, const void* W,
                             const float* bias, void* out, int M, int NW, int ldo, int glu, int dtype, int out_dtype, void* stream) {
  if (!X32) return SFM_ERR_ARG;
  return lin256_launch(nullptr, X32, ldx, lnw, lnb, eps, W, bias, out, M, NW, 0, ldo, glu, dtype, out_dtype, stream);
}
