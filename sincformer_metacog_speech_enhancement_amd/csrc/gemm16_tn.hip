// Weight-gradient GEMM of every Linear / 1x1 conv on the path (backward of models/conformer.py:44,47,
// 113,122 and of in_proj/out_proj):   dW[n, k] += sum_m G[m, n] * X[m, k]      (G = dL/dY, X = layer input)
// Both operands are row-major with the CONTRACTION index m as the slow dimension, so both MFMA
// fragments (8 consecutive m for one n / one k) are transposed reads: the [64 m][128] 16-bit tiles are
// staged row-major in LDS (320-byte rows: the 4 rows x 64 B of a transposed read tile the 256-B bank
// row) and fetched with ds_read_b64_tr_b16.  Output tile 128 (n) x 128 (k), 4 waves x 64x64.
// M is split over gridDim.z; partial tiles are accumulated into the fp32 dW with float atomics
// (128-byte contiguous per wave-instruction).  sfm_colsum gives the bias gradient.
#include "sfm_common.h"
#include <cstdlib>

#define TN_ROW 160      // u16 elements per LDS row (128 + 32 pad) = 320 B

// conv addressing of X (weight gradient of a Conv1d on channels-last activations): row m = (b, l) of the im2col matrix is
// x[b, l*stride - pad + tap, ch] with k = tap*Cin + ch; positions outside [0, Lin) are the conv's zero padding.
// Lout == 0: plain row-major X with row stride ldx.
// toeplitz != 0 (tap gradient of the single-channel sinc FIR, Cin == 1): row m = (b, t) is x[b, t + k - K/2], k = 0..K-1, read
// from 8 zero-padded copies of the utterance shifted by 0..7 samples (sfm_sinc_shift_pack) so that the 8-tap chunk starting
// at any sample is one aligned 16-byte load: chunk(o) = copy[o & 7][o & ~7 ...], o = t + k + pad.
struct TnConv {
  int Lout, Lin, Cin, stride, pad;
  long long x_batch_stride;
  int toeplitz;
};

template <class T>
__global__ __launch_bounds__(256) void gemm16_tn_kernel(const u16* __restrict__ G, const u16* __restrict__ X,
                                                        float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                        int K, int ldg, int ldx, int ldw, int rows_per_split, TnConv cv,
                                                        float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) u16 Gs[64 * TN_ROW];
  __shared__ __attribute__((aligned(16))) u16 Xs[64 * TN_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  // XCD-aware order: consecutive linear workgroup ids go round-robin over the 8 XCDs; give each XCD a contiguous run of
  // (n-tile, k-tile, m-split) ids so that the tiles of one m-split (same G and X rows) share an L2
  int bx, by, bz;
  {
    const int total = gridDim.x * gridDim.y * gridDim.z;
    int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    bx = id % gridDim.x;
    const int rest = id / gridDim.x;
    by = rest % gridDim.y;
    bz = rest / gridDim.y;
  }
  const int n0 = bx * 128, k0 = by * 128;
  const int m_begin = bz * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: 64 rows x 16 chunks (16 B) per operand -> 4 chunks per thread per operand
  int srow[4], scol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    srow[i] = c >> 4;
    scol[i] = (c & 15) * 8;
  }
  u32x4 rg[4], rx[4];
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mt + srow[i];
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
      if (m < m_end) {
        if (n0 + scol[i] + 8 <= N) a = *reinterpret_cast<const u32x4*>(G + (long long)m * ldg + n0 + scol[i]);
        else {
          u16 t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = (n0 + scol[i] + e < N) ? G[(long long)m * ldg + n0 + scol[i] + e] : (u16)0;
          a = *reinterpret_cast<const u32x4*>(t);
        }
        if (cv.toeplitz) {
          const int kk = k0 + scol[i];
          if (kk < K) {
            const int bb = m / cv.Lout, t = m - bb * cv.Lout;
            const int o = t + kk + cv.pad;
            b = *reinterpret_cast<const u32x4*>(X + ((long long)bb * 8 + (o & 7)) * cv.x_batch_stride + (o & ~7));
          }
        } else if (cv.Lout > 0) {                              // implicit im2col row (Cin % 8 == 0: a chunk stays in one tap)
          const int kk = k0 + scol[i];
          if (kk < K) {
            const int bb = m / cv.Lout, l = m - bb * cv.Lout;
            const int tap = kk / cv.Cin, ch = kk - tap * cv.Cin;
            const int pos = l * cv.stride - cv.pad + tap;
            if (pos >= 0 && pos < cv.Lin)
              b = *reinterpret_cast<const u32x4*>(X + (long long)bb * cv.x_batch_stride + (long long)pos * cv.Cin + ch);
          }
        } else if (k0 + scol[i] + 8 <= K) b = *reinterpret_cast<const u32x4*>(X + (long long)m * ldx + k0 + scol[i]);
        else {
          u16 t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = (k0 + scol[i] + e < K) ? X[(long long)m * ldx + k0 + scol[i] + e] : (u16)0;
          b = *reinterpret_cast<const u32x4*>(t);
        }
      }
      rg[i] = a;
      rx[i] = b;
    }
  };

  // transposed-read lane coordinates (see attention.hip): 16-lane group g -> column block (g&1)*16,
  // contraction rows 4*(g>>1) + q (and +8); lane 4q+p supplies row q, columns 4p..4p+3
  const int g16 = lane >> 4, i16 = lane & 15;
  const int qq = i16 >> 2, pp = i16 & 3;
  const int trow = 4 * (g16 >> 1) + qq;
  const int tcol = (g16 & 1) * 16 + 4 * pp;

  // bias gradient (column sums of G): the k-tile-0 workgroups already stage every G tile in LDS
  const bool do_bias = (db != nullptr) && (by == 0);
  float bsum = 0.f;
  load_tile(m_begin);
  for (int mt = m_begin; mt < m_end; mt += 64) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(&Gs[srow[i] * TN_ROW + scol[i]]) = rg[i];
      *reinterpret_cast<u32x4*>(&Xs[srow[i] * TN_ROW + scol[i]]) = rx[i];
    }
    __syncthreads();
    if (mt + 64 < m_end) load_tile(mt + 64);
    if (do_bias) {                                             // thread (column c, row half): 32 rows of the 64-row tile
      const int c = tid & 127, r0 = (tid >> 7) * 32;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) bsum += T::to_f32(Gs[(r0 + r) * TN_ROW + c]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u16* base = &Gs[(16 * s + trow) * TN_ROW + wn * 64 + i * 32 + tcol];
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * TN_ROW));
        const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
        fa[i] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u16* base = &Xs[(16 * s + trow) * TN_ROW + wk * 64 + j * 32 + tcol];
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * TN_ROW));
        const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
        fb[j] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = T::mfma(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + mfma_row(r, lane);
        if (n < N && k < K) {
          // ws: this M-split's partial [N][K] (folded in split order by sfm_fold_partials: deterministic); else fp32 atomics
          if (ws) ws[((long long)bz * N + n) * K + k] = acc[i][j][r];
          else atomicAdd(&dW[(long long)n * ldw + k], acc[i][j][r]);
        }
      }
    }
  if (do_bias) {
    const int n = n0 + (tid & 127);
    if (n < N) {
      // (two threads per column, one per row half of the staged tile: two partial rows per split)
      if (ws) ws[(long long)gridDim.z * N * K + ((long long)bz * 2 + (tid >> 7)) * N + n] = bsum;
      else atomicAdd(&db[n], bsum);
    }
  }
}

// Wide form: 256 (n) x 256 (k) output tile on 16 waves (4 x 4, each wave the same 64 x 64 tile and fragment reads as
// above), plain row-major operands with N % 256 == 0 and K % 256 == 0.  Per 64-row step a workgroup stages 64 KB for 8.4 MFLOP
// (7.8 B per kFLOP of the L2 -> LDS operand stream instead of 15.6).  LDS: four [64][128] sub-tiles per operand pair
// (G0, G1, X0, X1), each with the 320-byte rows of the narrow kernel = 80 KB, one workgroup per CU.
template <class T>
__global__ __launch_bounds__(1024) void gemm16_tn_wide_kernel(const u16* __restrict__ G, const u16* __restrict__ X,
                                                              float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                              int K, int ldg, int ldx, int ldw, int rows_per_split,
                                                              float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) u16 tn_smem[];
  u16* Gs = tn_smem;                                   // [2][64][TN_ROW]
  u16* Xs = tn_smem + 2 * 64 * TN_ROW;                 // [2][64][TN_ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 2, wk = wave & 3;             // 64-column block of the wave inside the 256 x 256 tile
  int bx, by, bz;
  {
    const int total = gridDim.x * gridDim.y * gridDim.z;
    int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    bx = id % gridDim.x;
    const int rest = id / gridDim.x;
    by = rest % gridDim.y;
    bz = rest / gridDim.y;
  }
  const int n0 = bx * 256, k0 = by * 256;
  const int m_begin = bz * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: 64 rows x 32 chunks (16 B) per operand -> 2 chunks per thread per operand
  int srow[2], scol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 1024 * i;
    srow[i] = c >> 5;
    scol[i] = (c & 31) * 8;
  }
  u32x4 rg[2], rx[2];
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mt + srow[i];
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
      if (m < m_end) {
        a = *reinterpret_cast<const u32x4*>(G + (long long)m * ldg + n0 + scol[i]);
        b = *reinterpret_cast<const u32x4*>(X + (long long)m * ldx + k0 + scol[i]);
      }
      rg[i] = a;
      rx[i] = b;
    }
  };
  const int g16 = lane >> 4, i16 = lane & 15;
  const int qq = i16 >> 2, pp = i16 & 3;
  const int trow = 4 * (g16 >> 1) + qq;
  const int tcol = (g16 & 1) * 16 + 4 * pp;
  const bool do_bias = (db != nullptr) && (by == 0);
  float bsum = 0.f;
  const u16* Gw = Gs + (wn >> 1) * 64 * TN_ROW + (wn & 1) * 64;   // the wave's 64 columns inside its 128-column sub-tile
  const u16* Xw = Xs + (wk >> 1) * 64 * TN_ROW + (wk & 1) * 64;
  load_tile(m_begin);
  for (int mt = m_begin; mt < m_end; mt += 64) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int sub = scol[i] >> 7, col = scol[i] & 127;
      *reinterpret_cast<u32x4*>(&Gs[sub * 64 * TN_ROW + srow[i] * TN_ROW + col]) = rg[i];
      *reinterpret_cast<u32x4*>(&Xs[sub * 64 * TN_ROW + srow[i] * TN_ROW + col]) = rx[i];
    }
    __syncthreads();
    if (mt + 64 < m_end) load_tile(mt + 64);
    if (do_bias && tid < 256) {                              // thread = column of the 256-column G tile: 64 rows
      const u16* col = Gs + (tid >> 7) * 64 * TN_ROW + (tid & 127);
#pragma unroll 8
      for (int r = 0; r < 64; ++r) bsum += T::to_f32(col[r * TN_ROW]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u16* base = &Gw[(16 * s + trow) * TN_ROW + i * 32 + tcol];
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * TN_ROW));
        const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
        fa[i] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u16* base = &Xw[(16 * s + trow) * TN_ROW + j * 32 + tcol];
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * TN_ROW));
        const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
        fb[j] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = T::mfma(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + mfma_row(r, lane);
        if (ws) ws[((long long)bz * N + n) * K + k] = acc[i][j][r];
        else atomicAdd(&dW[(long long)n * ldw + k], acc[i][j][r]);
      }
    }
  if (do_bias && tid < 256) {
    if (ws) ws[(long long)gridDim.z * N * K + (long long)bz * N + n0 + tid] = bsum;
    else atomicAdd(&db[n0 + tid], bsum);
  }
}

// out[n] += sum_m G[m, n]   (bias gradient); G fp32 or 16-bit
template <class T>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ G, float* __restrict__ out, int M, int N,
                                                     int ldg, int g_f32, int rows_per_block, float* __restrict__ ws) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (g_f32) {
    const float* g = reinterpret_cast<const float*>(G);
    for (int m = m0; m < m1; ++m) s += g[(long long)m * ldg + n];
  } else {
    const u16* g = reinterpret_cast<const u16*>(G);
    for (int m = m0; m < m1; ++m) s += T::to_f32(g[(long long)m * ldg + n]);
  }
  if (ws) ws[(long long)blockIdx.y * N + n] = s;
  else atomicAdd(&out[n], s);
}

// vector form: 256 columns x 256 rows per workgroup, 16-byte loads, LDS reduction over the row lanes
template <class T, int F32>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const void* __restrict__ G, float* __restrict__ out, int M, int N,
                                                         int ldg, float* __restrict__ ws) {
  constexpr int VW = F32 ? 4 : 8;                 // columns per 16-byte load
  constexpr int TX = 256 / VW;                    // threads across the 256 columns
  constexpr int TY = 256 / TX;                    // row lanes
  __shared__ float red[TY][256];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int n0 = blockIdx.x * 256 + tx * VW;
  const int m0 = blockIdx.y * 256, m1 = min(M, m0 + 256);
  float acc[VW];
#pragma unroll
  for (int i = 0; i < VW; ++i) acc[i] = 0.f;
  if (n0 < N) {
    for (int m = m0 + ty; m < m1; m += TY) {
      if (F32) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(G) + (long long)m * ldg + n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += v[i];
      } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(G) + (long long)m * ldg + n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] += T::to_f32((u16)(v[i] & 0xffffu));
          acc[2 * i + 1] += T::to_f32((u16)(v[i] >> 16));
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VW; ++i) red[ty][tx * VW + i] = acc[i];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < TY; ++j) s += red[j][threadIdx.x];
    if (ws) ws[(long long)blockIdx.y * N + n] = s;
    else atomicAdd(&out[n], s);
  }
}

// gemm16_tn2.hip: the LDS-DMA ring form (SFM_ERR_SHAPE = not a shape it takes)
int sfm_tn2_launch(const void* G, const void* X, float* dW, float* db, int M, int N, int K, int ldg, int ldx, int ldw, int dtype,
                   void* stream, const TnConv& cv, long long x_elems, int variant, float* ws, long long ws_floats);

// deterministic mode (ws != NULL): every M-split writes its partial dW [N][K] (+ db [N]) into ws and sfm_fold_partials adds the
// partials to dW / db in split order.  An upper bound of the floats any of the kernels below needs for (M, N, K):
extern "C" long long sfm_tn_ws_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const long long t128 = (long long)((N + 127) / 128) * ((K + 127) / 128), t256 = (long long)((N + 255) / 256) * ((K + 255) / 256);
  long long s = 768 / t128;
  if (256 / t256 > s) s = 256 / t256;
  if (s < 1) s = 1;
  const long long max_splits = (M + 255) / 256;
  if (s > max_splits) s = max_splits;
  return (s + 1) * ((long long)N * K + 4LL * N);            // (+ up to 4 bias partial rows per split)
}

// second pass of the deterministic mode; bias_rows = bias partial rows per split (threads that share a column in the kernel)
int tn_fold(float* ws, float* dW, float* db, int N, int K, int ldw, int splits, int bias_rows, void* stream) {
  int rc = sfm_fold_partials(ws, dW, N, K, ldw, splits, 1, stream);
  if (rc == SFM_OK && db) rc = sfm_fold_partials(ws + (long long)splits * N * K, db, 1, N, N, splits * bias_rows, 1, stream);
  return rc;
}

static int gemm16_tn_launch(const void* G, const void* X, float* dW, float* db, int M, int N, int K, int ldg, int ldx, int ldw,
                            int dtype, void* stream, TnConv cv, long long x_elems, float* ws, long long ws_floats) {
  if (!G || !X || !dW) return SFM_ERR_ARG;
  if (ws && (((uintptr_t)ws) % 16) != 0) return SFM_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (ldg % 8) != 0 || (cv.Lout == 0 && (ldx % 8) != 0)) return SFM_ERR_SHAPE;
  // The LDS-DMA ring kernel (gemm16_tn2.hip) takes the Conv1d / sinc-FIR weight gradients (M = 1 .. 16 M im2col rows: 1.3-1.4x the first
  // kernel, profiles/README.md round 3); on the plain K = 256 / 1024 linears the 256 x 256-tile kernel below stays ahead.
  // A/B knob (tools/gemm_tn_bench.py): SFM_TN2 = 0 first kernels only, 1 / 2 = ring kernel everywhere, 128 x 128 / 128 x 256 tiles
  static const int tn2 = getenv("SFM_TN2") ? atoi(getenv("SFM_TN2")) : -1;
  const bool tn2_auto = (tn2 < 0) && cv.Lout > 0;              // conv and the sinc bank's Toeplitz form (2.6 -> 1.8 ms at B 256 x 4 s)
  if ((tn2 > 0 || tn2_auto) && M >= 4096) {
    const int rc = sfm_tn2_launch(G, X, dW, db, M, N, K, ldg, ldx, ldw, dtype, stream, cv, x_elems, tn2 > 0 ? tn2 : 2, ws, ws_floats);
    if (rc != SFM_ERR_SHAPE) return rc;
  }
  static const int wide_on = getenv("SFM_TN_WIDE") ? atoi(getenv("SFM_TN_WIDE")) : 1;          // A/B knob
  if (wide_on && cv.Lout == 0 && cv.toeplitz == 0 && (N % 256) == 0 && (K % 256) == 0 && M >= 8192) {
    const int tiles_w = (N / 256) * (K / 256);
    int splits = 256 / tiles_w;                                // one 16-wave workgroup per CU: one full round
    if (splits < 1) splits = 1;
    const int max_splits = (M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    int rows = (M + splits - 1) / splits;
    rows = (rows + 63) / 64 * 64;
    splits = (M + rows - 1) / rows;
    if (ws && (long long)splits * ((long long)N * K + N) > ws_floats) return SFM_ERR_ARG;
    dim3 grid(N / 256, K / 256, splits), block(1024);
    const size_t lds = 4 * 64 * TN_ROW * sizeof(u16);
    static bool attr_set_dev[64] = {false};        // hipFuncSetAttribute is per device
  int attr_dev_ = 0;
  if (hipGetDevice(&attr_dev_) != hipSuccess || attr_dev_ < 0 || attr_dev_ >= 64) return SFM_ERR_LAUNCH;
  bool& attr_set = attr_set_dev[attr_dev_];
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)gemm16_tn_wide_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipFuncSetAttribute((const void*)gemm16_tn_wide_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SFM_ERR_LAUNCH;
      attr_set = true;
    }
    if (dtype == SFM_DT_F16)
      SFM_LAUNCH((gemm16_tn_wide_kernel<F16>), grid, block, lds, (hipStream_t)stream, (const u16*)G, (const u16*)X, dW, db, M, N, K,
                 ldg, ldx, ldw, rows, ws);
    else
      SFM_LAUNCH((gemm16_tn_wide_kernel<BF16>), grid, block, lds, (hipStream_t)stream, (const u16*)G, (const u16*)X, dW, db, M, N, K,
                 ldg, ldx, ldw, rows, ws);
    return ws ? tn_fold(ws, dW, db, N, K, ldw, splits, 1, stream) : SFM_OK;
  }
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  // 142 registers -> 3 workgroups per CU: aim at ONE full round of 768 resident workgroups (1024 was 1.3 rounds: the
  // second round ran on a third of the chip)
  static const int target = getenv("SFM_TN_TARGET") ? atoi(getenv("SFM_TN_TARGET")) : 768;   // A/B knob (tools/gemm_tn_bench.py)
  int splits = target / tiles;
  const int max_splits = (M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows = (M + splits - 1) / splits;
  rows = (rows + 63) / 64 * 64;
  splits = (M + rows - 1) / rows;
  if (ws && (long long)splits * ((long long)N * K + 2LL * N) > ws_floats) return SFM_ERR_ARG;
  dim3 grid((N + 127) / 128, (K + 127) / 128, splits), block(256);
  if (dtype == SFM_DT_F16)
    SFM_LAUNCH((gemm16_tn_kernel<F16>), grid, block, 0, (hipStream_t)stream, (const u16*)G, (const u16*)X, dW, db, M, N, K,
               ldg, ldx, ldw, rows, cv, ws);
  else
    SFM_LAUNCH((gemm16_tn_kernel<BF16>), grid, block, 0, (hipStream_t)stream, (const u16*)G, (const u16*)X, dW, db, M, N, K,
               ldg, ldx, ldw, rows, cv, ws);
  return ws ? tn_fold(ws, dW, db, N, K, ldw, splits, 2, stream) : SFM_OK;
}

// db (optional): bias gradient out[n] += sum_m G[m, n], computed from the G tiles the k-tile-0 workgroups stage anyway
extern "C" int sfm_gemm16_tn(const void* G, const void* X, float* dW, float* db, int M, int N, int K, int ldg, int ldx,
                             int ldw, int dtype, float* ws, long long ws_floats, void* stream) {
  TnConv cv = {0, 0, 0, 0, 0, 0, 0};
  return gemm16_tn_launch(G, X, dW, db, M, N, K, ldg, ldx, ldw, dtype, stream, cv, (long long)(M - 1) * ldx + K, ws, ws_floats);
}

// Conv1d weight gradient: G = dY [B*Lout, N] (16-bit), x [B, Lin, Cin] channels-last 16-bit;
// dW [N, ksize*Cin] (tap-major, the layout of the packed conv weight) += sum_m G[m, n] * im2col(x)[m, k]
extern "C" int sfm_conv_wgrad16(const void* G, const void* x, float* dW, float* db, int B, int Lout, int Lin, int Cin, int N,
                                int ksize, int stride, int pad, long long x_batch_stride, int ldg, int ldw, int dtype,
                                float* ws, long long ws_floats, void* stream) {
  if (B <= 0 || Lout <= 0 || Lin <= 0 || Cin <= 0 || (Cin % 8) != 0 || ksize <= 0 || stride <= 0) return SFM_ERR_SHAPE;
  TnConv cv = {Lout, Lin, Cin, stride, pad, x_batch_stride, 0};
  return gemm16_tn_launch(G, x, dW, db, B * Lout, N, ksize * Cin, ldg, 0, ldw, dtype, stream, cv,
                          (long long)(B - 1) * x_batch_stride + (long long)Lin * Cin, ws, ws_floats);
}

// Tap gradient of the sinc FIR bank (agents/perception.py:115-118 backward) on the matrix cores.
// sfm_sinc_shift_pack: x fp32 [B, L] -> xs 16-bit [B][8][Lc], xs[b][j][i] = x[b, i + j - 128] (0 outside [0, L)),
//   Lc = sfm_sinc_shift_len(L) (covers every chunk the GEMM reads).
// sfm_sinc_wgrad16: dW [C][256] fp32 += sum_{b,t} dy[b, t, c] * x[b, t + k - 125]   (columns 251..255: don't care)
template <class T>
__global__ __launch_bounds__(256) void sinc_shift_pack_kernel(const float* __restrict__ x, u16* __restrict__ xs, int L, int Lc) {
  const int b = blockIdx.z, j = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Lc) return;
  const int s = i + j - 128;
  const float v = (s >= 0 && s < L) ? x[(long long)b * L + s] : 0.f;
  xs[((long long)b * 8 + j) * Lc + i] = T::from_f32(v);
}

extern "C" long long sfm_sinc_shift_len(int L) { return ((long long)L + 272 + 7) / 8 * 8; }

extern "C" int sfm_sinc_shift_pack(const float* x, void* xs, int B, int L, int dtype, void* stream) {
  if (!x || !xs) return SFM_ERR_ARG;
  if (B <= 0 || L <= 0 || B > 65535) return SFM_ERR_SHAPE;
  const int Lc = (int)sfm_sinc_shift_len(L);
  dim3 grid((Lc + 255) / 256, 8, B), block(256);
  if (dtype == SFM_DT_F16) SFM_LAUNCH((sinc_shift_pack_kernel<F16>), grid, block, 0, (hipStream_t)stream, x, (u16*)xs, L, Lc);
  else SFM_LAUNCH((sinc_shift_pack_kernel<BF16>), grid, block, 0, (hipStream_t)stream, x, (u16*)xs, L, Lc);
  return SFM_OK;
}

extern "C" int sfm_sinc_wgrad16(const void* dy, const void* xs, float* dW, int B, int L, int C, int dtype, float* ws,
                                long long ws_floats, void* stream) {
  if (B <= 0 || L <= 0 || C <= 0 || (C % 8) != 0 || (long long)B * L > 2000000000LL) return SFM_ERR_SHAPE;
  TnConv cv = {L, L, 1, 1, 128 - 125, sfm_sinc_shift_len(L), 1};
  return gemm16_tn_launch(dy, xs, dW, nullptr, B * L, C, 256, C, 0, 256, dtype, stream, cv, (long long)B * 8 * sfm_sinc_shift_len(L),
                          ws, ws_floats);
}

// ws (optional, >= sfm_colsum_ws_floats): one partial row per row-block, folded in block order (deterministic) instead of atomics
extern "C" long long sfm_colsum_ws_floats(int M, int N) { return (long long)((M + 255) / 256 + 1) * (N > 0 ? N : 0); }

extern "C" int sfm_colsum(const void* G, float* out, int M, int N, int ldg, int g_f32, int dtype, float* ws, void* stream) {
  if (!G || !out) return SFM_ERR_ARG;
  if (M <= 0 || N <= 0) return SFM_ERR_SHAPE;
  const int vw = g_f32 ? 4 : 8;
  if (N % vw == 0 && ldg % vw == 0 && ((uintptr_t)G % 16) == 0) {
    dim3 grid((N + 255) / 256, (M + 255) / 256), block(256);
    if (g_f32) SFM_LAUNCH((colsum_vec_kernel<BF16, 1>), grid, block, 0, (hipStream_t)stream, G, out, M, N, ldg, ws);
    else if (dtype == SFM_DT_F16) SFM_LAUNCH((colsum_vec_kernel<F16, 0>), grid, block, 0, (hipStream_t)stream, G, out, M, N, ldg, ws);
    else SFM_LAUNCH((colsum_vec_kernel<BF16, 0>), grid, block, 0, (hipStream_t)stream, G, out, M, N, ldg, ws);
    return ws ? sfm_fold_partials(ws, out, 1, N, N, (int)grid.y, 1, stream) : SFM_OK;
  }
  int rpb = 512;
  dim3 grid((N + 255) / 256, (M + rpb - 1) / rpb), block(256);
  if (dtype == SFM_DT_F16)
    SFM_LAUNCH((colsum_kernel<F16>), grid, block, 0, (hipStream_t)stream, G, out, M, N, ldg, g_f32, rpb, ws);
  else
    SFM_LAUNCH((colsum_kernel<BF16>), grid, block, 0, (hipStream_t)stream, G, out, M, N, ldg, g_f32, rpb, ws);
  return ws ? sfm_fold_partials(ws, out, 1, N, N, (int)grid.y, 1, stream) : SFM_OK;
}
