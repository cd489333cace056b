// Weight-gradient GEMM, pipelined form:   dW[n, k] += sum_m G[m, n] * X[m, k]      (contract: gemm16_tn.hip)
// The first kernel stages both operands through registers into a single LDS tile (two workgroup barriers and 32 KB of
// ds_write_b128 per 64-row step); it is latency-bound at ~15 % of the matrix rate while the shapes of the training step are
// HBM-bound (M = 0.2 .. 8 M rows, N, K <= 1280).  Here:
//   * both operand tiles ([64 m][TN n] of G, [64 m][TK k] of X) go HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), no VGPR
//     staging, into a STAGES-deep ring: STAGES-1 steps (64 .. 96 KB per CU) stay in flight under the MFMAs, one raw s_barrier
//     and a counted s_waitcnt vmcnt per step;
//   * LDS-DMA writes lane-linear, so the rows are unpadded; the 64-byte segment s of row r lives at segment s ^ (r & 3)
//     (applied on the SOURCE address), which makes the ds_read_b64_tr_b16 fragment reads (4 rows x 64 B per 32 lanes)
//     conflict-free like the 320-byte rows of the first kernel;
//   * out-of-range rows / columns / conv padding are fetched with an out-of-range buffer offset (hardware returns 0);
//   * the im2col coordinates of a conv's X rows advance incrementally (no division in the loop).
// Tiles: (WNW x WKW) waves of 64 x 64; 2 x 2 (128 x 128, 4 stages) and 2 x 4 (128 n x 256 k, 3 stages), one workgroup per CU.
#include "sfm_common.h"

struct TnConv {            // = gemm16_tn.hip
  int Lout, Lin, Cin, stride, pad;
  long long x_batch_stride;
  int toeplitz;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void tn2_wait_vmcnt() {
  // + lgkmcnt(0): this wave's transposed reads of the slot refilled after the barrier have returned (WAR, see gemm16_epi.h wait_ring)
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <class T, int WNW, int WKW, int STAGES>
__global__ __launch_bounds__(WNW * WKW * 64) void gemm16_tn2_kernel(const u16* __restrict__ G, const u16* __restrict__ X,
                                                                    float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                                    int K, int ldg, int ldx, int ldw, int rows_per_split,
                                                                    unsigned g_records, unsigned x_records, TnConv cv,
                                                                    float* __restrict__ ws) {
  constexpr int NW = WNW * WKW;
  constexpr int TNc = 64 * WNW, TKc = 64 * WKW;               // tile columns of G (n) and X (k)
  constexpr int GROW = TNc * 2, XROW = TKc * 2;                 // LDS row bytes
  constexpr int G_STAGE = 64 * GROW, X_STAGE = 64 * XROW, STAGE = G_STAGE + X_STAGE;
  constexpr int NG = G_STAGE / 1024 / NW, NX = X_STAGE / 1024 / NW;   // LDS-DMA instructions per wave per stage
  constexpr int NLD = NG + NX;
  static_assert(G_STAGE % (1024 * NW) == 0 && X_STAGE % (1024 * NW) == 0, "whole DMA instructions per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char tn2_smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / WKW, wk = wave % WKW;
  // XCD-aware order (as the first kernel): each XCD gets a contiguous run of (n-tile, k-tile, m-split) ids, so the tiles of one
  // m-split (same G and X rows) share an L2
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
  const int n0 = bx * TNc, k0 = by * TKc;
  const int m_begin = bz * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  const int nsteps = (m_end - m_begin + 63) >> 6;

  auto g_rs = __builtin_amdgcn_make_buffer_rsrc((void*)G, 0, g_records, 0x00020000);
  auto x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, x_records, 0x00020000);

  // ---- per-lane source coordinates of the DMA pieces.  One instruction fills 1024 contiguous LDS bytes = 1024 / ROW rows; the
  //      lane at physical 16-byte chunk p of row r fetches the logical chunk ((p >> 2) ^ (r & 3)) << 2 | (p & 3) ----
  int g_row[NG], g_col[NG];                                    // row inside the 64-row step, first logical column (elements)
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int byte = (wave * NG + i) * 1024 + lane * 16;
    const int r = byte / GROW, p = (byte % GROW) >> 4;
    g_row[i] = r;
    const int c = ((((p >> 2) ^ (r & 3)) << 2) | (p & 3)) * 8;
    g_col[i] = (n0 + c + 8 <= N) ? n0 + c : -1;                // N % 8 == 0 (launcher): a chunk is inside or outside
  }
  int x_row[NX], x_col[NX];                                    // plain X: column (elements) or -1; conv: tap * Cin + ch
  int x_tapoff[NX], x_tap[NX];                                 // conv: tap, and ch (element offset inside a position)
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int byte = (wave * NX + i) * 1024 + lane * 16;
    const int r = byte / XROW, p = (byte % XROW) >> 4;
    x_row[i] = r;
    const int c = ((((p >> 2) ^ (r & 3)) << 2) | (p & 3)) * 8;
    const int kk = k0 + c;
    x_col[i] = (kk + 8 <= K || (cv.toeplitz && kk < K)) ? kk : -1;
    x_tap[i] = 0;
    x_tapoff[i] = 0;
    if (cv.Lout > 0 && !cv.toeplitz && x_col[i] >= 0) {
      x_tap[i] = kk / cv.Cin;
      x_tapoff[i] = kk - x_tap[i] * cv.Cin;
    }
  }
  // conv / toeplitz: (batch entry, position) of row m_begin + x_row[i] + 64 * step, advanced by 64 rows per issued step
  int cb[NX], cl[NX];
  if (cv.Lout > 0) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int m = m_begin + x_row[i];
      cb[i] = m / cv.Lout;
      cl[i] = m - cb[i] * cv.Lout;
    }
  }
  int issued = 0;                                              // steps issued so far
  auto issue = [&](int stage) {
    unsigned char* sg = tn2_smem + stage * STAGE;
    unsigned char* sx = sg + G_STAGE;
    const int mt = m_begin + issued * 64;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int m = mt + g_row[i];
      unsigned voff = 0xFFFFFFFFu;
      if (m < m_end && g_col[i] >= 0) voff = ((unsigned)m * (unsigned)ldg + (unsigned)g_col[i]) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_t)(sg + (wave * NG + i) * 1024), 16, (int)voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int m = mt + x_row[i];
      unsigned voff = 0xFFFFFFFFu;
      if (m < m_end && x_col[i] >= 0) {
        if (cv.toeplitz) {
          const int o = cl[i] + x_col[i] + cv.pad;
          voff = (unsigned)(((long long)cb[i] * 8 + (o & 7)) * cv.x_batch_stride + (o & ~7)) * 2u;
        } else if (cv.Lout > 0) {
          const int pos = cl[i] * cv.stride - cv.pad + x_tap[i];
          if (pos >= 0 && pos < cv.Lin)
            voff = (unsigned)((long long)cb[i] * cv.x_batch_stride + (long long)pos * cv.Cin + x_tapoff[i]) * 2u;
        } else {
          voff = ((unsigned)m * (unsigned)ldx + (unsigned)x_col[i]) * 2u;
        }
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(sx + (wave * NX + i) * 1024), 16, (int)voff, 0, 0, 0);
      if (cv.Lout > 0) {                                       // next step: 64 rows further
        cl[i] += 64;
        while (cl[i] >= cv.Lout) { cl[i] -= cv.Lout; ++cb[i]; }
      }
    }
    ++issued;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read lane coordinates: 16-lane group g -> column block (g & 1) * 16, contraction rows 4 * (g >> 1) + q (and + 8);
  // lane 4q + p supplies row q, columns 4p .. 4p + 3.  Row & 3 == q for every row a lane reads: its segment XOR is a constant.
  const int g16 = lane >> 4, i16 = lane & 15;
  const int qq = i16 >> 2, pp = i16 & 3;
  const int trow = 4 * (g16 >> 1) + qq;
  const int tcolb = ((g16 & 1) * 16 + 4 * pp) * 2;             // byte offset inside a 64-byte segment
  int ga_off[2], xb_off[2];                                    // byte offsets of the lane's fragment pieces inside a stage (row trow)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ga_off[i] = trow * GROW + ((((wn * 2 + i) ^ qq)) << 6) + tcolb;
    xb_off[i] = G_STAGE + trow * XROW + ((((wk * 2 + i) ^ qq)) << 6) + tcolb;
  }

  const bool do_bias = (db != nullptr) && (by == 0);
  float bsum = 0.f;
  constexpr int D = STAGES - 1;                                // prefetch distance
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nsteps) issue(s);

  int stage = 0;
  for (int t = 0; t < nsteps; ++t) {
    // step t must have landed; up to D - 1 younger steps may stay in flight
    const int younger = (nsteps - 1 - t) < (D - 1) ? (nsteps - 1 - t) : (D - 1);
    if (younger >= 2) tn2_wait_vmcnt<2 * NLD>();
    else if (younger == 1) tn2_wait_vmcnt<NLD>();
    else tn2_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                              // ... for every wave; and the slot refilled below is no longer read
    if (t + D < nsteps) {
      int st = stage + D;
      if (st >= STAGES) st -= STAGES;
      issue(st);
    }
    const unsigned char* sbase = tn2_smem + stage * STAGE;
    if (do_bias) {                                             // column sums of G from the staged tile (k-tile-0 workgroups)
      constexpr int RPT = 64 * TNc / (NW * 64);                // rows per thread: thread = (column, row block)
      const int c = tid % TNc, r0 = (tid / TNc) * RPT;
#pragma unroll 8
      for (int r = 0; r < RPT; ++r) {
        const int row = r0 + r;
        bsum += T::to_f32(*reinterpret_cast<const u16*>(sbase + row * GROW + ((((c >> 5) ^ (row & 3))) << 6) + ((2 * c) & 63)));
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* base = sbase + ga_off[i] + 16 * s * GROW;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * GROW));
        const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
        fa[i] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned char* base = sbase + xb_off[j] + 16 * s * XROW;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 8 * XROW));
        const u32x2 a0 = __builtin_bit_cast(u32x2, v0), a1 = __builtin_bit_cast(u32x2, v1);
        fb[j] = u32x4{a0[0], a0[1], a1[0], a1[1]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = T::mfma(fa[i], fb[j], acc[i][j]);
    }
    if (++stage == STAGES) stage = 0;
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
          if (ws) ws[((long long)bz * N + n) * K + k] = acc[i][j][r];        // this M-split's partial (gemm16_tn.hip: tn_fold)
          else atomicAdd(&dW[(long long)n * ldw + k], acc[i][j][r]);
        }
      }
    }
  if (do_bias) {
    const int n = n0 + tid % TNc;
    if (n < N) {
      if (ws) ws[(long long)gridDim.z * N * K + ((long long)bz * WKW + tid / TNc) * N + n] = bsum;   // WKW threads per column
      else atomicAdd(&db[n], bsum);
    }
  }
}

int tn_fold(float* ws, float* dW, float* db, int N, int K, int ldw, int splits, int bias_rows, void* stream);

template <class T, int WNW, int WKW, int STAGES>
static int tn2_go(const void* G, const void* X, float* dW, float* db, int M, int N, int K, int ldg, int ldx, int ldw, unsigned g_rec,
                  unsigned x_rec, hipStream_t st, const TnConv& cv, int n_cu, float* ws, long long ws_floats) {
  constexpr int TNc = 64 * WNW, TKc = 64 * WKW;
  constexpr int lds = STAGES * 64 * (TNc + TKc) * 2;
  static bool attr_set_dev[64] = {false};                      // hipFuncSetAttribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SFM_ERR_LAUNCH;
  if (!attr_set_dev[dev]) {
    if (hipFuncSetAttribute((const void*)gemm16_tn2_kernel<T, WNW, WKW, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
      return SFM_ERR_LAUNCH;
    attr_set_dev[dev] = true;
  }
  const int tiles = ((N + TNc - 1) / TNc) * ((K + TKc - 1) / TKc);
  // one workgroup per CU (the ring takes most of the LDS): ONE round of n_cu workgroups
  int splits = n_cu / tiles;
  const int max_splits = (M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows = (M + splits - 1) / splits;
  rows = (rows + 63) / 64 * 64;
  splits = (M + rows - 1) / rows;
  dim3 grid((N + TNc - 1) / TNc, (K + TKc - 1) / TKc, splits), block(WNW * WKW * 64);
  if (ws && (long long)splits * ((long long)N * K + (long long)WKW * N) > ws_floats) return SFM_ERR_ARG;
  SFM_LAUNCH((gemm16_tn2_kernel<T, WNW, WKW, STAGES>), grid, block, lds, st, (const u16*)G, (const u16*)X, dW, db, M, N, K, ldg, ldx,
             ldw, rows, g_rec, x_rec, cv, ws);
  return ws ? tn_fold(ws, dW, db, N, K, ldw, splits, WKW, (void*)st) : SFM_OK;
}

// called by gemm16_tn.hip's launcher; returns SFM_ERR_SHAPE when the shape is not one this kernel takes (the caller then
// uses the first kernel).  variant: 1 = 128 x 128 tiles, 2 = 128 (n) x 256 (k) tiles.
int sfm_tn2_launch(const void* G, const void* X, float* dW, float* db, int M, int N, int K, int ldg, int ldx, int ldw, int dtype,
                   void* stream, const TnConv& cv, long long x_elems, int variant, float* ws, long long ws_floats) {
  if ((N % 8) != 0 || (K % 8) != 0 || (ldg % 8) != 0) return SFM_ERR_SHAPE;
  if (cv.Lout == 0 && (ldx % 8) != 0) return SFM_ERR_SHAPE;
  if (cv.Lout > 0 && !cv.toeplitz && (cv.Cin % 8) != 0) return SFM_ERR_SHAPE;
  const long long g_bytes = ((long long)(M - 1) * ldg + N) * 2;
  const long long x_bytes = x_elems * 2;
  if (g_bytes >= 0xFFFFFFF0LL || x_bytes >= 0xFFFFFFF0LL || x_bytes <= 0) return SFM_ERR_SHAPE;   // 32-bit buffer offsets
  if ((((uintptr_t)G) % 16) != 0 || (((uintptr_t)X) % 16) != 0) return SFM_ERR_SHAPE;
  static int n_cu_dev[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SFM_ERR_LAUNCH;
  if (!n_cu_dev[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return SFM_ERR_LAUNCH;
    n_cu_dev[dev] = v;
  }
  const int n_cu = n_cu_dev[dev];
  hipStream_t st = (hipStream_t)stream;
  const unsigned gr = (unsigned)g_bytes, xr = (unsigned)x_bytes;
  if (variant == 2) {
    if (dtype == SFM_DT_F16) return tn2_go<F16, 2, 4, 3>(G, X, dW, db, M, N, K, ldg, ldx, ldw, gr, xr, st, cv, n_cu, ws, ws_floats);
    return tn2_go<BF16, 2, 4, 3>(G, X, dW, db, M, N, K, ldg, ldx, ldw, gr, xr, st, cv, n_cu, ws, ws_floats);
  }
  if (dtype == SFM_DT_F16) return tn2_go<F16, 2, 2, 4>(G, X, dW, db, M, N, K, ldg, ldx, ldw, gr, xr, st, cv, n_cu, ws, ws_floats);
  return tn2_go<BF16, 2, 2, 4>(G, X, dW, db, M, N, K, ldg, ldx, ldw, gr, xr, st, cv, n_cu, ws, ws_floats);
}
