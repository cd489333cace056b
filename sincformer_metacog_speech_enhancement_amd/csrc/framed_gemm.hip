// framed_gemm_f32: out[b, m, n] = bias[n] + sum_k sig[b, m*hop + k - padl] * Wt[k][n]
// with exact-fp32 arithmetic on the matrix cores (v_mfma_f32_32x32x2_f32, which is
// bit-for-bit an fmaf chain; there is no TF32 on gfx950, so this is full fp32).
// One kernel serves every fp32 "frames x matrix" product on the path:
//   * SincConv1d FIR filterbank  (hop 1, K 251, zero edges)   agents/perception.py:115-118
//   * torch.stft as a windowed DFT (hop 80, K 160, reflect)   training/conformer_pipeline.py:196-202
//   * the irfft x window stage of torch.istft (hop = row stride) conformer_pipeline.py:205-211
//   * generic row-major fp32 GEMM (hop = lda, padl 0)
// Tile 128 rows x 64 cols, k-chunks of 32; 4 waves each own 32 rows x 64 cols.
// A-tile is gathered from the signal with edge handling while staged into LDS
// (row stride 33 floats: conflict-free ds_read_b32 fragment reads).
#include "sfm_common.h"

struct FramedParams {
  const float* sig;
  const float* Wt;
  const float* bias;
  void* out;
  void* out2;
  float* gn_partial;
  long long sig_batch_stride, o_batch_stride, ldm, ldn;
  int B, M, Ls, hop, padl, K, Kpad, N, Npad, nsplit;
  int mode, out_f32, gn_group;
};

#define FBM 128
#define FBN 64
#define FKC 32
#define FAS 33

template <class T>
__global__ __launch_bounds__(256) void framed_gemm_kernel(FramedParams p) {
  __shared__ float As[FBM * FAS];
  __shared__ float Bs[FKC * FBN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * FBN;
  const int m0 = blockIdx.y * FBM;
  const int b = blockIdx.z;
  const float* sg = p.sig + (long long)b * p.sig_batch_stride;

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float ra[16], rb[8];
  const int nkc = p.Kpad / FKC;

  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int idx = tid + 256 * i;
      int row = idx >> 5, kk = idx & 31;
      int m = m0 + row;
      int k = kc * FKC + kk;
      float v = 0.f;
      if (m < p.M && k < p.K) {
        long long s = (long long)m * p.hop + k - p.padl;
        if (p.mode == 1) {
          if (s < 0) s = -s;
          if (s >= p.Ls) s = 2LL * (p.Ls - 1) - s;
          if (s >= 0 && s < p.Ls) v = sg[s];
        } else {
          if (s >= 0 && s < p.Ls) v = sg[s];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int idx = tid + 256 * i;
      int kk = idx >> 6, n = idx & 63;
      rb[i] = p.Wt[(long long)(kc * FKC + kk) * p.Npad + n0 + n];
    }
  };

  load_chunk(0);
  for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int idx = tid + 256 * i;
      As[(idx >> 5) * FAS + (idx & 31)] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) Bs[tid + 256 * i] = rb[i];
    __syncthreads();
    if (kc + 1 < nkc) load_chunk(kc + 1);
#pragma unroll
    for (int ks = 0; ks < FKC / 2; ++ks) {
      float a = As[(wave * 32 + (lane & 31)) * FAS + 2 * ks + (lane >> 5)];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float bv = Bs[(2 * ks + (lane >> 5)) * FBN + j * 32 + (lane & 31)];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + j * 32 + (lane & 31);
    const bool nok = n < p.N;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
    float gsum = 0.f, gsq = 0.f;
    void* dst = p.out;
    int nn = n;
    if (p.out2 && n >= p.nsplit) { dst = p.out2; nn = n - p.nsplit; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = m0 + wave * 32 + mfma_row(r, lane);
      if (nok && m < p.M) {
        float v = acc[j][r] + bv;
        gsum += v;
        gsq = __builtin_fmaf(v, v, gsq);               // (fmaf, not mul + add: see profiles/README.md, "the lost sums of squares")
        long long off = (long long)b * p.o_batch_stride + (long long)m * p.ldm + (long long)nn * p.ldn;
        if (p.out_f32) reinterpret_cast<float*>(dst)[off] = v;
        else reinterpret_cast<u16*>(dst)[off] = T::from_f32(v);
      }
    }
    if (p.gn_partial) {
      for (int o = 1; o < p.gn_group; o <<= 1) {
        gsum += __shfl_xor(gsum, o, 64);
        gsq += __shfl_xor(gsq, o, 64);
      }
      gsum += __shfl_xor(gsum, 32, 64);
      gsq += __shfl_xor(gsq, 32, 64);
      if (lane < 32 && (lane & (p.gn_group - 1)) == 0 && nok) {
        int ngroups = p.N / p.gn_group;
        long long slot = ((long long)b * (gridDim.y * 4) + blockIdx.y * 4 + wave) * ngroups + n / p.gn_group;
        p.gn_partial[slot * 2 + 0] = gsum;
        p.gn_partial[slot * 2 + 1] = gsq;
      }
    }
  }
}

extern "C" int sfm_framed_gemm_f32(const float* sig, const float* Wt, const float* bias, void* out, void* out2,
                                   float* gn_partial, int B, int M, int Ls, long long sig_batch_stride, int hop,
                                   int padl, int K, int Kpad, int N, int Npad, int nsplit, long long o_batch_stride,
                                   long long ldm, long long ldn, int mode, int out_f32, int gn_group, int dtype,
                                   void* stream) {
  if (!sig || !Wt || !out) return SFM_ERR_ARG;
  if (B <= 0 || M <= 0 || N <= 0 || K <= 0) return SFM_ERR_SHAPE;
  if (Kpad % FKC != 0 || Npad % FBN != 0 || K > Kpad || N > Npad) return SFM_ERR_SHAPE;
  if (mode == 1 && (padl >= Ls || Ls < 2)) return SFM_ERR_SHAPE;
  if (gn_partial && (gn_group <= 0 || gn_group > 32 || (32 % gn_group) != 0 || (N % gn_group) != 0)) return SFM_ERR_SHAPE;
  FramedParams p;
  p.sig = sig; p.Wt = Wt; p.bias = bias; p.out = out; p.out2 = out2; p.gn_partial = gn_partial;
  p.sig_batch_stride = sig_batch_stride; p.o_batch_stride = o_batch_stride; p.ldm = ldm; p.ldn = ldn;
  p.B = B; p.M = M; p.Ls = Ls; p.hop = hop; p.padl = padl; p.K = K; p.Kpad = Kpad; p.N = N; p.Npad = Npad;
  p.nsplit = nsplit; p.mode = mode; p.out_f32 = out_f32; p.gn_group = gn_group;
  dim3 grid((N + FBN - 1) / FBN, (M + FBM - 1) / FBM, B), block(256);
  if (dtype == SFM_DT_F16) SFM_LAUNCH((framed_gemm_kernel<F16>), grid, block, 0, (hipStream_t)stream, p);
  else SFM_LAUNCH((framed_gemm_kernel<BF16>), grid, block, 0, (hipStream_t)stream, p);
  SFM_CHECK_LAUNCH();
  return SFM_OK;
}

// ---------------------------------------------------------------------------
// framed_gemm_split16: the same "frames x matrix" product on the 16-bit matrix cores with split operands
// (x = x_hi + x_lo in bf16, 3 MFMAs per k-step: hi*hi + hi*lo + lo*hi, fp32 accumulate): |error| ~ 4e-6 relative,
// 5x the rate of v_mfma_f32_32x32x2_f32.  Used for the STFTs of the training objective (3 resolutions, prediction and
// target, forward and adjoint: training/conformer_pipeline.py:74-108), where fp32-exactness buys nothing; the model's
// own STFT / iSTFT stay on the exact fp32 kernel above.
// Tile 128 rows x 256 columns (a 1024-point spectrum is 4 column tiles, so the gathered signal tile is re-read 4x,
// not 17x), k-chunks of 32; 4 waves x (32 rows x 256 columns).  The signal tile is gathered (zero / reflect edges),
// split and packed in registers, the constant matrix arrives pre-split and n-major ([Npad][Kpad] u16) by 16-byte loads.
// LDS rows of 80 bytes: conflict-free ds_read_b128 fragments.
// ---------------------------------------------------------------------------
#define SBM 128
#define SBN 256
#define SKC 32
#define SSTR 40       // u16 per LDS row (32 + 8 pad)

struct Split16Params {
  const float* sig;
  const u16* Whi;
  const u16* Wlo;
  float* out;
  float* out2;
  long long sig_batch_stride, o_batch_stride, ldm;
  int B, M, Ls, hop, padl, K, Kpad, N, Npad, nsplit, col2_off, mode;
};

__global__ __launch_bounds__(256, 2) void framed_gemm_split16_kernel(Split16Params p) {
  __shared__ __attribute__((aligned(16))) u16 Ah[SBM * SSTR], Al[SBM * SSTR];
  __shared__ __attribute__((aligned(16))) u16 Bh[SBN * SSTR], Bl[SBN * SSTR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hl = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * SBN, m0 = blockIdx.y * SBM, b = blockIdx.z;
  const float* sg = p.sig + (long long)b * p.sig_batch_stride;

  f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float ra[16];
  u32x4 rbh[4], rbl[4];
  const int nkc = p.Kpad / SKC;

  auto fetch = [&](long long s) -> float {
    if (p.mode == 1) {
      if (s < 0) s = -s;
      if (s >= p.Ls) s = 2LL * (p.Ls - 1) - s;
    }
    return (s >= 0 && s < p.Ls) ? sg[s] : 0.f;
  };
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;                     // (row, k-pair)
      const int row = idx >> 4, kp = idx & 15;
      const int m = m0 + row, k = kc * SKC + 2 * kp;
      float v0 = 0.f, v1 = 0.f;
      if (m < p.M) {
        const long long s = (long long)m * p.hop + k - p.padl;
        if (k < p.K) v0 = fetch(s);
        if (k + 1 < p.K) v1 = fetch(s + 1);
      }
      ra[2 * i] = v0;
      ra[2 * i + 1] = v1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;                     // (n, 8-element k group)
      const int n = idx >> 2, kq = idx & 3;
      const long long off = (long long)(n0 + n) * p.Kpad + kc * SKC + kq * 8;
      rbh[i] = *reinterpret_cast<const u32x4*>(p.Whi + off);
      rbl[i] = *reinterpret_cast<const u32x4*>(p.Wlo + off);
    }
  };

  load_chunk(0);
  for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 4, kp = idx & 15;
      const float v0 = ra[2 * i], v1 = ra[2 * i + 1];
      const float h0 = BF16::to_f32(BF16::from_f32(v0)), h1 = BF16::to_f32(BF16::from_f32(v1));
      *reinterpret_cast<uint32_t*>(&Ah[row * SSTR + 2 * kp]) = BF16::pack(v0, v1);
      *reinterpret_cast<uint32_t*>(&Al[row * SSTR + 2 * kp]) = BF16::pack(v0 - h0, v1 - h1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int n = idx >> 2, kq = idx & 3;
      *reinterpret_cast<u32x4*>(&Bh[n * SSTR + kq * 8]) = rbh[i];
      *reinterpret_cast<u32x4*>(&Bl[n * SSTR + kq * 8]) = rbl[i];
    }
    __syncthreads();
    if (kc + 1 < nkc) load_chunk(kc + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ao = (wave * 32 + l31) * SSTR + ks * 16 + hl * 8;
      const u32x4 ah = *reinterpret_cast<const u32x4*>(&Ah[ao]);
      const u32x4 al = *reinterpret_cast<const u32x4*>(&Al[ao]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int bo = (j * 32 + l31) * SSTR + ks * 16 + hl * 8;
        const u32x4 bh = *reinterpret_cast<const u32x4*>(&Bh[bo]);
        const u32x4 bl = *reinterpret_cast<const u32x4*>(&Bl[bo]);
        acc[j] = BF16::mfma(ah, bh, acc[j]);
        acc[j] = BF16::mfma(ah, bl, acc[j]);
        acc[j] = BF16::mfma(al, bh, acc[j]);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = n0 + j * 32 + l31;
    if (n >= p.N) continue;
    float* dst = p.out;
    int nn = n;
    if (p.out2 && n >= p.nsplit) {
      dst = p.out2;
      nn = n - p.nsplit + p.col2_off;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + mfma_row(r, lane);
      if (m < p.M) dst[(long long)b * p.o_batch_stride + (long long)m * p.ldm + nn] = acc[j][r];
    }
  }
}

extern "C" int sfm_framed_gemm_split16(const float* sig, const void* Whi, const void* Wlo, float* out, float* out2, int B,
                                       int M, int Ls, long long sig_batch_stride, int hop, int padl, int K, int Kpad, int N,
                                       int Npad, int nsplit, int col2_off, long long o_batch_stride, long long ldm, int mode,
                                       void* stream) {
  if (!sig || !Whi || !Wlo || !out) return SFM_ERR_ARG;
  if (B <= 0 || M <= 0 || N <= 0 || K <= 0) return SFM_ERR_SHAPE;
  if (Kpad % SKC != 0 || Npad % SBN != 0 || K > Kpad || N > Npad) return SFM_ERR_SHAPE;
  if (mode == 1 && (padl >= Ls || Ls < 2)) return SFM_ERR_SHAPE;
  Split16Params p;
  p.sig = sig; p.Whi = (const u16*)Whi; p.Wlo = (const u16*)Wlo; p.out = out; p.out2 = out2;
  p.sig_batch_stride = sig_batch_stride; p.o_batch_stride = o_batch_stride; p.ldm = ldm;
  p.B = B; p.M = M; p.Ls = Ls; p.hop = hop; p.padl = padl; p.K = K; p.Kpad = Kpad; p.N = N; p.Npad = Npad;
  p.nsplit = nsplit; p.col2_off = col2_off; p.mode = mode;
  dim3 grid((N + SBN - 1) / SBN, (M + SBM - 1) / SBM, B), block(256);
  SFM_LAUNCH(framed_gemm_split16_kernel, grid, block, 0, (hipStream_t)stream, p);
  return SFM_OK;
}
