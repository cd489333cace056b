// gemm16: C = epilogue(A[M,K] * W[N,K]^T + bias), 16-bit (bf16|f16) MFMA operands,
// fp32 accumulate, for every dense layer on the path:
//   Linear (models/conformer.py:36-37,88,98,182,191; agents/msa.py:43-70),
//   and channels-last Conv1d as implicit GEMM (agents/perception.py:167-206):
//   with activations stored [B, L, Cin] the im2col row of output position l
//   is the CONTIGUOUS span x[b, l*stride-pad : +ksize, :], so A needs no
//   materialised im2col - only a row stride and an edge predicate.
// Tile 128 x BN x 32, 256 threads = 4 waves (2x2), each wave 64 x BN/2 via
// v_mfma_f32_32x32x16.  Register-staged global->LDS (loads of tile t+1 are in
// flight while tile t is multiplied), LDS rows padded to 80 B so that the
// ds_read_b128 fragment reads are bank-conflict free (16 lanes -> 16 slots).
#include "sfm_common.h"

#define EPI_NONE 0
#define EPI_SWISH 1
#define EPI_GELU 2
#define EPI_RESID 3     // out = resid + alpha * v          (fp32 out)
#define EPI_GLU 4       // packed (a|gate) 32-column pairs  -> N/2 outputs
#define EPI_SIGMOID 5
#define EPI_TANH_SCALE 6  // alpha * tanh(v)
#define EPI_SIGMA 7     // exp(0.5 * clamp(v, -10, 10))   (agents/perception.py:249)
#define EPI_CPEA 8      // cols < nsplit: sigmoid ; else alpha * tanh   (agents/cpea.py:102-105)

struct GemmParams {
  const u16* A;
  const u16* W;
  const float* bias;
  void* out;
  const float* resid;
  float* gn_partial;
  long long a_batch_stride, o_batch_stride, r_batch_stride;
  int B, Lout, Lin, Cin, lda, stride, pad, cin_shift;
  int K, Kpad, N, ldo, ldr;
  float alpha;
  int epi, out_f32, gn_group, nsplit;
};

#define BM 128
#define BK 32
#define LDS_ROW 40   // 32 + 8 pad (u16 elements) = 80 bytes

template <class T, int BN>
__global__ __launch_bounds__(256) void gemm16_kernel(GemmParams p) {
  constexpr int NJ = BN / 64;        // 32-col MFMA tiles per wave along N
  constexpr int BCH = BN / 64;       // W chunks per thread (BN*4 chunks / 256)
  __shared__ __attribute__((aligned(16))) u16 As[BM * LDS_ROW];
  __shared__ __attribute__((aligned(16))) u16 Bs[BN * LDS_ROW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BN;
  const int l0 = blockIdx.y * BM;
  const int b = blockIdx.z;

  const u16* Ab = p.A + (long long)b * p.a_batch_stride;

  // per-thread chunk coordinates
  int a_row[2], a_kc[2];
  long long a_base[2];
  int a_pos0[2];
  bool a_rowok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int c = tid + 256 * i;
    a_row[i] = c >> 2;
    a_kc[i] = (c & 3) * 8;
    int l = l0 + a_row[i];
    a_rowok[i] = l < p.Lout;
    a_pos0[i] = l * p.stride - p.pad;
    a_base[i] = (long long)a_pos0[i] * p.lda;
  }
  int b_row[BCH], b_kc[BCH];
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    int c = tid + 256 * i;
    b_row[i] = c >> 2;
    b_kc[i] = (c & 3) * 8;
  }

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[2], rb[BCH];
  const int nkt = p.Kpad / BK;

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int j = kt * BK + a_kc[i];
      int pos = a_pos0[i] + (j >> p.cin_shift);
      bool ok = a_rowok[i] && (j < p.K) && (pos >= 0) && (pos < p.Lin);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) {
        long long off = (p.lda == p.Cin || p.cin_shift >= 30)
                            ? (long long)j
                            : (long long)(j >> p.cin_shift) * p.lda + (j & (p.Cin - 1));
        v = *reinterpret_cast<const u32x4*>(Ab + a_base[i] + off);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      rb[i] = *reinterpret_cast<const u32x4*>(p.W + (long long)(n0 + b_row[i]) * p.Kpad + kt * BK + b_kc[i]);
    }
  };

  load_tile(0);
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<u32x4*>(&As[a_row[i] * LDS_ROW + a_kc[i]]) = ra[i];
#pragma unroll
    for (int i = 0; i < BCH; ++i)
      *reinterpret_cast<u32x4*>(&Bs[b_row[i] * LDS_ROW + b_kc[i]]) = rb[i];
    __syncthreads();
    if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fa[2], fb[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = *reinterpret_cast<const u32x4*>(&As[(wm * 64 + i * 32 + (lane & 31)) * LDS_ROW + s * 16 + (lane >> 5) * 8]);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        fb[j] = *reinterpret_cast<const u32x4*>(&Bs[(wn * (BN / 2) + j * 32 + (lane & 31)) * LDS_ROW + s * 16 + (lane >> 5) * 8]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ------------------------------ epilogue ------------------------------
  const int colb = n0 + wn * (BN / 2);
  if (p.epi == EPI_GLU) {
    // packed W rows: per 64-row block, rows 0..31 = 'a' channels, 32..63 = their gates
    // (host packing, pack_glu_weight in ops.py).  Requires NJ == 2.
    if constexpr (NJ == 2) {
      const int cout = (colb >> 1) + (lane & 31);
      const float ba = p.bias ? p.bias[colb + (lane & 31)] : 0.f;
      const float bg = p.bias ? p.bias[colb + 32 + (lane & 31)] : 0.f;
      u16* o16 = reinterpret_cast<u16*>(p.out) + (long long)b * p.o_batch_stride;
      float* o32 = reinterpret_cast<float*>(p.out) + (long long)b * p.o_batch_stride;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = l0 + wm * 64 + i * 32 + mfma_row(r, lane);
          if (m < p.Lout && cout < p.N) {
            float v = (acc[i][0][r] + ba) * sigmoid_f(acc[i][1][r] + bg);
            if (p.out_f32) o32[(long long)m * p.ldo + cout] = v;
            else o16[(long long)m * p.ldo + cout] = T::from_f32(v);
          }
        }
    }
    return;
  }

#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = colb + j * 32 + (lane & 31);
    const bool nok = n < p.N;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
    float gsum = 0.f, gsq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = l0 + wm * 64 + i * 32 + mfma_row(r, lane);
        bool ok = nok && (m < p.Lout);
        float v = acc[i][j][r] + bv;
        if (ok) { gsum += v; gsq += v * v; }
        switch (p.epi) {
          case EPI_SWISH: v = swish_f(v); break;
          case EPI_GELU: v = gelu_erf(v); break;
          case EPI_RESID:
            if (ok) v = p.resid[(long long)b * p.r_batch_stride + (long long)m * p.ldr + n] + p.alpha * v;
            break;
          case EPI_SIGMOID: v = sigmoid_f(v); break;
          case EPI_TANH_SCALE: v = p.alpha * tanhf(v); break;
          case EPI_SIGMA: v = expf(0.5f * fminf(fmaxf(v, -10.f), 10.f)); break;
          case EPI_CPEA: v = (n < p.nsplit) ? sigmoid_f(v) : p.alpha * tanhf(v); break;
          default: break;
        }
        if (ok) {
          long long off = (long long)b * p.o_batch_stride + (long long)m * p.ldo + n;
          if (p.out_f32) reinterpret_cast<float*>(p.out)[off] = v;
          else reinterpret_cast<u16*>(p.out)[off] = T::from_f32(v);
        }
      }
    }
    if (p.gn_partial) {
      // reduce over the lanes that share a GroupNorm group (gn_group consecutive
      // channels, gn_group | 32) and over both row halves of the wave
      for (int o = 1; o < p.gn_group; o <<= 1) {
        gsum += __shfl_xor(gsum, o, 64);
        gsq += __shfl_xor(gsq, o, 64);
      }
      gsum += __shfl_xor(gsum, 32, 64);
      gsq += __shfl_xor(gsq, 32, 64);
      if (lane < 32 && (lane & (p.gn_group - 1)) == 0 && nok) {
        int ngroups = p.N / p.gn_group;
        int g = n / p.gn_group;
        long long slot = ((long long)b * (gridDim.y * 2) + blockIdx.y * 2 + wm) * ngroups + g;
        p.gn_partial[slot * 2 + 0] = gsum;
        p.gn_partial[slot * 2 + 1] = gsq;
      }
    }
  }
}

template <class T>
static int launch_gemm16(const GemmParams& p, int bn, hipStream_t stream) {
  dim3 block(256);
  if (bn == 128) {
    dim3 grid((p.epi == EPI_GLU ? 2 * p.N + 127 : p.N + 127) / 128, (p.Lout + BM - 1) / BM, p.B);
    SFM_LAUNCH((gemm16_kernel<T, 128>), grid, block, 0, stream, p);
  } else {
    dim3 grid((p.N + 63) / 64, (p.Lout + BM - 1) / BM, p.B);
    SFM_LAUNCH((gemm16_kernel<T, 64>), grid, block, 0, stream, p);
  }
  SFM_CHECK_LAUNCH();
  return SFM_OK;
}

// See include/sincformer_hip.h for the contract.
extern "C" int sfm_gemm16_v1(const void* A, const void* W, const float* bias, void* out, const float* resid,
                          float* gn_partial, int B, int Lout, int Lin, int Cin, int lda, int ksize, int stride, int pad,
                          long long a_batch_stride, int Kpad, int N, int Npad, int ldo, long long o_batch_stride,
                          int ldr, long long r_batch_stride, float alpha, int epi, int out_f32, int gn_group,
                          int nsplit, int dtype, void* stream) {
  if (!A || !W || !out) return SFM_ERR_ARG;
  if (B <= 0 || Lout <= 0 || N <= 0) return SFM_ERR_SHAPE;
  if (Cin % 8 != 0 || lda % 8 != 0 || lda < Cin || Kpad % BK != 0) return SFM_ERR_SHAPE;
  int K = ksize * Cin;
  if (K > Kpad) return SFM_ERR_SHAPE;
  int shift = 30;
  if (ksize > 1) {
    if (Cin & (Cin - 1)) return SFM_ERR_SHAPE;   // conv mode needs power-of-two channels
    shift = 0;
    while ((1 << shift) < Cin) ++shift;
  }
  if (epi == EPI_RESID && !resid) return SFM_ERR_ARG;
  if (gn_partial && (gn_group <= 0 || gn_group > 32 || (32 % gn_group) != 0 || (N % gn_group) != 0)) return SFM_ERR_SHAPE;
  int bn = 128;
  if (epi != EPI_GLU && Npad % 128 != 0) bn = 64;
  if (Npad % 64 != 0) return SFM_ERR_SHAPE;
  if (epi == EPI_GLU && (Npad % 128 != 0 || Npad != 2 * N)) return SFM_ERR_SHAPE;
  GemmParams p;
  p.A = (const u16*)A; p.W = (const u16*)W; p.bias = bias; p.out = out; p.resid = resid; p.gn_partial = gn_partial;
  p.a_batch_stride = a_batch_stride; p.o_batch_stride = o_batch_stride; p.r_batch_stride = r_batch_stride;
  p.B = B; p.Lout = Lout; p.Lin = Lin; p.Cin = Cin; p.lda = lda; p.stride = stride; p.pad = pad; p.cin_shift = shift;
  p.K = K; p.Kpad = Kpad; p.N = N; p.ldo = ldo; p.ldr = ldr; p.alpha = alpha; p.epi = epi; p.out_f32 = out_f32;
  p.gn_group = gn_group; p.nsplit = nsplit;
  if (dtype == SFM_DT_BF16) return launch_gemm16<BF16>(p, bn, (hipStream_t)stream);
  if (dtype == SFM_DT_F16) return launch_gemm16<F16>(p, bn, (hipStream_t)stream);
  return SFM_ERR_ARG;
}
