// Ordered second pass of every split reduction of the training step (weight-gradient split-M partials, LayerNorm / GroupNorm
// dgamma / dbeta, BatchNorm column statistics, bias column sums, the objective's moments, ||g||^2): the producing kernels write one
// partial per workgroup (or per M-split) into a caller-provided workspace instead of issuing fp32 atomics, and this pass adds them
// in an order that depends on nothing but the number of partials - two runs of a step from the same state are then the same
// arithmetic, bit for bit, as the reference's CPU step is (training/conformer_pipeline.py:496-532 under torch's deterministic
// CPU reductions).
#include "sfm_common.h"

// ws: S partials of n = rows * cols elements, partial s at ws + s * stride.  One pass folds chunks of up to 64 consecutive
// partials: thread = (V consecutive elements, one of 4 contiguous quarters of the chunk), 8 loads in flight per thread, the four
// quarter sums combined through LDS in ascending order.  final == 0: chunk j's sum replaces its first partial (in place: a thread
// only ever touches its own elements of its own chunk); final == 1: S <= 64, out[r * ldo + c] (+)= the sum.
template <class A, int V>
__global__ __launch_bounds__(256) void fold_partials_kernel(A* __restrict__ ws, A* __restrict__ out, A* __restrict__ out2, int cols1,
                                                            long long n, int cols, long long ldo, int S, long long stride,
                                                            int accumulate, int final) {
  typedef A vec_t __attribute__((ext_vector_type(V)));
  __shared__ vec_t red[4][64];
  const int q = threadIdx.x >> 6, t = threadIdx.x & 63;
  const long long e = ((long long)blockIdx.x * 64 + t) * V;
  const int c_lo = blockIdx.y * 64, c_hi = min(S, c_lo + 64), len = c_hi - c_lo;
  const int per = (len + 3) / 4;
  const int s_lo = c_lo + q * per, s_hi = min(c_hi, s_lo + per);
  vec_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (vec_t)(A)0;
  if (e < n) {
    int s = s_lo;
    for (; s + 8 <= s_hi; s += 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += *reinterpret_cast<const vec_t*>(ws + (long long)(s + i) * stride + e);
    }
    for (int i = 0; s < s_hi; ++s, ++i) acc[i] += *reinterpret_cast<const vec_t*>(ws + (long long)s * stride + e);
  }
  red[q][t] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (q == 0 && e < n) {
    const vec_t v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if (!final) {
      *reinterpret_cast<vec_t*>(ws + (long long)c_lo * stride + e) = v;
    } else {
      const long long r = e / cols;
      const int c = (int)(e - r * cols);
      // out2 (rows == 1): columns >= cols1 belong to a second array (dgamma | dbeta partials side by side)
      vec_t* o = (out2 && c >= cols1) ? reinterpret_cast<vec_t*>(out2 + (c - cols1)) : reinterpret_cast<vec_t*>(out + r * ldo + c);
      *o = accumulate ? *o + v : v;
    }
  }
}

template <class A>
static int fold_go(A* ws, A* out, A* out2, int cols1, long long rows, int cols, long long ldo, int S, int accumulate, void* stream) {
  if (!ws || !out) return SFM_ERR_ARG;
  if (rows <= 0 || cols <= 0 || S <= 0 || (out2 && (rows != 1 || cols1 <= 0 || cols1 >= cols))) return SFM_ERR_SHAPE;
  const long long n = rows * cols;
  constexpr int VB = 16 / (int)sizeof(A);
  const bool vec = (cols % VB) == 0 && (ldo % VB) == 0 && (((uintptr_t)ws | (uintptr_t)out | (uintptr_t)out2) % 16) == 0 &&
                   (!out2 || (cols1 % VB) == 0);
  const long long quads = vec ? n / VB : n;
  if ((quads + 63) / 64 > 2147483647LL) return SFM_ERR_SHAPE;
  const unsigned gx = (unsigned)((quads + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  long long stride = n;
  while (true) {
    const int final = S <= 64;
    const unsigned gy = final ? 1u : (unsigned)((S + 63) / 64);
    if (vec) SFM_LAUNCH((fold_partials_kernel<A, VB>), dim3(gx, gy), dim3(256), 0, st, ws, out, out2, cols1, n, cols, ldo, S, stride, accumulate, final);
    else SFM_LAUNCH((fold_partials_kernel<A, 1>), dim3(gx, gy), dim3(256), 0, st, ws, out, out2, cols1, n, cols, ldo, S, stride, accumulate, final);
    if (final) break;
    S = (int)gy;
    stride *= 64;
  }
  return SFM_OK;
}

// out[r * ldo + c] (+)= sum_s ws[s][r][c]; ws [S][rows][cols] compact, DESTROYED (used as scratch of the multi-level fold)
int sfm_fold_partials(float* ws, float* out, long long rows, int cols, long long ldo, int S, int accumulate, void* stream) {
  return fold_go<float>(ws, out, nullptr, 0, rows, cols, ldo, S, accumulate, stream);
}

// one row of `cols` sums whose columns >= cols1 go to a second array: out[c] (c < cols1), out2[c - cols1]
int sfm_fold_partials2(float* ws, float* out, float* out2, int cols1, int cols, int S, int accumulate, void* stream) {
  return fold_go<float>(ws, out, out2, cols1, 1, cols, cols, S, accumulate, stream);
}

int sfm_fold_partials_f64(double* ws, double* out, long long rows, int cols, long long ldo, int S, int accumulate, void* stream) {
  return fold_go<double>(ws, out, nullptr, 0, rows, cols, ldo, S, accumulate, stream);
}
