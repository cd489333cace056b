// Gradient of the SincConv1d FIR bank with respect to its filter taps (training of agents/perception.py:79-118):
//   dfilt[c][k] = sum_{b, l} dy[b, l, c] * x[b, l + k - K/2]          (x = waveform, zero outside [0, L))
// i.e. the cross-correlation of every channel's output gradient with the input at the K lags of the kernel.  The chain
// rule through the analytic band-pass formula down to low_hz_ / band_hz_ (128 scalars) is applied to this [C, K] matrix
// on the host side (train.py, a few hundred FLOPs per tap).
// Thread (channel c, tap group kg): 32 consecutive taps in registers next to a 32-sample sliding window of x, indexed
// with compile-time (j + i) % 32 so the slide costs no moves; one broadcast load of x and one coalesced load of dy per
// output sample, issued one sample ahead.  Per-(utterance, span) partials go to a scratch [nparts][Kp][C] and are
// reduced by a second kernel (all workgroups adding into the same C*K addresses would serialise the atomics).
#include "sfm_common.h"

#define SW_TAPS 32

template <class T>
__global__ __launch_bounds__(512) void sinc_wgrad_kernel(const float* __restrict__ x, const void* __restrict__ dy, int dy_f32,
                                                         float* __restrict__ part, int L, int C, int K, int span) {
  const int c = threadIdx.x % 64, kg = threadIdx.x / 64;          // C == 64 channels per workgroup slice
  const int cc = blockIdx.x * 64 + c;
  const int b = blockIdx.z;
  const int l0 = blockIdx.y * span, l1 = min(L, l0 + span);
  const int pad = K / 2;
  const float* xb = x + (long long)b * L;
  auto ldx = [&](int s) -> float { return (s >= 0 && s < L) ? xb[s] : 0.f; };
  auto ldg = [&](int l) -> float {
    if (l >= l1 || cc >= C) return 0.f;
    const long long e = ((long long)b * L + l) * C + cc;
    return dy_f32 ? reinterpret_cast<const float*>(dy)[e] : T::to_f32(reinterpret_cast<const u16*>(dy)[e]);
  };
  float acc[SW_TAPS], w[SW_TAPS];
  const int base = l0 - pad + kg * SW_TAPS;                       // x index of tap (kg*32 + 0) at output sample l0
#pragma unroll
  for (int i = 0; i < SW_TAPS; ++i) {
    acc[i] = 0.f;
    w[i] = ldx(base + i);
  }
  float gn = ldg(l0), xn = ldx(base + SW_TAPS);
  for (int lb = l0; lb < l1; lb += SW_TAPS) {
#pragma unroll
    for (int j = 0; j < SW_TAPS; ++j) {
      const int l = lb + j;
      const float g = gn, xa = xn;
      gn = ldg(l + 1);
      xn = ldx(base + (l - l0) + SW_TAPS + 1);
#pragma unroll
      for (int i = 0; i < SW_TAPS; ++i) acc[i] += g * w[(j + i) % SW_TAPS];
      w[j] = xa;                                                  // slot j held the sample that just left the window
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (cc < C) {
    const int Kp = gridDim.x ? ((K + SW_TAPS - 1) / SW_TAPS) * SW_TAPS : 0;
    float* pp = part + (((long long)blockIdx.z * gridDim.y + blockIdx.y) * Kp) * C + cc;
#pragma unroll
    for (int i = 0; i < SW_TAPS; ++i) pp[(long long)(kg * SW_TAPS + i) * C] = acc[i];
  }
}

// dfilt[c][k] += sum_parts part[p][k][c]
__global__ __launch_bounds__(256) void sinc_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dfilt,
                                                                int nparts, int C, int K, int Kp, int parts_per_block) {
  const int e = blockIdx.x * 256 + threadIdx.x;                   // e = k * C + c
  if (e >= K * C) return;
  const int p0 = blockIdx.y * parts_per_block, p1 = min(nparts, p0 + parts_per_block);
  float s = 0.f;
  for (int p = p0; p < p1; ++p) s += part[(long long)p * Kp * C + e];
  const int k = e / C, c = e - k * C;
  atomicAdd(&dfilt[c * K + k], s);
}

static void sinc_wgrad_plan(int B, int L, int K, int* span, int* ny, int* Kp) {
  int nspan = (1024 + B - 1) / B;
  if (nspan > (L + 511) / 512) nspan = (L + 511) / 512;
  if (nspan < 1) nspan = 1;
  *span = ((L + nspan - 1) / nspan + SW_TAPS - 1) / SW_TAPS * SW_TAPS;
  *ny = (L + *span - 1) / *span;
  *Kp = (K + SW_TAPS - 1) / SW_TAPS * SW_TAPS;
}

extern "C" long long sfm_sinc_wgrad_scratch_floats(int B, int L, int C, int K) {
  int span, ny, Kp;
  sinc_wgrad_plan(B, L, K, &span, &ny, &Kp);
  return (long long)B * ny * Kp * C;
}

// x [B, L] fp32, dy [B, L, C] 16-bit or fp32 (gradient w.r.t. the raw filterbank output), dfilt [C, K] fp32 (accumulated)
extern "C" int sfm_sinc_wgrad(const float* x, const void* dy, int dy_f32, float* dfilt, float* scratch, int B, int L, int C,
                              int K, int dtype, void* stream) {
  if (!x || !dy || !dfilt || !scratch) return SFM_ERR_ARG;
  if (B <= 0 || L <= 0 || C <= 0 || K <= 0 || K > 8 * SW_TAPS) return SFM_ERR_SHAPE;
  int span, ny, Kp;
  sinc_wgrad_plan(B, L, K, &span, &ny, &Kp);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((C + 63) / 64, ny, B), block(64 * (Kp / SW_TAPS));
  if (dtype == SFM_DT_F16) SFM_LAUNCH((sinc_wgrad_kernel<F16>), grid, block, 0, st, x, dy, dy_f32, scratch, L, C, K, span);
  else SFM_LAUNCH((sinc_wgrad_kernel<BF16>), grid, block, 0, st, x, dy, dy_f32, scratch, L, C, K, span);
  const int nparts = B * ny, ppb = 64;
  dim3 g2((K * C + 255) / 256, (nparts + ppb - 1) / ppb);
  SFM_LAUNCH(sinc_wgrad_reduce_kernel, g2, dim3(256), 0, st, scratch, dfilt, nparts, C, K, Kp, ppb);
  return SFM_OK;
}
