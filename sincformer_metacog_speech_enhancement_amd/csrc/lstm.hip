// Persistent BiLSTM layer (nn.LSTM bidirectional, gate order i,f,g,o;
// agents/cpea.py:43-50,99).  The input projection x W_ih^T + b_ih + b_hh for
// BOTH directions is done beforehand by one GEMM (xg [B, T, 2, 4H] fp32); this
// kernel runs only the sequential recurrence.  Chains are independent per
// (utterance, direction): one 8H-thread workgroup per chain keeps the whole
// fp32 W_hh (4H x H) in registers and h in LDS; nothing but xg / h touches HBM.
// thread = (unit j, k-slice ks of H/8): it owns the 4 gate rows of its unit over
// its slice (4 x H/8 weights), reads only H/8 values of h per step (LDS traffic
// is what bounds the step), reduces the 4 partial sums over the unit's 8 adjacent
// lanes with DPP adds, evaluates one gate per lane with a single fast sigmoid
// (tanh(x) = 2 sigmoid(2x) - 1), and lane 0 of the unit updates c, h.  ONE
// workgroup barrier per time step.
#include "sfm_common.h"

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {          // v + v[permuted lane] (within a row of 16 lanes)
  const int x = __builtin_bit_cast(int, v);
  const int y = __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false);
  return v + __builtin_bit_cast(float, y);
}
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
  return __builtin_bit_cast(float, y);
}
#define DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141  // lane i <-> 7-i inside each group of 8
#define DPP_Q0 0x00          // quad_perm [0,0,0,0]
#define DPP_Q1 0x55
#define DPP_Q2 0xAA
#define DPP_Q3 0xFF

// 1 / (1 + 2^(-x log2 e)): v_exp_f32 + v_rcp_f32 (1 ulp each), no division sequence on the per-step critical path
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}

template <int H>
__global__ __launch_bounds__(8 * H) void bilstm_layer_kernel(const float* __restrict__ xg,
                                                             const float* __restrict__ whh,
                                                             float* __restrict__ out, int T) {
  constexpr int KS = H / 8;                                  // k-slice length per lane
  constexpr int SL = KS + 4;                                 // slice stride in LDS: slices ks and ks+4 on different banks
  __shared__ __attribute__((aligned(16))) float hs[2][8 * SL];
  const int tid = threadIdx.x;
  const int j = tid >> 3, ks = tid & 7;
  const int dir = blockIdx.x, b = blockIdx.y;
  float w[4][KS];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* wr = whh + ((long long)dir * 4 * H + g * H + j) * H + ks * KS;
#pragma unroll
    for (int i = 0; i < KS; ++i) w[g][i] = wr[i];
  }
  if (tid < 8 * SL) { hs[0][tid] = 0.f; hs[1][tid] = 0.f; }
  __syncthreads();
  float c = 0.f;
  const int mygate = ks & 3;                                  // lanes 0-3 (and 4-7) carry gate i,f,g,o
  const float* xb = xg + (long long)b * T * (8 * H) + (long long)dir * 4 * H + mygate * H + j;
  float* ob = out + (long long)b * T * (2 * H) + dir * H + j;
  int t = dir ? (T - 1) : 0;
  const int dt = dir ? -1 : 1;
  const float gsc = (mygate == 2) ? 2.0f : 1.0f, gof = (mygate == 2) ? -1.0f : 0.0f;
  const int hslot = (j / KS) * SL + (j % KS);                // where unit j's h lives
  float xnext = xb[(long long)t * (8 * H)];                   // every lane loads the input projection of ITS gate
  for (int s = 0; s < T; ++s, t += dt) {
    const float xcur = xnext;
    if (s + 1 < T) xnext = xb[(long long)(t + dt) * (8 * H)];
    const float* hc = hs[s & 1] + ks * SL;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KS; i += 4) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hc + i);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        a[g] += w[g][i] * hv[0];
        a[g] += w[g][i + 1] * hv[1];
        a[g] += w[g][i + 2] * hv[2];
        a[g] += w[g][i + 3] * hv[3];
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      a[g] = dpp_add<DPP_XOR1>(a[g]);
      a[g] = dpp_add<DPP_XOR2>(a[g]);
      a[g] = dpp_add<DPP_HALF_MIRROR>(a[g]);
    }
    // every lane now holds the 4 complete pre-activations of its unit; lane q of each quad activates gate q
    const float pre = ((mygate == 0) ? a[0] : (mygate == 1) ? a[1] : (mygate == 2) ? a[2] : a[3]) + xcur;
    const float act = gsc * fast_sigmoid(gsc * pre) + gof;
    const float ig = dpp_get<DPP_Q0>(act), fg = dpp_get<DPP_Q1>(act);
    const float cg = dpp_get<DPP_Q2>(act), og = dpp_get<DPP_Q3>(act);
    c = fg * c + ig * cg;
    const float h = og * (2.0f * fast_sigmoid(2.0f * c) - 1.0f);
    if (ks == 0) {
      hs[(s + 1) & 1][hslot] = h;
      ob[(long long)t * (2 * H)] = h;
    }
    // LDS-only barrier (__syncthreads() would also drain vmcnt: the global store of h and the x prefetch)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// xg [B, T, 2, 4H] fp32 (dir-major gates), whh [2, 4H, H] fp32, out [B, T, 2H] fp32
extern "C" int sfm_bilstm_layer(const float* xg, const float* whh, float* out, int B, int T, int H, int dtype,
                                void* stream) {
  (void)dtype;
  if (!xg || !whh || !out) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0) return SFM_ERR_SHAPE;
  if (H == 128) {
    SFM_LAUNCH((bilstm_layer_kernel<128>), dim3(2, B), dim3(1024), 0, (hipStream_t)stream, xg, whh, out, T);
  } else if (H == 64) {
    SFM_LAUNCH((bilstm_layer_kernel<64>), dim3(2, B), dim3(512), 0, (hipStream_t)stream, xg, whh, out, T);
  } else if (H == 32) {
    SFM_LAUNCH((bilstm_layer_kernel<32>), dim3(2, B), dim3(256), 0, (hipStream_t)stream, xg, whh, out, T);
  } else {
    return SFM_ERR_SHAPE;
  }
  SFM_CHECK_LAUNCH();
  return SFM_OK;
}
