// Persistent BiLSTM layer (nn.LSTM bidirectional, gate order i,f,g,o;
// agents/cpea.py:43-50,99).  The input projection x W_ih^T + b_ih + b_hh for
// BOTH directions is done beforehand by one GEMM (xg [B, T, 2, 4H] fp32); this
// kernel runs only the sequential recurrence.  Chains are independent per
// (utterance, direction): one 8H-thread workgroup per chain keeps the whole
// fp32 W_hh (4H x H) distributed in registers (H/2 weights per thread) and h in
// LDS, so a time step costs one H/2-long FMA chain, three shuffles and ONE
// workgroup barrier; nothing but xg / h traffic touches HBM.
// thread = (unit j, gate q, k-half): the 8 threads of a unit sit in adjacent lanes.
#include "sfm_common.h"

template <int H>
__global__ __launch_bounds__(8 * H) void bilstm_layer_kernel(const float* __restrict__ xg,
                                                             const float* __restrict__ whh,
                                                             float* __restrict__ out, int T) {
  __shared__ __attribute__((aligned(16))) float hs[2][H];
  constexpr int KH = H / 2;
  const int tid = threadIdx.x;
  const int j = tid >> 3, gate = (tid >> 1) & 3, half = tid & 1;
  const int dir = blockIdx.x, b = blockIdx.y;
  const int row = gate * H + j;
  float w[KH];
  {
    const float* wr = whh + ((long long)dir * 4 * H + row) * H + half * KH;
#pragma unroll
    for (int i = 0; i < KH; ++i) w[i] = wr[i];
  }
  if (tid < H) { hs[0][tid] = 0.f; hs[1][tid] = 0.f; }
  __syncthreads();
  float c = 0.f;
  const float* xb = xg + (long long)b * T * (8 * H) + (long long)dir * 4 * H + row;
  float* ob = out + (long long)b * T * (2 * H) + dir * H + j;
  int t = dir ? (T - 1) : 0;
  const int dt = dir ? -1 : 1;
  float xnext = (half == 0) ? xb[(long long)t * (8 * H)] : 0.f;
  for (int s = 0; s < T; ++s, t += dt) {
    const float xcur = xnext;
    if (s + 1 < T && half == 0) xnext = xb[(long long)(t + dt) * (8 * H)];
    const float* hc = hs[s & 1] + half * KH;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < KH; i += 4) {
      f32x4 hv = *reinterpret_cast<const f32x4*>(hc + i);
      acc += w[i] * hv[0];
      acc += w[i + 1] * hv[1];
      acc += w[i + 2] * hv[2];
      acc += w[i + 3] * hv[3];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += xcur;                       // valid on half==0 lanes (xcur = 0 on the others)
    // bring the four gates of unit j to its first lane (lane%8 == 0)
    float gi = acc;
    float gf = __shfl_down(acc, 2, 64);
    float gg = __shfl_down(acc, 4, 64);
    float go = __shfl_down(acc, 6, 64);
    if ((tid & 7) == 0) {
      float ig = 1.0f / (1.0f + expf(-gi));
      float fg = 1.0f / (1.0f + expf(-gf));
      float cg = tanhf(gg);
      float og = 1.0f / (1.0f + expf(-go));
      c = fg * c + ig * cg;
      float h = og * tanhf(c);
      hs[(s + 1) & 1][j] = h;
      ob[(long long)t * (2 * H)] = h;
    }
    __syncthreads();
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
