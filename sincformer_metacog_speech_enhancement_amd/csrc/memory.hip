// EpisodicMemory.forward (eval) in one launch (agents/memory.py:112-133,143-148):
// key_proj (Linear-LN-GELU-Linear) -> cosine similarity against the key bank ->
// softmax -> value read -> tanh bias -> sigmoid gate.  One workgroup per query
// row; every mat-vec is "one wave per output row, lanes over k" so weight reads
// are coalesced; all intermediates live in LDS.  fp32 throughout.
// params (packed by the host, fp32, in this order):
//   W0[kd,kd] b0[kd] lnw[kd] lnb[kd] W3[kd,kd] b3[kd] keys[S,kd] values[S,vd]
//   Wv[vd,vd] bv[vd] Wg[kd+vd] bg[1]
#include "sfm_common.h"

#define MEM_MAXD 512

__device__ __forceinline__ void matvec_wave(const float* __restrict__ W, const float* __restrict__ bias,
                                            const float* x, float* y, int rows, int cols, int wave, int lane,
                                            int nwaves) {
  for (int r = wave; r < rows; r += nwaves) {
    const float* wr = W + (long long)r * cols;
    float acc = 0.f;
    for (int k = lane; k < cols; k += 64) acc += wr[k] * x[k];
    acc = wave_sum(acc);
    if (lane == 0) y[r] = acc + (bias ? bias[r] : 0.f);
  }
}

__global__ __launch_bounds__(256) void memory_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ P,
                                                         float* bias_out, float* gate_out, int* top_idx,
                                                         float* sim_out, int kd, int vd, int S, float temperature) {
  __shared__ float x[MEM_MAXD], q[MEM_MAXD], t1[MEM_MAXD], sim[MEM_MAXD], ret[MEM_MAXD], red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* W0 = P;
  const float* b0 = W0 + (long long)kd * kd;
  const float* lnw = b0 + kd;
  const float* lnb = lnw + kd;
  const float* W3 = lnb + kd;
  const float* b3 = W3 + (long long)kd * kd;
  const float* keys = b3 + kd;
  const float* vals = keys + (long long)S * kd;
  const float* Wv = vals + (long long)S * vd;
  const float* bv = Wv + (long long)vd * vd;
  const float* Wg = bv + vd;
  const float* bg = Wg + kd + vd;
  for (int i = tid; i < kd; i += 256) x[i] = emb[(long long)b * kd + i];
  __syncthreads();
  matvec_wave(W0, b0, x, t1, kd, kd, wave, lane, 4);
  __syncthreads();
  // LayerNorm + GELU over t1 (wave 0)
  if (wave == 0) {
    float s = 0.f;
    for (int i = lane; i < kd; i += 64) s += t1[i];
    float mean = wave_sum(s) / (float)kd;
    float v = 0.f;
    for (int i = lane; i < kd; i += 64) { float c = t1[i] - mean; v += c * c; }
    float rstd = rsqrtf(wave_sum(v) / (float)kd + 1e-5f);
    for (int i = lane; i < kd; i += 64) x[i] = gelu_erf((t1[i] - mean) * rstd * lnw[i] + lnb[i]);
  }
  __syncthreads();
  matvec_wave(W3, b3, x, q, kd, kd, wave, lane, 4);
  __syncthreads();
  // |q|
  if (wave == 0) {
    float s = 0.f;
    for (int i = lane; i < kd; i += 64) s += q[i] * q[i];
    s = wave_sum(s);
    if (lane == 0) red[0] = fmaxf(sqrtf(s), 1e-12f);
  }
  __syncthreads();
  const float qn = red[0];
  // cosine similarity with each key (F.normalize on both)
  for (int r = wave; r < S; r += 4) {
    const float* kr = keys + (long long)r * kd;
    float dot = 0.f, kk = 0.f;
    for (int k = lane; k < kd; k += 64) { float kv = kr[k]; dot += kv * q[k]; kk += kv * kv; }
    dot = wave_sum(dot);
    kk = wave_sum(kk);
    if (lane == 0) sim[r] = dot / (qn * fmaxf(sqrtf(kk), 1e-12f)) / temperature;
  }
  __syncthreads();
  // softmax + argmax (wave 0; first maximal index as torch.argmax)
  if (wave == 0) {
    float m = -1e30f;
    int mi = 0;
    for (int r = lane; r < S; r += 64) if (sim[r] > m) { m = sim[r]; mi = r; }
    for (int o = 32; o > 0; o >>= 1) {
      float om = __shfl_xor(m, o, 64);
      int oi = __shfl_xor(mi, o, 64);
      if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    float s = 0.f;
    for (int r = lane; r < S; r += 64) s += expf(sim[r] - m);
    s = wave_sum(s);
    for (int r = lane; r < S; r += 64) t1[r] = expf(sim[r] - m) / s;
    if (lane == 0) {
      if (top_idx) top_idx[b] = mi;
      if (sim_out) sim_out[b] = m;
    }
  }
  __syncthreads();
  // retrieved = attention @ values
  for (int v = tid; v < vd; v += 256) {
    float acc = 0.f;
    for (int r = 0; r < S; ++r) acc += t1[r] * vals[(long long)r * vd + v];
    ret[v] = acc;
  }
  __syncthreads();
  matvec_wave(Wv, bv, ret, x, vd, vd, wave, lane, 4);     // x <- value_proj linear
  if (wave == 0) {
    float s = 0.f;
    for (int k = lane; k < kd; k += 64) s += Wg[k] * q[k];
    for (int k = lane; k < vd; k += 64) s += Wg[kd + k] * ret[k];
    s = wave_sum(s);
    if (lane == 0) red[1] = 1.0f / (1.0f + expf(-(s + bg[0])));
  }
  __syncthreads();
  const float g = red[1];
  for (int v = tid; v < vd; v += 256) bias_out[(long long)b * vd + v] = tanhf(x[v]) * g;
  if (tid == 0 && gate_out) gate_out[b] = g;
}

extern "C" int sfm_memory_fwd(const float* emb, const float* params, float* bias_out, float* gate_out, int* top_idx,
                              float* sim_out, int B, int key_dim, int value_dim, int slots, float temperature,
                              void* stream) {
  if (!emb || !params || !bias_out) return SFM_ERR_ARG;
  if (B <= 0 || key_dim <= 0 || key_dim > MEM_MAXD || value_dim <= 0 || value_dim > MEM_MAXD || slots <= 0 ||
      slots > MEM_MAXD)
    return SFM_ERR_SHAPE;
  SFM_LAUNCH(memory_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, emb, params, bias_out, gate_out,
                     top_idx, sim_out, key_dim, value_dim, slots, temperature);
  SFM_CHECK_LAUNCH();
  return SFM_OK;
}
