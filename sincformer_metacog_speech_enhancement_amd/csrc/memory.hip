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

// ---------------------------------------------------------------------------
// Backward of the same function (training: EpisodicMemory in train() mode).  One workgroup per query row recomputes the
// forward intermediates in LDS (the whole forward is ~0.4 MFLOP per row) and propagates
//   d(bias * gate) [vd], d(gate) [1]  ->  d(emb) [kd]  and the gradient of every parameter,
// accumulated over the rows into `dparams`, a zero-filled blob with the layout of `params`: with a workspace every row writes its
// own copy of the blob (each element exactly once, plain stores) and reduce.hip folds the rows in order - bit-reproducible, no
// contention; without one, ~170 k fp32 atomics per row on the same addresses.
// top_indices / similarity are not differentiable (agents/memory.py:136-146 uses them for bookkeeping only).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void matvec_t(const float* __restrict__ W, const float* d, float* y, int rows, int cols, int tid) {
  // y[k] = sum_i W[i][k] d[i]   (thread per column: consecutive threads read consecutive addresses of a row)
  for (int k = tid; k < cols; k += 256) {
    float acc = 0.f;
    for (int i = 0; i < rows; ++i) acc += W[(long long)i * cols + k] * d[i];
    y[k] = acc;
  }
}
// own != 0: dst is this row's private copy of the blob (store); else the shared blob (atomic add)
__device__ __forceinline__ void mem_acc(float* dst, float v, int own) {
  if (own) *dst = v;
  else atomicAdd(dst, v);
}
__device__ __forceinline__ void outer_atomic(float* __restrict__ dW, const float* d, const float* x, int rows, int cols, int tid,
                                             int own) {
  // dW[i][k] += d[i] x[k]
  for (int e = tid; e < rows * cols; e += 256) {
    const int i = e / cols, k = e - i * cols;
    mem_acc(dW + e, d[i] * x[k], own);
  }
}

__global__ __launch_bounds__(256) void memory_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ P,
                                                         const float* __restrict__ d_out, const float* __restrict__ d_gate,
                                                         float* __restrict__ d_emb, float* __restrict__ dP_shared, int kd, int vd,
                                                         int S, float temperature, float* __restrict__ ws, long long n_params) {
  __shared__ float x[MEM_MAXD], t1[MEM_MAXD], xhat[MEM_MAXD], yv[MEM_MAXD], u[MEM_MAXD], q[MEM_MAXD], cosv[MEM_MAXD], knorm[MEM_MAXD],
      att[MEM_MAXD], ret[MEM_MAXD], tb[MEM_MAXD], da[MEM_MAXD], db[MEM_MAXD], dq[MEM_MAXD], red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int own = ws != nullptr;
  float* dP = own ? ws + (long long)b * n_params : dP_shared;
  const long long oW0 = 0, ob0 = oW0 + (long long)kd * kd, olnw = ob0 + kd, olnb = olnw + kd, oW3 = olnb + kd,
                  ob3 = oW3 + (long long)kd * kd, okeys = ob3 + kd, ovals = okeys + (long long)S * kd,
                  oWv = ovals + (long long)S * vd, obv = oWv + (long long)vd * vd, oWg = obv + vd, obg = oWg + kd + vd;
  const float *W0 = P + oW0, *b0 = P + ob0, *lnw = P + olnw, *lnb = P + olnb, *W3 = P + oW3, *b3 = P + ob3, *keys = P + okeys,
              *vals = P + ovals, *Wv = P + oWv, *bv = P + obv, *Wg = P + oWg, *bg = P + obg;
  // ------------------------------ forward, keeping the intermediates ------------------------------
  for (int i = tid; i < kd; i += 256) x[i] = emb[(long long)b * kd + i];
  __syncthreads();
  matvec_wave(W0, b0, x, t1, kd, kd, wave, lane, 4);
  __syncthreads();
  if (wave == 0) {
    float s_ = 0.f;
    for (int i = lane; i < kd; i += 64) s_ += t1[i];
    const float mean = wave_sum(s_) / (float)kd;
    float v = 0.f;
    for (int i = lane; i < kd; i += 64) { const float c = t1[i] - mean; v += c * c; }
    const float rstd = rsqrtf(wave_sum(v) / (float)kd + 1e-5f);
    for (int i = lane; i < kd; i += 64) {
      xhat[i] = (t1[i] - mean) * rstd;
      yv[i] = xhat[i] * lnw[i] + lnb[i];
      u[i] = gelu_erf(yv[i]);
    }
    if (lane == 0) red[2] = rstd;
  }
  __syncthreads();
  matvec_wave(W3, b3, u, q, kd, kd, wave, lane, 4);
  __syncthreads();
  if (wave == 0) {
    float s_ = 0.f;
    for (int i = lane; i < kd; i += 64) s_ += q[i] * q[i];
    s_ = wave_sum(s_);
    if (lane == 0) red[0] = fmaxf(sqrtf(s_), 1e-12f);
  }
  __syncthreads();
  const float qn = red[0];
  for (int r = wave; r < S; r += 4) {
    const float* kr = keys + (long long)r * kd;
    float dot = 0.f, kk = 0.f;
    for (int k = lane; k < kd; k += 64) { const float kv = kr[k]; dot += kv * q[k]; kk += kv * kv; }
    dot = wave_sum(dot);
    kk = wave_sum(kk);
    if (lane == 0) {
      knorm[r] = fmaxf(sqrtf(kk), 1e-12f);
      cosv[r] = dot / (qn * knorm[r]);
    }
  }
  __syncthreads();
  if (wave == 0) {
    float m = -1e30f;
    for (int r = lane; r < S; r += 64) m = fmaxf(m, cosv[r] / temperature);
    m = wave_max(m);
    float s_ = 0.f;
    for (int r = lane; r < S; r += 64) s_ += expf(cosv[r] / temperature - m);
    s_ = wave_sum(s_);
    for (int r = lane; r < S; r += 64) att[r] = expf(cosv[r] / temperature - m) / s_;
  }
  __syncthreads();
  for (int v = tid; v < vd; v += 256) {
    float acc = 0.f;
    for (int r = 0; r < S; ++r) acc += att[r] * vals[(long long)r * vd + v];
    ret[v] = acc;
  }
  __syncthreads();
  matvec_wave(Wv, bv, ret, tb, vd, vd, wave, lane, 4);
  if (wave == 0) {
    float s_ = 0.f;
    for (int k = lane; k < kd; k += 64) s_ += Wg[k] * q[k];
    for (int k = lane; k < vd; k += 64) s_ += Wg[kd + k] * ret[k];
    s_ = wave_sum(s_);
    if (lane == 0) red[1] = 1.0f / (1.0f + expf(-(s_ + bg[0])));
  }
  __syncthreads();
  const float gate = red[1];
  // ------------------------------ backward ------------------------------
  // out = tanh(vp) * gate: d vp -> da[0..vd), d gate (total)
  float part = 0.f;
  for (int v = tid; v < vd; v += 256) {
    const float t = tanhf(tb[v]);
    const float g = d_out[(long long)b * vd + v];
    da[v] = g * gate * (1.0f - t * t);
    part += g * t;
  }
  part = wave_sum(part);
  if (lane == 0) red[4 + wave] = part;
  __syncthreads();
  const float dgate = (d_gate ? d_gate[b] : 0.f) + red[4] + red[5] + red[6] + red[7];
  const float dgpre = dgate * gate * (1.0f - gate);
  // value_proj: dWv += da (x) ret, dbv += da, d ret = Wv^T da (-> db[0..vd))
  outer_atomic(dP + oWv, da, ret, vd, vd, tid, own);
  for (int v = tid; v < vd; v += 256) mem_acc(dP + obv + v, da[v], own);
  matvec_t(Wv, da, db, vd, vd, tid);
  // gate: dWg += dgpre * [q | ret], dbg += dgpre, dq = dgpre * Wg[:kd], d ret += dgpre * Wg[kd:]
  for (int k = tid; k < kd; k += 256) {
    mem_acc(dP + oWg + k, dgpre * q[k], own);
    dq[k] = dgpre * Wg[k];
  }
  __syncthreads();
  for (int v = tid; v < vd; v += 256) {
    mem_acc(dP + oWg + kd + v, dgpre * ret[v], own);
    db[v] += dgpre * Wg[kd + v];
  }
  if (tid == 0) mem_acc(dP + obg, dgpre, own);
  __syncthreads();
  // ret = att @ values: d att[r] = <d ret, values[r]> (-> da[0..S)), dvalues[r] += att[r] d ret
  for (int r = wave; r < S; r += 4) {
    float acc = 0.f;
    for (int v = lane; v < vd; v += 64) acc += db[v] * vals[(long long)r * vd + v];
    acc = wave_sum(acc);
    if (lane == 0) da[r] = acc;
  }
  for (int e = tid; e < S * vd; e += 256) {
    const int r = e / vd, v = e - r * vd;
    mem_acc(dP + ovals + e, att[r] * db[v], own);
  }
  __syncthreads();
  // softmax: d sim[r] = att[r] (d att[r] - sum att d att); c[r] = d sim[r] / temperature -> da[r]
  if (wave == 0) {
    float s_ = 0.f;
    for (int r = lane; r < S; r += 64) s_ += att[r] * da[r];
    s_ = wave_sum(s_);
    for (int r = lane; r < S; r += 64) da[r] = att[r] * (da[r] - s_) / temperature;
  }
  __syncthreads();
  // cosine: cos_r = <q, k_r> / (|q| |k_r|):  dq += sum_r c_r (k_r / (|q||k_r|) - cos_r q / |q|^2),
  //                                           dk_r += c_r (q / (|q||k_r|) - cos_r k_r / |k_r|^2)
  for (int k = tid; k < kd; k += 256) {
    float acc = 0.f;
    for (int r = 0; r < S; ++r) {
      const float kv = keys[(long long)r * kd + k];
      acc += da[r] * (kv / (qn * knorm[r]) - cosv[r] * q[k] / (qn * qn));
      mem_acc(dP + okeys + (long long)r * kd + k, da[r] * (q[k] / (qn * knorm[r]) - cosv[r] * kv / (knorm[r] * knorm[r])), own);
    }
    dq[k] += acc;
  }
  __syncthreads();
  // q = W3 u + b3
  outer_atomic(dP + oW3, dq, u, kd, kd, tid, own);
  for (int k = tid; k < kd; k += 256) mem_acc(dP + ob3 + k, dq[k], own);
  matvec_t(W3, dq, db, kd, kd, tid);                      // db = d u
  __syncthreads();
  // u = gelu(y), y = xhat * lnw + lnb, xhat = LayerNorm(t1)
  float p1 = 0.f, p2 = 0.f;
  for (int i = tid; i < kd; i += 256) {
    const float z = yv[i];
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    const float dy = db[i] * (cdf + z * 0.39894228040143267794f * expf(-0.5f * z * z));
    mem_acc(dP + olnw + i, dy * xhat[i], own);
    mem_acc(dP + olnb + i, dy, own);
    const float dxh = dy * lnw[i];
    da[i] = dxh;
    p1 += dxh;
    p2 += dxh * xhat[i];
  }
  p1 = wave_sum(p1);
  p2 = wave_sum(p2);
  __syncthreads();
  if (lane == 0) { red[4 + wave] = p1; }
  __syncthreads();
  const float s1 = (red[4] + red[5] + red[6] + red[7]) / (float)kd;
  __syncthreads();
  if (lane == 0) { red[4 + wave] = p2; }
  __syncthreads();
  const float s2 = (red[4] + red[5] + red[6] + red[7]) / (float)kd;
  const float rstd = red[2];
  for (int i = tid; i < kd; i += 256) dq[i] = rstd * (da[i] - s1 - xhat[i] * s2);      // dq = d t1
  __syncthreads();
  // t1 = W0 x + b0
  outer_atomic(dP + oW0, dq, x, kd, kd, tid, own);
  for (int k = tid; k < kd; k += 256) mem_acc(dP + ob0 + k, dq[k], own);
  matvec_t(W0, dq, db, kd, kd, tid);
  __syncthreads();
  if (d_emb)
    for (int k = tid; k < kd; k += 256) d_emb[(long long)b * kd + k] = db[k];
}

extern "C" long long sfm_memory_param_floats(int kd, int vd, int S) {
  return 2LL * kd * kd + 4LL * kd + (long long)S * kd + (long long)S * vd + (long long)vd * vd + vd + kd + vd + 1;
}

// ws (optional, >= B * sfm_memory_param_floats(key_dim, value_dim, slots) floats): per-row copies of the gradient blob, folded in row order
extern "C" int sfm_memory_bwd(const float* emb, const float* params, const float* d_out, const float* d_gate, float* d_emb,
                              float* dparams, int B, int key_dim, int value_dim, int slots, float temperature, float* ws,
                              void* stream) {
  if (!emb || !params || !d_out || !dparams) return SFM_ERR_ARG;
  if (ws && (((uintptr_t)ws) % 16) != 0) return SFM_ERR_ARG;
  if (B <= 0 || key_dim <= 0 || key_dim > MEM_MAXD || value_dim <= 0 || value_dim > MEM_MAXD || slots <= 0 ||
      slots > MEM_MAXD)
    return SFM_ERR_SHAPE;
  const long long np_ = sfm_memory_param_floats(key_dim, value_dim, slots);
  SFM_LAUNCH(memory_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, emb, params, d_out, d_gate, d_emb, dparams, key_dim,
             value_dim, slots, temperature, ws, np_);
  // (the blob is one row of np_ floats; the fold takes the scalar path when np_ is odd)
  return ws ? sfm_fold_partials(ws, dparams, 1, (int)np_, np_, B, 1, stream) : SFM_OK;
}
