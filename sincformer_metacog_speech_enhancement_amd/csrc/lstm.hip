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

#ifndef SFM_LSTM_LPU
#define SFM_LSTM_LPU 4                       // lanes per hidden unit of the forward recurrence (8 = round 1's mapping)
#endif

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

// save (training): [B, T, 2, 5, H] fp32 = the activated gates i, f, g, o and the cell state c of every step
// LPU lanes per hidden unit: thread = (unit j, k-slice ks of H / LPU).  8 lanes per unit (16 waves at H 128) is 4 waves per
// SIMD x ~100 instructions per step = issue bound at ~1600 cycles; 4 lanes per unit (8 waves, 128 multiply-adds per lane in four
// independent chains, one DPP stage fewer) issues ~1330 (tools/lstm_bench.py: 0.60 -> see profiles/README.md).
template <int H, int LPU>
__global__ __launch_bounds__(LPU * H) void bilstm_layer_kernel(const float* __restrict__ xg,
                                                               const float* __restrict__ whh,
                                                               float* __restrict__ out, int T, float* __restrict__ save) {
  constexpr int KS = H / LPU;                                // k-slice length per lane
  constexpr int SL = KS + 4;                                 // slice stride in LDS: the slices of a unit's lanes on different banks
  __shared__ __attribute__((aligned(16))) float hs[2][LPU * SL];
  const int tid = threadIdx.x;
  const int j = tid / LPU, ks = tid % LPU;
  const int dir = blockIdx.x, b = blockIdx.y;
  float w[4][KS];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* wr = whh + ((long long)dir * 4 * H + g * H + j) * H + ks * KS;
#pragma unroll
    for (int i = 0; i < KS; ++i) w[g][i] = wr[i];
  }
  if (tid < LPU * SL) { hs[0][tid] = 0.f; hs[1][tid] = 0.f; }
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
      if (LPU == 8) a[g] = dpp_add<DPP_HALF_MIRROR>(a[g]);
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
    if (save) {
      float* sv = save + ((((long long)b * T + t) * 2 + dir) * 5) * H + j;
      if (ks < 4) sv[mygate * H] = act;
      if (ks == LPU - 4) sv[4 * H] = c;                      // (8 lanes: lane 4; 4 lanes: lane 0)
    }
    // LDS-only barrier (__syncthreads() would also drain vmcnt: the global store of h and the x prefetch)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// Inference form with the recurrent product on fp16 operands: W_hh as packed half pairs in registers (64 instead of 128 per lane:
// the kernel fits 4 waves per SIMD, so TWO chains share a CU - at batch 256 there are 512 chains for 256 CUs and the fp32 form runs
// them in two rounds), h published in LDS as fp16 (half the LDS bytes of a step), `v_dot2_f32_f16` with fp32 accumulation (half the
// multiply-add instructions).  c, the gates and the h written to `out` stay fp32; what is rounded is the recurrent operand pair, to
// the format every MFMA operand of the 16-bit path already has.  SAVE: the training forward (gates and cell states for the BPTT).
typedef _Float16 lstm_h2 __attribute__((ext_vector_type(2)));
template <int H, int LPU, bool SAVE>
__global__ __launch_bounds__(LPU * H, 4) void bilstm_layer16_kernel(const float* __restrict__ xg, const float* __restrict__ whh,
                                                                    float* __restrict__ out, int T, float* __restrict__ save) {
  constexpr int KS = H / LPU;                                // h values per lane
  constexpr int KW = KS / 2;                                 // = dwords (half pairs) per slice
  constexpr int SLW = KW + 4;                                // dword stride of a slice
  static_assert(KW % 4 == 0, "16-byte slice reads");
  __shared__ __attribute__((aligned(16))) uint32_t hs[2][LPU * SLW];
  const int tid = threadIdx.x;
  const int j = tid / LPU, ks = tid % LPU;
  const int dir = blockIdx.x, b = blockIdx.y;
  lstm_h2 w[4][KW];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* wr = whh + ((long long)dir * 4 * H + g * H + j) * H + ks * KS;
#pragma unroll
    for (int i = 0; i < KW; ++i) w[g][i] = lstm_h2{(_Float16)wr[2 * i], (_Float16)wr[2 * i + 1]};
    __builtin_amdgcn_sched_barrier(0);                       // one gate's fp32 rows in flight at a time (all four at once spill)
  }
  if (tid < LPU * SLW) { hs[0][tid] = 0u; hs[1][tid] = 0u; }
  __syncthreads();
  float c = 0.f;
  const int mygate = ks & 3;
  const float* xb = xg + (long long)b * T * (8 * H) + (long long)dir * 4 * H + mygate * H + j;
  float* ob = out + (long long)b * T * (2 * H) + dir * H + j;
  int t = dir ? (T - 1) : 0;
  const int dt = dir ? -1 : 1;
  const float gsc = (mygate == 2) ? 2.0f : 1.0f, gof = (mygate == 2) ? -1.0f : 0.0f;
  // unit j's h as a byte address in LDS.  The 16-bit store is an asm statement: a C++ store through a _Float16 lvalue into the
  // uint32_t array is an aliasing violation, and the compiler then reads ONE dword per 16-byte group and reuses it
  const uint32_t hbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&hs[0][0];
  const uint32_t hslot = hbase + ((j / KS) * SLW * 2 + (j % KS)) * 2;
  const uint32_t hread = hbase + ks * SLW * 4;                // this lane's slice (the reads are asm too: 16-byte, counted waits)
  float xnext = xb[(long long)t * (8 * H)];
  for (int s = 0; s < T; ++s, t += dt) {
    const float xcur = xnext;
    if (s + 1 < T) xnext = xb[(long long)(t + dt) * (8 * H)];
    const uint32_t hc = hread + (uint32_t)((s & 1) * LPU * SLW * 4);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    u32x4 hv[KW / 4];
#pragma unroll
    for (int q = 0; q < KW / 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hv[q]) : "v"(hc), "n"(q * 16) : "memory");
    // every group passes through a wait statement (the compiler may move the dot products across an asm they do not depend on)
    if constexpr (KW / 4 == 4) {
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(hv[0]), "+v"(hv[1])::"memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hv[2]), "+v"(hv[3])::"memory");
    } else {
      static_assert(KW / 4 == 2, "H 128 or 64 with 4 lanes per unit");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hv[0]), "+v"(hv[1])::"memory");
    }
#pragma unroll
    for (int q = 0; q < KW / 4; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t hd = hv[q][e];                          // (a scalar copy first: __builtin_bit_cast applied to a vector ELEMENT
        const lstm_h2 hp = __builtin_bit_cast(lstm_h2, hd);    //  reads element 0 with this hipcc)
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = __builtin_amdgcn_fdot2(w[g][4 * q + e], hp, a[g], false);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      a[g] = dpp_add<DPP_XOR1>(a[g]);
      a[g] = dpp_add<DPP_XOR2>(a[g]);
      if (LPU == 8) a[g] = dpp_add<DPP_HALF_MIRROR>(a[g]);
    }
    const float pre = ((mygate == 0) ? a[0] : (mygate == 1) ? a[1] : (mygate == 2) ? a[2] : a[3]) + xcur;
    const float act = gsc * fast_sigmoid(gsc * pre) + gof;
    const float ig = dpp_get<DPP_Q0>(act), fg = dpp_get<DPP_Q1>(act);
    const float cg = dpp_get<DPP_Q2>(act), og = dpp_get<DPP_Q3>(act);
    c = fg * c + ig * cg;
    const float h = og * (2.0f * fast_sigmoid(2.0f * c) - 1.0f);
    if (ks == 0) {
      const uint32_t hb = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)h);
      asm volatile("ds_write_b16 %0, %1" ::"v"(hslot + (uint32_t)(((s + 1) & 1) * LPU * SLW * 4)), "v"(hb) : "memory");
      ob[(long long)t * (2 * H)] = h;
    }
    if (SAVE) {                                              // training: activated gates and cell state for the BPTT (fp32)
      float* sv = save + ((((long long)b * T + t) * 2 + dir) * 5) * H + j;
      if (ks < 4) sv[mygate * H] = act;
      if (ks == LPU - 4) sv[4 * H] = c;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// xg [B, T, 2, 4H] fp32 (dir-major gates), whh [2, 4H, H] fp32, out [B, T, 2H] fp32
// ---------------------------------------------------------------------------
// BPTT of one BiLSTM layer (training of agents/cpea.py:43-50): same workgroup = chain mapping as the forward.
// Per step (reverse of the chain's own time order):  dh = dout[t] + W_hh^T da[t+];  from the saved gates / cell states
// the pre-activation gradients da = (di i(1-i), df f(1-f), dg (1-g^2), do o(1-o)) are formed by lane 0 of each unit,
// published through LDS, written to dxg (= gradient w.r.t. the input projection, from which dW_ih, db, dx and dW_hh are
// GEMMs afterwards) and every thread multiplies its 4 x H/8 slice of W_hh^T into the next step's recurrent dh.
// ---------------------------------------------------------------------------
// LPU lanes per hidden unit (as the forward kernel): 8 = 16 waves at H 128, 64 multiply-adds per lane and step; 4 = 8 waves, 128
// multiply-adds in two chains, one DPP stage fewer - the per-wave cost of the gate-gradient section (8 or 16 active lanes) is
// paid by half as many waves.
template <int H, int LPU>
__global__ __launch_bounds__(LPU * H) void bilstm_layer_bwd_kernel(const float* __restrict__ save,
                                                                 const float* __restrict__ whh,
                                                                 const float* __restrict__ dout,
                                                                 float* __restrict__ dxg, int T) {
  constexpr int G4 = 4 * H;                                   // contraction length of W_hh^T
  constexpr int KS = G4 / LPU;                                // slice per lane
  constexpr int SL = KS + 4;
  __shared__ __attribute__((aligned(16))) float das[2][LPU * SL];
  const int tid = threadIdx.x;
  const int j = tid / LPU, ks = tid % LPU;
  const int dir = blockIdx.x, b = blockIdx.y;
  float w[KS];                                                // W_hh[r, j] for r in this lane's slice of the 4H gate rows
#pragma unroll
  for (int i = 0; i < KS; ++i) w[i] = whh[((long long)dir * G4 + ks * KS + i) * H + j];
  for (int i = tid; i < LPU * SL; i += LPU * H) { das[0][i] = 0.f; das[1][i] = 0.f; }
  __syncthreads();
  const long long cb = (long long)b * T;
  float dh_rec = 0.f, dc_next = 0.f;
  // the chain ran t = t_first, t_first + dt, ...; BPTT walks it backwards
  const int dt = dir ? -1 : 1;
  int t = dir ? 0 : (T - 1);
  // lane 0 of each unit streams the step's 7 saved values one step ahead of their use
  float nx[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int tt) {
    const float* sv = save + (((cb + tt) * 2 + dir) * 5) * H + j;
    nx[0] = sv[0]; nx[1] = sv[H]; nx[2] = sv[2 * H]; nx[3] = sv[3 * H]; nx[4] = sv[4 * H];
    const int tp = tt - dt;                                    // previous step of the chain
    nx[5] = (tp >= 0 && tp < T) ? save[(((cb + tp) * 2 + dir) * 5 + 4) * H + j] : 0.f;
    nx[6] = dout[(cb + tt) * (2 * H) + dir * H + j];
  };
  if (ks == 0) fetch(t);
  for (int s = 0; s < T; ++s, t -= dt) {
    if (ks == 0) {
      const float ig = nx[0], fg = nx[1], gg = nx[2], og = nx[3], c = nx[4], cp = nx[5];
      const float dh = nx[6] + dh_rec;
      if (s + 1 < T) fetch(t - dt);
      const float tc = 2.0f * fast_sigmoid(2.0f * c) - 1.0f;
      const float dc = dh * og * (1.0f - tc * tc) + dc_next;
      const float dai = dc * gg * ig * (1.0f - ig);
      const float daf = dc * cp * fg * (1.0f - fg);
      const float dag = dc * ig * (1.0f - gg * gg);
      const float dao = dh * tc * og * (1.0f - og);
      dc_next = dc * fg;
      float* dx = dxg + (cb + t) * (8 * H) + (long long)dir * G4 + j;
      dx[0] = dai; dx[H] = daf; dx[2 * H] = dag; dx[3 * H] = dao;
      float* da = das[s & 1];                                  // row r = g*H + j lives at (r / KS) * SL + r % KS
      da[((0 * H + j) / KS) * SL + (0 * H + j) % KS] = dai;
      da[((1 * H + j) / KS) * SL + (1 * H + j) % KS] = daf;
      da[((2 * H + j) / KS) * SL + (2 * H + j) % KS] = dag;
      da[((3 * H + j) / KS) * SL + (3 * H + j) % KS] = dao;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float* dc_ = das[s & 1] + ks * SL;
    float a = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < KS; i += 8) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(dc_ + i);
      const f32x4 u = *reinterpret_cast<const f32x4*>(dc_ + i + 4);
      a += w[i] * v[0];
      a2 += w[i + 4] * u[0];
      a += w[i + 1] * v[1];
      a2 += w[i + 5] * u[1];
      a += w[i + 2] * v[2];
      a2 += w[i + 6] * u[2];
      a += w[i + 3] * v[3];
      a2 += w[i + 7] * u[3];
    }
    a += a2;
    a = dpp_add<DPP_XOR1>(a);
    a = dpp_add<DPP_XOR2>(a);
    if (LPU == 8) a = dpp_add<DPP_HALF_MIRROR>(a);
    dh_rec = a;                                                // every lane of the unit holds it; lane 0 uses it
  }
}

template <int H>
static int bilstm_fwd_go(const float* xg, const float* whh, float* out, float* save, int B, int T, hipStream_t st, bool w16 = false) {
  constexpr int LPU = H >= 64 ? SFM_LSTM_LPU : 8;           // H 32: 4 x H / 4 = 8-float slices, too short for the float4 reads
  if constexpr (H >= 64 && (H / LPU) % 8 == 0) {
    if (w16) {
      if (save) SFM_LAUNCH((bilstm_layer16_kernel<H, LPU, true>), dim3(2, B), dim3(LPU * H), 0, st, xg, whh, out, T, save);
      else SFM_LAUNCH((bilstm_layer16_kernel<H, LPU, false>), dim3(2, B), dim3(LPU * H), 0, st, xg, whh, out, T, save);
      return SFM_OK;
    }
  }
  SFM_LAUNCH((bilstm_layer_kernel<H, LPU>), dim3(2, B), dim3(LPU * H), 0, st, xg, whh, out, T, save);
  return SFM_OK;
}

// xg [B, T, 2, 4H] fp32 (dir-major gates), whh [2, 4H, H] fp32, out [B, T, 2H] fp32
extern "C" int sfm_bilstm_layer_train(const float* xg, const float* whh, float* out, float* save, int B, int T, int H,
                                      int dtype, void* stream) {
  (void)dtype;
  if (!xg || !whh || !out) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0) return SFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (H == 128) return bilstm_fwd_go<128>(xg, whh, out, save, B, T, st);
  if (H == 64) return bilstm_fwd_go<64>(xg, whh, out, save, B, T, st);
  if (H == 32) return bilstm_fwd_go<32>(xg, whh, out, save, B, T, st);
  return SFM_ERR_SHAPE;
}

extern "C" int sfm_bilstm_layer(const float* xg, const float* whh, float* out, int B, int T, int H, int dtype,
                                void* stream) {
  return sfm_bilstm_layer_train(xg, whh, out, nullptr, B, T, H, dtype, stream);
}

// training forward with the recurrent product on fp16 operands when w16 != 0 (the reference trains its nn.LSTM under fp16 autocast:
// training/conformer_pipeline.py:504); the saved gates / cell states and the BPTT stay fp32
extern "C" int sfm_bilstm_layer_train_ex(const float* xg, const float* whh, float* out, float* save, int B, int T, int H, int w16,
                                         void* stream) {
  if (!xg || !whh || !out || !save) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0) return SFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (H == 128) return bilstm_fwd_go<128>(xg, whh, out, save, B, T, st, w16 != 0);
  if (H == 64) return bilstm_fwd_go<64>(xg, whh, out, save, B, T, st, w16 != 0);
  if (H == 32) return bilstm_fwd_go<32>(xg, whh, out, save, B, T, st, false);
  return SFM_ERR_SHAPE;
}

// inference, recurrent product on fp16 operands when w16 != 0 (bilstm_layer16_kernel; H 32 keeps the fp32 kernel)
extern "C" int sfm_bilstm_layer_ex(const float* xg, const float* whh, float* out, int B, int T, int H, int w16, void* stream) {
  if (!xg || !whh || !out) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0) return SFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (H == 128) return bilstm_fwd_go<128>(xg, whh, out, nullptr, B, T, st, w16 != 0);
  if (H == 64) return bilstm_fwd_go<64>(xg, whh, out, nullptr, B, T, st, w16 != 0);
  if (H == 32) return bilstm_fwd_go<32>(xg, whh, out, nullptr, B, T, st, false);
  return SFM_ERR_SHAPE;
}

// save [B, T, 2, 5, H] from sfm_bilstm_layer_train, dout [B, T, 2H] fp32 -> dxg [B, T, 2, 4H] fp32
extern "C" int sfm_bilstm_layer_bwd(const float* save, const float* whh, const float* dout, float* dxg, int B, int T, int H,
                                    void* stream) {
  if (!save || !whh || !dout || !dxg) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0) return SFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  static const int lpu = getenv("SFM_LSTM_BWD_LPU") ? atoi(getenv("SFM_LSTM_BWD_LPU")) : 4;    // A/B knob (tools/lstm_bench.py)
  if (H == 128 && lpu == 4) SFM_LAUNCH((bilstm_layer_bwd_kernel<128, 4>), dim3(2, B), dim3(512), 0, st, save, whh, dout, dxg, T);
  else if (H == 128) SFM_LAUNCH((bilstm_layer_bwd_kernel<128, 8>), dim3(2, B), dim3(1024), 0, st, save, whh, dout, dxg, T);
  else if (H == 64) SFM_LAUNCH((bilstm_layer_bwd_kernel<64, 8>), dim3(2, B), dim3(512), 0, st, save, whh, dout, dxg, T);
  else if (H == 32) SFM_LAUNCH((bilstm_layer_bwd_kernel<32, 8>), dim3(2, B), dim3(256), 0, st, save, whh, dout, dxg, T);
  else return SFM_ERR_SHAPE;
  return SFM_OK;
}
