// Common device helpers for the gfx950 (MI355X / CDNA4) kernels.
// wave = 64 lanes everywhere; MFMA fragment maps follow the CDNA4 ISA:
//   v_mfma_f32_32x32x16_{bf16,f16}: A lane l -> row l&31, k = 8*(l>>5)+j (j<8)
//                                   B lane l -> col l&31, k = 8*(l>>5)+j
//   C/D (16 regs): col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
//   v_mfma_f32_32x32x2_f32: A lane l -> A[l&31][l>>5], B lane l -> B[l>>5][l&31]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SFM_OK 0
#define SFM_ERR_ARG -1
#define SFM_ERR_SHAPE -2
#define SFM_ERR_LAUNCH -3

#define SFM_DT_BF16 0
#define SFM_DT_F16 1

typedef uint16_t u16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;   // 8 x 16-bit packed
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

struct BF16 {
  static constexpr int id = SFM_DT_BF16;
  static __device__ __forceinline__ u16 from_f32(float f) {
    __bf16 h = (__bf16)f;                         // v_cvt_pk_bf16_f32, RNE, NaN-preserving
    return __builtin_bit_cast(u16, h);
  }
  static __device__ __forceinline__ float to_f32(u16 b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {   // one v_cvt_pk_bf16_f32 (RNE)
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  // v_mfma_f32_16x16x32: A lane l -> row l&15, k = 8*(l>>4)+j; B lane l -> col l&15, same k; C/D col = l&15, row = 4*(l>>4)+r
  static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

struct F16 {
  static constexpr int id = SFM_DT_F16;
  static __device__ __forceinline__ u16 from_f32(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(u16, h);
  }
  static __device__ __forceinline__ float to_f32(u16 b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
  static __device__ __forceinline__ uint32_t pack(float lo, float hi) {   // RNE (not the RTZ pk convert)
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

template <class T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return T::pack(lo, hi);
}

// ---- counter-based dropout: keep(seed, idx) is a pure function, so forward and backward agree and no
//      mask is stored.  (Statistically equivalent to, not bit-identical with, torch's Philox stream.)
__device__ __forceinline__ uint32_t sfm_hash(uint32_t seed, unsigned long long idx) {
  uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ (uint32_t)(idx >> 32) * 0x85EBCA77u ^ seed;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
// keep(seed, idx): full hash of the 8-element group idx >> 3, then one multiply-add + one xorshift-multiply round per element
// (the epilogues and ew_train handle 8 consecutive elements per thread: 1 full hash + 8 cheap rounds instead of 8 full hashes).
__device__ __forceinline__ uint32_t sfm_keep_threshold(float p) { return (uint32_t)ceilf(p * 16777216.0f); }
__device__ __forceinline__ float sfm_keep_from_group(uint32_t gh, uint32_t j, uint32_t thr24, float inv_keep) {
  uint32_t x = gh + j * 0x9E3779B1u;
  x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return ((x >> 8) >= thr24) ? inv_keep : 0.f;
}
__device__ __forceinline__ float sfm_keep_scale(uint32_t seed, unsigned long long idx, float p, float inv_keep) {
  // 0 if dropped, 1/(1-p) if kept
  return sfm_keep_from_group(sfm_hash(seed, idx >> 3), (uint32_t)(idx & 7ull), sfm_keep_threshold(p), inv_keep);
}
// the same decisions for 8 consecutive elements; one group hash when e0 is a multiple of 8
__device__ __forceinline__ void sfm_keep_scale8(uint32_t seed, unsigned long long e0, float p, float inv_keep, float (&k)[8]) {
  if (e0 & 7ull) {
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = sfm_keep_scale(seed, e0 + i, p, inv_keep);
    return;
  }
  const uint32_t gh = sfm_hash(seed, e0 >> 3), thr = sfm_keep_threshold(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = sfm_keep_from_group(gh, (uint32_t)i, thr, inv_keep);
}


// row of C/D register r for lane l (32x32 tiles)
__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// Normal CDF without a transcendental: Phi(z) - 1/2 = z Q(z^2) on |z| <= 3.75 (clamped beyond), Q of degree 6 fitted minimax to
// the erf form: |error| < 5.7e-5 over all z - below the spacing of the 16-bit values it is used in front of (the GroupNorm
// backward's GELU derivative, the training-mode GroupNorm + GELU pass with a 16-bit result).
__device__ __forceinline__ float normal_cdf_poly(float z) {
  const float zc = __builtin_amdgcn_fmed3f(z, -3.75f, 3.75f);
  const float u = zc * zc;
  float q = 3.912424329e-08f;
  q = fmaf(q, u, -2.376248530e-06f);
  q = fmaf(q, u, 6.234773063e-05f);
  q = fmaf(q, u, -9.441793120e-04f);
  q = fmaf(q, u, 9.362551949e-03f);
  q = fmaf(q, u, -6.578987097e-02f);
  q = fmaf(q, u, 3.987064729e-01f);
  return fmaf(zc, q, 0.5f);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float swish_f(float x) { return x * sigmoid_f(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// wave-wide sum without LDS traffic: 4 DPP steps leave every 16-lane row's total in all of its lanes,
// then the four row totals are combined through scalar registers (v_readlane)
template <int CTRL>
__device__ __forceinline__ float dpp_perm_add(float v) {
  const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
  return v + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_perm_add<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_perm_add<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_perm_add<0x141>(v);    // row_half_mirror
  v = dpp_perm_add<0x140>(v);    // row_mirror
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}

// Sixteen wave-wide sums at once: lane l returns the sum over all 64 lanes of v[l & 15].  A transposing butterfly: at each
// of the four in-row stages a lane keeps half of its values and hands the other half to a partner lane that keeps those, so
// the work halves per stage (8 + 4 + 2 + 1 DPP adds instead of 16 x 4).  gfx9 DPP has no xor-4 / xor-8 lane permutation; the
// mirrors (xor 15, xor 7) do, provided they come FIRST - a partner must agree with the lane on every bit already used for
// selecting (row_mirror flips bits 3..0: used when none is; row_half_mirror flips 2..0: after bit 3 only; then xor 2, xor 1).
// The four 16-lane rows are then added by the gfx950 lane swaps.  ~55 VALU instructions against ~190 for 16 wave_sum_dpp.
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum16_transpose(const float (&v)[16], int lane) {
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
  float w[8], x[4], y[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = (b3 ? v[i + 8] : v[i]) + dpp_perm<0x140>(b3 ? v[i] : v[i + 8]);      // row_mirror
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = (b2 ? w[i + 4] : w[i]) + dpp_perm<0x141>(b2 ? w[i] : w[i + 4]);      // row_half_mirror
#pragma unroll
  for (int i = 0; i < 2; ++i) y[i] = (b1 ? x[i + 2] : x[i]) + dpp_perm<0x4E>(b1 ? x[i] : x[i + 2]);       // quad_perm [2,3,0,1]
  float z = (b0 ? y[1] : y[0]) + dpp_perm<0xB1>(b0 ? y[0] : y[1]);                                         // quad_perm [1,0,3,2]
  // (inline asm: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane{16,32}_swap into one register when both
  //  inputs are the same value; s_nop 1 = the VALU-write -> permlane-swap hazard the compiler would have covered)
  float a = z, c = z;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(c));     // a: rows (0,0,2,2), c: rows (1,1,3,3)
  z = a + c;
  a = z, c = z;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));     // a: halves (lo,lo), c: (hi,hi)
  return a + c;
}

// hipGetLastError() is sticky on ROCm 7 (it reports the last *error* of any earlier runtime call
// in this thread, e.g. a benign hipErrorNotReady from an event query made by the caller's
// framework), so the state is cleared right before the launch and read right after it.
#define SFM_LAUNCH(kernel, grid, block, shmem, stream, ...)                    \
  do {                                                                         \
    (void)hipGetLastError();                                                   \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);       \
    if (hipGetLastError() != hipSuccess) return SFM_ERR_LAUNCH;                \
  } while (0)
#define SFM_CHECK_LAUNCH() do { } while (0)

// reduce.hip: the ordered second pass of the split reductions (deterministic training step).  ws [S][rows][cols] compact.
// (ws is used as scratch by the multi-level fold: its contents are destroyed)
int sfm_fold_partials(float* ws, float* out, long long rows, int cols, long long ldo, int S, int accumulate, void* stream);
int sfm_fold_partials2(float* ws, float* out, float* out2, int cols1, int cols, int S, int accumulate, void* stream);
int sfm_fold_partials_f64(double* ws, double* out, long long rows, int cols, long long ldo, int S, int accumulate, void* stream);
