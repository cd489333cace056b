// SincConv1d FIR filterbank on the 16-bit matrix cores with split operands
// (agents/perception.py:115-118: conv1d(1 -> C channels, K taps, stride 1, pad K/2)).
//
// x and h are each split into hi + lo 16-bit parts (x = xh + xl exactly to 2x the mantissa) and the
// product is xh*hh + xh*hl + xl*hh with fp32 accumulation: ~2^-15 (bf16) / ~2^-21 (fp16) relative
// error at 3 MFMA passes of the 16x faster 16-bit rate, instead of the fp32 MFMA of framed_gemm.
//
// Toeplitz trick: MFMA row r of a tile is output sample m0 + o + 8r (o = shift 0..7), so the A
// fragment x[m0 + 8r + k' .. +8] is a 16-byte ALIGNED run of the single LDS copy of the signal for
// every lane; the shift o is moved into the filters (W_o[k'] = h[k' - o]), which a prep kernel
// writes once per forward.  Workgroup (8 waves) = (shift o, chunk of samples, utterance): W_o (hi, lo)
// lives in LDS for the whole chunk; each wave owns 32 strided rows x all C=64 channels; the next
// sub-tile's samples are prefetched into registers under the MFMAs.
// Epilogue: * 2^-12 (filters are pre-scaled by 2^12 to stay out of fp16 subnormals), GroupNorm
// partial sums, LDS transpose, full 128-byte channels-last row stores.
#include "sfm_common.h"

#define FIR_C 64
#define FIR_KP 272          // taps (<= 265 incl. shift) padded to 17 k-steps of 16
#define FIR_WROW 280        // u16 elements per filter row in LDS (560 B: conflict-free b128 reads)
#define FIR_WAVES 8
#define FIR_SUB (FIR_WAVES * 256)   // samples per sub-tile
#define FIR_NSUB 4                 // sub-tiles per workgroup chunk
#define FIR_XPT ((FIR_SUB + FIR_KP + FIR_WAVES * 64 - 1) / (FIR_WAVES * 64))   // staged samples per thread
#define FIR_SPAN (FIR_SUB + FIR_KP)

// filt [C][K] fp32 -> wsh [8][2 (hi,lo)][C][FIR_KP] 16-bit, W_o[k'] = 2^12 * h[k' - o]
template <class T>
__global__ __launch_bounds__(256) void sinc_fir16_prep_kernel(const float* __restrict__ filt, u16* __restrict__ wsh, int K) {
  const int o = blockIdx.x, c = blockIdx.y;
  for (int kp = threadIdx.x; kp < FIR_KP; kp += 256) {
    const int k = kp - o;
    const float h = (k >= 0 && k < K) ? filt[c * K + k] * 4096.0f : 0.f;
    const u16 hi = T::from_f32(h);
    const u16 lo = T::from_f32(h - T::to_f32(hi));
    wsh[(((long long)o * 2 + 0) * FIR_C + c) * FIR_KP + kp] = hi;
    wsh[(((long long)o * 2 + 1) * FIR_C + c) * FIR_KP + kp] = lo;
  }
}

// PASSES = 3: hi x hi + hi x lo + lo x hi (fp32-class accuracy); PASSES = 1: hi x hi only - waveform and taps rounded ONCE to the
// operand format.  With fp16 operands and a 16-bit result that rounding (2^-11) is the size of the output's own, and the mask
// RMSE of the whole path does not move (2.127e-4 -> 2.125e-4, profiles/README.md round 3) at 0.62 x the time.
template <class T, int PASSES>
__global__ __launch_bounds__(FIR_WAVES * 64) void sinc_fir16_kernel(const float* __restrict__ sig, const u16* __restrict__ wsh,
                                                         void* __restrict__ out, float* __restrict__ gn_partial,
                                                         int L, int pad, int out_f32, int tiles_per_batch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  u16* Wh = reinterpret_cast<u16*>(dsm);                       // [C][FIR_WROW]
  u16* Wl = Wh + FIR_C * FIR_WROW;
  u16* Xh = Wl + FIR_C * FIR_WROW;                             // [FIR_SPAN]
  u16* Xl = Xh + FIR_SPAN;
  float* img = reinterpret_cast<float*>(Xl + FIR_SPAN);        // FIR_WAVES x [32][68] floats

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hl = lane >> 5;
  const int o = blockIdx.x, b = blockIdx.z;
  const int c0 = blockIdx.y * (FIR_NSUB * FIR_SUB);
  const float* xb = sig + (long long)b * L;

  // filters for this shift -> LDS (16-byte chunks)
  {
    const u16* src = wsh + (long long)o * 2 * FIR_C * FIR_KP;
    for (int e = tid; e < 2 * FIR_C * (FIR_KP / 8); e += FIR_WAVES * 64) {
      const int part = e / (FIR_C * (FIR_KP / 8));
      const int rem = e - part * (FIR_C * (FIR_KP / 8));
      const int c = rem / (FIR_KP / 8), k8 = (rem - c * (FIR_KP / 8)) * 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + ((long long)part * FIR_C + c) * FIR_KP + k8);
      *reinterpret_cast<u32x4*>((part ? Wl : Wh) + c * FIR_WROW + k8) = v;
    }
  }

  constexpr int NT = FIR_WAVES * 64;
  float xr[FIR_XPT];
  auto load_span = [&](int m0) {
#pragma unroll
    for (int q = 0; q < FIR_XPT; ++q) {
      const int i = tid + q * NT;
      const int sidx = m0 + i - pad;
      xr[q] = (i < FIR_SPAN && sidx >= 0 && sidx < L) ? xb[sidx] : 0.f;
    }
  };
  load_span(c0);
  for (int sub = 0; sub < FIR_NSUB; ++sub) {
    const int m0 = c0 + sub * FIR_SUB;                         // first sample of this sub-tile (before the shift)
    if (m0 >= L) break;
    __syncthreads();                                           // previous sub-tile's readers are done
#pragma unroll
    for (int q = 0; q < FIR_XPT; ++q) {
      const int i = tid + q * NT;
      if (i < FIR_SPAN) {
        const u16 hi = T::from_f32(xr[q]);
        Xh[i] = hi;
        Xl[i] = T::from_f32(xr[q] - T::to_f32(hi));
      }
    }
    __syncthreads();
    if (sub + 1 < FIR_NSUB && m0 + FIR_SUB < L) load_span(m0 + FIR_SUB);   // lands under the MFMAs below

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int xoff = wave * 256 + 8 * l31 + 8 * hl;            // + k0: 16-byte aligned for every lane
#pragma unroll
    for (int k0 = 0; k0 < FIR_KP; k0 += 16) {
      const u32x4 ah = *reinterpret_cast<const u32x4*>(Xh + xoff + k0);
      const u32x4 al = *reinterpret_cast<const u32x4*>(Xl + xoff + k0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int wo = (j * 32 + l31) * FIR_WROW + k0 + 8 * hl;
        const u32x4 bh = *reinterpret_cast<const u32x4*>(Wh + wo);
        const u32x4 bl = *reinterpret_cast<const u32x4*>(Wl + wo);
        acc[j] = T::mfma(ah, bh, acc[j]);
        if (PASSES == 3) {
          acc[j] = T::mfma(ah, bl, acc[j]);
          acc[j] = T::mfma(al, bh, acc[j]);
        }
      }
    }

    // ---- epilogue: scale back, per-wave image [32 rows][64 ch], stats, row stores ----
    float* im = img + wave * (32 * 68);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) im[mfma_row(r, lane) * 68 + j * 32 + l31] = acc[j][r] * (1.0f / 4096.0f);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int c8 = (lane & 7) * 8, rsub = lane >> 3;           // 8 channels per lane, 8 rows per pass
    float gsum = 0.f, gsq = 0.f;
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += 8) {
      const int row = r0 + rsub;
      const int m = m0 + wave * 256 + o + 8 * row;             // output sample of this image row
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&im[row * 68 + c8]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&im[row * 68 + c8 + 4]);
      if (m < L) {
        const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) { gsum += v[e]; gsq += v[e] * v[e]; }
        const long long off = ((long long)b * L + m) * FIR_C + c8;
        if (out_f32) {
          float* op = reinterpret_cast<float*>(out) + off;
          *reinterpret_cast<f32x4*>(op) = x0;
          *reinterpret_cast<f32x4*>(op + 4) = x1;
        } else {
          u32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
          *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(out) + off) = pk;
        }
      }
    }
    if (gn_partial) {                                          // GroupNorm(8, 64): one 8-channel chunk = one group
      for (int s = 8; s < 64; s <<= 1) {
        gsum += __shfl_xor(gsum, s, 64);
        gsq += __shfl_xor(gsq, s, 64);
      }
      if (lane < 8) {
        const long long tile = ((long long)blockIdx.y * FIR_NSUB + sub) * (8 * FIR_WAVES) + o * FIR_WAVES + wave;
        const long long slot = (((long long)b * tiles_per_batch + tile) * 8 + lane) * 2;
        gn_partial[slot] = gsum;
        gn_partial[slot + 1] = gsq;
      }
    }
  }
}

// wave [B, L] fp32, filt [64, K] fp32 (sfm_sinc_filters), wsh: workspace of 8*2*64*272 u16,
// out [B, L, 64] channels-last (16-bit or fp32), gn_partial [B][P][8][2] with P = sfm_sinc_fir16_tiles(L).
extern "C" int sfm_sinc_fir16_tiles(int L) {
  const int chunk = FIR_NSUB * FIR_SUB;
  const int nchunks = (L + chunk - 1) / chunk;
  return nchunks * FIR_NSUB * 8 * FIR_WAVES;                  // (chunks x sub-tiles) x 8 shifts x waves; ZERO-FILLED by the caller
}

template <class T, int PASSES>
static int sinc_fir16_go(const float* wave, const float* filt, void* wsh, void* out, float* gn_partial, int L, int K, int out_f32,
                         dim3 grid, int lds, int tpb, hipStream_t st) {
  static bool attr_set_dev[64] = {false};              // hipFuncSetAttribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SFM_ERR_LAUNCH;
  if (!attr_set_dev[dev]) {
    if (hipFuncSetAttribute((const void*)sinc_fir16_kernel<T, PASSES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return SFM_ERR_LAUNCH;
    attr_set_dev[dev] = true;
  }
  SFM_LAUNCH((sinc_fir16_prep_kernel<T>), dim3(8, FIR_C), dim3(256), 0, st, filt, (u16*)wsh, K);
  SFM_LAUNCH((sinc_fir16_kernel<T, PASSES>), grid, dim3(FIR_WAVES * 64), lds, st, wave, (const u16*)wsh, out, gn_partial, L, K / 2,
             out_f32, tpb);
  return SFM_OK;
}

// passes: 3 = split operands (hi x hi + hi x lo + lo x hi, fp32-class accuracy), 1 = hi x hi only (one rounding of the waveform
// and the taps to the operand format), 0 = auto: 1 for fp16 operands with a 16-bit result, 3 otherwise
extern "C" int sfm_sinc_fir16_ex(const float* wave, const float* filt, void* wsh, void* out, float* gn_partial, int B, int L, int C,
                                 int K, int out_f32, int dtype, int passes, void* stream) {
  if (!wave || !filt || !wsh || !out) return SFM_ERR_ARG;
  if (C != FIR_C || K < 1 || K + 7 > FIR_KP || (K & 1) == 0 || B <= 0 || L <= 0 || (passes != 0 && passes != 1 && passes != 3))
    return SFM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int chunk = FIR_NSUB * FIR_SUB;
  const int nchunks = (L + chunk - 1) / chunk;
  const int lds = (2 * FIR_C * FIR_WROW + 2 * FIR_SPAN) * 2 + FIR_WAVES * 32 * 68 * 4;
  const int tpb = nchunks * FIR_NSUB * 8 * FIR_WAVES;
  dim3 grid(8, nchunks, B);
  if (passes == 0) passes = (dtype == SFM_DT_F16 && out_f32 == 0) ? 1 : 3;
  if (dtype == SFM_DT_F16)
    return passes == 1 ? sinc_fir16_go<F16, 1>(wave, filt, wsh, out, gn_partial, L, K, out_f32, grid, lds, tpb, st)
                       : sinc_fir16_go<F16, 3>(wave, filt, wsh, out, gn_partial, L, K, out_f32, grid, lds, tpb, st);
  return passes == 1 ? sinc_fir16_go<BF16, 1>(wave, filt, wsh, out, gn_partial, L, K, out_f32, grid, lds, tpb, st)
                     : sinc_fir16_go<BF16, 3>(wave, filt, wsh, out, gn_partial, L, K, out_f32, grid, lds, tpb, st);
}

extern "C" int sfm_sinc_fir16(const float* wave, const float* filt, void* wsh, void* out, float* gn_partial, int B,
                              int L, int C, int K, int out_f32, int dtype, void* stream) {
  return sfm_sinc_fir16_ex(wave, filt, wsh, out, gn_partial, B, L, C, K, out_f32, dtype, 0, stream);
}
