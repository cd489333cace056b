// Shared by gemm16v2.hip and conv16p.hip: the launch parameters of the 16-bit GEMM / conv kernels, the epilogue codes and
// the strip epilogue (bias / activation / GLU / residual / dropout / GroupNorm partials, 16-byte row stores).
#pragma once
#include "sfm_common.h"

#define EPI_NONE 0
#define EPI_SWISH 1
#define EPI_GELU 2
#define EPI_RESID 3
#define EPI_GLU 4
#define EPI_SIGMOID 5
#define EPI_TANH_SCALE 6
#define EPI_SIGMA 7
#define EPI_CPEA 8
#define EPI_SWISH_DUAL 9     // training forward of an FFN's first Linear: out = keep * swish(z), out2 = d = keep * swish'(z) (16-bit)
#define EPI_SWISH_BWD 10     // training backward: out = v * aux, aux = the saved derivative factor d

struct Gemm2Params {
  const u16* A;
  const u16* W;
  const float* bias;
  void* out;
  const float* resid;
  float* gn_partial;
  long long a_batch_stride, o_batch_stride, r_batch_stride;
  int B, Lout, Lin, Cin, lda, stride, pad, cin_shift;
  int K, Kpad, N, Npad, ldo, ldr;
  float alpha;
  int epi, out_f32, gn_group, nsplit;
  int nMt, nNt, a_records, w_records, vec_ok, gn_slots;
  float p_drop;                                   // EPI_RESID: out = resid + alpha * keep(m*N + n) * v; EPI_SWISH_*: hidden dropout
  unsigned int seed;
  const u16* aux;                                 // EPI_SWISH_BWD: saved derivative factor keep * swish'(z), layout of `out`
  u16* out2;                                      // EPI_SWISH_DUAL: second output (that derivative factor), layout of `out`
};

// 16-bit results are written in the operands' format T (out_f32 == 0) or in the OTHER 16-bit format (out_f32 == 2: a stage
// boundary of the precision policy, e.g. fp16 projections feeding a bf16 attention core); out_f32 == 1 is fp32.
template <class T>
__device__ __forceinline__ uint32_t pack2_out(float lo, float hi, bool other) {
  if (T::id == SFM_DT_BF16) return other ? F16::pack(lo, hi) : BF16::pack(lo, hi);
  return other ? BF16::pack(lo, hi) : F16::pack(lo, hi);
}
template <class T>
__device__ __forceinline__ u16 from_f32_out(float v, bool other) {
  if (T::id == SFM_DT_BF16) return other ? F16::from_f32(v) : BF16::from_f32(v);
  return other ? BF16::from_f32(v) : F16::from_f32(v);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// The wait in front of a ring barrier: this wave's LDS-DMA pieces of the stage to consume have landed (counted vmcnt) AND its
// own ds_reads of the stage it read last have RETURNED (lgkmcnt(0)).  The second half is what makes the refill right after the
// barrier safe (WAR): the compiler sinks the last MFMAs of a k-tile below the next barrier and leaves their operand reads in
// flight across it; without the lgkmcnt a fast wave's refill of that slot (L2-hit latency: a few hundred cycles) could land
// before a slow wave's reads were served - rare wrong 64 x 32 accumulator blocks (found in round 3 by
// tools/pa_determinism_probe.py: 3-7 passes of 300 at B 64 in conv16p; cdna guide: "restage ... 1 phase after when an lgkmcnt
// before the reading phase's first barrier retired those reads").
template <int N>
__device__ __forceinline__ void wait_ring() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;


// Epilogue of one wave tile (64 rows x WN columns, accumulators acc[2][NJ]) in 8 passes of 8 rows through the wave's private
// LDS strip `img` (8 x (WN + 4) floats): bias / activation / GLU / residual / dropout / GroupNorm partials, 16-byte row stores.
// colb = first packed column of the wave tile, row_base = its first output row (within batch entry b).
template <class T, int NJ>
__device__ __forceinline__ void gemm16_epilogue_strips(const Gemm2Params& p, f32x16 (&acc)[2][NJ], float* img, int lane, int b,
                                                       int colb, int row_base) {
  constexpr int WN = NJ * 32, IMG_LD = WN + 4;
  const int l31 = lane & 31, hl = lane >> 5;
  // ------------------------------ epilogue: 8 passes of 8 rows through the wave's LDS strip ------------------------------
    const bool glu = (p.epi == EPI_GLU);
  const int ecols = glu ? 32 : WN;
  const int cpr = ecols >> 3;
  const int c8 = (lane % cpr) * 8, rsub = lane / cpr;
  const bool lane_on = rsub < 8;
  
  const int ncol0 = glu ? ((colb >> 1) + c8) : (colb + c8);
  const long long obase = (long long)b * p.o_batch_stride;
  float gsum = 0.f, gsq = 0.f;
  // The 8 passes are a RUNTIME loop: only the strip write (which needs compile-time accumulator indices) is expanded 8 times,
  // behind a wave-uniform switch; the heavy part - activations, residual, dropout, stores - exists once.  Fully unrolled,
  // this epilogue was 28 k instructions per kernel and the instruction fetch of its sparse paths, not the arithmetic,
  // bounded the short-K GEMMs of the path.
#pragma unroll 1
  for (int pass = 0; pass < 8; ++pass) {
    {
      __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the previous pass has been read out
      __builtin_amdgcn_wave_barrier();
#define SFM_STRIP_WRITE(I_, Q_)                                                                                      \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                     \
  _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) img[(hl * 4 + rr) * IMG_LD + j * 32 + l31] = acc[I_][j][4 * Q_ + rr];
      switch (pass) {
        case 0: { SFM_STRIP_WRITE(0, 0) } break;
        case 1: { SFM_STRIP_WRITE(0, 1) } break;
        case 2: { SFM_STRIP_WRITE(0, 2) } break;
        case 3: { SFM_STRIP_WRITE(0, 3) } break;
        case 4: { SFM_STRIP_WRITE(1, 0) } break;
        case 5: { SFM_STRIP_WRITE(1, 1) } break;
        case 6: { SFM_STRIP_WRITE(1, 2) } break;
        default: { SFM_STRIP_WRITE(1, 3) } break;
      }
#undef SFM_STRIP_WRITE
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_wave_barrier();
      float bia[8], big[8];                            // (re-read per pass from L1: keeps 16 registers out of the loop)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bia[e] = p.bias ? p.bias[colb + c8 + e] : 0.f;
        big[e] = (glu && p.bias) ? p.bias[colb + 32 + c8 + e] : 0.f;
      }
      const int row = pass * 8 + rsub;                 // row inside the wave tile
      const int m = row_base + row;
      const bool mok = lane_on && m < p.Lout;
      float v[8];
      {
        const int rs = lane_on ? rsub : 0;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(&img[rs * IMG_LD + c8]);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(&img[rs * IMG_LD + c8 + 4]);
        v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3];
        v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bia[e];
        if (glu) {
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(&img[rs * IMG_LD + 32 + c8]);
          const f32x4 g1 = *reinterpret_cast<const f32x4*>(&img[rs * IMG_LD + 32 + c8 + 4]);
          const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= sigmoid_f(g[e] + big[e]);
        }
      }
      if (p.gn_partial && mok) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (ncol0 + e < p.N) { gsum += v[e]; gsq = __builtin_fmaf(v[e], v[e], gsq); }   // (fmaf: see profiles/README.md, "the lost sums of squares")
      }
      switch (p.epi) {
        case EPI_SWISH:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = swish_f(v[e]);
          break;
        case EPI_GELU:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
          break;
        case EPI_SIGMOID:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = sigmoid_f(v[e]);
          break;
        case EPI_TANH_SCALE:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = p.alpha * tanhf(v[e]);
          break;
        case EPI_SIGMA:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = expf(0.5f * fminf(fmaxf(v[e], -10.f), 10.f));
          break;
        case EPI_CPEA:
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (ncol0 + e < p.nsplit) ? sigmoid_f(v[e]) : p.alpha * tanhf(v[e]);
          break;
        default: break;
      }
      if (mok && (p.epi == EPI_SWISH_DUAL || p.epi == EPI_SWISH_BWD)) {
        // fused Swish of the FFN (training): vector path only (the launcher guarantees N % 8 == 0, aligned rows, 16-bit out)
        const long long orow = obase + (long long)m * p.ldo + ncol0;
        if (ncol0 + 8 <= p.N) {
          if (p.epi == EPI_SWISH_DUAL) {
            // forward: u = keep * swish(z) and, for the backward, the derivative factor d = keep * swish'(z) (NOT z itself:
            // the backward's epilogue is then a single multiply instead of exp + rcp + the dropout hash per element)
            const float ik = (p.p_drop > 0.f) ? 1.0f / (1.0f - p.p_drop) : 1.0f;
            const unsigned long long e0 = ((unsigned long long)b * p.Lout + m) * p.N + ncol0;
            float kp[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (p.p_drop > 0.f) sfm_keep_scale8(p.seed, e0, p.p_drop, ik, kp);
            float dd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float sg = sigmoid_f(v[e]);
              dd[e] = kp[e] * sg * (1.0f + v[e] * (1.0f - sg));
              v[e] = v[e] * sg * kp[e];
            }
            u32x4 pd;
#pragma unroll
            for (int e = 0; e < 4; ++e) pd[e] = pack2<T>(dd[2 * e], dd[2 * e + 1]);
            *reinterpret_cast<u32x4*>(p.out2 + orow) = pd;
          } else {
            const u32x4 pd = *reinterpret_cast<const u32x4*>(p.aux + orow);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] *= T::to_f32((u16)(pd[e] & 0xffffu));
              v[2 * e + 1] *= T::to_f32((u16)(pd[e] >> 16));
            }
          }
          u32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
          *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.out) + orow) = pk;
        }
      } else if (mok) {
        const long long orow = obase + (long long)m * p.ldo + ncol0;
        if (p.vec_ok && ncol0 + 8 <= p.N) {
          if (p.epi == EPI_RESID) {
            if (p.p_drop > 0.f) {
              const float ik = 1.0f / (1.0f - p.p_drop);
              const unsigned long long e0 = ((unsigned long long)b * p.Lout + m) * p.N + ncol0;
float kp[8];
              sfm_keep_scale8(p.seed, e0, p.p_drop, ik, kp);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= kp[e];
            }
            const float* rp = p.resid + (long long)b * p.r_batch_stride + (long long)m * p.ldr + ncol0;
            const f32x4 r0v = *reinterpret_cast<const f32x4*>(rp);
            const f32x4 r1v = *reinterpret_cast<const f32x4*>(rp + 4);
            v[0] = r0v[0] + p.alpha * v[0]; v[1] = r0v[1] + p.alpha * v[1];
            v[2] = r0v[2] + p.alpha * v[2]; v[3] = r0v[3] + p.alpha * v[3];
            v[4] = r1v[0] + p.alpha * v[4]; v[5] = r1v[1] + p.alpha * v[5];
            v[6] = r1v[2] + p.alpha * v[6]; v[7] = r1v[3] + p.alpha * v[7];
          }
          if (p.out_f32 == 1) {
            float* op = reinterpret_cast<float*>(p.out) + orow;
            f32x4 a = {v[0], v[1], v[2], v[3]}, c = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4*>(op) = a;
            *reinterpret_cast<f32x4*>(op + 4) = c;
          } else {
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = pack2_out<T>(v[2 * e], v[2 * e + 1], p.out_f32 == 2);
            *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.out) + orow) = pk;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (ncol0 + e < p.N) {
              float y = v[e];
              if (p.epi == EPI_RESID) {
                if (p.p_drop > 0.f)
                  y *= sfm_keep_scale(p.seed, ((unsigned long long)b * p.Lout + m) * p.N + ncol0 + e, p.p_drop, 1.0f / (1.0f - p.p_drop));
                y = p.resid[(long long)b * p.r_batch_stride + (long long)m * p.ldr + ncol0 + e] + p.alpha * y;
              }
              if (p.out_f32 == 1) reinterpret_cast<float*>(p.out)[orow + e] = y;
              else reinterpret_cast<u16*>(p.out)[orow + e] = from_f32_out<T>(y, p.out_f32 == 2);
            }
          }
        }
      }
    }
  }
  if (p.gn_partial) {
    for (int o = cpr; o < 64; o <<= 1) {               // lanes with the same column chunk hold different rows
      gsum += __shfl_xor(gsum, o, 64);
      gsq += __shfl_xor(gsq, o, 64);
    }
    const int cpg = p.gn_group >> 3;
    for (int o = 1; o < cpg; o <<= 1) {
      gsum += __shfl_xor(gsum, o, 64);
      gsq += __shfl_xor(gsq, o, 64);
    }
    if (lane < cpr && (lane % cpg) == 0 && ncol0 < p.N && (row_base >> 6) < p.gn_slots) {
      const int ngroups = p.N / p.gn_group;
      const long long sl = ((long long)b * p.gn_slots + (row_base >> 6)) * ngroups + ncol0 / p.gn_group;
      p.gn_partial[sl * 2 + 0] = gsum;
      p.gn_partial[sl * 2 + 1] = gsq;
    }
  }

}

