// On-device quality metrics of the step right after the path (main.py:141-151 evaluates every utterance with
// evaluation/ssnr.py:26 and the fallback STOI of evaluation/stoi.py:53 in per-utterance numpy loops): batched here.
//   ssnr_frames : per 160-sample frame (hop 80) 10 log10(sum c^2 / sum (c - e)^2), clipped, silence skipped;
//                 per-utterance sum and count in fp64 (evaluation/ssnr.py:53-92)
//   stoi_frames : per analysis frame the normalised spectral correlation of evaluation/stoi.py:76-94 from the two
//                 magnitude spectra (the DFTs run on framed_gemm_f32 with the symmetric-Hann DFT operand)
#include "sfm_common.h"

__device__ __forceinline__ double wsum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// acc[b] = { sum of kept frame SNRs, number of kept frames }
__global__ __launch_bounds__(256) void ssnr_frames_kernel(const float* __restrict__ clean, const float* __restrict__ enh,
                                                          double* __restrict__ acc, int L, int nframes, int frame, int hop,
                                                          float upper, float lower) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const float* c = clean + (long long)b * L;
  const float* e = enh + (long long)b * L;
  double ssum = 0.0, scnt = 0.0;
  for (int n = blockIdx.x * 4 + wave; n < nframes; n += gridDim.x * 4) {
    const int start = n * hop;
    double sp = 0.0, ep = 0.0;
    for (int i = lane; i < frame; i += 64) {
      const double cv = c[start + i], d = cv - (double)e[start + i];
      sp += cv * cv;
      ep += d * d;
    }
    sp = wsum_d(sp);
    ep = wsum_d(ep);
    if (sp < 1e-10) continue;                                  // silence frame: skipped (evaluation/ssnr.py:72)
    double snr = (ep < 1e-10) ? (double)upper : 10.0 * log10(sp / ep);
    snr = fmin(fmax(snr, (double)lower), (double)upper);
    ssum += snr;
    scnt += 1.0;
  }
  if (lane == 0 && scnt > 0.0) {
    atomicAdd(&acc[2 * b + 0], ssum);
    atomicAdd(&acc[2 * b + 1], scnt);
  }
}

// spectra [B, nframes, F] (real, imag) of the RAW signals; sc[b], se[b] = the rms normalisation factors
// 1/(rms + 1e-10) of evaluation/stoi.py:65-66 (spectra are linear in the signal).  acc[b] += clip(corr, -1, 1).
__global__ __launch_bounds__(256) void stoi_frames_kernel(const float* __restrict__ cr, const float* __restrict__ ci,
                                                          const float* __restrict__ er, const float* __restrict__ ei,
                                                          const double* __restrict__ sc, const double* __restrict__ se,
                                                          double* __restrict__ acc, int nframes, int F) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const double kc = sc[b], ke = se[b];
  double total = 0.0;
  for (int n = blockIdx.x * 4 + wave; n < nframes; n += gridDim.x * 4) {
    const long long off = ((long long)b * nframes + n) * F;
    double A = 0.0, Bq = 0.0, C = 0.0;
    for (int f = lane; f < F; f += 64) {
      const double a0 = cr[off + f], a1 = ci[off + f], b0 = er[off + f], b1 = ei[off + f];
      const double cm = sqrt(a0 * a0 + a1 * a1) * kc, em = sqrt(b0 * b0 + b1 * b1) * ke;
      A += cm * cm;
      Bq += em * em;
      C += cm * em;
    }
    A = wsum_d(A);
    Bq = wsum_d(Bq);
    C = wsum_d(C);
    const double clean_energy = sqrt(A + 1e-10);
    const double k = clean_energy / (sqrt(Bq) + 1e-10);        // enh_norm = enh_spec * k
    double corr = (k * C) / (sqrt(A * (k * k * Bq)) + 1e-10);
    corr = fmin(fmax(corr, -1.0), 1.0);
    total += corr;
  }
  if (lane == 0) atomicAdd(&acc[b], total);
}

extern "C" int sfm_ssnr_frames(const float* clean, const float* enh, double* acc, int B, int L, int frame, int hop,
                               float upper, float lower, void* stream) {
  if (!clean || !enh || !acc) return SFM_ERR_ARG;
  if (B <= 0 || L <= 0 || frame <= 0 || hop <= 0) return SFM_ERR_SHAPE;
  const int nframes = (L - frame) / hop + 1;
  if (L < frame || nframes < 1) return SFM_OK;                 // fewer samples than one frame: acc stays {0, 0}
  int nb = (nframes + 3) / 4;
  if (nb > 256) nb = 256;
  SFM_LAUNCH(ssnr_frames_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, clean, enh, acc, L, nframes, frame, hop, upper,
             lower);
  return SFM_OK;
}

extern "C" int sfm_stoi_frames(const float* cr, const float* ci, const float* er, const float* ei, const double* sc,
                               const double* se, double* acc, int B, int nframes, int F, void* stream) {
  if (!cr || !ci || !er || !ei || !sc || !se || !acc) return SFM_ERR_ARG;
  if (B <= 0 || nframes <= 0 || F <= 0) return SFM_ERR_SHAPE;
  int nb = (nframes + 3) / 4;
  if (nb > 256) nb = 256;
  SFM_LAUNCH(stoi_frames_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, cr, ci, er, ei, sc, se, acc, nframes, F);
  return SFM_OK;
}
