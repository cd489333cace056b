// Forward of the training objective of training/conformer_pipeline.py (A21):
//   si_snr_loss :52-71, MultiResolutionSTFTLoss :74-108, L1 magnitude :562-564, total :570.
// The STFTs run on framed_gemm_f32; these kernels are the reductions (fp64 accumulation through
// wave shuffles + one f64 atomic per workgroup) and the scalar finalisation on device.
#include "sfm_common.h"

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// per utterance b: S[b] = { sum e, sum t, sum e^2, sum t^2, sum e*t }   (accumulated with atomics: zero S first)
__global__ __launch_bounds__(256) void wave_moments_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                           double* __restrict__ S, int L, double* __restrict__ ws) {
  __shared__ double red[4][5];
  const int b = blockIdx.y;
  const float* e = est + (long long)b * L;
  const float* t = tgt + (long long)b * L;
  double s[5] = {0, 0, 0, 0, 0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
    const double a = e[i], c = t[i];
    s[0] += a; s[1] += c; s[2] += a * a; s[3] += c * c; s[4] += a * c;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    s[k] = wave_sum_d(s[k]);
    if (lane == 0) red[wave][k] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (ws) ws[((long long)blockIdx.x * gridDim.y + b) * 5 + threadIdx.x] = v;       // partial [block][B][5], folded in block order
    else atomicAdd(&S[b * 5 + threadIdx.x], v);
  }
}

// spectra of prediction (pr, pi) and target (tr, ti), n bins each:
// S = { sum (|T|-|P|)^2, sum |T|^2, sum |log(|P|+1e-8) - log(|T|+1e-8)|, sum |sqrt(P^2+1e-8) - sqrt(T^2+1e-8)| }
__global__ __launch_bounds__(256) void spec_sums_kernel(const float* __restrict__ pr, const float* __restrict__ pi,
                                                        const float* __restrict__ tr, const float* __restrict__ ti,
                                                        double* __restrict__ S, long long n, double* __restrict__ ws) {
  __shared__ double red[4][4];
  double s[4] = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float a = pr[i], b = pi[i], c = tr[i], d = ti[i];
    const float p2 = a * a + b * b, t2 = c * c + d * d;
    const float pm = sqrtf(p2), tm = sqrtf(t2);
    const float df = tm - pm;
    s[0] += (double)df * df;
    s[1] += (double)t2;
    s[2] += (double)fabsf(logf(pm + 1e-8f) - logf(tm + 1e-8f));
    s[3] += (double)fabsf(sqrtf(p2 + 1e-8f) - sqrtf(t2 + 1e-8f));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s[k] = wave_sum_d(s[k]);
    if (lane == 0) red[wave][k] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (ws) ws[(long long)blockIdx.x * 4 + threadIdx.x] = v;
    else atomicAdd(&S[threadIdx.x], v);
  }
}

// out[0] = total, out[1] = neg SI-SNR, out[2] = L1 magnitude, out[3] = MR-STFT
// wave moments Sw [B][5]; framing spectra sums Sm [4] (n_mag bins, uses element 3); multi-res sums Sr [R][4] with counts nr[R]
__global__ void enhancer_loss_finalize_kernel(const double* __restrict__ Sw, const double* __restrict__ Sm,
                                              const double* __restrict__ Sr, const long long* __restrict__ nr, int B,
                                              int L, long long n_mag, int R, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double acc = 0.0;
  for (int b = 0; b < B; ++b) {
    const double se = Sw[b * 5 + 0], st = Sw[b * 5 + 1], see = Sw[b * 5 + 2], stt = Sw[b * 5 + 3], set = Sw[b * 5 + 4];
    const double me = se / L, mt = st / L;
    const double Et = stt - L * mt * mt;                     // sum (t - mean t)^2
    const double Ee = see - L * me * me;
    const double dot = set - L * me * mt;
    const double s_energy = Et + 1e-8;
    const double k = dot / s_energy;                         // s_target = k * t'
    const double star = k * k * Et;
    const double noise = Ee - 2.0 * k * dot + k * k * Et;
    acc += 10.0 * log10(star / (noise + 1e-8) + 1e-8);
  }
  const double neg_sisnr = -acc / B;
  const double l1mag = Sm[3] / (double)n_mag;
  double mr = 0.0;
  for (int r = 0; r < R; ++r) {
    const double sc = sqrt(Sr[r * 4 + 0]) / (sqrt(Sr[r * 4 + 1]) + 1e-8);
    const double lm = Sr[r * 4 + 2] / (double)nr[r];
    mr += sc + lm;
  }
  mr /= (double)(R > 0 ? R : 1);
  out[0] = (float)(neg_sisnr + 0.5 * l1mag + mr);
  out[1] = (float)neg_sisnr;
  out[2] = (float)l1mag;
  out[3] = (float)mr;
}

// ws (optional; >= 64 * B * 5 doubles for sfm_wave_moments, >= 2048 * 4 doubles for sfm_spec_sums): one partial per workgroup,
// folded in workgroup order by reduce.hip - the sums (and with them the loss value and its gradient) are then bit-reproducible;
// NULL = f64 atomics
extern "C" int sfm_wave_moments(const float* est, const float* tgt, double* S, int B, int L, double* ws, void* stream) {
  if (!est || !tgt || !S) return SFM_ERR_ARG;
  if (B <= 0 || L <= 0) return SFM_ERR_SHAPE;
  int nb = (L + 255) / 256;
  if (nb > 64) nb = 64;
  SFM_LAUNCH(wave_moments_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, est, tgt, S, L, ws);
  return ws ? sfm_fold_partials_f64(ws, S, B, 5, 5, nb, 1, stream) : SFM_OK;
}

extern "C" int sfm_spec_sums(const float* pr, const float* pi, const float* tr, const float* ti, double* S, long long n,
                             double* ws, void* stream) {
  if (!pr || !pi || !tr || !ti || !S) return SFM_ERR_ARG;
  if (n <= 0) return SFM_ERR_SHAPE;
  long long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  SFM_LAUNCH(spec_sums_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pr, pi, tr, ti, S, n, ws);
  return ws ? sfm_fold_partials_f64(ws, S, 1, 4, 4, (int)nb, 1, stream) : SFM_OK;
}

extern "C" int sfm_enhancer_loss_finalize(const double* Sw, const double* Sm, const double* Sr, const long long* nr, int B,
                                          int L, long long n_mag, int R, float* out, void* stream) {
  if (!Sw || !Sm || !Sr || !nr || !out) return SFM_ERR_ARG;
  if (B <= 0 || L <= 0 || n_mag <= 0 || R < 0) return SFM_ERR_SHAPE;
  SFM_LAUNCH(enhancer_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, Sw, Sm, Sr, nr, B, L, n_mag, R, out);
  return SFM_OK;
}
