// Backward of the training objective (training/conformer_pipeline.py:52-108, 539-572) and of the bounded polar
// mask (training/conformer_pipeline.py:283-295).  All kernels are streaming (HBM-bound) passes; the reductions they
// need were already produced by the forward (wave_moments / spec_sums in losses.hip) and are read from device
// memory, so the whole loss backward enqueues without a host synchronisation.
//   d loss / d est      (SI-SNR)            : sisnr_bwd          closed form from the per-utterance moments
//   d loss / d |STFT|   (spectral conv. + log-magnitude, or L1 magnitude) -> d (real, imag) : spec_loss_bwd
//   adjoint of the reflect-padded framing of torch.stft (overlap-add with the reflected images folded in) : stft_adjoint_ola
//   polar mask + complex multiply backward  : polar_mask_bwd
// The two matrix products of each STFT adjoint run on framed_gemm_f32 with the transposed DFT operands.
#include "sfm_common.h"

// dwave[b, s] = scale * d(neg SI-SNR mean over B)/d est[b, s]
//   e' = est - mean, t' = tgt - mean, dot = <e', t'>, Et = <t', t'>, se = Et + 1e-8, k = dot / se,
//   star = k^2 Et, noise = |e' - k t'|^2, q = noise + 1e-8, ratio = star / q + 1e-8,
//   d ln(ratio)/d e' = (1/ratio) * [ (2 k Et / se) / q * t'  -  star / q^2 * ( 2 e' - (2 k + 2 (dot - k Et)/se) t' ) ]
// (both t' and e' have zero mean, so the mean-removal Jacobian is the identity on this gradient)
__global__ __launch_bounds__(256) void sisnr_bwd_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                        const double* __restrict__ Sw, float* __restrict__ dwave, int B,
                                                        int L, float scale) {
  const int b = blockIdx.y;
  const double se_ = Sw[b * 5 + 0], st_ = Sw[b * 5 + 1], see = Sw[b * 5 + 2], stt = Sw[b * 5 + 3], set = Sw[b * 5 + 4];
  const double me = se_ / L, mt = st_ / L;
  const double Et = stt - L * mt * mt, Ee = see - L * me * me, dot = set - L * me * mt;
  const double se = Et + 1e-8;
  const double k = dot / se;
  const double star = k * k * Et;
  const double noise = Ee - 2.0 * k * dot + k * k * Et;
  const double q = noise + 1e-8;
  const double ratio = star / q + 1e-8;
  const double pre = -(10.0 / 2.302585092994046) / (double)B / ratio * (double)scale;
  const double ct = pre * ((2.0 * k * Et / se) / q + star / (q * q) * (2.0 * k + 2.0 * (dot - k * Et) / se));
  const double ce = pre * (-2.0 * star / (q * q));
  const float fct = (float)ct, fce = (float)ce, fme = (float)me, fmt = (float)mt;
  const float* e = est + (long long)b * L;
  const float* t = tgt + (long long)b * L;
  float* o = dwave + (long long)b * L;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) o[i] = fct * (t[i] - fmt) + fce * (e[i] - fme);
}

// mode 0: loss = scale * ( sqrt(S0) / (sqrt(S1) + 1e-8)  +  S2 / n )        S = spec_sums of this resolution
// mode 1: loss = scale * S3 / n                                              (L1 of sqrt(.^2 + 1e-8) magnitudes)
// writes (or accumulates) d loss/d pr, d loss/d pi at dr/di[m * ld + f]
__global__ __launch_bounds__(256) void spec_loss_bwd_kernel(const float* __restrict__ pr, const float* __restrict__ pi,
                                                            const float* __restrict__ tr, const float* __restrict__ ti,
                                                            const double* __restrict__ S, float* __restrict__ dr,
                                                            float* __restrict__ di, long long n, int F, long long ld,
                                                            int mode, int accumulate, float scale) {
  float c_sc = 0.f, c_n = scale / (float)n;
  if (mode == 0) {
    const double nrm = sqrt(S[0]), tn = sqrt(S[1]) + 1e-8;
    c_sc = (nrm > 0.0) ? (float)((double)scale / (nrm * tn)) : 0.f;
  }
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const float a = pr[e], b = pi[e], c = tr[e], d = ti[e];
    const float p2 = a * a + b * b, t2 = c * c + d * d;
    float ga, gb;
    if (mode == 0) {
      const float pm = sqrtf(p2), tm = sqrtf(t2);
      const float dl = logf(pm + 1e-8f) - logf(tm + 1e-8f);
      const float sg = (dl > 0.f) ? 1.f : ((dl < 0.f) ? -1.f : 0.f);
      const float gP = c_sc * (pm - tm) + c_n * sg / (pm + 1e-8f);
      const float inv = (pm > 0.f) ? 1.0f / pm : 0.f;          // torch.abs of a complex zero has zero gradient
      ga = gP * a * inv;
      gb = gP * b * inv;
    } else {
      const float pm = sqrtf(p2 + 1e-8f), tm = sqrtf(t2 + 1e-8f);
      const float df = pm - tm;
      const float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
      const float gP = c_n * sg / pm;
      ga = gP * a;
      gb = gP * b;
    }
    const long long m = e / F;
    const int f = (int)(e - m * F);
    if (accumulate) {
      dr[m * ld + f] += ga;
      di[m * ld + f] += gb;
    } else {
      dr[m * ld + f] = ga;
      di[m * ld + f] = gb;
    }
  }
}

// Adjoint of "reflect-pad by n_fft/2, cut frames of `win` samples every `hop`" (torch.stft center=True with the window
// support offset woff = (n_fft - win)/2):  frames [B, T, win] (gradient w.r.t. the windowed frame samples; the window
// is already folded into the GEMM operand) -> dwave [B, L].
//   ola(q) = sum_t frames[t][q - woff - t*hop]              q = index in the padded signal
//   dwave[s] (+)= ola(pad + s) + [1 <= s <= pad] ola(pad - s) + [L-1-pad <= s <= L-2] ola(pad + 2(L-1) - s)
// post (optional, [L]): the result is multiplied by post[s] (used to apply 1/envelope of the iSTFT once at the end).
__device__ __forceinline__ float ola_at(const float* __restrict__ fr, int q, int Tn, int hop, int win, int woff) {
  const int p = q - woff;
  if (p < 0) return 0.f;
  int t = p / hop;
  if (t > Tn - 1) t = Tn - 1;
  float acc = 0.f;
  for (; t >= 0; --t) {
    const int n = p - t * hop;
    if (n >= win) break;
    acc += fr[(long long)t * win + n];
  }
  return acc;
}

__global__ __launch_bounds__(256) void stft_adjoint_ola_kernel(const float* __restrict__ frames, float* __restrict__ dwave,
                                                               const float* __restrict__ post, int Tn, int L, int n_fft,
                                                               int hop, int win, int accumulate) {
  const int b = blockIdx.y;
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= L) return;
  const int pad = n_fft / 2, woff = (n_fft - win) / 2;
  const float* fr = frames + (long long)b * Tn * win;
  float v = ola_at(fr, pad + s, Tn, hop, win, woff);
  if (s >= 1 && s <= pad) v += ola_at(fr, pad - s, Tn, hop, win, woff);
  if (s >= L - 1 - pad && s <= L - 2) v += ola_at(fr, pad + 2 * (L - 1) - s, Tn, hop, win, woff);
  float* o = dwave + (long long)b * L + s;
  if (accumulate) v += *o;
  if (post) v *= post[s];
  *o = v;
}

// backward of  mg = sigmoid(a), ph = tanh(p) * phase_scale, mask = mg (cos ph, sin ph), enh = mask (x) noisy
// dlog[m, f] = d/da, dlog[m, F + f] = d/dp      (row stride ld_d; the columns >= 2F are left untouched)
// mag_bias (optional, [M / rows_per_batch, F]): a = lm + bias as in the forward kernel; nr == NULL: noisy = 1 + 0j (the mask
// itself is the output, agents/msa.py:166-172)
__global__ __launch_bounds__(256) void polar_mask_bwd_kernel(const float* __restrict__ lm, const float* __restrict__ lp,
                                                             const float* __restrict__ mag_bias, const float* __restrict__ nr,
                                                             const float* __restrict__ ni, const float* __restrict__ der,
                                                             const float* __restrict__ dei, float* __restrict__ dlog, int F,
                                                             long long total, long long rows_per_batch, float phase_scale,
                                                             long long ld_l, long long ld_d) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long m = e / F;
    const int f = (int)(e - m * F);
    float a = lm[m * ld_l + f];
    if (mag_bias) a += mag_bias[(m / rows_per_batch) * F + f];
    const float mg = 1.0f / (1.0f + expf(-a));
    const float th = tanhf(lp[m * ld_l + f]);
    const float ph = th * phase_scale;
    const float c = cosf(ph), s = sinf(ph);
    const float r = nr ? nr[e] : 1.0f, i = nr ? ni[e] : 0.0f, gr = der[e], gi = dei[e];
    const float dxr = gr * r + gi * i;
    const float dxi = gi * r - gr * i;
    const float dmg = dxr * c + dxi * s;
    const float dph = mg * (dxi * c - dxr * s);
    dlog[m * ld_d + f] = dmg * mg * (1.0f - mg);
    dlog[m * ld_d + F + f] = dph * phase_scale * (1.0f - th * th);
  }
}

extern "C" int sfm_sisnr_bwd(const float* est, const float* tgt, const double* Sw, float* dwave, int B, int L, float scale,
                             void* stream) {
  if (!est || !tgt || !Sw || !dwave) return SFM_ERR_ARG;
  if (B <= 0 || L <= 0) return SFM_ERR_SHAPE;
  int nb = (L + 255) / 256;
  if (nb > 256) nb = 256;
  SFM_LAUNCH(sisnr_bwd_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, est, tgt, Sw, dwave, B, L, scale);
  return SFM_OK;
}

extern "C" int sfm_spec_loss_bwd(const float* pr, const float* pi, const float* tr, const float* ti, const double* S,
                                 float* dr, float* di, long long n, int F, long long ld, int mode, int accumulate,
                                 float scale, void* stream) {
  if (!pr || !pi || !tr || !ti || !S || !dr || !di) return SFM_ERR_ARG;
  if (n <= 0 || F <= 0 || ld < F || (mode != 0 && mode != 1)) return SFM_ERR_SHAPE;
  long long nb = (n + 255) / 256;
  if (nb > 16384) nb = 16384;
  SFM_LAUNCH(spec_loss_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pr, pi, tr, ti, S, dr, di, n, F, ld,
             mode, accumulate, scale);
  return SFM_OK;
}

extern "C" int sfm_stft_adjoint_ola(const float* frames, float* dwave, const float* post, int B, int T, int L, int n_fft,
                                    int hop, int win, int accumulate, void* stream) {
  if (!frames || !dwave) return SFM_ERR_ARG;
  if (B <= 0 || T <= 0 || L <= n_fft / 2 || hop <= 0 || win <= 0 || win > n_fft) return SFM_ERR_SHAPE;
  SFM_LAUNCH(stft_adjoint_ola_kernel, dim3((L + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, frames, dwave, post, T, L,
             n_fft, hop, win, accumulate);
  return SFM_OK;
}

extern "C" int sfm_polar_mask_bwd(const float* lm, const float* lp, const float* mag_bias, const float* nr, const float* ni,
                                  const float* der, const float* dei, float* dlog, long long M, long long rows_per_batch, int F,
                                  float phase_scale, long long ld_logits, long long ld_dlog, void* stream) {
  if (!lm || !lp || !der || !dei || !dlog || ((nr == nullptr) != (ni == nullptr))) return SFM_ERR_ARG;
  if (M <= 0 || F <= 0 || ld_dlog < 2 * F || rows_per_batch <= 0 || M % rows_per_batch) return SFM_ERR_SHAPE;
  const long long total = M * F;
  long long nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  SFM_LAUNCH(polar_mask_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lm, lp, mag_bias, nr, ni, der, dei,
             dlog, F, total, rows_per_batch, phase_scale, ld_logits, ld_dlog);
  return SFM_OK;
}
