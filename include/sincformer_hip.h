/* libsincformer_hip.so — C ABI of the MI355X (gfx950) hot path.
 *
 * The reference (pure Python / PyTorch) has no FFI layer: its operator boundary
 * for this path is the set of aten dispatch sites listed in SURVEY.md §2b
 * (K1..K17).  Each entry point below replaces the cited reference call site.
 * Conventions:
 *   - plain pointers to DEVICE memory + sizes; no torch types; `stream` is a
 *     hipStream_t passed as void* (0 = default stream).
 *   - enqueue-only: every call is asynchronous on `stream`, allocates nothing,
 *     keeps no global state (thread-safe per stream), never throws.
 *   - returns 0 (SFM_OK) or a negative error: -1 bad argument, -2 unsupported
 *     shape, -3 launch failure.
 *   - `dtype` selects the 16-bit MFMA operand / activation format:
 *       0 = bf16, 1 = fp16 (accumulation is always fp32).
 *   - "16-bit" tensors are raw uint16_t words of that format.
 */
#ifndef SINCFORMER_HIP_H
#define SINCFORMER_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define SFM_OK 0
#define SFM_ERR_ARG (-1)
#define SFM_ERR_SHAPE (-2)
#define SFM_ERR_LAUNCH (-3)
#define SFM_DT_BF16 0
#define SFM_DT_F16 1

/* epilogues of sfm_gemm16 */
#define SFM_EPI_NONE 0
#define SFM_EPI_SWISH 1      /* x*sigmoid(x)           models/conformer.py:45          */
#define SFM_EPI_GELU 2       /* erf GELU               agents/msa.py:45,64,69          */
#define SFM_EPI_RESID 3      /* resid + alpha*v        models/conformer.py:49,71,128   */
#define SFM_EPI_GLU 4        /* a*sigmoid(g)           models/conformer.py:114         */
#define SFM_EPI_SIGMOID 5
#define SFM_EPI_TANH_SCALE 6
#define SFM_EPI_SIGMA 7      /* exp(.5*clamp(v,-10,10)) agents/perception.py:249       */
#define SFM_EPI_CPEA 8       /* sigmoid | alpha*tanh   agents/cpea.py:102-105          */

int sfm_abi_version(void);

/* Dense layer / channels-last Conv1d as implicit GEMM on 16-bit MFMA.
 * Replaces nn.Linear / nn.Conv1d call sites: models/conformer.py:44,47,113,122,
 * 209,222; in_proj/out_proj of nn.MultiheadAttention (:69); agents/msa.py:143,
 * 154-160; agents/perception.py:195,198,203 (conv_blocks), :168 (downsample),
 * :175,179 (real/imag proj), :185,187 (uncertainty head); agents/cpea.py:56-76.
 *   A   [B, Lin, Cin] 16-bit (position stride lda >= Cin, batch stride a_batch_stride; elements)
 *   W   [Npad, Kpad] 16-bit, row n = output channel, K order = (tap, cin);
 *       zero padded; Kpad % 64 == 0, Npad % 64 == 0
 *   out row (b, l) = epi( sum_{tap,ci} A[b, l*stride-pad+tap, ci] W[n,(tap,ci)] + bias[n] )
 *   gn_partial (optional): per (batch, row-half-tile, group) {sum, sumsq} of the
 *       pre-activation outputs, [B][2*ceil(Lout/128)][N/gn_group][2] floats.
 *   out_f32: 0 = 16-bit result in the operands' format `dtype`, 1 = fp32, 2 = 16-bit in the OTHER format (bf16 <-> fp16:
 *       a stage boundary of the host's precision policy; one rounding of the fp32 accumulators).
 */
int sfm_gemm16(const void* A, const void* W, const float* bias, void* out, const float* resid,
               float* gn_partial, int B, int Lout, int Lin, int Cin, int lda, int ksize, int stride, int pad,
               long long a_batch_stride, int Kpad, int N, int Npad, int ldo, long long o_batch_stride,
               int ldr, long long r_batch_stride, float alpha, int epi, int out_f32, int gn_group,
               int nsplit, int dtype, void* stream);

/* Same contract with an explicit kernel variant (A/B measurements): 0 auto, 2 = 128-row tiles with a 2-stage LDS-DMA ring,
 * 6 = persistent form, 9 = 256-row wide tiles, 10 = 512 x 128 tiles (gemm16v2.hip).  sfm_gemm16 == variant 0. */
int sfm_gemm16_ex(const void* A, const void* W, const float* bias, void* out, const float* resid,
                  float* gn_partial, int B, int Lout, int Lin, int Cin, int lda, int ksize, int stride, int pad,
                  long long a_batch_stride, int Kpad, int N, int Npad, int ldo, long long o_batch_stride,
                  int ldr, long long r_batch_stride, float alpha, int epi, int out_f32, int gn_group,
                  int nsplit, int dtype, int variant, void* stream);
/* sfm_gemm16_ex + residual-branch dropout in the EPI_RESID epilogue (training forward of models/conformer.py:48,70,127):
 * out = resid + alpha * keep(seed, m*N + n)/(1 - p_drop) * (A W^T + bias), the same counter-based keep as sfm_ew_train. */
int sfm_gemm16_train(const void* A, const void* W, const float* bias, void* out, const float* resid,
                  float* gn_partial, int B, int Lout, int Lin, int Cin, int lda, int ksize, int stride, int pad,
                  long long a_batch_stride, int Kpad, int N, int Npad, int ldo, long long o_batch_stride,
                  int ldr, long long r_batch_stride, float alpha, int epi, int out_f32, int gn_group,
                  int nsplit, int dtype, int variant, float p_drop, unsigned int seed,
                     void* stream);
/* Linear (A [M, Cin] 16-bit, row stride lda) with the FeedForwardModule's Swish and hidden dropout fused into the epilogue
 * (training; models/conformer.py:44-46 and its backward).  backward == 0: z = A W^T + bias; out 16-bit = keep/(1-p) * swish(z),
 * out2 [M, N] 16-bit = d = keep/(1-p) * swish'(z), the derivative factor kept for the backward;  backward != 0: out 16-bit =
 * (A W^T) * aux with aux = the saved d.  Dropout counters m * N + n, the same as sfm_ew_train modes 0 / 1.  N % 8 == 0,
 * 16-byte aligned rows. */
int sfm_gemm16_swish(const void* A, const void* W, const float* bias, void* out, const void* aux, void* out2, int M, int Cin,
                     int lda, int Kpad, int N, int Npad, int ldo, int backward, float p_drop, unsigned int seed, int dtype,
                     void* stream);
/* Conv1d (channels-last) whose INPUT is normalised while the operand is staged: replaces the
 * GroupNorm -> GELU -> Conv1d chains of the PerceptionAgent (agents/perception.py:192-206 `_make_block`, :121-129
 * `_ResidualBlock.forward`, :167-171 `downsample`) without materialising the normalised activation:
 *     x = GELU(sc1[b,c] * x1 + sh1[b,c]  [+ sc2[b,c] * x2 + sh2[b,c]]),   out = Conv1d(x; W, bias, ksize, stride, pad)
 *   x1, x2 [B, Lin, Cin] 16-bit raw outputs of the producing conv(s); sc / sh [B, Cin] fp32 from sfm_gn_finalize
 *   W [N][ksize * Cin] 16-bit tap-major (sfm_gemm16's layout), bias [N]; out [B, Lout, N] (out_f32 as sfm_gemm16)
 *   gn_partial: as sfm_gemm16 ([B][2 ceil(Lout/128)][N/gn_group][2]) or NULL
 *   Ws, bias_s, out_s, gn_partial_s: optional Conv1d(Cin -> N, 1, stride 2) of a residual block's skip branch on the same
 *   x, computed from the same staged input (N = 128, ksize 7 only)
 * Shapes: Cin % 64 == 0; (ksize, stride, pad, N, #inputs) in {(7,2,3,128 + skip,1|2), (7,2,3,256,2), (3,1,1,128|256,1),
 * (5,2,2,256,2), (1,2,0,256,2)}; anything else returns SFM_ERR_SHAPE (-2). */
int sfm_conv16p(const void* x1, const float* sc1, const float* sh1, const void* x2, const float* sc2, const float* sh2,
                const void* W, const float* bias, void* out, float* gn_partial, const void* Ws, const float* bias_s,
                void* out_s, float* gn_partial_s, int B, int Lin, int Cin, int N, int ksize, int stride, int pad,
                int out_f32, int gn_group, int dtype, void* stream);

/* out[b,m,n] = bias[n] + sum_k sig[b, m*hop + k - padl] * Wt[k][n], exact fp32 on
 * v_mfma_f32_32x32x2_f32.  Replaces F.conv1d of SincConv1d (agents/perception.py:117),
 * torch.stft (training/conformer_pipeline.py:199), the irfft stage of torch.istft
 * (:210) and fp32 matmuls.  mode 0: zero outside [0,Ls); 1: reflect.
 * Columns >= nsplit go to out2 (when non-NULL) at column n - nsplit.
 * gn_partial layout: [B][4*ceil(M/128)][N/gn_group][2]. */
int sfm_framed_gemm_f32(const float* sig, const float* Wt, const float* bias, void* out, void* out2,
                        float* gn_partial, int B, int M, int Ls, long long sig_batch_stride, int hop,
                        int padl, int K, int Kpad, int N, int Npad, int nsplit, long long o_batch_stride,
                        long long ldm, long long ldn, int mode, int out_f32, int gn_group, int dtype,
                        void* stream);

/* softmax(Q K^T * scale) V per (batch, head); replaces the attention core of
 * nn.MultiheadAttention (models/conformer.py:69).  qkv [B,T,ldqkv] 16-bit with
 * q at column h*hd, k at koff+h*hd, v at voff+h*hd; out [B,T,ldo] 16-bit at h*hd.
 * scale <= 0 means Q is pre-multiplied by softmax_scale*log2(e) (folded into W_q/b_q). */
int sfm_attention_fwd(const void* qkv, void* out, int B, int T, int H, int hd, int ldqkv, int ldo,
                      int koff, int voff, long long qkv_batch_stride, long long o_batch_stride,
                      float scale, int dtype, void* stream);
/* Same, with the result written in `out_dtype` (SFM_DT_BF16 / SFM_DT_F16) when it differs from the operands' `dtype`
 * (precision policy: a bf16 attention core behind fp16 projections; head_dim 64 kernels only), and with the head_dim-64
 * kernel chosen per call by `variant` (A/B measurements; the library keeps no selection state): 0 = by shape (the pipelined
 * persistent kernel attn_fwd_hd64p for T 400..512, 800..1024 and >= 1024 with 512-row query tiles, for T 231..256 with 256-row
 * tiles; 32 query rows per wave otherwise), 1 = always 32 query rows per wave, 3 (or 2) = the persistent ring kernel of
 * round 2, 4 / 5 = the pipelined persistent kernel with one 8-wave / two 4-wave workgroups per CU. */
int sfm_attention_fwd_ex(const void* qkv, void* out, int B, int T, int H, int hd, int ldqkv, int ldo,
                         int koff, int voff, long long qkv_batch_stride, long long o_batch_stride,
                         float scale, int dtype, int out_dtype, int variant, void* stream);

/* nn.LayerNorm (+ optional erf GELU when act==1): models/conformer.py:43,68,107,150;
 * agents/msa.py:44-47; training/conformer_pipeline.py:274,282. */
int sfm_layernorm(const float* x, const float* w, const float* b, void* out16, float* out32, int M, int D,
                  int ldx, int ld16, int ld32, float eps, int act, int dtype, void* stream);

/* nn.GroupNorm split in (stats from GEMM epilogue) -> finalize -> apply
 * (agents/perception.py:157,169,176,180,196,199,204 and the residual add+GELU :129). */
int sfm_gn_finalize(const float* partial, const float* w, const float* b, float* scale, float* shift, int B,
                    int P, int G, int C, long long rows, float eps, void* stream);
int sfm_gn_apply(const void* x1, const float* sc1, const float* sh1, const void* x2, const float* sc2,
                 const float* sh2, void* out, int B, long long rows_per_batch, int C, int in_f32, int out_f32,
                 int act, int dtype, void* stream);

/* depthwise Conv1d + BatchNorm1d(eval) + Swish, channels-last: models/conformer.py:117-119 */
int sfm_dwconv_bn_swish(const void* x, const float* wdw, const float* bdw, const float* bnw,
                        const float* bnb, const float* bnm, const float* bnv, void* out, int B, int T,
                        int C, int KS, float eps, int dtype, void* stream);

/* same operation with host-folded operands (wT [KS][C], BatchNorm+bias folded into sc/sh), register-resident
 * taps; KS in {7, 31}, C in {64, 128, 256, 512} */
int sfm_dwconv_folded(const void* x, const float* wT, const float* sc, const float* sh, void* out, int B, int T,
                      int C, int KS, int act, int out_f32, int dtype, void* stream);

/* layout / packing */
int sfm_convert_rows(const float* src, void* dst, long long M, int C, int Cz, long long ld_src,
                     long long ld_dst, int dtype, void* stream);
int sfm_transpose(const void* src, void* dst, int B, int R, int C, long long src_batch, long long src_row,
                  long long dst_batch, long long dst_row, int src_f32, int dst_f32, int dtype, void* stream);
int sfm_pool_time(const float* src, void* dst16, float* dst32, int B, int Tin, int Tout, int C,
                  long long ld_src, long long ld_dst, int dtype, void* stream);
/* as sfm_pool_time with out = scale[b, c] * avg + shift[b, c]: pooled GroupNorm'd latents straight from the raw layer
 * output (glue G1 applied to agents/perception.py:244-246 without materialising the full-rate normalised tensor) */
int sfm_pool_time_affine(const float* src, const float* scale, const float* shift, void* dst16, float* dst32, int B,
                         int Tin, int Tout, int C, long long ld_src, long long ld_dst, int dtype, void* stream);
/* sfm_pool_time_affine on a 16-bit source [B, Tin, ld_src] in the format `src_dtype` (the fused path writes the raw latent
 * heads in the PerceptionAgent stage's operand format: agents/perception.py:244-246 + glue G1); dst16 in the format `dtype` */
int sfm_pool_time_affine16(const void* src16, int src_dtype, const float* scale, const float* shift, void* dst16, float* dst32,
                           int B, int Tin, int Tout, int C, long long ld_src, long long ld_dst, int dtype, void* stream);
/* mean over time (glue G2: episodic-memory key), src fp32 [B, T, ld_src] cols [0, C) -> dst fp32 [B, C]; deterministic
 * two-pass sum, scratch: sfm_mean_time_scratch_floats floats */
long long sfm_mean_time_scratch_floats(int B, int T, int C);
int sfm_mean_time(const float* src, float* dst, float* scratch, int B, int T, int C, long long ld_src, void* stream);
/* the same reduction without the 1/T (training: gradient of a per-utterance bias broadcast over the frames) */
int sfm_sum_time(const float* src, float* dst, float* scratch, int B, int T, int C, long long ld_src, void* stream);
/* adjoint of sfm_pool_time (training: gradient of the pooled latents back to the full-rate latents), fp32 */
int sfm_pool_time_bwd(const float* dout, float* dsrc, int B, int Tin, int Tout, int C, long long ld_dout,
                      long long ld_dsrc, void* stream);
/* agents/msa.py:134-137 */
int sfm_stft_lognorm_pack(const float* re, const float* im, void* dst, long long M, int F, int zpad,
                          long long ld_dst, int dtype, void* stream);
/* adjoint of sfm_stft_lognorm_pack: g [M, ld_g] fp32 holds d(real') in columns [0,F) and d(imag') in [F,2F) */
int sfm_stft_lognorm_bwd(const float* re, const float* im, const float* g, float* dre, float* dim_, long long M, int F,
                         long long ld_g, void* stream);
/* agents/msa.py:166-172, training/conformer_pipeline.py:287-296 */
int sfm_polar_mask(const float* lm, const float* lp, const float* mag_bias, const float* nr, const float* ni,
                   float* mr, float* mi, float* er, float* ei, float* mmag, int B, long long rows_per_batch,
                   int F, float phase_scale, long long ld_logits, long long ld_enh, void* stream);
/* models/conformer.py:243-244 */
int sfm_complex_mul(const float* sr, const float* si, const float* mr, const float* mi, float* er, float* ei,
                    long long total, void* stream);
/* overlap-add + envelope division of torch.istft: training/conformer_pipeline.py:210 */
int sfm_istft_ola(const float* frames, const float* win2, float* out, int B, int T, int L, int n_fft, int hop,
                  int win, long long ld_frames, void* stream);
int sfm_pack_spec(const float* re, const float* im, float* dst, long long M, int F, int ld, long long ld_src,
                  void* stream);
/* agents/perception.py:88-112 */
int sfm_sinc_filters(const float* low_hz, const float* band_hz, const float* window, const float* n_,
                     float* filt, float* Wt, int C, int K, int Npad, float sample_rate, float min_low_hz,
                     float min_band_hz, void* stream);
/* Whole FeedForwardModule in one launch (models/conformer.py:41-49, eval): out = x + alpha*(W2 swish(W1 LN(x)+b1)+b2).
 * x/out [M,256] fp32; W1 [FF,256], W2 [256,FF] 16-bit row-major (nn.Linear layout); D must be 256, FF % 64 == 0. */
int sfm_ffn_fused(const float* x, const float* lnw, const float* lnb, const void* W1, const float* b1,
                  const void* W2, const float* b2, float* out, int M, int D, int FF, float alpha, float eps,
                  int dtype, void* stream);
/* sfm_ffn_fused + the LayerNorm that follows the module inside the block (mhsa.layer_norm after ff1, final_norm after ff2;
 * models/conformer.py:66,151) applied to y in the epilogue: ln_out [M, 256] 16-bit (ln_out_f32 = 0) or fp32; out (y) may be
 * NULL when only the normalised rows are needed. */
int sfm_ffn_fused_ln(const float* x, const float* lnw, const float* lnb, const void* W1, const float* b1, const void* W2,
                     const float* b2, float* out, int M, int D, int FF, float alpha, float eps, const float* ln2w,
                     const float* ln2b, void* ln_out, int ln_out_f32, int dtype, void* stream);
/* SincConv1d FIR (agents/perception.py:117) on the 16-bit matrix cores with hi/lo split operands
 * (3 MFMA passes, ~fp32 accuracy).  filt [64,K] fp32 from sfm_sinc_filters; wsh = workspace of
 * 8*2*64*272 uint16; out [B,L,64] channels-last; gn_partial [B][sfm_sinc_fir16_tiles(L)][8][2],
 * which the caller must zero-fill (tiles past L are not written). */
int sfm_sinc_fir16_tiles(int L);
int sfm_sinc_fir16(const float* wave, const float* filt, void* wsh, void* out, float* gn_partial, int B,
                   int L, int C, int K, int out_f32, int dtype, void* stream);
/* passes: 3 = split operands (hi x hi + hi x lo + lo x hi: 1e-7 / 1e-6 max error), 1 = one rounding of waveform and taps to the
 * operand format, 0 = auto (1 for fp16 operands with a 16-bit result, else 3) - sfm_sinc_fir16 = passes 0 */
int sfm_sinc_fir16_ex(const float* wave, const float* filt, void* wsh, void* out, float* gn_partial, int B, int L, int C, int K,
                      int out_f32, int dtype, int passes, void* stream);
/* The same frames-x-matrix product on the 16-bit matrix cores with split bf16 operands (hi + lo, 3 MFMAs per k-step,
 * ~4e-6 relative error): the STFTs of the training objective (training/conformer_pipeline.py:74-108) and their adjoints.
 * Whi / Wlo: the constant matrix pre-split and n-major, [Npad (x256)][Kpad (x32)] uint16; out fp32 [b][m][n] with row
 * stride ldm; columns >= nsplit go to out2 at column n - nsplit + col2_off (when out2 != NULL). */
int sfm_framed_gemm_split16(const float* sig, const void* Whi, const void* Wlo, float* out, float* out2, int B, int M,
                            int Ls, long long sig_batch_stride, int hop, int padl, int K, int Kpad, int N, int Npad,
                            int nsplit, int col2_off, long long o_batch_stride, long long ldm, int mode, void* stream);
/* Forward of the training objective (training/conformer_pipeline.py:52-108, 539-572): reductions in fp64.
 * S buffers must be zero-filled by the caller (the kernels accumulate into them).
 * ORDERED REDUCTIONS (every entry point of the training step that sums over workgroups takes an optional workspace `ws`): with
 * it, each workgroup writes its partial into ws and a second pass folds the partials in workgroup order - the step is then
 * bit-reproducible, like the reference's CPU step (training/conformer_pipeline.py:496-532); ws == NULL selects fp32 / f64
 * atomics.  Sizes: sfm_*_ws_floats, or as stated; ws contents are scratch (destroyed), 16-byte aligned.
 * here: ws >= 64 * B * 5 doubles (wave_moments), >= 2048 * 4 doubles (spec_sums). */
int sfm_wave_moments(const float* est, const float* tgt, double* S, int B, int L, double* ws, void* stream);
int sfm_spec_sums(const float* pr, const float* pi, const float* tr, const float* ti, double* S, long long n, double* ws,
                  void* stream);
int sfm_enhancer_loss_finalize(const double* Sw, const double* Sm, const double* Sr, const long long* nr, int B,
                               int L, long long n_mag, int R, float* out, void* stream);
/* Backward of the objective and of the bounded polar mask (training/conformer_pipeline.py:52-108, 283-295, 539-572).
 * sisnr_bwd: dwave = scale * d(neg SI-SNR)/d est from the moments Sw of sfm_wave_moments.
 * spec_loss_bwd: mode 0 spectral-convergence + log-magnitude of one resolution (S = its sfm_spec_sums), mode 1 the
 *   L1 magnitude term; writes/accumulates d/d(real, imag) at dr/di[m*ld + f].
 * stft_adjoint_ola: adjoint of the reflect-padded framing of torch.stft: frames [B,T,win] -> dwave [B,L]
 *   (optionally accumulated, optionally multiplied by post[L] afterwards).
 * polar_mask_bwd: (d enh_real, d enh_imag) -> d logits [M, ld_dlog >= 2F] (magnitude | phase columns). */
int sfm_sisnr_bwd(const float* est, const float* tgt, const double* Sw, float* dwave, int B, int L, float scale,
                  void* stream);
int sfm_spec_loss_bwd(const float* pr, const float* pi, const float* tr, const float* ti, const double* S, float* dr,
                      float* di, long long n, int F, long long ld, int mode, int accumulate, float scale, void* stream);
int sfm_stft_adjoint_ola(const float* frames, float* dwave, const float* post, int B, int T, int L, int n_fft, int hop,
                         int win, int accumulate, void* stream);
/* mag_bias (or NULL): [M / rows_per_batch, F] added to the magnitude logit as in sfm_polar_mask; nr = ni = NULL: the mask itself
 * is the output (noisy = 1 + 0j, agents/msa.py:166-172) */
int sfm_polar_mask_bwd(const float* lm, const float* lp, const float* mag_bias, const float* nr, const float* ni,
                       const float* der, const float* dei, float* dlog, long long M, long long rows_per_batch, int F,
                       float phase_scale, long long ld_logits, long long ld_dlog, void* stream);
/* small-shape attention backward (any head_dim <= 256; the MFMA kernels of sfm_attention_bwd need head_dim 64):
 * dkv32 = zero-filled fp32 scratch [B*T, 2*H*hd]; dqkv gets dQ' | dK | dV like sfm_attention_bwd. */
int sfm_attention_bwd_generic(const void* qkv, const void* O, const void* dO, const float* lse, float* dkv32, void* dqkv,
                              int B, int T, int H, int hd, int ldqkv, int ldo, int koff, int voff, float p_drop,
                              unsigned int seed, int dtype, void* stream);
/* Batched quality metrics (evaluation/ssnr.py:26-92, fallback STOI evaluation/stoi.py:53-99).
 * ssnr_frames: acc [B][2] fp64 (zero-filled) += { sum of clipped frame SNRs over non-silent frames, their count }.
 * stoi_frames: spectra [B, nframes, F] of the raw signals, sc / se [B] fp64 = 1/(rms + 1e-10) of clean / enhanced,
 *   acc [B] fp64 (zero-filled) += sum over frames of the clipped spectral correlation. */
int sfm_ssnr_frames(const float* clean, const float* enh, double* acc, int B, int L, int frame, int hop, float upper,
                    float lower, void* stream);
int sfm_stoi_frames(const float* cr, const float* ci, const float* er, const float* ei, const double* sc, const double* se,
                    double* acc, int B, int nframes, int F, void* stream);
/* Gradient of the SincConv1d FIR bank w.r.t. its taps (training of agents/perception.py:79-118):
 * dfilt [C, K] += sum_{b,l} dy[b, l, c] x[b, l + k - K/2]; x [B, L] fp32, dy [B, L, C] 16-bit or fp32; scratch:
 * sfm_sinc_wgrad_scratch_floats floats (need not be zeroed).  K <= 256. */
long long sfm_sinc_wgrad_scratch_floats(int B, int L, int C, int K);
int sfm_sinc_wgrad(const float* x, const void* dy, int dy_f32, float* dfilt, float* scratch, int B, int L, int C, int K,
                   int dtype, void* stream);
/* The same tap gradient on the matrix cores for a 16-bit dy (K = 251 taps, centre 125): sfm_sinc_shift_pack writes 8 copies
 * of every utterance, zero-padded and shifted by 0..7 samples, xs 16-bit [B][8][sfm_sinc_shift_len(L)], so that the 8-tap
 * chunk at any sample is one aligned 16-byte load; sfm_sinc_wgrad16: dW fp32 [C][256] += dy^T * Toeplitz(x) (columns
 * 251..255 are scratch), dW zeroed by the caller. */
long long sfm_sinc_shift_len(int L);
int sfm_sinc_shift_pack(const float* x, void* xs, int B, int L, int dtype, void* stream);
int sfm_sinc_wgrad16(const void* dy, const void* xs, float* dW, int B, int L, int C, int dtype, float* ws, long long ws_floats,
                     void* stream);                       /* ws: sfm_tn_ws_floats(B * L, C, 256) */
/* Backward of the PerceptionAgent's GroupNorm nodes out = act(GN(x1) [+ GN(x2)]) (agents/perception.py:121-129, 157,
 * 192-206), channels-last [B, L, C], C a power of two in [64, 2048].  sc / sh [B, C] fp32: the forward's scale and shift;
 * mean / rstd [B, G].  reduce: S [B][3][C] += { sum dp, sum dp xhat1, sum dp xhat2 } with dp = dout * act'(p) (S zeroed by
 * the caller);  coefs: coef = [3][B][C] (a, b, c) and dparam [3][C] += (dbeta, dgamma1, dgamma2) (zeroed by the caller);
 * apply: dx_i = a dp - b - xhat_i c. */
int sfm_gn_bwd_reduce(const void* dout, int dout_f32, const void* x1, int x1_f32, const float* sc1, const float* sh1,
                      const float* mean1, const float* rstd1, const void* x2, int x2_f32, const float* sc2,
                      const float* sh2, const float* mean2, const float* rstd2, float* S, int B, int L, int C, int G,
                      int act, int dtype, float* ws, void* stream);
long long sfm_gn_bwd_reduce_ws_floats(int B, int L, int C);
/* (S is DESTROYED: after the coefficient tables it is the scratch of the ordered fold over b that produces dparam) */
int sfm_gn_bwd_coefs(float* S, const float* gamma1, const float* rstd1, const float* gamma2, const float* rstd2,
                     float* coef1, float* coef2, float* dparam, int B, int L, int C, int G, void* stream);
int sfm_gn_bwd_apply(const void* dout, int dout_f32, const void* x1, int x1_f32, const float* sc1, const float* sh1,
                     const float* mean1, const float* rstd1, const float* coef1, void* dx1, int dx1_f32, const void* x2,
                     int x2_f32, const float* sc2, const float* sh2, const float* mean2, const float* rstd2,
                     const float* coef2, void* dx2, int dx2_f32, int B, int L, int C, int G, int act, int dtype,
                     void* stream);
/* SURVEY 8f N4.  MetacognitiveArbitrationAgent (agents/maa.py:70-135) on N = B * T uncertainty values; stats = (running_mean,
 * running_var) fp32 on the device; params = W1[64] b1[64] W2[64*64] b2[64] W3[4*64] b3[4] fp32 (the decision_net).
 * update_stats: train()-mode EMA (momentum 0.1, unbiased batch variance; acc = 2 zeroed doubles, re-zeroed on return);
 * forward: logits / probs [N,4], decisions int64 [N], confidence [N];  backward: dsigma [N] and the 16-bit operands of the
 * weight-gradient GEMMs (H1, H2, dZ1, dZ2 [N,64]; GL [N,8] = total logit gradient; XN [N,8], column 0 = normalised sigma). */
int sfm_maa_update_stats(const float* sigma, long long n, double* acc, float* stats, long long* num_updates, float momentum,
                         void* stream);
int sfm_maa_forward(const float* sigma, const float* stats, const float* params, float* logits, float* probs,
                    long long* decisions, float* confidence, long long n, void* stream);
int sfm_maa_backward(const float* sigma, const float* stats, const float* params, const float* g_logits, const float* g_probs,
                     const float* g_conf, float* dsigma, void* H1, void* H2, void* dZ1, void* dZ2, void* GL, void* XN,
                     long long n, int dtype, void* stream);
/* VectorQuantizer (models/vq.py:54-96), M <= 16 scalar centroids: q = nearest centroid (first minimum), idx int64,
 * acc[0] += sum (x - q)^2 (double, zeroed by the caller);  backward: dx = g_q + g_loss beta 2 (x - q) / n (straight-through +
 * commitment term), dcent[k] += g_loss 2 (q - x) / n over the elements assigned to k (codebook term; zeroed by the caller). */
int sfm_vq_forward(const float* x, const float* centroids, int M, float* q, long long* idx, double* acc, long long n,
                   void* stream);
int sfm_vq_backward(const float* x, const long long* idx, const float* centroids, int M, const float* g_q,
                    const float* g_loss, float beta, float* dx, float* dcent, long long n, void* stream);
/* Optimiser step (training/conformer_pipeline.py:424-429 AdamW, :509 NaN/Inf skip, :514 clip_grad_norm_) on flat fp32
 * buffers.  ctl = 8 doubles: [0] step count, [1] sum of squares (sfm_sumsq accumulates; zeroed by the step), [2] flag > 0
 * forces a skip, [3] applied gradient scale, [4] skipped (0/1), [5],[6] bias corrections, [7] gradient norm. */
int sfm_sumsq(const float* g, long long n, double* out, double* ws, void* stream);        /* ws: >= 2048 doubles */
int sfm_adamw_step(float* p, float* g, float* m, float* v, long long n, double* ctl, float lr, float beta1, float beta2,
                   float eps, float wd, float inv_scale, float max_norm, int write_back_grad, void* stream);
/* Same, leaving alone - p, m and v - the parameters that received no gradient in this step (torch.optim.AdamW skips a
 * parameter whose grad is None): spans[k] = first flat element of parameter k (n_params + 1 ascending entries, device),
 * touched[k] > 0 = stepped (device; in data-parallel runs the per-rank 0/1 masks summed by the gradient all-reduce, so
 * that every replica takes the same decision). */
int sfm_adamw_step_masked(float* p, float* g, float* m, float* v, long long n, double* ctl, float lr, float beta1, float beta2,
                          float eps, float wd, float inv_scale, float max_norm, int write_back_grad, const long long* spans,
                          const float* touched, int n_params, void* stream);
/* The same step under the reference's AMP recipe (training/conformer_pipeline.py:442 torch.amp.GradScaler, :504
 * scaler.scale(loss).backward(), :512-517 unscale_ / clip_grad_norm_ / scaler.step / scaler.update) with the loss scale kept on the
 * device: loss_scale = 4 floats {S, clean steps since S last changed, steps skipped for Inf / NaN gradients, steps skipped for a
 * non-finite loss}.  g holds S (x world; inv_world = 1 / world) times the gradient: the prepare thread unscales, takes the
 * clip / skip decision, halves S (backoff_factor) after a step with Inf / NaN gradients, multiplies it by growth_factor after
 * growth_interval clean steps - GradScaler.update() without a host round trip.  spans / touched may both be NULL. */
int sfm_adamw_step_scaled(float* p, float* g, float* m, float* v, long long n, double* ctl, float lr, float beta1, float beta2,
                          float eps, float wd, float inv_world, float max_norm, int write_back_grad, const long long* spans,
                          const float* touched, int n_params, float* loss_scale, float growth_factor, float backoff_factor,
                          int growth_interval, void* stream);
/* ---- training path of the ConformerBlock (backward of models/conformer.py:28-151) ---- */
/* dW[n,k] += sum_m G[m,n] X[m,k] (weight gradient; M is split over workgroups: with ws (>= sfm_tn_ws_floats(M, N, K) floats) every
 * split writes its partial [N][K] and the partials are added to dW in split order, without ws by fp32 atomics; zero dW first);
 * db (optional): db[n] += sum_m G[m,n] in the same launch; sfm_colsum = the stand-alone bias gradient */
long long sfm_tn_ws_floats(int M, int N, int K);
int sfm_gemm16_tn(const void* G, const void* X, float* dW, float* db, int M, int N, int K, int ldg, int ldx, int ldw,
                  int dtype, float* ws, long long ws_floats, void* stream);
/* Conv1d weight gradient (training of agents/perception.py:121-129, 160-188 convs): G = dY [B*Lout, N] 16-bit, x
 * [B, Lin, Cin] channels-last 16-bit, dW [N, ksize*Cin] tap-major fp32 += G^T im2col(x) with the im2col rows addressed
 * in place (zero padding by range check); db (optional) += column sums of G. */
int sfm_conv_wgrad16(const void* G, const void* x, float* dW, float* db, int B, int Lout, int Lin, int Cin, int N, int ksize,
                     int stride, int pad, long long x_batch_stride, int ldg, int ldw, int dtype, float* ws, long long ws_floats,
                     void* stream);                       /* ws: sfm_tn_ws_floats(B * Lout, N, ksize * Cin) */
long long sfm_colsum_ws_floats(int M, int N);
int sfm_colsum(const void* G, float* out, int M, int N, int ldg, int g_f32, int dtype, float* ws, void* stream);
/* LayerNorm backward; ws (optional, sfm_layernorm_bwd_ws_floats(M, D) floats): ordered dgamma / dbeta */
long long sfm_layernorm_bwd_ws_floats(int M, int D);
int sfm_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* dres, float* dx,
                      float* dgamma, float* dbeta, int M, int D, int ldx, int ld, float eps, float* ws, void* stream);
/* the same with dy in the 16-bit format `dtype` (dy_16 != 0) and its own row stride ldy; ld = row stride of dres and dx */
int sfm_layernorm_bwd_ex(const float* x, const float* gamma, const void* dy, int dy_16, const float* dres, float* dx,
                         float* dgamma, float* dbeta, int M, int D, int ldx, int ldy, int ld, float eps, int dtype,
                         float* ws, void* stream);
/* the same, also writing next16 [M, D] (format `dtype`, contiguous rows) = next_alpha * dropout(dx; next_p, next_seed) with the
 * counters of sfm_ew_train mode 4: the 16-bit operand the next backward node of the residual chain starts from */
int sfm_layernorm_bwd_next(const float* x, const float* gamma, const void* dy, int dy_16, const float* dres, float* dx,
                           float* dgamma, float* dbeta, int M, int D, int ldx, int ldy, int ld, float eps, int dtype,
                           void* next16, float next_alpha, float next_p, unsigned int next_seed, float* ws, void* stream);
/* mode 0 swish fwd, 1 swish bwd, 2 GLU fwd, 3 GLU bwd, 4 alpha*g*dropout; counter-based dropout (p, seed) */
int sfm_ew_train(const void* z, const void* g, void* out, long long M, int N, int mode, int g_f32, int out_f32,
                 float alpha, float p, unsigned int seed, int dtype, void* stream);
/* BatchNorm1d training statistics / backward (through the following Swish); ws (optional): sfm_col_stats_ws_floats(M, C) floats */
long long sfm_col_stats_ws_floats(int M, int C);
int sfm_col_stats(const float* y, const float* aux, const float* mean, const float* rstd, float* S, int M, int C, float* ws,
                  void* stream);
/* gradient fan-in: out[m, c] = a[m, c] + (c < Cb ? b[m, c] : 0), fp32 rows with strides lda / ldb / ldo (C, Cb multiples of 4) */
int sfm_add_cols(const float* a, const float* b, float* out, long long M, int C, int Cb, long long lda, long long ldb, long long ldo,
                 void* stream);
/* BiLSTM dW_hh operand: previous output of each chain, h fp32 [B, T, 2H] -> out 16-bit [B*T, 2H]
 * (out[b,t,:H] = h[b,t-1,:H], out[b,t,H:] = h[b,t+1,H:], zero at the chain's first step) */
int sfm_lstm_hprev16(const float* h, void* out, int B, int T, int H, int dtype, void* stream);
/* nn.BatchNorm1d training statistics from sfm_col_stats' sums: mean, rstd, folded affine (sc, sh) and the in-place update of the
 * running statistics (unbiased variance; run_* may be NULL); eval_mode: statistics = run_mean / run_var, nothing updated */
int sfm_bn_finalize(const float* S, const float* gamma, const float* beta, float* run_mean, float* run_var, float* mean,
                    float* rstd, float* sc, float* sh, int C, long long M, float eps, float momentum, int eval_mode,
                    void* stream);
int sfm_bn_swish_bwd(const void* g, const float* y, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, float* S, float* dy, int M, int C, int g_f32, int pass, int dtype,
                     float* ws, void* stream);             /* ws: as sfm_col_stats (pass 0 only) */
/* depthwise-conv weight/bias gradient: per-(utterance, span) partial sums go to `scratch`
 * (sfm_dwconv_wgrad_scratch_floats floats, need not be zeroed) and are reduced into dw [C, KS] / db [C] (accumulated). */
long long sfm_dwconv_wgrad_scratch_floats(int B, int T, int C, int KS);
int sfm_dwconv_wgrad(const void* x, const float* dy, float* dw, float* db, float* scratch, int B, int T, int C, int KS,
                     int dtype, void* stream);
/* attention forward for training (writes log2-domain LSE [B,H,T], applies attention dropout) and its backward */
int sfm_attention_fwd_train(const void* qkv, void* out, float* lse, int B, int T, int H, int hd, int ldqkv,
                            int ldo, int koff, int voff, long long qkv_batch_stride, long long o_batch_stride,
                            float scale, float p_drop, unsigned int seed, int dtype, void* stream);
int sfm_attention_bwd(const void* qkv, const void* O, const void* dO, const float* lse, float* delta, void* dqkv,
                      int B, int T, int H, int hd, int ldqkv, int ldo, int koff, int voff, float p_drop,
                      unsigned int seed, int dtype, void* stream);
/* Linear layer with K = 256 inputs and a 16-bit result, one 8-wave workgroup per 128 rows for ALL output columns (A tile resident in
   registers, W streamed through an LDS-DMA ring, epilogue of a 64-row W chunk under the next chunk's MFMAs; csrc/lin256.hip):
     glu == 0: out[M, NW]     = A[M, 256] W^T + b                       (models/conformer.py:57-59, the Q | K | V projection)
     glu != 0: out[M, NW / 2] = GLU(A W^T + b), W rows in 64-row groups of 32 values | 32 gates
                                                                         (models/conformer.py:92-96, pointwise_conv1 + GLU)
   A rows have stride lda >= 256 (16-bit elements, multiple of 8), W is [NW, 256] row-major, NW % 64 == 0 (% 128 with glu),
   bias [NW] fp32 or NULL; out_dtype may be the other 16-bit format (one rounding of the fp32 accumulators) or, with glu == 0,
   2 = fp32 (ldo a multiple of 4; the BiLSTM input projections of agents/cpea.py:43-50).  The same contract as
   sfm_gemm16 on these shapes, and the same bits (same MFMA, same k order, same epilogue expressions). */
int sfm_lin256(const void* A, const void* W, const float* bias, void* out, int M, int NW, int lda, int ldo, int glu, int dtype,
               int out_dtype, void* stream);
/* The same with LayerNorm(X32[m, :256]; lnw, lnb, eps) as the operand rows (models/conformer.py:88 conv.layer_norm -> :92 pointwise_conv1):
   the workgroup that owns 128 rows for all output columns normalises them once, in its prologue, from the fp32 residual stream
   [M, ldx] - the arithmetic of sfm_layernorm on D = 256, bit for bit -, so sfm_layernorm + sfm_lin256 in one launch and without the
   16-bit normalised tensor in HBM.  16-bit results only. */
int sfm_ln_lin256(const float* X32, int ldx, const float* lnw, const float* lnb, float eps, const void* W, const float* bias,
                  void* out, int M, int NW, int ldo, int glu, int dtype, int out_dtype, void* stream);
/* PerceptionAgent latent heads (agents/perception.py:183-199, real_proj | imag_proj 1x1 convs stacked: NW = 2 * latent_dim rows of W) with
   the time pooling of the fused path (glue G1) in the same launch: pooled[b, i, :] = mean over the adaptive-average window of frame i
   of xd[b, t, :256] W^T + bias (16-bit, RAW: the GroupNorm that follows has no activation, is affine per (utterance, channel) and is
   applied to `pooled` afterwards - sfm_pool_time_affine16 with Tin == Tout), and the GroupNorm statistics of the FULL-RATE outputs:
   gn_partial [B, P, NW / gcols, 2] (sum, sum of squares per row tile; P = ceil(Tout / sfm_headpool_frames_per_tile(Tin, Tout)); reduce
   with sfm_gn_finalize(rows = Tin)).  The full-rate head outputs are never written.  gcols = 16; Tin >= Tout. */
int sfm_headpool_frames_per_tile(int Tin, int Tout);
int sfm_headpool(const void* xd, const void* W, const float* bias, void* pooled, float* gn_partial, int B, int Tin, int Tout, int NW,
                 int lda, int ldp, int gcols, int dtype, void* stream);

/* one direction-pair of an nn.LSTM layer (agents/cpea.py:43-50,99), see lstm.hip */
int sfm_bilstm_layer(const float* xg, const float* whh, float* out, int B, int T, int H, int dtype, void* stream);
/* inference only: w16 != 0 runs the recurrent product W_hh h on fp16 operands (weights and h rounded once, fp32 accumulation, gates /
   cell state / output fp32): half the registers, so two chains share a CU - 1.6 x at batch 256; H 32 keeps the fp32 kernel.
   (agents/cpea.py:43-50,99: the same nn.LSTM recurrence) */
int sfm_bilstm_layer_ex(const float* xg, const float* whh, float* out, int B, int T, int H, int w16, void* stream);
/* sfm_bilstm_layer_train with the same fp16-operand recurrent product (the reference's LSTM runs under fp16 autocast in training:
   training/conformer_pipeline.py:504); `save` and the BPTT (sfm_bilstm_layer_bwd) stay fp32 */
int sfm_bilstm_layer_train_ex(const float* xg, const float* whh, float* out, float* save, int B, int T, int H, int w16, void* stream);
/* training: as sfm_bilstm_layer, also saving the activated gates and cell states [B, T, 2, 5, H] fp32; and the BPTT of
 * the layer: dout [B, T, 2H] -> dxg [B, T, 2, 4H] = gradient w.r.t. the input projection (dW_ih, dW_hh, biases and dx
 * are GEMMs / column sums of dxg afterwards). */
int sfm_bilstm_layer_train(const float* xg, const float* whh, float* out, float* save, int B, int T, int H, int dtype,
                           void* stream);
int sfm_bilstm_layer_bwd(const float* save, const float* whh, const float* dout, float* dxg, int B, int T, int H,
                         void* stream);
/* EpisodicMemory.forward eval (agents/memory.py:112-133) in one launch, see memory.hip */
int sfm_memory_fwd(const float* emb, const float* params, float* bias_out, float* gate_out, int* top_idx,
                   float* sim_out, int B, int key_dim, int value_dim, int slots, float temperature, void* stream);
/* Backward of sfm_memory_fwd (EpisodicMemory in train() mode): d_out = gradient of the gated bias [B, value_dim], d_gate = of
 * the gate [B] (or NULL); d_emb [B, key_dim] (or NULL) and dparams = a ZERO-FILLED blob with the layout of `params`
 * (gradients of key_proj, keys, values, value_proj, gate, accumulated over the rows: with ws - B x sfm_memory_param_floats floats -
 * every row writes its own copy of the blob and the copies are folded in row order; ws == NULL: fp32 atomics). */
long long sfm_memory_param_floats(int key_dim, int value_dim, int slots);
int sfm_memory_bwd(const float* emb, const float* params, const float* d_out, const float* d_gate, float* d_emb,
                   float* dparams, int B, int key_dim, int value_dim, int slots, float temperature, float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif
