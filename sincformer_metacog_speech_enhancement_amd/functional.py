"""Forward compositions of the HIP kernels for every module on the path.

Internal layouts (DESIGN.md "Data layout in HBM"):
  * time-major / channels-last everywhere: [B, T, C] (never [B, C, T]),
  * fp32 residual stream, 16-bit (bf16|f16) GEMM operands and wide activations,
  * weights packed once per (parameter version, dtype) into MFMA-friendly
    [Npad, Kpad] 16-bit tiles (`pack_*`), cached by the nn.Module mirrors.
All functions take/return device tensors and only enqueue kernels on the
current stream; there is no CPU code path.
"""
import math
import torch

from . import ops

N_FFT, HOP, WIN = 256, 80, 160
N_FREQ = N_FFT // 2 + 1
SINC_ON_16BIT_MFMA = True  # False: exact-fp32 matrix instruction (framed_gemm) for the sinc FIR
FUSE_LD = 1088           # 2*256 + 4*64 + 2*129 = 1026 -> padded to 17*64


def _f32(t):
    return t.detach().float().contiguous()


def sub(sd, prefix):
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def add_to16(x32, y32, C):
    """(x + y) -> 16-bit, via the two-branch scale/shift kernel with unit scales."""
    M = x32.numel() // C
    dev = x32.device
    one = _const(dev, C, 1.0)
    zero = _const(dev, C, 0.0)
    out = torch.empty(M, C, device=dev, dtype=ops.compute_dtype())
    ops.gn_apply(x32, one, zero, out, 1, M, C, act=0, x2=y32, sc2=one, sh2=zero)
    return out


_const_cache = {}


def _const(dev, C, val):
    key = (str(dev), C, val)
    t = _const_cache.get(key)
    if t is None:
        t = torch.full((1, C), val, device=dev, dtype=torch.float32)
        _const_cache[key] = t
    return t


# ---------------------------------------------------------------------------
# STFT / iSTFT  (training/conformer_pipeline.py:196-211)
# ---------------------------------------------------------------------------
_stft_cache = {}


def _stft_consts(n_fft, win, dev):
    key = (n_fft, win, str(dev))
    c = _stft_cache.get(key)
    if c is None:
        wi, win2 = ops.istft_matrix(n_fft, win, dev)
        c = {"fwd": ops.stft_matrix(n_fft, win, dev), "inv": wi, "win2": win2}
        _stft_cache[key] = c
    return c


def stft(wave, n_fft=N_FFT, hop=HOP, win=WIN):
    """wave [B, L] fp32 -> real, imag [B, T, n_fft/2+1] fp32 (contiguous), T = 1 + L//hop."""
    wave = wave.contiguous()
    B, L = wave.shape
    if L <= n_fft // 2:
        raise RuntimeError("stft: signal shorter than the reflect padding (L=%d)" % L)
    T = 1 + L // hop
    F = n_fft // 2 + 1
    c = _stft_consts(n_fft, win, wave.device)
    re = torch.empty(B, T, F, device=wave.device, dtype=torch.float32)
    im = torch.empty(B, T, F, device=wave.device, dtype=torch.float32)
    # frame t covers xpad[t*hop + woff + k], k in [0, win): sample index t*hop + k - win/2  (reflected)
    ops.framed_gemm(wave, c["fwd"], re, B=B, M=T, Ls=L, sig_batch_stride=L, hop=hop, padl=n_fft // 2 - (n_fft - win) // 2,
                    K=win, N=2 * F, o_batch_stride=T * F, ldm=F, ldn=1, mode=1, out2=im, nsplit=F)
    return re, im


def stft_split16(wave, n_fft=N_FFT, hop=HOP, win=WIN):
    """stft() on the 16-bit matrix cores with split bf16 operands (~4e-6 relative error, 5x the fp32 MFMA rate): the
    STFTs of the training objective.  The matrix drops the two identically-zero columns (imag of DC and Nyquist), so a
    spectrum is exactly n_fft columns = whole 256-column tiles."""
    wave = wave.contiguous()
    B, L = wave.shape
    if L <= n_fft // 2:
        raise RuntimeError("stft: signal shorter than the reflect padding (L=%d)" % L)
    T = 1 + L // hop
    F = n_fft // 2 + 1
    c = _stft_consts(n_fft, win, wave.device)
    if "fwd16" not in c:
        base = c["fwd"][:win, :2 * F]
        c["fwd16"] = ops.pack_split16_matrix(torch.cat([base[:, :F], base[:, F + 1:2 * F - 1]], dim=1).contiguous())
    re = torch.empty(B, T, F, device=wave.device, dtype=torch.float32)
    im = torch.empty(B, T, F, device=wave.device, dtype=torch.float32)
    ops.framed_gemm_split16(wave, c["fwd16"], re, B=B, M=T, Ls=L, sig_batch_stride=L, hop=hop,
                            padl=n_fft // 2 - (n_fft - win) // 2, o_batch_stride=T * F, ldm=F, mode=1, out2=im, nsplit=F,
                            col2_off=1)
    im[..., 0] = 0.0
    im[..., F - 1] = 0.0
    return re, im


def istft_from_packed(spec, B, T, length, n_fft=N_FFT, hop=HOP, win=WIN):
    """spec [B*T, ld>=2F] fp32 rows (real | imag | zero pad) -> waveform [B, length]."""
    F = n_fft // 2 + 1
    c = _stft_consts(n_fft, win, spec.device)
    ld = spec.stride(0)
    M = B * T
    frames = torch.empty(M, win, device=spec.device, dtype=torch.float32)
    ops.framed_gemm(spec, c["inv"], frames, B=1, M=M, Ls=M * ld, sig_batch_stride=0, hop=ld, padl=0, K=2 * F, N=win,
                    o_batch_stride=0, ldm=win, ldn=1, mode=0)
    out = torch.empty(B, length, device=spec.device, dtype=torch.float32)
    ops.istft_ola(frames, c["win2"], out, B, T, length, n_fft, hop, win, win)
    return out


def istft_from_packed_split16(spec, B, T, length, n_fft=N_FFT, hop=HOP, win=WIN):
    """istft_from_packed with the irfft product on the 16-bit matrix cores (split bf16 operands, ~4e-6 relative error)."""
    F = n_fft // 2 + 1
    c = _stft_consts(n_fft, win, spec.device)
    if "inv16" not in c:
        c["inv16"] = ops.pack_split16_matrix(c["inv"][:2 * F, :win].contiguous())
    ld = spec.stride(0)
    M = B * T
    frames = torch.empty(M, win, device=spec.device, dtype=torch.float32)
    ops.framed_gemm_split16(spec, c["inv16"], frames, B=1, M=M, Ls=M * ld, sig_batch_stride=0, hop=ld, padl=0, o_batch_stride=0,
                            ldm=win, mode=0)
    out = torch.empty(B, length, device=spec.device, dtype=torch.float32)
    ops.istft_ola(frames, c["win2"], out, B, T, length, n_fft, hop, win, win)
    return out


# The fused inference path (enhance_path) takes its STFT / iSTFT on split bf16 operands: 4e-6 relative error against the exact
# fp32 matrix instruction, far below the path's 16-bit activations, at a third of the time.  The module API (batch_stft,
# batch_istft, the loss mirrors) keeps the exact form.
FUSED_STFT_SPLIT16 = True


def istft(real, imag, length, n_fft=N_FFT, hop=HOP, win=WIN):
    real, imag = real.contiguous(), imag.contiguous()
    B, T, F = real.shape
    ld = ops.round_up(2 * F, 8)
    spec = torch.empty(B * T, ld, device=real.device, dtype=torch.float32)
    ops.pack_spec(real, imag, spec, B * T, F, ld, F)
    return istft_from_packed(spec, B, T, length, n_fft, hop, win)


# ---------------------------------------------------------------------------
# Conformer (models/conformer.py)
# ---------------------------------------------------------------------------
FUSED_FFN = True      # False: LayerNorm + two gemm16 launches


def pack_ffn(sd):
    pk = {"ln_w": _f32(sd["layer_norm.weight"]), "ln_b": _f32(sd["layer_norm.bias"]),
          "w1": ops.pack_linear(sd["linear1.weight"], sd["linear1.bias"]),
          "w2": ops.pack_linear(sd["linear2.weight"], sd["linear2.bias"])}
    FF, D = sd["linear1.weight"].shape
    if D == 256 and FF % 64 == 0:          # operands of the fused kernel: plain nn.Linear layouts in 16-bit
        dt = ops.compute_dtype()
        pk["fused"] = (sd["linear1.weight"].detach().to(dt).contiguous(), _f32(sd["linear1.bias"]),
                       sd["linear2.weight"].detach().to(dt).contiguous(), _f32(sd["linear2.bias"]))
    return pk


def pack_mhsa(sd, num_heads=4):
    w, b = sd["attention.in_proj_weight"].detach().float(), sd["attention.in_proj_bias"].detach().float()
    D = w.shape[1]
    # fold softmax_scale * log2(e) into the Q rows so the attention kernel's exp2 needs no per-score multiply
    qs = ops.ATTN_QSCALE_LOG2E / math.sqrt(D // num_heads)
    w = torch.cat([w[:D] * qs, w[D:]], dim=0)
    b = torch.cat([b[:D] * qs, b[D:]], dim=0)
    return {"ln_w": _f32(sd["layer_norm.weight"]), "ln_b": _f32(sd["layer_norm.bias"]),
            "win": ops.pack_linear(w, b), "heads": num_heads,
            "wout": ops.pack_linear(sd["attention.out_proj.weight"], sd["attention.out_proj.bias"])}


def pack_convmod(sd):
    D = sd["pointwise2.weight"].shape[0]
    KS = sd["depthwise.weight"].shape[-1]
    folded = None
    if KS in (7, 31) and D in (64, 128, 256, 512):
        # BatchNorm(eval) and the depthwise bias folded into one scale/shift per channel; taps transposed to [KS][C]
        bsc = _f32(sd["batch_norm.weight"]) * torch.rsqrt(_f32(sd["batch_norm.running_var"]) + 1e-5)
        bsh = _f32(sd["batch_norm.bias"]) - _f32(sd["batch_norm.running_mean"]) * bsc + _f32(sd["depthwise.bias"]) * bsc
        folded = (_f32(sd["depthwise.weight"].reshape(D, KS)).t().contiguous(), bsc.contiguous(), bsh.contiguous())
    return {"dw_folded": folded, "ln_w": _f32(sd["layer_norm.weight"]), "ln_b": _f32(sd["layer_norm.bias"]),
            "pw1": ops.pack_linear(sd["pointwise1.weight"].reshape(2 * D, D), sd["pointwise1.bias"], glu=True),
            "dw_w": _f32(sd["depthwise.weight"].reshape(D, -1)), "dw_b": _f32(sd["depthwise.bias"]),
            "bn_w": _f32(sd["batch_norm.weight"]), "bn_b": _f32(sd["batch_norm.bias"]),
            "bn_m": _f32(sd["batch_norm.running_mean"]), "bn_v": _f32(sd["batch_norm.running_var"]),
            "pw2": ops.pack_linear(sd["pointwise2.weight"].reshape(D, D), sd["pointwise2.bias"])}


def pack_block(sd, num_heads=4, index=None):
    """index: position of the block in its stack (precision policy overrides "block<i>" / "attn<i>", ops.STAGES)"""
    with ops.stage("block", index):
        return {"ff1": pack_ffn(sub(sd, "ff1")), "mhsa": pack_mhsa(sub(sd, "mhsa"), num_heads), "conv": pack_convmod(sub(sd, "conv")),
                "ff2": pack_ffn(sub(sd, "ff2")), "fn_w": _f32(sd["final_norm.weight"]), "fn_b": _f32(sd["final_norm.bias"]),
                "index": index}


def _ln16(x32, w, b, act=0):
    out = torch.empty(x32.shape, device=x32.device, dtype=ops.compute_dtype())
    ops.layernorm(x32, w, b, out16=out, act=act)
    return out


def _ffn_fusable(x32, pk):
    return FUSED_FFN and "fused" in pk and x32.is_contiguous()


def ffn_forward_ln(x32, pk, ln_w, ln_b, ln_f32, want_y):
    """FeedForwardModule + the LayerNorm that follows it in the block, one launch: (y or None, LN(y))."""
    w1, b1, w2, b2 = pk["fused"]
    return ops.ffn_fused_ln(x32, pk["ln_w"], pk["ln_b"], w1, b1, w2, b2, ln_w, ln_b, ln_f32, want_y=want_y, alpha=0.5)


def ffn_forward(x32, pk):
    """FeedForwardModule.forward (eval): x + 0.5 * W2 swish(W1 LN(x))   [M, D] fp32 -> fp32"""
    if FUSED_FFN and "fused" in pk and x32.is_contiguous():
        w1, b1, w2, b2 = pk["fused"]
        return ops.ffn_fused(x32, pk["ln_w"], pk["ln_b"], w1, b1, w2, b2, alpha=0.5)
    h = _ln16(x32, pk["ln_w"], pk["ln_b"])
    u = ops.linear16(h, pk["w1"], epi=ops.EPI_SWISH)
    return ops.linear16(u, pk["w2"], epi=ops.EPI_RESID, resid=x32, alpha=0.5)


def mhsa_forward(x32, pk, B, T, H, h=None, index=None):
    """h: LN(x32) in 16-bit when the producer already normalised (fused into the preceding FFN's epilogue).
    The projections run in the caller's ("block") operand format, the Q | K | V buffer and the attention core in the
    "attn" stage's format: the in-projection's epilogue writes that format, the attention epilogue writes the block's."""
    D = x32.shape[1]
    bdt = ops.compute_dtype()
    adt = ops.stage_dtype("attn", index) if D // H == 64 else bdt      # the small-shape attention kernel keeps one format
    if pk["heads"] != H:
        raise RuntimeError("packed attention weights were scaled for %d heads, got %d" % (pk["heads"], H))
    if h is None:
        qkv = ops.ln_linear16(x32, pk["ln_w"], pk["ln_b"], pk["win"], out_dtype=adt)
    else:
        qkv = ops.linear16(h, pk["win"], out_dtype=adt)
    if adt is bdt:
        o = ops.attention(qkv, B, T, H, D // H, prescaled=True)
    else:
        with ops.stage("attn", index):
            o = ops.attention(qkv, B, T, H, D // H, prescaled=True, out_dtype=bdt)
    return ops.linear16(o, pk["wout"], epi=ops.EPI_RESID, resid=x32, alpha=1.0)


def convmod_forward(x32, pk, B, T):
    D = x32.shape[1]
    g = ops.ln_linear16(x32, pk["ln_w"], pk["ln_b"], pk["pw1"], epi=ops.EPI_GLU)      # LayerNorm = the GEMM kernel's prologue
    if pk["dw_folded"] is not None:
        d = ops.dwconv_folded(g, pk["dw_folded"][0], pk["dw_folded"][1], pk["dw_folded"][2], B, T, D)
    else:
        d = ops.dwconv_bn_swish(g, pk["dw_w"], pk["dw_b"], pk["bn_w"], pk["bn_b"], pk["bn_m"], pk["bn_v"], B, T, D)
    return ops.linear16(d, pk["pw2"], epi=ops.EPI_RESID, resid=x32, alpha=1.0)


def block_forward(x32, pk, B, T, H, want16=False):
    """ConformerBlock.forward (eval) on the flattened [B*T, D] fp32 stream."""
    with ops.stage("block", pk.get("index")):
        return _block_forward(x32, pk, B, T, H, want16)


def _block_forward(x32, pk, B, T, H, want16):
    if _ffn_fusable(x32, pk["ff1"]) and x32.shape[1] == 256:
        # ff1 and mhsa.layer_norm in one launch; likewise ff2 and final_norm below (the sum ff2 produces is not kept)
        x, h = ffn_forward_ln(x32, pk["ff1"], pk["mhsa"]["ln_w"], pk["mhsa"]["ln_b"], False, True)
        x = mhsa_forward(x, pk["mhsa"], B, T, H, h=h, index=pk.get("index"))
        x = convmod_forward(x, pk["conv"], B, T)
        if not want16 and _ffn_fusable(x, pk["ff2"]):
            return ffn_forward_ln(x, pk["ff2"], pk["fn_w"], pk["fn_b"], True, False)[1]
        x = ffn_forward(x, pk["ff2"])
    else:
        x = ffn_forward(x32, pk["ff1"])
        x = mhsa_forward(x, pk["mhsa"], B, T, H, index=pk.get("index"))
        x = convmod_forward(x, pk["conv"], B, T)
        x = ffn_forward(x, pk["ff2"])
    out = torch.empty_like(x)
    out16 = torch.empty(x.shape, device=x.device, dtype=ops.compute_dtype()) if want16 else None
    ops.layernorm(x, pk["fn_w"], pk["fn_b"], out16=out16, out32=out)
    return (out, out16) if want16 else out


def pack_complex_conformer(sd, num_blocks, num_heads=4):
    nf2 = sd["input_proj.weight"].shape[1]
    with ops.stage("front"):
        win = ops.pack_linear(sd["input_proj.weight"], sd["input_proj.bias"], k_pad_to=ops.round_up(nf2, 64))
    with ops.stage("tail"):
        wout = ops.pack_linear(sd["output_proj.weight"], sd["output_proj.bias"])
    return {"in": win, "out": wout,
            "blocks": [pack_block(sub(sd, "blocks.%d" % i), num_heads, i) for i in range(num_blocks)], "nf2": nf2}


def complex_conformer_core(x16, pk, B, T, H, out_dtype=None):
    """x16: [M, ld>=round32(2*n_freq)] 16-bit operand of input_proj -> output_proj result [M, 2*n_freq]."""
    with ops.stage("front"):
        x = ops.linear16(x16, pk["in"], out_dtype=torch.float32)
    skip = x
    for bp in pk["blocks"]:
        x = block_forward(x, bp, B, T, H)
    with ops.stage("tail"):
        y16 = add_to16(x, skip, x.shape[1])
        return ops.linear16(y16, pk["out"], out_dtype=out_dtype or ops.compute_dtype())


def complex_conformer_forward(stft_real, stft_imag, pk, H):
    """ComplexConformer.forward: ([B,T,nf], [B,T,nf]) fp32 -> (mask_real, mask_imag) fp32 views."""
    B, T, nf = stft_real.shape
    M = B * T
    sr, si = stft_real.contiguous(), stft_imag.contiguous()
    ld = pk["in"].Kpad
    with ops.stage("front"):
        x16 = torch.empty(M, ld, device=sr.device, dtype=ops.compute_dtype())
        ops.convert_rows(sr, x16, M, nf, nf, nf, ld)
        ops.convert_rows(si, x16[:, nf:], M, nf, ld - nf, nf, ld)
    y = complex_conformer_core(x16, pk, B, T, H, out_dtype=torch.float32).reshape(B, T, 2 * nf)
    return y[..., :nf], y[..., nf:]


# ---------------------------------------------------------------------------
# PerceptionAgent (agents/perception.py:216-251)
# ---------------------------------------------------------------------------
def pack_perception(sd, sample_rate):
    with ops.stage("pa"):
        return _pack_perception(sd, sample_rate)


def _pack_perception(sd, sample_rate):
    pk = {"fs": float(sample_rate)}
    for k in ("low_hz_", "band_hz_", "window", "n_"):
        pk[k] = _f32(sd["sinc_conv." + k]).reshape(-1)
    pk["K"] = pk["window"].numel()
    pk["C0"] = pk["low_hz_"].numel()
    pk["sn_w"], pk["sn_b"] = _f32(sd["sinc_norm.weight"]), _f32(sd["sinc_norm.bias"])
    pk["blocks"] = []
    for i in range(3):
        s = sub(sd, "conv_blocks.%d" % i)
        cout = s["main.0.weight"].shape[0]
        pk["blocks"].append({
            "c1": ops.pack_linear(s["main.0.weight"], s["main.0.bias"]), "g1w": _f32(s["main.1.weight"]), "g1b": _f32(s["main.1.bias"]),
            "c2": ops.pack_linear(s["main.3.weight"], s["main.3.bias"]), "g2w": _f32(s["main.4.weight"]), "g2b": _f32(s["main.4.bias"]),
            "cs": ops.pack_linear(s["skip.0.weight"], s["skip.0.bias"]), "gsw": _f32(s["skip.1.weight"]), "gsb": _f32(s["skip.1.bias"]),
            "cout": cout, "groups": min(16, cout)})
    pk["down"] = ops.pack_linear(sd["downsample.0.weight"], sd["downsample.0.bias"])
    pk["dn_w"], pk["dn_b"] = _f32(sd["downsample.1.weight"]), _f32(sd["downsample.1.bias"])
    D = sd["real_proj.0.weight"].shape[0]
    pk["D"] = D
    wz = torch.cat([sd["real_proj.0.weight"], sd["imag_proj.0.weight"]], dim=0)
    bz = torch.cat([sd["real_proj.0.bias"], sd["imag_proj.0.bias"]], dim=0)
    pk["zproj"] = ops.pack_linear(wz, bz)
    pk["z_w"] = torch.cat([_f32(sd["real_proj.1.weight"]), _f32(sd["imag_proj.1.weight"])])
    pk["z_b"] = torch.cat([_f32(sd["real_proj.1.bias"]), _f32(sd["imag_proj.1.bias"])])
    pk["u0"] = ops.pack_linear(sd["uncertainty_head.0.weight"], sd["uncertainty_head.0.bias"])
    pk["u2"] = ops.pack_linear(sd["uncertainty_head.2.weight"], sd["uncertainty_head.2.bias"])
    return pk


def _conv_gn(x16, pw, B, Lin, stride, pad, groups, raw_dtype):
    """conv (implicit GEMM) + GroupNorm partial statistics; returns raw [B, Lout, C], partial, P, Lout."""
    cin, k = pw.cin, pw.ksize
    Lout = (Lin + 2 * pad - k) // stride + 1
    C = pw.N
    raw = torch.empty(B, Lout, C, device=x16.device, dtype=raw_dtype)
    P = 2 * ((Lout + 127) // 128)
    part = torch.empty(B, P, groups, 2, device=x16.device, dtype=torch.float32)
    ops.gemm16(x16, pw, raw, B=B, Lout=Lout, Lin=Lin, a_batch_stride=Lin * cin, ldo=C, o_batch_stride=Lout * C,
               stride=stride, pad=pad, gn_partial=part, gn_group=C // groups)
    return raw, part, P, Lout


def perception_forward(wave, pk, keep_sinc=False, latents=True, pool_to=None):
    """wave [B, L] fp32 -> zcat [B, T_pa, 2D] fp32 (z_real | z_imag, channels-last), sigma [B, T_pa] fp32.
    latents=False: zcat = (raw head outputs [B, T_pa, 2D] 16-bit, scale, shift [B, 2D]) - the caller only pools the latents, and the
    GroupNorm behind the heads is affine per (utterance, channel); with pool_to = T (the STFT frame count) the heads and the pooling
    are one launch (ops.headpool) and the first element is already the POOLED raw output [B, T, 2D]."""
    with ops.stage("pa"):
        return _perception_forward(wave, pk, keep_sinc, latents, pool_to)


PATCH_CONV = True     # False: one gn_apply pass + implicit-GEMM convs (sfm_gemm16) per layer, as in round 1


def _convp(inp, pw, B, Lin, stride, pad, groups, dt, skip_pw=None):
    """conv16p on inp = (raw1, sc1, sh1[, raw2, sc2, sh2]) -> raw output + GroupNorm partials (and the same for the skip conv)"""
    dev = inp[0].device
    k, N = pw.ksize, pw.N
    Lout = (Lin + 2 * pad - k) // stride + 1
    P = 2 * ((Lout + 127) // 128)
    raw = torch.empty(B, Lout, N, device=dev, dtype=dt)
    part = torch.empty(B, P, groups, 2, device=dev, dtype=torch.float32)
    raw_s = part_s = None
    if skip_pw is not None:
        raw_s = torch.empty(B, Lout, N, device=dev, dtype=dt)
        part_s = torch.empty(B, P, groups, 2, device=dev, dtype=torch.float32)
    x2, sc2, sh2 = (inp[3], inp[4], inp[5]) if len(inp) == 6 else (None, None, None)
    ops.conv16p(inp[0], inp[1], inp[2], pw, raw, B=B, Lin=Lin, stride=stride, pad=pad, x2=x2, sc2=sc2, sh2=sh2,
                gn_partial=part, gn_group=N // groups, skip_pw=skip_pw, out_s=raw_s, gn_partial_s=part_s)
    return raw, part, P, Lout, raw_s, part_s


def _patch_conv_ok(pk):
    if not PATCH_CONV or pk["C0"] != 64:
        return False
    cin = 64
    for i, bp in enumerate(pk["blocks"]):
        C = bp["cout"]
        fused_skip = C == 128
        if not ops.conv16p_supported(cin, C, 7, 2, 3, i > 0, fused_skip):
            return False
        if not fused_skip and not ops.conv16p_supported(cin, C, 1, 2, 0, i > 0, False):
            return False
        if not ops.conv16p_supported(C, C, 3, 1, 1, False, False):
            return False
        cin = C
    return ops.conv16p_supported(cin, pk["D"], 5, 2, 2, True, False)


def _perception_forward(wave, pk, keep_sinc, latents, pool_to=None):
    dt = ops.compute_dtype()
    wave = wave.contiguous()
    B, L = wave.shape
    C0, K = pk["C0"], pk["K"]
    dev = wave.device
    filt, Wt = ops.sinc_filters(pk["low_hz_"], pk["band_hz_"], pk["window"], pk["n_"], C0, K, pk["fs"], 50.0, 50.0,
                                want_filt=True)
    raw = torch.empty(B, L, C0, device=dev, dtype=dt)
    if C0 == 64 and K + 7 <= 272 and SINC_ON_16BIT_MFMA:
        # split-operand 16-bit MFMA FIR (~fp32 accuracy, 3 passes of the 16x faster matrix rate)
        part, P0 = ops.sinc_fir16(wave, filt, raw, B, L, C0, K)
    else:
        P0 = 4 * ((L + 127) // 128)
        part = torch.empty(B, P0, 8, 2, device=dev, dtype=torch.float32)
        ops.framed_gemm(wave, Wt, raw, B=B, M=L, Ls=L, sig_batch_stride=L, hop=1, padl=K // 2, K=K, N=C0,
                        o_batch_stride=L * C0, ldm=C0, ldn=1, mode=0, gn_partial=part, gn_group=C0 // 8)
    sc, sh = ops.gn_finalize(part, pk["sn_w"], pk["sn_b"], B, P0, 8, C0, L)
    Lc = L
    D = pk["D"]
    if _patch_conv_ok(pk):
        # Every conv reads the RAW output of its producer(s) and applies their GroupNorm (+ the residual add) + GELU while it
        # stages its operand (sfm_conv16p): the activations between the layers are never materialised.
        inp = (raw, sc, sh)
        for bp in pk["blocks"]:
            G, C = bp["groups"], bp["cout"]
            if C == 128:                                  # main k7 s2 conv and the 1x1 s2 skip conv from one staged input
                r1, p1, P1, L1, rs, ps = _convp(inp, bp["c1"], B, Lc, 2, 3, G, dt, skip_pw=bp["cs"])
            else:
                r1, p1, P1, L1, _, _ = _convp(inp, bp["c1"], B, Lc, 2, 3, G, dt)
                rs, ps, _, _, _, _ = _convp(inp, bp["cs"], B, Lc, 2, 0, G, dt)
            s1, h1 = ops.gn_finalize(p1, bp["g1w"], bp["g1b"], B, P1, G, C, L1)
            ss, hs = ops.gn_finalize(ps, bp["gsw"], bp["gsb"], B, P1, G, C, L1)
            r2, p2, P2, _, _, _ = _convp((r1, s1, h1), bp["c2"], B, L1, 1, 1, G, dt)
            s2, h2 = ops.gn_finalize(p2, bp["g2w"], bp["g2b"], B, P2, G, C, L1)
            inp = (r2, s2, h2, rs, ss, hs)
            Lc = L1
        rd, pd, Pd, Tpa, _, _ = _convp(inp, pk["down"], B, Lc, 2, 2, 16, dt)
    else:
        x = torch.empty(B, L, C0, device=dev, dtype=dt)
        ops.gn_apply(raw, sc, sh, x, B, L, C0, act=1)
        del raw
        for bp in pk["blocks"]:
            G, C = bp["groups"], bp["cout"]
            r1, p1, P1, L1 = _conv_gn(x, bp["c1"], B, Lc, 2, 3, G, dt)
            s1, h1 = ops.gn_finalize(p1, bp["g1w"], bp["g1b"], B, P1, G, C, L1)
            a1 = torch.empty(B, L1, C, device=dev, dtype=dt)
            ops.gn_apply(r1, s1, h1, a1, B, L1, C, act=1)
            r2, p2, P2, _ = _conv_gn(a1, bp["c2"], B, L1, 1, 1, G, dt)
            rs, ps, Ps, _ = _conv_gn(x, bp["cs"], B, Lc, 2, 0, G, dt)
            s2, h2 = ops.gn_finalize(p2, bp["g2w"], bp["g2b"], B, P2, G, C, L1)
            ss, hs = ops.gn_finalize(ps, bp["gsw"], bp["gsb"], B, Ps, G, C, L1)
            x = torch.empty(B, L1, C, device=dev, dtype=dt)
            ops.gn_apply(r2, s2, h2, x, B, L1, C, act=1, x2=rs, sc2=ss, sh2=hs)
            Lc = L1
        rd, pd, Pd, Tpa = _conv_gn(x, pk["down"], B, Lc, 2, 2, 16, dt)
    sd_, hd_ = ops.gn_finalize(pd, pk["dn_w"], pk["dn_b"], B, Pd, 16, D, Tpa)
    xd = torch.empty(B, Tpa, D, device=dev, dtype=dt)
    ops.gn_apply(rd, sd_, hd_, xd, B, Tpa, D, act=1)
    # complex latent heads: one GEMM for (real | imag), GroupNorm(16) per half = 32 groups over 2D channels
    # (raw output fp32 when the full-rate latents are returned; otherwise in the operands' format: it is only pooled, and the
    #  GroupNorm statistics come from the fp32 accumulators either way)
    tiles = None
    if not latents and pool_to is not None and D == 256 and pk["zproj"].Npad == 2 * D and pk["zproj"].ksize == 1:
        tiles = ops.headpool_tiles(Tpa, pool_to)
    if tiles is not None:
        # heads + time pooling in one launch: the full-rate raw latents are never written (their GroupNorm statistics are)
        rz = torch.empty(B, pool_to, 2 * D, device=dev, dtype=dt)
        pz = torch.empty(B, tiles[1], 32, 2, device=dev, dtype=torch.float32)
        ops.headpool(xd, pk["zproj"], rz, pz, B, Tpa, pool_to, gcols=(2 * D) // 32)
        Pz = tiles[1]
    else:
        rz, pz, Pz, _ = _conv_gn(xd, pk["zproj"], B, Tpa, 1, 0, 32, torch.float32 if latents else dt)
    sz, hz = ops.gn_finalize(pz, pk["z_w"], pk["z_b"], B, Pz, 32, 2 * D, Tpa)
    if latents:
        zcat = torch.empty(B, Tpa, 2 * D, device=dev, dtype=torch.float32)
        ops.gn_apply(rz, sz, hz, zcat, B, Tpa, 2 * D, act=0)
    else:
        # latents=False: the caller only pools the latents (glue G1); GroupNorm without activation is affine per
        # (utterance, channel), so it commutes with the average: hand back raw output + scale/shift instead
        zcat = (rz, sz, hz)
    # uncertainty head
    u = torch.empty(B, Tpa, pk["u0"].N, device=dev, dtype=dt)
    ops.gemm16(xd, pk["u0"], u, B=B, Lout=Tpa, Lin=Tpa, a_batch_stride=Tpa * D, ldo=pk["u0"].N,
               o_batch_stride=Tpa * pk["u0"].N, stride=1, pad=1, epi=ops.EPI_GELU)
    sigma = torch.empty(B * Tpa, 1, device=dev, dtype=torch.float32)
    ops.linear16(u.reshape(B * Tpa, -1), pk["u2"], epi=ops.EPI_SIGMA, out=sigma)
    return zcat, sigma.reshape(B, Tpa)


# ---------------------------------------------------------------------------
# CPEA (agents/cpea.py:79-112)
# ---------------------------------------------------------------------------
def pack_cpea(sd, num_layers=2):
    with ops.stage("front"):
        return _pack_cpea(sd, num_layers)


def _pack_cpea(sd, num_layers):
    pk = {"layers": [], "H": sd["lstm.weight_hh_l0"].shape[1]}
    for l in range(num_layers):
        wih = torch.cat([sd["lstm.weight_ih_l%d" % l], sd["lstm.weight_ih_l%d_reverse" % l]], dim=0)
        bias = torch.cat([sd["lstm.bias_ih_l%d" % l] + sd["lstm.bias_hh_l%d" % l],
                          sd["lstm.bias_ih_l%d_reverse" % l] + sd["lstm.bias_hh_l%d_reverse" % l]], dim=0)
        whh = torch.stack([_f32(sd["lstm.weight_hh_l%d" % l]), _f32(sd["lstm.weight_hh_l%d_reverse" % l])], dim=0)
        pk["layers"].append({"wih": ops.pack_linear(wih, bias), "whh": whh.contiguous()})
    wh = torch.cat([sd["rho_s_head.0.weight"], sd["rho_n_head.0.weight"], sd["phi1_head.0.weight"],
                    sd["phi2_head.0.weight"]], dim=0)
    bh = torch.cat([sd["rho_s_head.0.bias"], sd["rho_n_head.0.bias"], sd["phi1_head.0.bias"], sd["phi2_head.0.bias"]], dim=0)
    pk["heads"] = ops.pack_linear(wh, bh)
    pk["oc"] = sd["rho_s_head.0.weight"].shape[0]
    return pk


def cpea_forward(z16, pk, B, T, out=None):
    """z16: [B*T, ld] 16-bit rows whose first input_dim columns are the latent ->
    [B*T, 4*oc] (rho_s | rho_n | phi1 | phi2); `out` may be a strided 16-bit/fp32 view."""
    with ops.stage("front"):
        return _cpea_forward(z16, pk, B, T, out)


def _cpea_forward(z16, pk, B, T, out):
    H = pk["H"]
    M = B * T
    dt = ops.compute_dtype()
    x16 = z16
    for lp in pk["layers"]:
        xg = ops.linear16(x16, lp["wih"], out_dtype=torch.float32)                  # [M, 8H] = [B,T,2,4H]
        h = ops.bilstm_layer(xg, lp["whh"], B, T, H, w16=ops.lstm_w16())            # [B, T, 2H] fp32
        x16 = torch.empty(M, 2 * H, device=h.device, dtype=dt)
        ops.convert_rows(h, x16, M, 2 * H, 2 * H, 2 * H, 2 * H)
    oc = pk["oc"]
    if out is None:
        out = torch.empty(M, 4 * oc, device=x16.device, dtype=torch.float32)
    ops.linear16(x16, pk["heads"], epi=ops.EPI_CPEA, alpha=math.pi, nsplit=2 * oc, out=out)
    return out


# ---------------------------------------------------------------------------
# EpisodicMemory (agents/memory.py:95-148)
# ---------------------------------------------------------------------------
def pack_memory_params(sd):
    order = ["key_proj.0.weight", "key_proj.0.bias", "key_proj.1.weight", "key_proj.1.bias", "key_proj.3.weight",
             "key_proj.3.bias", "keys", "values", "value_proj.0.weight", "value_proj.0.bias", "gate.0.weight",
             "gate.0.bias"]
    return torch.cat([_f32(sd[k]).reshape(-1) for k in order]).contiguous()


# ---------------------------------------------------------------------------
# MaskSynthesisAgent (agents/msa.py:106-174)
# ---------------------------------------------------------------------------
def pack_msa(sd, num_blocks, num_heads=4):
    with ops.stage("front"):
        pk = {"f0": ops.pack_linear(sd["fusion.0.weight"], sd["fusion.0.bias"], k_pad_to=FUSE_LD),
              "f1w": _f32(sd["fusion.1.weight"]), "f1b": _f32(sd["fusion.1.bias"]),
              "f3": ops.pack_linear(sd["fusion.3.weight"], sd["fusion.3.bias"]),
              "f4w": _f32(sd["fusion.4.weight"]), "f4b": _f32(sd["fusion.4.bias"])}
    pk["conf"] = pack_complex_conformer(sub(sd, "conformer"), num_blocks, num_heads)
    with ops.stage("tail"):
        pk.update({"r0": ops.pack_linear(sd["mask_proj_real.0.weight"], sd["mask_proj_real.0.bias"]),
                   "r2": ops.pack_linear(sd["mask_proj_real.2.weight"], sd["mask_proj_real.2.bias"]),
                   "i0": ops.pack_linear(sd["mask_proj_imag.0.weight"], sd["mask_proj_imag.0.bias"]),
                   "i2": ops.pack_linear(sd["mask_proj_imag.2.weight"], sd["mask_proj_imag.2.bias"])})
    pk["d_model"] = sd["fusion.3.weight"].shape[0]
    return pk


def msa_logits(fused16, pk, B, T, H):
    """fused16 [M, FUSE_LD] 16-bit (the 8-way concat of agents/msa.py:140, zero padded)
    -> magnitude / phase logits [M, 129] fp32 each."""
    D = pk["d_model"]
    with ops.stage("front"):
        h = ops.linear16(fused16, pk["f0"], out_dtype=torch.float32)
        h16 = _ln16(h, pk["f1w"], pk["f1b"], act=1)
        h = ops.linear16(h16, pk["f3"], out_dtype=torch.float32)
        hf16 = _ln16(h, pk["f4w"], pk["f4b"])
    y16 = complex_conformer_core(hf16, pk["conf"], B, T, H)   # [M, D]: mask_r | mask_i, 16-bit in the tail's format
    half = D // 2
    with ops.stage("tail"):
        gr = ops.linear16(y16[:, :half], pk["r0"], epi=ops.EPI_GELU)
        lm = ops.linear16(gr, pk["r2"], out_dtype=torch.float32)
        gi = ops.linear16(y16[:, half:], pk["i0"], epi=ops.EPI_GELU)
        lp = ops.linear16(gi, pk["i2"], out_dtype=torch.float32)
    return lm, lp


def msa_pack_inputs(z_real, z_imag, cpea, noisy_real, noisy_imag):
    """Module-API inputs (reference layouts) -> fused 16-bit operand [B*T, FUSE_LD]."""
    with ops.stage("front"):
        return _msa_pack_inputs(z_real, z_imag, cpea, noisy_real, noisy_imag)


def _msa_pack_inputs(z_real, z_imag, cpea, noisy_real, noisy_imag):
    B, D, T = z_real.shape
    M = B * T
    dev = z_real.device
    fused = torch.empty(M, FUSE_LD, device=dev, dtype=ops.compute_dtype())
    zr, zi = z_real.float().contiguous(), z_imag.float().contiguous()
    ops.transpose(zr, fused, B, D, T, D * T, T, T * FUSE_LD, FUSE_LD)
    ops.transpose(zi, fused[:, D:], B, D, T, D * T, T, T * FUSE_LD, FUSE_LD)
    col = 2 * D
    for k in ("rho_s", "rho_n", "phi1", "phi2"):
        c = cpea[k].float().contiguous()
        oc = c.shape[-1]
        ops.convert_rows(c, fused[:, col:], M, oc, oc, oc, FUSE_LD)
        col += oc
    nr, ni = noisy_real.float().contiguous(), noisy_imag.float().contiguous()
    F = nr.shape[-1]
    ops.stft_lognorm_pack(nr, ni, fused[:, col:], M, F, FUSE_LD - col - 2 * F, FUSE_LD)
    return fused, nr, ni


PHASE_SCALE_MSA = 3.14159 / 8.0     # literal of agents/msa.py:168


def msa_forward(z_real, z_imag, cpea, noisy_real, noisy_imag, pk, H, mag_bias=None):
    B, D, T = z_real.shape
    fused, nr, ni = msa_pack_inputs(z_real, z_imag, cpea, noisy_real, noisy_imag)
    lm, lp = msa_logits(fused, pk, B, T, H)
    F = nr.shape[-1]
    mr = torch.empty(B, T, F, device=lm.device, dtype=torch.float32)
    mi = torch.empty(B, T, F, device=lm.device, dtype=torch.float32)
    ops.polar_mask(lm, lp, B, T, F, PHASE_SCALE_MSA, lm.stride(0), mag_bias=mag_bias, mr=mr, mi=mi)
    return mr, mi


# ---------------------------------------------------------------------------
# North-star composition with the build-defined glue
# ---------------------------------------------------------------------------
def enhance_path(wave, packs, H=4, use_memory=False, want=("mask", "wave")):
    """wave [B, L] fp32 -> dict(mask_real, mask_imag [B,T,129], enhanced [B,L], ...).
    packs: dict(pa=, cpea=, msa=[, memory=(params, kd, vd, slots, temp)])."""
    dt = ops.stage_dtype("front")
    wave = wave.contiguous()
    B, L = wave.shape
    dev = wave.device
    T = 1 + L // HOP
    M = B * T
    pa = packs["pa"]
    D = pa["D"]
    want_lat = "latents" in want
    zcat, sigma = perception_forward(wave, pa, latents=want_lat, pool_to=T)   # [B, Tpa, 2D] fp32 (or raw / pooled raw + GroupNorm affine)
    zsrc, zsc, zsh = (zcat, None, None) if want_lat else zcat
    Tpa = zsrc.shape[1]                                                # (= T when the heads kernel pooled already: identity windows below)
    fused = torch.empty(M, FUSE_LD, device=dev, dtype=dt)
    with ops.stage("front"):
        ops.pool_time(zsrc, fused, None, B, Tpa, T, 2 * D, 2 * D, FUSE_LD, scale=zsc, shift=zsh)   # G1 -> fused[:, :2D]
    oc4 = 4 * packs["cpea"]["oc"]
    cpea_forward(fused, packs["cpea"], B, T, out=fused[:, 2 * D:2 * D + oc4])   # CPEA(z_real pooled) -> fused cols
    nr, ni = (stft_split16 if FUSED_STFT_SPLIT16 else stft)(wave)
    col = 2 * D + oc4
    with ops.stage("front"):
        ops.stft_lognorm_pack(nr, ni, fused[:, col:], M, N_FREQ, FUSE_LD - col - 2 * N_FREQ, FUSE_LD)
    bias = None
    out = {}
    if use_memory:
        params, kd, vd, slots, temp = packs["memory"]
        zpool = torch.empty(B, T, kd, device=dev, dtype=torch.float32)  # fp32 pooled z_real columns the key is made of
        ops.pool_time(zsrc, None, zpool, B, Tpa, T, kd, 2 * D, kd, scale=zsc[:, :kd].contiguous(), shift=zsh[:, :kd].contiguous())
        emb_r = ops.mean_time(zpool, B, T, kd, kd)                      # G2: key = mean over frames
        bias, gate, top, sim = ops.memory_fwd(emb_r, params, kd, vd, slots, temp)
        out.update(mem_bias=bias, mem_gate=gate, mem_top=top, mem_sim=sim)
    lm, lp = msa_logits(fused, packs["msa"], B, T, H)
    mr = torch.empty(B, T, N_FREQ, device=dev, dtype=torch.float32)
    mi = torch.empty(B, T, N_FREQ, device=dev, dtype=torch.float32)
    ld = ops.round_up(2 * N_FREQ, 8)
    spec = torch.zeros(M, ld, device=dev, dtype=torch.float32)
    ops.polar_mask(lm, lp, B, T, N_FREQ, PHASE_SCALE_MSA, lm.stride(0), mag_bias=bias, nr=nr, ni=ni, mr=mr, mi=mi,
                   er=spec, ei=spec[:, N_FREQ:], ld_enh=ld)            # G3 bias; enhanced spectrum packed for iSTFT
    out.update(mask_real=mr, mask_imag=mi, sigma=sigma, noisy_real=nr, noisy_imag=ni, spec=spec)
    if want_lat:
        out["zcat"] = zcat
    if "wave" in want:
        out["enhanced"] = (istft_from_packed_split16 if FUSED_STFT_SPLIT16 else istft_from_packed)(spec, B, T, L)
    return out


# ---------------------------------------------------------------------------
# Training objective, forward only (training/conformer_pipeline.py:52-108, 539-572)
# ---------------------------------------------------------------------------
MR_STFT = ((256, 64, 256), (512, 128, 512), (1024, 256, 1024))



_COUNTS = {}


def counts_tensor(counts, device):
    """element counts of the multi-resolution spectra as a device tensor, cached per shape: a torch.tensor(list, device=)
    per step is a blocking pageable copy, i.e. one host-device synchronisation per step"""
    key = (counts, str(device))
    t = _COUNTS.get(key)
    if t is None:
        t = _COUNTS[key] = torch.tensor(list(counts), device=device, dtype=torch.int64)
    return t


def enhancer_loss(enh_real, enh_imag, clean_wave, clean_real, clean_imag):
    """_compute_loss after the model call: iSTFT -> SI-SNR + 0.5 * L1 magnitude + multi-resolution STFT.
    Returns (losses [4] = total, neg_sisnr, l1_mag, mr_stft — a device tensor, no host sync), enhanced waveform."""
    B, L = clean_wave.shape
    enh_wav = istft(enh_real, enh_imag, L)
    Sw = ops.wave_moments(enh_wav, clean_wave.contiguous())
    Sm = ops.spec_sums(enh_real.contiguous(), enh_imag.contiguous(), clean_real.contiguous(), clean_imag.contiguous())
    Sr = torch.zeros(len(MR_STFT), 4, device=clean_wave.device, dtype=torch.float64)
    counts = []
    for i, (nf, hp, wn) in enumerate(MR_STFT):
        pr, pi = stft(enh_wav, nf, hp, wn)
        tr, ti = stft(clean_wave, nf, hp, wn)
        s = ops.spec_sums(pr, pi, tr, ti)
        Sr[i].copy_(s)
        counts.append(pr.numel())
    nr = counts_tensor(tuple(counts), clean_wave.device)
    return ops.enhancer_loss_finalize(Sw, Sm, Sr, nr, B, L, enh_real.numel()), enh_wav
