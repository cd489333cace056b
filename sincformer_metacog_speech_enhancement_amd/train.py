"""Training-mode ConformerBlock (forward with dropout / BatchNorm batch statistics + full backward) on the
HIP kernels, exposed to torch autograd as ONE Function per block so that the reference's
training/conformer_pipeline.py (autocast + GradScaler + loss.backward()) can train with it unchanged.

Backward of models/conformer.py:28-151 (reference line numbers in the per-module functions).
Dropout uses a counter-based keep function (statistically equivalent to torch's Philox stream, not
bit-identical: SURVEY H5); with p = 0 the gradients match torch autograd of the reference math
(tests/test_train_gpu.py).  Gradients flow between kernels in fp32 for [M, D] tensors and in the 16-bit
compute dtype where they feed the matrix cores.
"""
import math
import os
import torch

from . import ops

_NAMES = [
    "ff1.layer_norm.weight", "ff1.layer_norm.bias", "ff1.linear1.weight", "ff1.linear1.bias",
    "ff1.linear2.weight", "ff1.linear2.bias",
    "mhsa.layer_norm.weight", "mhsa.layer_norm.bias", "mhsa.attention.in_proj_weight", "mhsa.attention.in_proj_bias",
    "mhsa.attention.out_proj.weight", "mhsa.attention.out_proj.bias",
    "conv.layer_norm.weight", "conv.layer_norm.bias", "conv.pointwise1.weight", "conv.pointwise1.bias",
    "conv.depthwise.weight", "conv.depthwise.bias", "conv.batch_norm.weight", "conv.batch_norm.bias",
    "conv.pointwise2.weight", "conv.pointwise2.bias",
    "ff2.layer_norm.weight", "ff2.layer_norm.bias", "ff2.linear1.weight", "ff2.linear1.bias",
    "ff2.linear2.weight", "ff2.linear2.bias",
    "final_norm.weight", "final_norm.bias",
]
PARAM_NAMES = tuple(_NAMES)


def _f32(t):
    return t.detach().float().contiguous()


def _lin_pack(w2d, b):
    """forward pack (y = x W^T + b) and dgrad pack (dx = dy W)"""
    return ops.pack_linear(w2d, b), ops.pack_linear(w2d.t().contiguous())


def _ln16(x32, w, b):
    out = torch.empty(x32.shape, device=x32.device, dtype=ops.compute_dtype())
    ops.layernorm(x32, w, b, out16=out)
    return out


def _resid_gemm(a16, pw, x32, alpha, p, seed):
    """x + alpha * dropout(a16 @ W^T + b): the residual add and the branch dropout ride in the GEMM epilogue (same
    counter-based keep(seed, m*N + n) that the backward's ew_train(EW_SCALE_DROP) evaluates)."""
    return ops.linear16(a16, pw, epi=ops.EPI_RESID, resid=x32, alpha=alpha, out_dtype=torch.float32, p_drop=p, seed=seed)


class _Seeds:
    def __init__(self, base):
        self.base, self.i = int(base) & 0x7FFFFFFF, 0

    def next(self):
        self.i += 1
        return (self.base * 2654435761 + self.i * 40503) & 0xFFFFFFFF


# ---------------------------------------------------------------------------
# FeedForwardModule (models/conformer.py:41-49)
# ---------------------------------------------------------------------------
FUSE_FFN_SWISH = os.environ.get("SFM_FUSE_FFN_SWISH", "1") != "0"     # Swish (+ hidden dropout) of the FFN in the epilogues of its GEMMs (forward: dual output; backward)


def _ffn_fwd(x, P, pre, p, seeds):
    dt = ops.compute_dtype()
    M, D = x.shape
    lw, lb = _f32(P[pre + "layer_norm.weight"]), _f32(P[pre + "layer_norm.bias"])
    f1, b1 = _lin_pack(_f32(P[pre + "linear1.weight"]), _f32(P[pre + "linear1.bias"]))
    f2, b2 = _lin_pack(_f32(P[pre + "linear2.weight"]), _f32(P[pre + "linear2.bias"]))
    h16 = _ln16(x, lw, lb)
    s1, s2 = seeds.next(), seeds.next()
    if FUSE_FFN_SWISH and f1.N % 8 == 0:
        # one epilogue writes u = dropout(swish(z)) and, in the slot of the pre-activation, the derivative factor
        # d = keep * swish'(z): the backward's epilogue is then a single multiply
        z1, u = ops.linear16_swish(h16, f1, p_drop=p, seed=s1)
    else:
        z1 = ops.linear16(h16, f1)                                        # [M, FF] 16-bit pre-activation
        u = torch.empty_like(z1)
        ops.ew_train(ops.EW_SWISH_FWD, u, z=z1, p=p, seed=s1)
    y = _resid_gemm(u, f2, x, 0.5, p, s2)                                 # x + 0.5 * dropout(linear2(u)), one launch
    return y, dict(x=x, lw=lw, h16=h16, z1=z1, u=u, b1=b1, b2=b2, s1=s1, s2=s2, p=p)


FUSE_NEXT_DROP = os.environ.get("SFM_FUSE_NEXT_DROP", "1") != "0"     # A/B knob: the LayerNorm backward also writes the next node's 16-bit operand


def _branch_grad(dy, do, alpha, p, seed):
    """alpha * dropout(dy) in the compute format: handed over by the previous node's LayerNorm backward (`do`), or one pass over dy"""
    if do is not None:
        return do
    do = torch.empty(dy.shape, device=dy.device, dtype=ops.compute_dtype())
    ops.ew_train(ops.EW_SCALE_DROP, do, g=dy, alpha=alpha, p=p, seed=seed)
    return do


def _ln_bwd_chain(x, lw, dh, dy, gw, gb, nxt):
    """LayerNorm backward at the end of a module's backward; nxt = (alpha, p, seed) of the NEXT node's branch dropout -> (dx, do16)"""
    if nxt is not None and FUSE_NEXT_DROP:
        return ops.layernorm_bwd(x, lw, dh, dy, gw, gb, next_drop=nxt)
    return ops.layernorm_bwd(x, lw, dh, dy, gw, gb), None


def _ffn_bwd(dy, c, G, pre, do=None, nxt=None):
    dt = ops.compute_dtype()
    M, D = dy.shape
    FF = c["z1"].shape[1]
    do = _branch_grad(dy, do, 0.5, c["p"], c["s2"])
    # input gradient first, weight gradient second: the K = 1024 weight-gradient GEMM launched right behind the streaming
    # ew_train ran 0.28 ms against 0.21 ms behind a GEMM (tools/ffn_bwd_probe.py --variant late; profiles/README.md, round 3)
    if FUSE_FFN_SWISH and FF % 8 == 0:
        dz = ops.linear16_swish(do, c["b2"], p_drop=c["p"], seed=c["s1"], aux=c["z1"])   # (do W2) * d, d saved by the forward
    else:
        du = ops.linear16(do, c["b2"], out_dtype=torch.float32)             # [M, FF]
        dz = torch.empty(M, FF, device=dy.device, dtype=dt)
        ops.ew_train(ops.EW_SWISH_BWD, dz, z=c["z1"], g=du, p=c["p"], seed=c["s1"])
    ops.gemm16_tn(do, c["u"], G[pre + "linear2.weight"], G[pre + "linear2.bias"])
    ops.gemm16_tn(dz, c["h16"], G[pre + "linear1.weight"], G[pre + "linear1.bias"])
    dh = ops.linear16(dz, c["b1"])                                          # [M, D] 16-bit: read once, by the LayerNorm backward
    return _ln_bwd_chain(c["x"], c["lw"], dh, dy, G[pre + "layer_norm.weight"], G[pre + "layer_norm.bias"], nxt)


# ---------------------------------------------------------------------------
# MultiHeadSelfAttention (models/conformer.py:66-71)
# ---------------------------------------------------------------------------
def _mhsa_fwd(x, P, B, T, H, p, seeds):
    M, D = x.shape
    hd = D // H
    lw, lb = _f32(P["mhsa.layer_norm.weight"]), _f32(P["mhsa.layer_norm.bias"])
    qs = ops.ATTN_QSCALE_LOG2E / math.sqrt(hd)
    w, bias = _f32(P["mhsa.attention.in_proj_weight"]).clone(), _f32(P["mhsa.attention.in_proj_bias"]).clone()
    w[:D] *= qs
    bias[:D] *= qs
    fin, bin_ = _lin_pack(w, bias)
    fout, bout = _lin_pack(_f32(P["mhsa.attention.out_proj.weight"]), _f32(P["mhsa.attention.out_proj.bias"]))
    h16 = _ln16(x, lw, lb)
    qkv = ops.linear16(h16, fin)
    sa, sd = seeds.next(), seeds.next()
    O, lse = ops.attention_train(qkv, B, T, H, hd, p_drop=p, seed=sa)
    y = _resid_gemm(O, fout, x, 1.0, p, sd)
    return y, dict(x=x, lw=lw, h16=h16, qkv=qkv, O=O, lse=lse, bin=bin_, bout=bout, sa=sa, sd=sd, p=p, qs=qs, B=B, T=T, H=H)


def _mhsa_bwd(dy, c, G, do=None, nxt=None):
    dt = ops.compute_dtype()
    M, D = dy.shape
    do = _branch_grad(dy, do, 1.0, c["p"], c["sd"])
    ops.gemm16_tn(do, c["O"], G["mhsa.attention.out_proj.weight"], G["mhsa.attention.out_proj.bias"])
    dO = ops.linear16(do, c["bout"])                                         # 16-bit [M, D]
    dqkv = ops.attention_bwd(c["qkv"], c["O"], dO, c["lse"], c["B"], c["T"], c["H"], D // c["H"], p_drop=c["p"], seed=c["sa"])
    gw, gb = G["mhsa.attention.in_proj_weight"], G["mhsa.attention.in_proj_bias"]
    # the Q rows of W were multiplied by qs in the forward: accumulate into scratch, then scale those rows
    tw = torch.zeros_like(gw)
    tb = torch.zeros_like(gb)
    ops.gemm16_tn(dqkv, c["h16"], tw, tb)
    with ops.after_wgrad(tw, tb):
        tw[:D] *= c["qs"]
        tb[:D] *= c["qs"]
        gw += tw
        gb += tb
    dh = ops.linear16(dqkv, c["bin"])
    return _ln_bwd_chain(c["x"], c["lw"], dh, dy, G["mhsa.layer_norm.weight"], G["mhsa.layer_norm.bias"], nxt)


# ---------------------------------------------------------------------------
# ConvolutionModule (models/conformer.py:101-128), BatchNorm1d in training mode
# ---------------------------------------------------------------------------
def _conv_fwd(x, P, B, T, p, seeds, bn_buffers, momentum=0.1, eps=1e-5, bn_eval=False):
    dt = ops.compute_dtype()
    M, D = x.shape
    lw, lb = _f32(P["conv.layer_norm.weight"]), _f32(P["conv.layer_norm.bias"])
    f1, b1 = _lin_pack(_f32(P["conv.pointwise1.weight"]).reshape(2 * D, D), _f32(P["conv.pointwise1.bias"]))
    f2, b2 = _lin_pack(_f32(P["conv.pointwise2.weight"]).reshape(D, D), _f32(P["conv.pointwise2.bias"]))
    wdw = _f32(P["conv.depthwise.weight"]).reshape(D, -1)
    KS = wdw.shape[1]
    bdw = _f32(P["conv.depthwise.bias"])
    gam, bet = _f32(P["conv.batch_norm.weight"]), _f32(P["conv.batch_norm.bias"])
    h16 = _ln16(x, lw, lb)
    pre = ops.linear16(h16, f1)                                              # [M, 2D] 16-bit (a | gate)
    g16 = torch.empty(M, D, device=x.device, dtype=dt)
    ops.ew_train(ops.EW_GLU_FWD, g16, z=pre)
    ones = torch.ones(D, device=x.device)
    yc = torch.empty(M, D, device=x.device, dtype=torch.float32)
    ops.dwconv_folded(g16, wdw.t().contiguous(), ones, bdw, B, T, D, out=yc, act=0)   # conv + bias, fp32
    # statistics -> mean, rstd, folded affine and the running-statistics update (unbiased var, momentum) in one launch
    rm, rv = (bn_buffers[0], bn_buffers[1]) if bn_buffers is not None else (None, None)
    fp32_buffers = rm is not None and rm.dtype == torch.float32 and rv.dtype == torch.float32
    rm32, rv32 = (rm, rv) if fp32_buffers or rm is None else (rm.detach().float(), rv.detach().float())
    S = None if bn_eval else ops.col_stats(yc)
    mean, rstd, sc, sh = ops.bn_finalize(S, gam, bet, rm32, rv32, M, eps, momentum, eval_mode=bn_eval)
    if bn_buffers is not None and not bn_eval:
        with torch.no_grad():
            if not fp32_buffers:
                rm.copy_(rm32)
                rv.copy_(rv32)
            bn_buffers[2].add_(1)
    s16 = torch.empty(M, D, device=x.device, dtype=dt)
    ops.gn_apply(yc, sc, sh, s16, 1, M, D, act=2)                            # BatchNorm + Swish
    sd = seeds.next()
    y = _resid_gemm(s16, f2, x, 1.0, p, sd)
    return y, dict(x=x, lw=lw, h16=h16, pre=pre, g16=g16, yc=yc, mean=mean, rstd=rstd, gam=gam, bet=bet, s16=s16, b1=b1, b2=b2,
                   wdw=wdw, KS=KS, sd=sd, p=p, B=B, T=T, bn_eval=bn_eval)


def _conv_bwd(dy, c, G, do=None, nxt=None):
    dt = ops.compute_dtype()
    M, D = dy.shape
    B, T, KS = c["B"], c["T"], c["KS"]
    do = _branch_grad(dy, do, 1.0, c["p"], c["sd"])
    ops.gemm16_tn(do, c["s16"], G["conv.pointwise2.weight"].view(D, D), G["conv.pointwise2.bias"])
    ds = ops.linear16(do, c["b2"], out_dtype=torch.float32)
    dyc, dgam, dbet = ops.bn_swish_bwd(ds, c["yc"], c["mean"], c["rstd"], c["gam"], c["bet"], eval_mode=c["bn_eval"])
    G["conv.batch_norm.weight"] += dgam
    G["conv.batch_norm.bias"] += dbet
    dw, db = ops.dwconv_wgrad(c["g16"], dyc, B, T, D, KS)
    G["conv.depthwise.weight"] += dw.view_as(G["conv.depthwise.weight"])
    G["conv.depthwise.bias"] += db
    # input gradient of the depthwise conv = correlation with the flipped taps
    dyc16 = torch.empty(M, D, device=dy.device, dtype=dt)
    ops.convert_rows(dyc, dyc16, M, D, D, D, D)
    wflipT = torch.flip(c["wdw"], dims=[1]).t().contiguous()
    ones, zeros = torch.ones(D, device=dy.device), torch.zeros(D, device=dy.device)
    dg = ops.dwconv_folded(dyc16, wflipT, ones, zeros, B, T, D, act=0)
    dpre = torch.empty(M, 2 * D, device=dy.device, dtype=dt)
    ops.ew_train(ops.EW_GLU_BWD, dpre, z=c["pre"], g=dg, N=D)
    ops.gemm16_tn(dpre, c["h16"], G["conv.pointwise1.weight"].view(2 * D, D), G["conv.pointwise1.bias"])
    dh = ops.linear16(dpre, c["b1"])
    return _ln_bwd_chain(c["x"], c["lw"], dh, dy, G["conv.layer_norm.weight"], G["conv.layer_norm.bias"], nxt)


# ---------------------------------------------------------------------------
# whole block
# ---------------------------------------------------------------------------
def block_train_forward(x32, P, B, T, H, p, seed, bn_buffers=None, momentum=0.1, eps=1e-5, bn_eval=False):
    seeds = _Seeds(seed)
    x1, c1 = _ffn_fwd(x32, P, "ff1.", p, seeds)
    x2, c2 = _mhsa_fwd(x1, P, B, T, H, p, seeds)
    x3, c3 = _conv_fwd(x2, P, B, T, p, seeds, bn_buffers, momentum, eps, bn_eval)
    x4, c4 = _ffn_fwd(x3, P, "ff2.", p, seeds)
    fw, fb = _f32(P["final_norm.weight"]), _f32(P["final_norm.bias"])
    y = torch.empty_like(x4)
    ops.layernorm(x4, fw, fb, out32=y)
    return y, dict(c1=c1, c2=c2, c3=c3, c4=c4, x4=x4, fw=fw)


def _zero_grads(shapes, device):
    """{name: zero-filled fp32 tensor of that shape}, all views of ONE buffer (one fill launch instead of one per
    parameter; every view starts on a 256-byte boundary)"""
    offs, n = {}, 0
    for k, shp in shapes.items():
        offs[k] = n
        n += (int(torch.Size(shp).numel()) + 63) // 64 * 64
    flat = torch.zeros(n, device=device, dtype=torch.float32)
    return {k: flat[o:o + torch.Size(shapes[k]).numel()].view(shapes[k]) for k, o in offs.items()}


def block_train_backward(dy32, ctx, P):
    G = _zero_grads({k: P[k].shape for k in PARAM_NAMES}, dy32.device)
    # the block's weight-gradient GEMMs overlap its input-gradient chain on a second stream (B 256 x T 801: -1 ms of 51.6 per
    # step; at B 64 the extra events cost more than the overlap returns)
    with ops.wgrad_side_stream(enabled=dy32.shape[0] >= 100000):
        # every LayerNorm backward of the chain also writes the 16-bit alpha * dropout(dx) the NEXT module's backward starts from
        c1, c2, c3, c4 = ctx["c1"], ctx["c2"], ctx["c3"], ctx["c4"]
        d4, o4 = _ln_bwd_chain(ctx["x4"], ctx["fw"], dy32, None, G["final_norm.weight"], G["final_norm.bias"], (0.5, c4["p"], c4["s2"]))
        d3, o3 = _ffn_bwd(d4, c4, G, "ff2.", do=o4, nxt=(1.0, c3["p"], c3["sd"]))
        d2, o2 = _conv_bwd(d3, c3, G, do=o3, nxt=(1.0, c2["p"], c2["sd"]))
        d1, o1 = _mhsa_bwd(d2, c2, G, do=o2, nxt=(0.5, c1["p"], c1["s2"]))
        dx, _ = _ffn_bwd(d1, c1, G, "ff1.", do=o1)
    return dx, G


class ConformerBlockFunction(torch.autograd.Function):
    """y = ConformerBlock(x) in training mode; saved state lives on ctx as plain tensors (no autograd graph inside)."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        B, T, D = x.shape
        H, p, seed, bn_buffers, momentum, eps, bn_eval = meta
        P = dict(zip(PARAM_NAMES, params))
        x32 = x.detach().float().reshape(B * T, D).contiguous()
        y, saved = block_train_forward(x32, P, B, T, H, p, seed, bn_buffers, momentum, eps, bn_eval)
        ctx.saved, ctx.P, ctx.in_dtype, ctx.shape = saved, P, x.dtype, (B, T, D)
        ctx.param_dtypes = [t.dtype for t in params]
        return y.reshape(B, T, D).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        B, T, D = ctx.shape
        dx, G = block_train_backward(dy.detach().float().reshape(B * T, D).contiguous(), ctx.saved, ctx.P)
        grads = [G[k].to(dt_) for k, dt_ in zip(PARAM_NAMES, ctx.param_dtypes)]
        ctx.saved = None
        return (dx.reshape(B, T, D).to(ctx.in_dtype), None) + tuple(grads)


# ---------------------------------------------------------------------------
# LayerNorm -> Linear (input_norm + input_proj, output_norm + heads of SpeechEnhancer,
# training/conformer_pipeline.py:273-284)
# ---------------------------------------------------------------------------
class LNLinearFunction(torch.autograd.Function):
    """y[M, N] = LayerNorm(x[:, :K]) @ W^T + b, fp32 in / fp32 out, GEMMs on 16-bit operands.
    ln_w = ln_b = None: plain Linear (input_proj / output_proj of ComplexConformer, models/conformer.py:211,222)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, W, b):
        M = x.shape[0]
        N, K = W.shape
        x32 = x.detach().float()
        if x32.stride(1) != 1:
            x32 = x32.contiguous()
        has_ln = ln_w is not None
        lw, lb, w32 = (_f32(ln_w), _f32(ln_b), _f32(W)) if has_ln else (None, None, _f32(W))
        Kp = ops.round_up(K, 64)
        fwd = ops.pack_linear(w32, _f32(b), k_pad_to=Kp)
        h16 = torch.zeros(M, Kp, device=x.device, dtype=ops.compute_dtype()) if Kp != K else \
            torch.empty(M, K, device=x.device, dtype=ops.compute_dtype())
        if has_ln:
            ops.layernorm(x32, lw, lb, out16=h16)
        else:
            ops.convert_rows(x32, h16, M, K, Kp, x32.stride(0), Kp)
        Np = ops.round_up(N, 8)
        ybuf = torch.empty(M, Np, device=x.device, dtype=torch.float32)
        y = ybuf[:, :N]
        ops.linear16(h16, fwd, out=y)
        ctx.saved = (x32, lw, w32, h16)
        ctx.dims = (M, N, K)
        ctx.dtypes = (x.dtype, ln_w.dtype if has_ln else None, ln_b.dtype if has_ln else None, W.dtype, b.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x32, lw, w32, h16 = ctx.saved
        M, N, K = ctx.dims
        dev = dy.device
        dy = dy.float()
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        Np = ops.round_up(N, 64)
        dy16 = torch.empty(M, Np, device=dev, dtype=ops.compute_dtype())
        ops.convert_rows(dy, dy16, M, N, Np, dy.stride(0), Np)
        dW = torch.zeros(N, K, device=dev, dtype=torch.float32)
        db = torch.zeros(N, device=dev, dtype=torch.float32)
        ops.gemm16_tn(dy16[:, :N], h16[:, :K], dW, db)
        bwd = ops.pack_linear(w32.t().contiguous(), k_pad_to=Np)
        dh = ops.linear16(dy16, bwd, out_dtype=torch.float32 if lw is None else None)   # [M, K]; 16-bit into the LayerNorm backward
        t = ctx.dtypes
        ctx.saved = None
        if lw is None:
            return dh.to(t[0]), None, None, dW.to(t[3]), db.to(t[4])
        dg = torch.zeros(K, device=dev, dtype=torch.float32)
        dbt = torch.zeros(K, device=dev, dtype=torch.float32)
        dx = ops.layernorm_bwd(x32, lw, dh, None, dg, dbt)
        return dx.to(t[0]), dg.to(t[1]), dbt.to(t[2]), dW.to(t[3]), db.to(t[4])


class PolarMaskFunction(torch.autograd.Function):
    """bounded polar mask applied to the noisy STFT (training/conformer_pipeline.py:283-295):
    logits [M, 2F] (magnitude | phase) -> enh_real, enh_imag, mask_mag (monitoring only, not differentiable)."""

    @staticmethod
    def forward(ctx, logits, noisy_real, noisy_imag, phase_scale):
        B, T, F = noisy_real.shape
        lg = logits.detach()
        nr, ni = noisy_real.detach().float().contiguous(), noisy_imag.detach().float().contiguous()
        dev = lg.device
        er = torch.empty(B, T, F, device=dev, dtype=torch.float32)
        ei = torch.empty(B, T, F, device=dev, dtype=torch.float32)
        mm = torch.empty(B, T, F, device=dev, dtype=torch.float32)
        ops.polar_mask(lg, lg[:, F:], B, T, F, phase_scale, lg.stride(0), nr=nr, ni=ni, er=er, ei=ei, mmag=mm, ld_enh=F)
        ctx.saved = (lg, nr, ni)
        ctx.ps = phase_scale
        ctx.mark_non_differentiable(mm)
        return er, ei, mm

    @staticmethod
    def backward(ctx, der, dei, _dmm):
        lg, nr, ni = ctx.saved
        B, T, F = nr.shape
        M = B * T
        dl = torch.zeros(M, ops.round_up(2 * F, 8), device=lg.device, dtype=torch.float32)
        ops.polar_mask_bwd(lg, lg[:, F:], nr, ni, der.float().contiguous(), dei.float().contiguous(), dl, M, F, ctx.ps,
                           lg.stride(0))
        ctx.saved = None
        return dl[:, :2 * F], None, None, None


# ---------------------------------------------------------------------------
# objective: SI-SNR + 0.5 L1 magnitude + multi-resolution STFT (training/conformer_pipeline.py:52-108, 539-572)
# ---------------------------------------------------------------------------
_adj_cache = {}
# STFTs of the objective (3 resolutions x prediction/target, forward and adjoint) on split-bf16 MFMA operands
# (functional.stft_split16, ~4e-6 relative error) instead of the exact-fp32 matrix instruction: 3.6x faster.
LOSS_STFT_SPLIT16 = True


def _adjoint_consts(n_fft, win, dev):
    """transposed DFT operands of functional._stft_consts (adjoint GEMMs of stft / istft)."""
    from . import functional as Fn
    key = (n_fft, win, str(dev))
    c = _adj_cache.get(key)
    if c is None:
        base = Fn._stft_consts(n_fft, win, dev)
        F2 = n_fft + 2
        c = {"fwdT": ops.pack_f32_matrix(base["fwd"][:win, :F2].t().contiguous()),      # [2F, win]
             "fwdT16": ops.pack_split16_matrix(base["fwd"][:win, :F2].t().contiguous()),
             "invT": ops.pack_f32_matrix(base["inv"][:F2, :win].t().contiguous())}      # [win, 2F]
        _adj_cache[key] = c
    return c


_env_cache = {}


def _inv_envelope(L, T, n_fft, hop, win, dev):
    """1 / (sum of squared windows) per output sample of torch.istft (0 where the envelope vanishes)."""
    import numpy as np
    key = (L, T, n_fft, hop, win, str(dev))
    v = _env_cache.get(key)
    if v is None:
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win, dtype=np.float64) / win)
        woff = (n_fft - win) // 2
        env = np.zeros(max(L + n_fft, (T - 1) * hop + n_fft) + n_fft, dtype=np.float64)
        for t in range(T):
            env[t * hop + woff: t * hop + woff + win] += w * w
        env = env[n_fft // 2: n_fft // 2 + L]
        inv = np.where(env > 1e-11, 1.0 / np.maximum(env, 1e-30), 0.0)
        v = torch.from_numpy(inv.astype(np.float32)).to(dev)
        _env_cache[key] = v
    return v


class EnhancerLossFunction(torch.autograd.Function):
    """(enh_real, enh_imag, clean_wave, clean_real, clean_imag) -> total loss (0-dim, differentiable w.r.t. the enhanced
    spectrum) and aux = [neg_sisnr, l1_mag, mr_stft] (monitoring).  The gradient is produced during the forward pass
    (every reduction it needs is already on the device) and only scaled by the incoming grad in backward."""

    @staticmethod
    def forward(ctx, enh_real, enh_imag, clean_wave, clean_real, clean_imag, n_fft, hop, win):
        from . import functional as Fn
        er, ei = enh_real.detach().float().contiguous(), enh_imag.detach().float().contiguous()
        cw = clean_wave.detach().float().contiguous()
        cr, ci = clean_real.detach().float().contiguous(), clean_imag.detach().float().contiguous()
        B, L = cw.shape
        _, T, F = er.shape
        dev = er.device
        need_grad = enh_real.requires_grad or enh_imag.requires_grad
        enh_wav = Fn.istft(er, ei, L, n_fft, hop, win)
        Sw = ops.wave_moments(enh_wav, cw)
        Sm = ops.spec_sums(er, ei, cr, ci)
        R = len(Fn.MR_STFT)
        Sr = torch.zeros(R, 4, device=dev, dtype=torch.float64)
        counts = []
        dwave = None
        if need_grad:
            dwave = torch.empty(B, L, device=dev, dtype=torch.float32)
            ops.sisnr_bwd(enh_wav, cw, Sw, dwave)
        inv_env = _inv_envelope(L, T, n_fft, hop, win, dev) if need_grad else None
        for i, (nf, hp, wn) in enumerate(Fn.MR_STFT):
            stft_ = Fn.stft_split16 if LOSS_STFT_SPLIT16 else Fn.stft
            pr, pi = stft_(enh_wav, nf, hp, wn)
            tr, ti = stft_(cw, nf, hp, wn)
            ops.spec_sums(pr, pi, tr, ti, out=Sr[i])
            counts.append(pr.numel())
            if need_grad:
                Tr, Fr = pr.shape[1], pr.shape[2]
                Mr = B * Tr
                ld = ops.round_up(2 * Fr, 8)
                g = torch.zeros(Mr, ld, device=dev, dtype=torch.float32)
                ops.spec_loss_bwd(pr, pi, tr, ti, Sr[i], g, g[:, Fr:], Fr, ld, 0, scale=1.0 / R)
                del pr, pi, tr, ti
                frames = torch.empty(Mr, wn, device=dev, dtype=torch.float32)
                if LOSS_STFT_SPLIT16:
                    ops.framed_gemm_split16(g, _adjoint_consts(nf, wn, dev)["fwdT16"], frames, B=1, M=Mr, Ls=Mr * ld,
                                            sig_batch_stride=0, hop=ld, padl=0, o_batch_stride=0, ldm=wn, mode=0)
                else:
                    ops.framed_gemm(g, _adjoint_consts(nf, wn, dev)["fwdT"], frames, B=1, M=Mr, Ls=Mr * ld, sig_batch_stride=0,
                                    hop=ld, padl=0, K=2 * Fr, N=wn, o_batch_stride=0, ldm=wn, ldn=1, mode=0)
                ops.stft_adjoint_ola(frames, dwave, B, Tr, L, nf, hp, wn, accumulate=True,
                                     post=inv_env if i == R - 1 else None)
                del g, frames
        nr = Fn.counts_tensor(tuple(counts), dev)
        losses = ops.enhancer_loss_finalize(Sw, Sm, Sr, nr, B, L, er.numel())
        if need_grad:
            d_er = torch.empty(B, T, F, device=dev, dtype=torch.float32)
            d_ei = torch.empty(B, T, F, device=dev, dtype=torch.float32)
            ops.framed_gemm(dwave, _adjoint_consts(n_fft, win, dev)["invT"], d_er, B=B, M=T, Ls=L, sig_batch_stride=L, hop=hop,
                            padl=n_fft // 2 - (n_fft - win) // 2, K=win, N=2 * F, o_batch_stride=T * F, ldm=F, ldn=1, mode=0,
                            out2=d_ei, nsplit=F)
            ops.spec_loss_bwd(er, ei, cr, ci, Sm, d_er, d_ei, F, F, 1, accumulate=True, scale=0.5)
            ctx.grads = (d_er, d_ei)
        aux = losses[1:].clone()
        ctx.mark_non_differentiable(aux, enh_wav)
        ctx.dtypes = (enh_real.dtype, enh_imag.dtype)
        return losses[0].clone(), aux, enh_wav

    @staticmethod
    def backward(ctx, g_total, _g_aux, _g_wav):
        d_er, d_ei = ctx.grads
        ctx.grads = None
        s = g_total.float()
        return (d_er * s).to(ctx.dtypes[0]), (d_ei * s).to(ctx.dtypes[1]), None, None, None, None, None, None


# ---- the objective's three terms as stand-alone nodes (the reference's si_snr_loss / MultiResolutionSTFTLoss /
#      F.l1_loss on magnitudes can be called on their own; EnhancerLossFunction above is their fused form) ----
def _dummy_moments(dev):
    """wave moments of a (1 x 1-sample) pair with a finite SI-SNR: placeholder for terms a call does not use"""
    return torch.tensor([[0.0, 0.0, 1.0, 1.0, 0.0]], device=dev, dtype=torch.float64)


class SiSnrFunction(torch.autograd.Function):
    """si_snr_loss (training/conformer_pipeline.py:52-71): negative mean SI-SNR of estimated [.., L] against target."""

    @staticmethod
    def forward(ctx, estimated, target):
        L = estimated.shape[-1]
        e = estimated.detach().float().reshape(-1, L).contiguous()
        t = target.detach().float().reshape(-1, L).contiguous()
        B = e.shape[0]
        Sw = ops.wave_moments(e, t)
        z4 = torch.zeros(1, 4, device=e.device, dtype=torch.float64)
        nr = torch.ones(1, device=e.device, dtype=torch.int64)
        losses = ops.enhancer_loss_finalize(Sw, z4[0], z4, nr, B, L, 1, R=0)
        ctx.grad = None
        if estimated.requires_grad:
            dw = torch.empty(B, L, device=e.device, dtype=torch.float32)
            ops.sisnr_bwd(e, t, Sw, dw)
            ctx.grad = dw.reshape(estimated.shape)
        ctx.dtype = estimated.dtype
        return losses[1].clone()

    @staticmethod
    def backward(ctx, g):
        d, ctx.grad = ctx.grad, None
        return (None if d is None else (d * g.float()).to(ctx.dtype)), None


class MrStftFunction(torch.autograd.Function):
    """MultiResolutionSTFTLoss.forward (training/conformer_pipeline.py:94-108): mean over the resolutions of spectral
    convergence + L1 log-magnitude, predicted [B, L] against target; sizes = ((n_fft, hop, win), ...)."""

    @staticmethod
    def forward(ctx, predicted, target, sizes):
        from . import functional as Fn
        L = predicted.shape[-1]
        pw = predicted.detach().float().reshape(-1, L).contiguous()
        tw = target.detach().float().reshape(-1, L).contiguous()
        B, dev, R = pw.shape[0], pw.device, len(sizes)
        need_grad = predicted.requires_grad
        Sr = torch.zeros(R, 4, device=dev, dtype=torch.float64)
        dwave = torch.zeros(B, L, device=dev, dtype=torch.float32) if need_grad else None
        counts = []
        for i, (nf, hp, wn) in enumerate(sizes):
            split = LOSS_STFT_SPLIT16 and nf == wn and nf % 256 == 0
            stft_ = Fn.stft_split16 if split else Fn.stft
            pr, pi = stft_(pw, nf, hp, wn)
            tr, ti = stft_(tw, nf, hp, wn)
            ops.spec_sums(pr, pi, tr, ti, out=Sr[i])
            counts.append(pr.numel())
            if need_grad:
                Tr, Fr = pr.shape[1], pr.shape[2]
                Mr = B * Tr
                ld = ops.round_up(2 * Fr, 8)
                g = torch.zeros(Mr, ld, device=dev, dtype=torch.float32)
                ops.spec_loss_bwd(pr, pi, tr, ti, Sr[i], g, g[:, Fr:], Fr, ld, 0, scale=1.0 / R)
                frames = torch.empty(Mr, wn, device=dev, dtype=torch.float32)
                if split:
                    ops.framed_gemm_split16(g, _adjoint_consts(nf, wn, dev)["fwdT16"], frames, B=1, M=Mr, Ls=Mr * ld,
                                            sig_batch_stride=0, hop=ld, padl=0, o_batch_stride=0, ldm=wn, mode=0)
                else:
                    ops.framed_gemm(g, _adjoint_consts(nf, wn, dev)["fwdT"], frames, B=1, M=Mr, Ls=Mr * ld, sig_batch_stride=0,
                                    hop=ld, padl=0, K=2 * Fr, N=wn, o_batch_stride=0, ldm=wn, ldn=1, mode=0)
                ops.stft_adjoint_ola(frames, dwave, B, Tr, L, nf, hp, wn, accumulate=True)
        nr = Fn.counts_tensor(tuple(counts), dev)
        losses = ops.enhancer_loss_finalize(_dummy_moments(dev), Sr[0], Sr, nr, 1, 1, 1)
        ctx.grad = dwave.reshape(predicted.shape) if need_grad else None
        ctx.dtype = predicted.dtype
        return losses[3].clone()

    @staticmethod
    def backward(ctx, g):
        d, ctx.grad = ctx.grad, None
        return (None if d is None else (d * g.float()).to(ctx.dtype)), None, None


class L1MagnitudeFunction(torch.autograd.Function):
    """F.l1_loss(sqrt(er^2 + ei^2 + 1e-8), sqrt(cr^2 + ci^2 + 1e-8)) of training/conformer_pipeline.py:562-564."""

    @staticmethod
    def forward(ctx, er, ei, cr, ci):
        a = [t.detach().float().contiguous() for t in (er, ei, cr, ci)]
        dev = a[0].device
        Sm = ops.spec_sums(*a)
        nr = torch.ones(1, device=dev, dtype=torch.int64)
        losses = ops.enhancer_loss_finalize(_dummy_moments(dev), Sm, torch.zeros(1, 4, device=dev, dtype=torch.float64), nr, 1, 1,
                                            a[0].numel(), R=0)
        ctx.grads = None
        if er.requires_grad or ei.requires_grad:
            F = a[0].shape[-1]
            d_er, d_ei = torch.empty_like(a[0]), torch.empty_like(a[0])
            ops.spec_loss_bwd(a[0], a[1], a[2], a[3], Sm, d_er, d_ei, F, F, 1, accumulate=False, scale=1.0)
            ctx.grads = (d_er, d_ei)
        ctx.dtypes = (er.dtype, ei.dtype)
        return losses[2].clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is None:
            return None, None, None, None
        d_er, d_ei = ctx.grads
        ctx.grads = None
        return (d_er * g.float()).to(ctx.dtypes[0]), (d_ei * g.float()).to(ctx.dtypes[1]), None, None


class ComplexMulFunction(torch.autograd.Function):
    """ComplexConformer.apply_mask (models/conformer.py:230-245) with its backward: the gradient of a complex product
    is the incoming gradient times the conjugate of the other factor — the same HIP kernel."""

    @staticmethod
    def forward(ctx, sr, si, mr, mi):
        a = [t.detach().float().contiguous() for t in (sr, si, mr, mi)]
        ctx.saved = a
        ctx.dtypes = [t.dtype for t in (sr, si, mr, mi)]
        er, ei = ops.complex_mul(*a)
        return er.reshape(sr.shape), ei.reshape(sr.shape)

    @staticmethod
    def backward(ctx, ger, gei):
        sr, si, mr, mi = ctx.saved
        ger, gei = ger.float().contiguous(), gei.float().contiguous()
        dsr, dsi = ops.complex_mul(ger, gei, mr, -mi)
        dmr, dmi = ops.complex_mul(ger, gei, sr, -si)
        t = ctx.dtypes
        ctx.saved = None
        return (dsr.reshape(sr.shape).to(t[0]), dsi.reshape(sr.shape).to(t[1]), dmr.reshape(sr.shape).to(t[2]),
                dmi.reshape(sr.shape).to(t[3]))


# ---------------------------------------------------------------------------
# stand-alone sub-modules (models/conformer.py:28-128) in training mode: the same forward / backward pieces as the block
# ---------------------------------------------------------------------------
_SUB = {
    "ffn": ("ff1.", ["layer_norm.weight", "layer_norm.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias"]),
    "mhsa": ("mhsa.", ["layer_norm.weight", "layer_norm.bias", "attention.in_proj_weight", "attention.in_proj_bias",
                       "attention.out_proj.weight", "attention.out_proj.bias"]),
    "conv": ("conv.", ["layer_norm.weight", "layer_norm.bias", "pointwise1.weight", "pointwise1.bias", "depthwise.weight",
                       "depthwise.bias", "batch_norm.weight", "batch_norm.bias", "pointwise2.weight", "pointwise2.bias"]),
}


def submodule_param_names(kind):
    return list(_SUB[kind][1])


class SubmoduleFunction(torch.autograd.Function):
    """FeedForwardModule / MultiHeadSelfAttention / ConvolutionModule in train() mode (or eval() under autograd)."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        kind, H, p, seed, bn_buffers, momentum, eps, bn_eval = meta
        prefix, names = _SUB[kind]
        B, T, D = x.shape
        P = {prefix + n: t for n, t in zip(names, params)}
        x32 = x.detach().float().reshape(B * T, D).contiguous()
        seeds = _Seeds(seed)
        if kind == "ffn":
            y, c = _ffn_fwd(x32, P, prefix, p, seeds)
        elif kind == "mhsa":
            y, c = _mhsa_fwd(x32, P, B, T, H, p, seeds)
        else:
            y, c = _conv_fwd(x32, P, B, T, p, seeds, bn_buffers, momentum, eps, bn_eval)
        ctx.saved, ctx.P, ctx.kind = c, P, kind
        ctx.shape, ctx.in_dtype, ctx.param_dtypes = (B, T, D), x.dtype, [t.dtype for t in params]
        return y.reshape(B, T, D).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        B, T, D = ctx.shape
        prefix, names = _SUB[ctx.kind]
        G = _zero_grads({prefix + n: ctx.P[prefix + n].shape for n in names}, dy.device)
        d = dy.detach().float().reshape(B * T, D).contiguous()
        if ctx.kind == "ffn":
            dx, _ = _ffn_bwd(d, ctx.saved, G, prefix)
        elif ctx.kind == "mhsa":
            dx, _ = _mhsa_bwd(d, ctx.saved, G)
        else:
            dx, _ = _conv_bwd(d, ctx.saved, G)
        grads = [G[prefix + n].to(t) for n, t in zip(names, ctx.param_dtypes)]
        ctx.saved = None
        return (dx.reshape(B, T, D).to(ctx.in_dtype), None) + tuple(grads)


# ---------------------------------------------------------------------------
# small nodes for the fusion MLP / mask heads of MaskSynthesisAgent (agents/msa.py:42-71, 134-172)
# ---------------------------------------------------------------------------
class GeluFunction(torch.autograd.Function):
    """exact-erf GELU on fp32 rows [M, N]"""

    @staticmethod
    def forward(ctx, x):
        x32 = x.detach().float().contiguous()
        y = torch.empty_like(x32)
        ops.ew_train(ops.EW_GELU_FWD, y, z=x32)
        ctx.saved = x32
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x32 = ctx.saved
        dx = torch.empty_like(x32)
        ops.ew_train(ops.EW_GELU_BWD, dx, z=x32, g=dy.detach().float().contiguous())
        ctx.saved = None
        return dx.to(ctx.in_dtype)


class LayerNormFunction(torch.autograd.Function):
    """nn.LayerNorm on fp32 rows [M, D <= 512], fp32 out"""

    @staticmethod
    def forward(ctx, x, w, b):
        x32 = x.detach().float().contiguous()
        lw, lb = _f32(w), _f32(b)
        y = torch.empty_like(x32)
        ops.layernorm(x32, lw, lb, out32=y)
        ctx.saved = (x32, lw)
        ctx.dtypes = (x.dtype, w.dtype, b.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x32, lw = ctx.saved
        D = lw.numel()
        dg = torch.zeros(D, device=dy.device, dtype=torch.float32)
        db = torch.zeros(D, device=dy.device, dtype=torch.float32)
        dx = ops.layernorm_bwd(x32, lw, dy.detach().float().contiguous(), None, dg, db)
        t = ctx.dtypes
        ctx.saved = None
        return dx.to(t[0]), dg.to(t[1]), db.to(t[2])


class MaskHeadFunction(torch.autograd.Function):
    """bounded polar mask of agents/msa.py:166-172 from the two heads' logits (lm, lp [M, F] each, or lp = None and lm = the
    merged [M, 2F] = magnitude | phase) (+ the per-utterance magnitude-logit bias [B, F] of glue G3, or None):
    (mask_real, mask_imag) [B, T, F] = sigmoid(lm + bias) * (cos, sin)(tanh(lp) * pi/8).
    Backward: sfm_polar_mask_bwd with the bias and no noisy spectrum; d bias = sum over the frames (sfm_sum_time)."""

    @staticmethod
    def forward(ctx, lm, lp, mag_bias, B, T, phase_scale):
        merged = lp is None
        if merged:
            lg = lm.detach().float()
            if lg.stride(1) != 1:
                lg = lg.contiguous()
            F = lg.shape[-1] // 2
            lm32, lp32 = lg[:, :F], lg[:, F:]
        else:
            lm32, lp32 = lm.detach().float(), lp.detach().float()
            if lm32.stride(1) != 1 or lp32.stride(1) != 1 or lm32.stride(0) != lp32.stride(0):
                lm32, lp32 = lm32.contiguous(), lp32.contiguous()
            F = lm32.shape[-1]
        bias = mag_bias.detach().float().contiguous() if mag_bias is not None else None
        mr = torch.empty(B, T, F, device=lm.device, dtype=torch.float32)
        mi = torch.empty(B, T, F, device=lm.device, dtype=torch.float32)
        ops.polar_mask(lm32, lp32, B, T, F, phase_scale, lm32.stride(0), mag_bias=bias, mr=mr, mi=mi)
        ctx.saved = (lm32, lp32, bias)
        ctx.meta = (B, T, F, phase_scale, lm.dtype, None if merged else lp.dtype, mag_bias.dtype if mag_bias is not None else None, merged)
        return mr, mi

    @staticmethod
    def backward(ctx, dmr, dmi):
        lm32, lp32, bias = ctx.saved
        B, T, F, ps, t0, t1, t2, merged = ctx.meta
        M = B * T
        ld = ops.round_up(2 * F, 8)
        dl = torch.empty(M, ld, device=lm32.device, dtype=torch.float32)
        ops.polar_mask_bwd(lm32, lp32, None, None, dmr.detach().float().contiguous(), dmi.detach().float().contiguous(), dl, M, F,
                           ps, lm32.stride(0), mag_bias=bias, rows_per_batch=T)
        db = None
        if bias is not None and ctx.needs_input_grad[2]:
            db = ops.sum_time(dl, B, T, F, ld).to(t2)
        ctx.saved = None
        if merged:
            return dl[:, :2 * F].to(t0), None, db, None, None, None
        return dl[:, :F].to(t0), dl[:, F:2 * F].to(t1), db, None, None, None


class FusionInputLinearFunction(torch.autograd.Function):
    """first Linear of MaskSynthesisAgent.fusion applied to the 8-way concatenation of agents/msa.py:134-141:
    y [M, N] = cat(z_real^T, z_imag^T, rho_s, rho_n, phi1, phi2, lognorm(noisy_real), lognorm(noisy_imag)) @ W^T + b.
    The concatenation never exists in fp32: the eight sources are written as 16-bit columns of the one [M, 1088] GEMM operand
    by the transposing / converting / log1p-normalising kernels the inference path uses (functional._msa_pack_inputs), and
    the backward GEMM's [M, 1026] result is handed back as column views (the two latents through a transposing copy, the
    noisy STFT through sfm_stft_lognorm_bwd when it carries a gradient)."""

    @staticmethod
    def forward(ctx, z_real, z_imag, rho_s, rho_n, phi1, phi2, noisy_real, noisy_imag, W, b):
        from . import functional as Fn
        B, D, T = z_real.shape
        M = B * T
        cp = {"rho_s": rho_s.detach(), "rho_n": rho_n.detach(), "phi1": phi1.detach(), "phi2": phi2.detach()}
        fused16, nr, ni = Fn._msa_pack_inputs(z_real.detach(), z_imag.detach(), cp, noisy_real.detach(), noisy_imag.detach())
        N, K = W.shape
        w32 = _f32(W)
        y = ops.linear16(fused16, ops.pack_linear(w32, _f32(b), k_pad_to=Fn.FUSE_LD), out_dtype=torch.float32)
        ctx.saved = (fused16, w32, nr, ni)
        ctx.dims = (B, D, T, N, K, rho_s.shape[-1], nr.shape[-1])
        ctx.dtypes = [t.dtype for t in (z_real, z_imag, rho_s, rho_n, phi1, phi2, noisy_real, noisy_imag, W, b)]
        return y

    @staticmethod
    def backward(ctx, dy):
        fused16, w32, nr, ni = ctx.saved
        B, D, T, N, K, oc, F = ctx.dims
        M = B * T
        dev = dy.device
        dy = dy.float()
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        Np = ops.round_up(N, 64)
        dy16 = torch.empty(M, Np, device=dev, dtype=ops.compute_dtype())
        ops.convert_rows(dy, dy16, M, N, Np, dy.stride(0), Np)
        dW = torch.zeros(N, K, device=dev, dtype=torch.float32)
        db = torch.zeros(N, device=dev, dtype=torch.float32)
        ops.gemm16_tn(dy16[:, :N], fused16[:, :K], dW, db)
        need = ctx.needs_input_grad
        t = ctx.dtypes
        grads = [None] * 8
        if any(need[:8]):
            df = ops.linear16(dy16, ops.pack_linear(w32.t().contiguous(), k_pad_to=Np), out_dtype=torch.float32)   # [M, K]
            dfb = df.reshape(B, T, -1)
            for i in (0, 1):
                if need[i]:
                    grads[i] = dfb[..., i * D:(i + 1) * D].transpose(1, 2).contiguous().to(t[i])
            for i in range(4):
                if need[2 + i]:
                    grads[2 + i] = dfb[..., 2 * D + i * oc:2 * D + (i + 1) * oc].to(t[2 + i])
            if need[6] or need[7]:
                c0 = 2 * D + 4 * oc
                dre, dim_ = ops.stft_lognorm_bwd(nr.reshape(M, F), ni.reshape(M, F), df[:, c0:], M, F)
                grads[6] = dre.reshape(B, T, F).to(t[6]) if need[6] else None
                grads[7] = dim_.reshape(B, T, F).to(t[7]) if need[7] else None
        ctx.saved = None
        return (*grads, dW.to(t[8]), db.to(t[9]))


class FusionInputPackedFunction(torch.autograd.Function):
    """FusionInputLinearFunction for callers that hold the inputs in the path's own layouts: the pooled latents channels-last
    and whole (zcat [B, T, 2D] = z_real | z_imag) and the four CPEA outputs as the one [B, T, 4 oc] tensor the fused heads
    write.  Two converting copies and the log1p pack build the operand; the backward hands the two column blocks of the input
    gradient back as views (no transposes, no per-output slices for autograd to re-assemble)."""

    @staticmethod
    def forward(ctx, zcat, cp_all, noisy_real, noisy_imag, W, b):
        from . import functional as Fn
        B, T, D2 = zcat.shape
        M = B * T
        C4 = cp_all.shape[-1]
        dt = ops.compute_dtype()
        fused16 = torch.empty(M, Fn.FUSE_LD, device=zcat.device, dtype=dt)
        z32, c32 = zcat.detach().float().reshape(M, D2), cp_all.detach().float().reshape(M, C4)
        if z32.stride(1) != 1:
            z32 = z32.contiguous()
        if c32.stride(1) != 1:
            c32 = c32.contiguous()
        ops.convert_rows(z32, fused16, M, D2, D2, z32.stride(0), Fn.FUSE_LD)
        ops.convert_rows(c32, fused16[:, D2:], M, C4, C4, c32.stride(0), Fn.FUSE_LD)
        nr, ni = noisy_real.detach().float().contiguous(), noisy_imag.detach().float().contiguous()
        F = nr.shape[-1]
        col = D2 + C4
        ops.stft_lognorm_pack(nr, ni, fused16[:, col:], M, F, Fn.FUSE_LD - col - 2 * F, Fn.FUSE_LD)
        N, K = W.shape
        w32 = _f32(W)
        y = ops.linear16(fused16, ops.pack_linear(w32, _f32(b), k_pad_to=Fn.FUSE_LD), out_dtype=torch.float32)
        ctx.saved = (fused16, w32, nr, ni)
        ctx.dims = (B, T, D2, C4, N, K, F)
        ctx.dtypes = [t.dtype for t in (zcat, cp_all, noisy_real, noisy_imag, W, b)]
        return y

    @staticmethod
    def backward(ctx, dy):
        fused16, w32, nr, ni = ctx.saved
        B, T, D2, C4, N, K, F = ctx.dims
        M = B * T
        dev = dy.device
        dy = dy.float()
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        Np = ops.round_up(N, 64)
        dy16 = torch.empty(M, Np, device=dev, dtype=ops.compute_dtype())
        ops.convert_rows(dy, dy16, M, N, Np, dy.stride(0), Np)
        dW = torch.zeros(N, K, device=dev, dtype=torch.float32)
        db = torch.zeros(N, device=dev, dtype=torch.float32)
        ops.gemm16_tn(dy16[:, :N], fused16[:, :K], dW, db)
        need = ctx.needs_input_grad
        t = ctx.dtypes
        g = [None] * 4
        if any(need[:4]):
            df = torch.empty(M, ops.round_up(K, 8), device=dev, dtype=torch.float32)[:, :K]      # 16-byte aligned rows
            ops.linear16(dy16, ops.pack_linear(w32.t().contiguous(), k_pad_to=Np), out=df)          # [M, K]
            if need[0]:
                g[0] = df[:, :D2].reshape(B, T, D2).to(t[0])                 # row-strided views of df
            if need[1]:
                g[1] = df[:, D2:D2 + C4].reshape(B, T, C4).to(t[1])
            if need[2] or need[3]:
                dre, dim_ = ops.stft_lognorm_bwd(nr.reshape(M, F), ni.reshape(M, F), df[:, D2 + C4:], M, F)
                g[2] = dre.reshape(B, T, F).to(t[2]) if need[2] else None
                g[3] = dim_.reshape(B, T, F).to(t[3]) if need[3] else None
        ctx.saved = None
        return (*g, dW.to(t[4]), db.to(t[5]))


class LatentFanoutFunction(torch.autograd.Function):
    """the pooled latents zp [B, T, 2D] feed the mask-synthesis fusion whole and the CPEA by their real half (glue G1 of
    DESIGN.md): returns (zp, zp[..., :D]) and folds the two gradients in ONE pass (sfm_add_cols) - left to autograd the fan-in is
    a zero-fill and a strided copy per slice plus an add per consumer, each over the [B, T, 2D] tensor."""

    @staticmethod
    def forward(ctx, zp, D):
        ctx.D = D
        return zp.view_as(zp), zp[..., :D]

    @staticmethod
    def backward(ctx, g_all, g_half):
        D = ctx.D
        if g_half is None:
            return g_all, None
        B, T = g_half.shape[0], g_half.shape[1]
        M = B * T
        gh = g_half.float().reshape(M, D)
        if gh.stride(1) != 1:
            gh = gh.contiguous()
        if g_all is None:
            out = torch.zeros(B, T, 2 * D, device=g_half.device, dtype=torch.float32)
            out[..., :D].copy_(gh.reshape(B, T, D))
            return out, None
        ga = g_all.float().reshape(M, 2 * D)                                   # a row-strided view stays a view
        if ga.stride(1) != 1 or ga.stride(0) % 4 or ga.data_ptr() % 16:
            ga = ga.contiguous()
        return ops.add_cols(ga, gh, M, 2 * D, D).reshape(B, T, 2 * D).to(g_all.dtype), None


def msa_train_forward(msa, z_real, z_imag, cpea_outputs, noisy_real, noisy_imag, mag_logit_bias=None, latents_cl=None):
    """MaskSynthesisAgent.forward (agents/msa.py:106-174) built from HIP autograd nodes: fusion MLP (its first Linear reads
    the eight inputs directly) -> ComplexConformer (its own training / eval-autograd path, real | imag kept side by side in one
    [M, d_model] tensor) -> the two GELU heads as block-diagonal GEMMs on that tensor -> bounded polar mask.
    latents_cl (optional, [B, T, 2D] = z_real | z_imag channels-last) and cpea_outputs.packed ([B, T, 4 oc], set by
    cpea_train_forward) are the path's own layouts of the same inputs: with them no transposed / sliced copies are made."""
    B, D, T = z_real.shape
    M = B * T
    f = msa.fusion
    packed = getattr(cpea_outputs, "packed", None)
    if latents_cl is not None and packed is not None and latents_cl.shape == (B, T, 2 * D):
        x = FusionInputPackedFunction.apply(latents_cl, packed, noisy_real, noisy_imag, f[0].weight, f[0].bias)
    else:
        x = FusionInputLinearFunction.apply(z_real, z_imag, cpea_outputs["rho_s"], cpea_outputs["rho_n"], cpea_outputs["phi1"],
                                            cpea_outputs["phi2"], noisy_real, noisy_imag, f[0].weight, f[0].bias)
    x = GeluFunction.apply(LayerNormFunction.apply(x, f[1].weight, f[1].bias))
    x = LNLinearFunction.apply(x, None, None, f[3].weight, f[3].bias)
    x = LayerNormFunction.apply(x, f[4].weight, f[4].bias)
    y = msa.conformer.train_core(x, B, T)                                       # [M, d_model] = mask_r | mask_i
    pr, pi = msa.mask_proj_real, msa.mask_proj_imag
    # the two heads side by side: block-diagonal weights (the off-diagonal zeros contribute exact zeros), one GEMM per layer
    h = LNLinearFunction.apply(y, None, None, torch.block_diag(pr[0].weight, pi[0].weight), torch.cat([pr[0].bias, pi[0].bias]))
    logits = LNLinearFunction.apply(GeluFunction.apply(h), None, None, torch.block_diag(pr[2].weight, pi[2].weight),
                                    torch.cat([pr[2].bias, pi[2].bias]))         # [M, 2F] = magnitude | phase logits
    return MaskHeadFunction.apply(logits, None, mag_logit_bias, B, T, 3.14159 / 8.0)


# ---------------------------------------------------------------------------
# CorrelationPhaseEstimationAgent (agents/cpea.py:43-112): BiLSTM layers with BPTT, inter-layer dropout, 4 heads
# ---------------------------------------------------------------------------
class BiLSTMLayerFunction(torch.autograd.Function):
    """one bidirectional nn.LSTM layer: x [B, T, Din] -> h [B, T, 2H].  Input projection, input gradient and all weight
    gradients are GEMMs on the matrix cores; the recurrence and its BPTT are the persistent kernels of lstm.hip."""

    @staticmethod
    def forward(ctx, x, wif, whf, bif, bhf, wir, whr, bir, bhr):
        B, T, Din = x.shape
        H = whf.shape[1]
        M = B * T
        dt = ops.compute_dtype()
        x32 = x.detach().float()
        if x32.stride(2) != 1 or x32.stride(0) != T * x32.stride(1):              # (a column slice of a wider row is fine as it is)
            x32 = x32.contiguous()
        wih = torch.cat([_f32(wif), _f32(wir)], dim=0)                               # [8H, Din]
        bias = torch.cat([_f32(bif) + _f32(bhf), _f32(bir) + _f32(bhr)], dim=0)
        whh = torch.stack([_f32(whf), _f32(whr)], dim=0).contiguous()                # [2, 4H, H]
        x16 = torch.empty(M, Din, device=x.device, dtype=dt)
        ops.convert_rows(x32, x16, M, Din, Din, x32.stride(1), Din)
        xg = ops.linear16(x16, ops.pack_linear(wih, bias), out_dtype=torch.float32)   # [M, 8H] = [B, T, 2, 4H]
        h, save = ops.bilstm_layer_train(xg, whh, B, T, H)
        ctx.saved = (x16, wih, whh, save, h)
        ctx.dims = (B, T, Din, H)
        ctx.dtypes = [t.dtype for t in (x, wif, whf, bif, bhf, wir, whr, bir, bhr)]
        return h

    @staticmethod
    def backward(ctx, dh):
        x16, wih, whh, save, h = ctx.saved
        B, T, Din, H = ctx.dims
        M = B * T
        dev, dt = dh.device, ops.compute_dtype()
        dxg = ops.bilstm_layer_bwd(save, whh, dh.detach().float().contiguous(), B, T, H).reshape(M, 8 * H)
        dxg16 = torch.empty(M, 8 * H, device=dev, dtype=dt)
        ops.convert_rows(dxg, dxg16, M, 8 * H, 8 * H, 8 * H, 8 * H)
        dwih = torch.zeros(8 * H, Din, device=dev, dtype=torch.float32)
        db = torch.zeros(8 * H, device=dev, dtype=torch.float32)
        ops.gemm16_tn(dxg16, x16, dwih, db)
        dx = ops.linear16(dxg16, ops.pack_linear(wih.t().contiguous()), out_dtype=torch.float32).reshape(B, T, Din)
        # dW_hh[dir] = sum_t da[t] (x) h_prev[t]: the chain's previous output (zero at its first step)
        hp16 = ops.lstm_hprev16(h, B, T, H)
        dwhh = torch.zeros(2, 4 * H, H, device=dev, dtype=torch.float32)
        for d in range(2):
            ops.gemm16_tn(dxg16[:, d * 4 * H:(d + 1) * 4 * H], hp16[:, d * H:(d + 1) * H], dwhh[d])
        t = ctx.dtypes
        ctx.saved = None
        g = 4 * H
        return (dx.to(t[0]), dwih[:g].to(t[1]), dwhh[0].to(t[2]), db[:g].to(t[3]), db[:g].to(t[4]),
                dwih[g:].to(t[5]), dwhh[1].to(t[6]), db[g:].to(t[7]), db[g:].to(t[8]))


class DropoutFunction(torch.autograd.Function):
    """inverted dropout with the counter-based keep function (inter-layer dropout of nn.LSTM, agents/cpea.py:49)"""

    @staticmethod
    def forward(ctx, x, p, seed):
        x32 = x.detach().float().contiguous()
        y = torch.empty_like(x32)
        n = x32.shape[-1]
        ops.ew_train(ops.EW_SCALE_DROP, y.reshape(-1, n), g=x32.reshape(-1, n), alpha=1.0, p=p, seed=seed)
        ctx.meta = (p, seed, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, dtp = ctx.meta
        d = dy.detach().float().contiguous()
        dx = torch.empty_like(d)
        n = d.shape[-1]
        ops.ew_train(ops.EW_SCALE_DROP, dx.reshape(-1, n), g=d.reshape(-1, n), alpha=1.0, p=p, seed=seed)
        return dx.to(dtp), None, None


class CpeaHeadsFunction(torch.autograd.Function):
    """the four 64-wide heads of agents/cpea.py:57-76,100-105 as ONE GEMM with the bounded activations in its epilogue
    (EPI_CPEA: sigmoid on the first half of the columns, pi * tanh on the second) -> [M, 4 * oc] fp32 = rho_s | rho_n | phi1 | phi2.
    Backward: the activation derivative from the saved OUTPUTS (sfm_ew_train mode 7), then the two GEMMs of a Linear."""

    @staticmethod
    def forward(ctx, x, W, b):
        import math
        M, K = x.shape
        N = W.shape[0]
        x32 = x.detach().float()
        if x32.stride(1) != 1:
            x32 = x32.contiguous()
        w32 = _f32(W)
        x16 = torch.empty(M, K, device=x.device, dtype=ops.compute_dtype())
        ops.convert_rows(x32, x16, M, K, K, x32.stride(0), K)
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        ops.linear16(x16, ops.pack_linear(w32, _f32(b)), epi=ops.EPI_CPEA, alpha=math.pi, nsplit=N // 2, out=y)
        ctx.saved = (x16, w32, y)
        ctx.dtypes = (x.dtype, W.dtype, b.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        import math
        x16, w32, y = ctx.saved
        M, K = x16.shape
        N = w32.shape[0]
        dev = dy.device
        dz = torch.empty(M, N, device=dev, dtype=torch.float32)
        ops.ew_train(ops.EW_CPEA_BWD, dz, z=y, g=dy.detach().float().contiguous(), alpha=math.pi)
        Np = ops.round_up(N, 64)
        dz16 = torch.empty(M, Np, device=dev, dtype=ops.compute_dtype())
        ops.convert_rows(dz, dz16, M, N, Np, N, Np)
        dW = torch.zeros(N, K, device=dev, dtype=torch.float32)
        db = torch.zeros(N, device=dev, dtype=torch.float32)
        ops.gemm16_tn(dz16[:, :N], x16, dW, db)
        dx = ops.linear16(dz16, ops.pack_linear(w32.t().contiguous(), k_pad_to=Np), out_dtype=torch.float32)
        t = ctx.dtypes
        ctx.saved = None
        return dx.to(t[0]), dW.to(t[1]), db.to(t[2])


class CpeaOutputs(dict):
    """the reference's output dict (agents/cpea.py:107-112) plus `.packed`: the same four tensors as one [B, T, 4 oc] tensor"""
    packed = None


def cpea_train_forward(cpea, z_t):
    """CorrelationPhaseEstimationAgent.forward (agents/cpea.py:79-112) from HIP autograd nodes."""
    if z_t.dim() == 3 and z_t.shape[-1] != cpea.input_dim:
        z_t = z_t.transpose(1, 2)
    x = z_t.float().contiguous()
    B, T, _ = x.shape
    L = cpea.lstm
    for l in range(cpea.num_layers):
        ps = [getattr(L, n % l) for n in ("weight_ih_l%d", "weight_hh_l%d", "bias_ih_l%d", "bias_hh_l%d",
                                          "weight_ih_l%d_reverse", "weight_hh_l%d_reverse", "bias_ih_l%d_reverse",
                                          "bias_hh_l%d_reverse")]
        x = BiLSTMLayerFunction.apply(x, *ps)
        if cpea.training and L.dropout > 0 and l + 1 < cpea.num_layers:
            x = DropoutFunction.apply(x, float(L.dropout), int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
    M = B * T
    heads = [cpea.rho_s_head[0], cpea.rho_n_head[0], cpea.phi1_head[0], cpea.phi2_head[0]]
    W = torch.cat([h.weight for h in heads], dim=0)
    b = torch.cat([h.bias for h in heads], dim=0)
    y = CpeaHeadsFunction.apply(x.reshape(M, -1), W, b).reshape(B, T, -1)
    oc = heads[0].weight.shape[0]
    out = CpeaOutputs({"rho_s": y[..., :oc], "rho_n": y[..., oc:2 * oc], "phi1": y[..., 2 * oc:3 * oc], "phi2": y[..., 3 * oc:]})
    out.packed = y                     # the four outputs side by side, as the fused heads wrote them (msa_train_forward)
    return out


MEMORY_PARAM_ORDER = ("key_proj.0.weight", "key_proj.0.bias", "key_proj.1.weight", "key_proj.1.bias", "key_proj.3.weight",
                      "key_proj.3.bias", "keys", "values", "value_proj.0.weight", "value_proj.0.bias", "gate.0.weight",
                      "gate.0.bias")                                  # = functional.pack_memory_params


class MemoryFunction(torch.autograd.Function):
    """EpisodicMemory.forward (agents/memory.py:95-148) as the one-workgroup-per-utterance kernel of the inference path
    (sfm_memory_fwd) and its adjoint (sfm_memory_bwd: forward recomputed in LDS, parameter gradients accumulated with
    fp32 atomics into a blob laid out like the packed parameters).  Returns (gated bias [B, vd], gate [B, 1], top_indices,
    similarity); the last two are bookkeeping outputs and carry no gradient."""

    @staticmethod
    def forward(ctx, emb, temperature, *params):
        e = emb.detach().float().contiguous()
        blob = torch.cat([_f32(p).reshape(-1) for p in params]).contiguous()
        kd, vd, S = params[0].shape[0], params[7].shape[1], params[6].shape[0]
        bias, gate, top, sim = ops.memory_fwd(e, blob, kd, vd, S, temperature)
        ctx.saved = (e, blob)
        ctx.meta = (kd, vd, S, float(temperature), [tuple(p.shape) for p in params], [p.dtype for p in params], emb.dtype)
        ctx.mark_non_differentiable(top, sim)
        return bias, gate, top, sim

    @staticmethod
    def backward(ctx, d_bias, d_gate, _d_top, _d_sim):
        e, blob = ctx.saved
        kd, vd, S, temp, shapes, dtypes, edt = ctx.meta
        Bn = e.shape[0]
        dob = d_bias.detach().float().contiguous() if d_bias is not None else torch.zeros(Bn, vd, device=e.device)
        dg = d_gate.detach().float().reshape(Bn).contiguous() if d_gate is not None else None
        d_emb, dblob = ops.memory_bwd(e, blob, dob, dg, kd, vd, S, temp, want_d_emb=ctx.needs_input_grad[0])
        outs, off = [], 0
        for shp, dt in zip(shapes, dtypes):
            k = 1
            for d_ in shp:
                k *= d_
            outs.append(dblob[off:off + k].view(shp).to(dt))
            off += k
        ctx.saved = None
        return (d_emb.to(edt) if d_emb is not None else None, None, *outs)


def memory_train_forward(mem, emb):
    """EpisodicMemory.forward (agents/memory.py:95-148) with autograd: one HIP kernel each way (MemoryFunction)."""
    sd = dict(mem.named_parameters())
    params = [sd[k] for k in MEMORY_PARAM_ORDER]
    bias, gate, top, sim = MemoryFunction.apply(emb, float(mem.temperature), *params)
    return {"bias": bias, "gate": gate, "top_indices": top.long(), "similarity": sim}


class IstftFunction(torch.autograd.Function):
    """batch_istft (training/conformer_pipeline.py:205-211) with its adjoint: irfft-by-matrix + overlap-add forward;
    backward = frames-x-matrix product of g / envelope with the transposed operand (the objective's own path in
    EnhancerLossFunction does the same inline)."""

    @staticmethod
    def forward(ctx, real, imag, length, n_fft, hop, win):
        from . import functional as Fn
        r, i = real.detach().float().contiguous(), imag.detach().float().contiguous()
        ctx.meta = (tuple(r.shape), length, n_fft, hop, win, real.dtype, imag.dtype)
        return Fn.istft(r, i, length, n_fft, hop, win)

    @staticmethod
    def backward(ctx, g):
        (B, T, F), L, n_fft, hop, win, d0, d1 = ctx.meta
        genv = (g.detach().float() * _inv_envelope(L, T, n_fft, hop, win, g.device)).contiguous()
        dr = torch.empty(B, T, F, device=g.device, dtype=torch.float32)
        di = torch.empty(B, T, F, device=g.device, dtype=torch.float32)
        ops.framed_gemm(genv, _adjoint_consts(n_fft, win, g.device)["invT"], dr, B=B, M=T, Ls=L, sig_batch_stride=L, hop=hop,
                        padl=n_fft // 2 - (n_fft - win) // 2, K=win, N=2 * F, o_batch_stride=T * F, ldm=F, ldn=1, mode=0,
                        out2=di, nsplit=F)
        return dr.to(d0), di.to(d1), None, None, None, None


# ---------------------------------------------------------------------------
# PerceptionAgent (agents/perception.py:132-254) in training mode
# ---------------------------------------------------------------------------
def sinc_filters_autograd(low_hz_, band_hz_, window, n_, sample_rate, min_low_hz=50.0, min_band_hz=50.0):
    """The analytic band-pass bank of SincConv1d.forward (agents/perception.py:88-112) as differentiable torch ops on the
    [C, K] = [64, 251] filter matrix: the chain rule from the tap gradient (sfm_sinc_wgrad) to the 2 x 64 cut-off
    parameters.  Same quirk as the reference: the cut-offs are divided by the sample rate although n_ already is."""
    low = min_low_hz + low_hz_.abs()
    high = torch.clamp(low + min_band_hz + band_hz_.abs(), max=sample_rate / 2.0)
    f_lo, f_hi = low / sample_rate, high / sample_rate
    left = (torch.sin(f_hi * n_) - torch.sin(f_lo * n_)) / (n_ / 2.0 + 1e-8)
    bp = torch.cat([left, 2.0 * (f_hi - f_lo), left.flip(dims=[1])], dim=1) * window
    return bp / (bp.abs().sum(dim=1, keepdim=True) + 1e-8)


class SincConvFunction(torch.autograd.Function):
    """The stand-alone SincConv1d in train() mode (agents/perception.py:79-118): forward = the module's fp32 FIR
    (framed_gemm_f32, exact); backward = the tap gradient dfilt[c, k] = sum_{b,l} dy[b, c, l] wave[b, l + k - K//2] (fp32 vector
    kernel sfm_sinc_wgrad, exact) and the chain rule through the analytic filter bank to low_hz_ / band_hz_
    (sinc_filters_autograd).  The waveform is the network's input: it gets no gradient (as in PerceptionFunction)."""

    @staticmethod
    def forward(ctx, waveform, mod, low_hz_, band_hz_):
        B, _, L = waveform.shape
        C, K = mod.out_channels, mod.kernel_size
        w = waveform.detach().float().reshape(B, L).contiguous()
        _, Wt = ops.sinc_filters(low_hz_.detach().reshape(-1).contiguous(), band_hz_.detach().reshape(-1).contiguous(),
                                 mod.window.contiguous(), mod.n_.reshape(-1).contiguous(), C, K, mod.sample_rate, mod.min_low_hz,
                                 mod.min_band_hz, want_filt=False)
        out = torch.empty(B, C, L, device=w.device, dtype=torch.float32)
        ops.framed_gemm(w, Wt, out, B=B, M=L, Ls=L, sig_batch_stride=L, hop=1, padl=K // 2, K=K, N=C, o_batch_stride=C * L, ldm=1,
                        ldn=L, mode=0)
        ctx.mod, ctx.wave = mod, w
        ctx.save_for_backward(low_hz_, band_hz_)
        return out

    @staticmethod
    def backward(ctx, dy):
        mod, w = ctx.mod, ctx.wave
        lo0, bw0 = ctx.saved_tensors
        C, K = mod.out_channels, mod.kernel_size
        dy_cl = torch.empty(dy.shape[0], dy.shape[2], C, device=dy.device, dtype=torch.float32)
        Bn, Ln = dy.shape[0], dy.shape[2]
        ops.transpose(dy.float().contiguous(), dy_cl, Bn, C, Ln, C * Ln, Ln, Ln * C, C)   # [B, C, L] -> channels-last [B, L, C]
        dfilt = ops.sinc_wgrad(w, dy_cl, C, K, exact=True)
        with torch.enable_grad():
            lo = lo0.detach().float().clone().requires_grad_(True)
            bw = bw0.detach().float().clone().requires_grad_(True)
            filt = sinc_filters_autograd(lo, bw, mod.window.float(), mod.n_.float(), float(mod.sample_rate),
                                         float(mod.min_low_hz), float(mod.min_band_hz))
            glo, gbw = torch.autograd.grad(filt, [lo, bw], grad_outputs=dfilt)
        return None, None, glo.to(lo0.dtype), gbw.to(bw0.dtype)


def _pa_param_names(pa):
    return [k for k, _ in pa.named_parameters()]


class PerceptionFunction(torch.autograd.Function):
    """waveform [B, L] -> latents zcat [B, T_pa, 2D] fp32 (z_real | z_imag).  Forward = the inference kernels of
    functional.perception_forward, keeping every raw conv output and its GroupNorm statistics; backward = GroupNorm /
    GELU backward kernels, Conv1d input and weight gradients as implicit GEMMs, the FIR tap gradient and, through
    sinc_filters_autograd, the gradients of the 128 sinc parameters.  The waveform itself gets no gradient."""

    @staticmethod
    def forward(ctx, wave, pa, *params):
        from . import functional as Fn
        names = _pa_param_names(pa)
        P = dict(zip(names, params))
        dt = ops.compute_dtype()
        wave = wave.detach().float().contiguous()
        B, L = wave.shape
        dev = wave.device
        fs = float(pa.sample_rate)
        sc_ = pa.sinc_conv
        with torch.no_grad():
            filt = sinc_filters_autograd(_f32(P["sinc_conv.low_hz_"]), _f32(P["sinc_conv.band_hz_"]), sc_.window.float(),
                                         sc_.n_.float(), fs).contiguous()
        C0, K = filt.shape
        raw0 = torch.empty(B, L, C0, device=dev, dtype=dt)
        part0, P0 = ops.sinc_fir16(wave, filt, raw0, B, L, C0, K)
        nodes = []

        def norm(raws, parts, Ps, gws, gbs, G, rows, act, out_dtype):
            """GroupNorm (+ second branch) + activation; remembers what the backward needs"""
            C = raws[0].shape[-1]
            tabs = []
            for r, p_, pn, gw, gb in zip(raws, parts, Ps, gws, gbs):
                s, h = ops.gn_finalize(p_, _f32(gw), _f32(gb), B, pn, G, C, rows)
                mean, rstd = ops.gn_stats(p_, rows, C, G)
                tabs.append((s, h, mean, rstd))
            out = torch.empty(B, rows, C, device=dev, dtype=out_dtype)
            if len(raws) == 2:
                ops.gn_apply(raws[0], tabs[0][0], tabs[0][1], out, B, rows, C, act=act, x2=raws[1], sc2=tabs[1][0], sh2=tabs[1][1])
            else:
                ops.gn_apply(raws[0], tabs[0][0], tabs[0][1], out, B, rows, C, act=act)
            return out, tabs

        x, t0 = norm([raw0], [part0], [P0], [P["sinc_norm.weight"]], [P["sinc_norm.bias"]], 8, L, 1, dt)
        saved = {"wave": wave, "raw0": raw0, "t0": t0, "x0": x, "blocks": []}
        Lc = L
        for i in range(3):
            pre = "conv_blocks.%d." % i
            w1, b1 = P[pre + "main.0.weight"], P[pre + "main.0.bias"]
            w2, b2 = P[pre + "main.3.weight"], P[pre + "main.3.bias"]
            ws, bs = P[pre + "skip.0.weight"], P[pre + "skip.0.bias"]
            C = w1.shape[0]
            G = min(16, C)
            r1, p1, P1, L1 = Fn._conv_gn(x, ops.pack_linear(_f32(w1), _f32(b1)), B, Lc, 2, 3, G, dt)
            a1, t1 = norm([r1], [p1], [P1], [P[pre + "main.1.weight"]], [P[pre + "main.1.bias"]], G, L1, 1, dt)
            r2, p2, P2, _ = Fn._conv_gn(a1, ops.pack_linear(_f32(w2), _f32(b2)), B, L1, 1, 1, G, dt)
            rs, ps, Ps, _ = Fn._conv_gn(x, ops.pack_linear(_f32(ws), _f32(bs)), B, Lc, 2, 0, G, dt)
            xo, t2 = norm([r2, rs], [p2, ps], [P2, Ps], [P[pre + "main.4.weight"], P[pre + "skip.1.weight"]],
                          [P[pre + "main.4.bias"], P[pre + "skip.1.bias"]], G, L1, 1, dt)
            saved["blocks"].append(dict(xin=x, Lin=Lc, L1=L1, C=C, G=G, r1=r1, t1=t1, a1=a1, r2=r2, rs=rs, t2=t2))
            x, Lc = xo, L1
        D = P["downsample.0.weight"].shape[0]
        rd, pd, Pd, Tpa = Fn._conv_gn(x, ops.pack_linear(_f32(P["downsample.0.weight"]), _f32(P["downsample.0.bias"])), B, Lc, 2,
                                      2, 16, dt)
        xd, td = norm([rd], [pd], [Pd], [P["downsample.1.weight"]], [P["downsample.1.bias"]], 16, Tpa, 1, dt)
        wz = torch.cat([_f32(P["real_proj.0.weight"]), _f32(P["imag_proj.0.weight"])], dim=0)
        bz = torch.cat([_f32(P["real_proj.0.bias"]), _f32(P["imag_proj.0.bias"])], dim=0)
        gz = torch.cat([_f32(P["real_proj.1.weight"]), _f32(P["imag_proj.1.weight"])])
        hz = torch.cat([_f32(P["real_proj.1.bias"]), _f32(P["imag_proj.1.bias"])])
        rz, pz, Pz, _ = Fn._conv_gn(xd, ops.pack_linear(wz, bz), B, Tpa, 1, 0, 32, torch.float32)
        zcat, tz = norm([rz], [pz], [Pz], [gz], [hz], 32, Tpa, 0, torch.float32)
        saved.update(x3=x, L3=Lc, rd=rd, td=td, xd=xd, Tpa=Tpa, D=D, rz=rz, tz=tz, wz=wz, gz=gz, filt_shape=(C0, K))
        ctx.saved, ctx.P, ctx.names, ctx.pa = saved, P, names, pa
        ctx.dtypes = [t.dtype for t in params]
        return zcat

    @staticmethod
    def backward(ctx, dz):
        S, P, pa = ctx.saved, ctx.P, ctx.pa
        dt = ops.compute_dtype()
        B, L = S["wave"].shape
        Tpa, D = S["Tpa"], S["D"]
        G = {}

        def gnb(dout, act, groups, raws, tabs, gammas):
            a = []
            for r, t, g in zip(raws, tabs, gammas):
                a += [r, t[0], t[1], t[2], t[3], _f32(g)]
            return ops.gn_act_backward(dout, act, groups, *a, dx_dtype=dt)

        # latent heads: GroupNorm(16) per half, 1x1 conv (real | imag stacked)
        d_rz, dgz, dbz = gnb(dz.detach().float().contiguous(), 0, 32, [S["rz"]], S["tz"], [S["gz"]])
        G["real_proj.1.weight"], G["imag_proj.1.weight"] = dgz[:D], dgz[D:]
        G["real_proj.1.bias"], G["imag_proj.1.bias"] = dbz[:D], dbz[D:]
        dWz, dbz0 = ops.conv_wgrad16(d_rz.reshape(B * Tpa, 2 * D), S["xd"], B, Tpa, Tpa, D, 2 * D, 1, 1, 0)
        G["real_proj.0.weight"], G["imag_proj.0.weight"] = dWz[:D], dWz[D:]
        G["real_proj.0.bias"], G["imag_proj.0.bias"] = dbz0[:D], dbz0[D:]
        d_xd = ops.conv_dgrad16(d_rz, S["wz"], B, Tpa, Tpa, 1, 0, out_dtype=dt)
        # downsample: conv k5 s2 p2 -> GN(16) -> GELU
        d_rd, G["downsample.1.weight"], G["downsample.1.bias"] = gnb(d_xd, 1, 16, [S["rd"]], S["td"], [P["downsample.1.weight"]])
        L3 = S["L3"]
        G["downsample.0.weight"], G["downsample.0.bias"] = ops.conv_wgrad16(d_rd.reshape(B * Tpa, D), S["x3"], B, Tpa, L3, D, D, 5, 2, 2)
        dx = ops.conv_dgrad16(d_rd, P["downsample.0.weight"], B, Tpa, L3, 2, 2, out_dtype=dt)
        # residual blocks, last to first
        for i in (2, 1, 0):
            b_ = S["blocks"][i]
            pre = "conv_blocks.%d." % i
            C, Gp, L1, Lin = b_["C"], b_["G"], b_["L1"], b_["Lin"]
            Cin = b_["xin"].shape[-1]
            res = gnb(dx, 1, Gp, [b_["r2"], b_["rs"]], b_["t2"], [P[pre + "main.4.weight"], P[pre + "skip.1.weight"]])
            d_r2, G[pre + "main.4.weight"], G[pre + "main.4.bias"], d_rs, G[pre + "skip.1.weight"], G[pre + "skip.1.bias"] = res
            G[pre + "main.3.weight"], G[pre + "main.3.bias"] = ops.conv_wgrad16(d_r2.reshape(B * L1, C), b_["a1"], B, L1, L1, C, C, 3, 1, 1)
            d_a1 = ops.conv_dgrad16(d_r2, P[pre + "main.3.weight"], B, L1, L1, 1, 1, out_dtype=dt)
            d_r1, G[pre + "main.1.weight"], G[pre + "main.1.bias"] = gnb(d_a1, 1, Gp, [b_["r1"]], b_["t1"], [P[pre + "main.1.weight"]])
            G[pre + "main.0.weight"], G[pre + "main.0.bias"] = ops.conv_wgrad16(d_r1.reshape(B * L1, C), b_["xin"], B, L1, Lin, Cin, C, 7, 2, 3)
            G[pre + "skip.0.weight"], G[pre + "skip.0.bias"] = ops.conv_wgrad16(d_rs.reshape(B * L1, C), b_["xin"], B, L1, Lin, Cin, C, 1, 2, 0)
            # input gradient: the skip path (k 1, stride 2: even rows only) is a plain GEMM whose fp32 result rides into the main
            # conv's even-row GEMM as its epilogue addend; dx is written once, in the 16-bit format the next node reads
            skip_even = ops.conv_dgrad16(d_rs, P[pre + "skip.0.weight"], B, L1, L1, 1, 0)
            dx = ops.conv_dgrad16(d_r1, P[pre + "main.0.weight"], B, L1, Lin, 2, 3, out_dtype=dt, add_even=skip_even)
        # sinc stage: GN(8) + GELU, then the FIR tap gradient and the chain rule to the cut-off parameters
        d_raw0, G["sinc_norm.weight"], G["sinc_norm.bias"] = gnb(dx, 1, 8, [S["raw0"]], S["t0"], [P["sinc_norm.weight"]])
        C0, K = S["filt_shape"]
        dfilt = ops.sinc_wgrad(S["wave"], d_raw0, C0, K)
        sc_ = pa.sinc_conv
        with torch.enable_grad():
            lo = _f32(P["sinc_conv.low_hz_"]).clone().requires_grad_(True)
            bw = _f32(P["sinc_conv.band_hz_"]).clone().requires_grad_(True)
            filt = sinc_filters_autograd(lo, bw, sc_.window.float(), sc_.n_.float(), float(pa.sample_rate))
            glo, gbw = torch.autograd.grad(filt, [lo, bw], grad_outputs=dfilt)
        G["sinc_conv.low_hz_"], G["sinc_conv.band_hz_"] = glo, gbw
        out = []
        for k, dtp in zip(ctx.names, ctx.dtypes):
            g = G.get(k)
            out.append(None if g is None else g.reshape(P[k].shape).to(dtp))     # uncertainty head: no gradient path here
        ctx.saved = None
        return (None, None) + tuple(out)


class PoolTimeFunction(torch.autograd.Function):
    """glue G1 under autograd: adaptive average pooling of channels-last latents [B, Tin, C] fp32 to Tout frames"""

    @staticmethod
    def forward(ctx, x, Tout):
        x = x.detach().float().contiguous()
        B, Tin, C = x.shape
        out = torch.empty(B, Tout, C, device=x.device, dtype=torch.float32)
        ops.pool_time(x, None, out, B, Tin, Tout, C, C, C)
        ctx.dims = (B, Tin, Tout, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Tin, Tout, C = ctx.dims
        return ops.pool_time_bwd(dout.detach().float(), B, Tin, Tout, C), None


class MeanTimeFunction(torch.autograd.Function):
    """glue G2 under autograd: mean over the frames of channels-last rows, x [B, T, C] fp32 (a column slice of wider rows is
    read in place) -> [B, C]; backward = the adjoint of pooling to one frame (sfm_pool_time_bwd)"""

    @staticmethod
    def forward(ctx, x):
        x32 = x.detach().float()
        B, T, C = x32.shape
        if x32.stride(2) != 1 or x32.stride(0) != T * x32.stride(1):
            x32 = x32.contiguous()
        ctx.dims = (B, T, C, x.dtype)
        return ops.mean_time(x32, B, T, C, x32.stride(1))

    @staticmethod
    def backward(ctx, g):
        B, T, C, dtp = ctx.dims
        return ops.pool_time_bwd(g.detach().float().reshape(B, 1, C), B, T, 1, C).to(dtp)


def perception_latents_train(pa, waveform):
    """channels-last latents [B, T_pa, 2D] (z_real | z_imag) of the PerceptionAgent as an autograd node"""
    if waveform.dim() == 3:
        waveform = waveform.squeeze(1)
    return PerceptionFunction.apply(waveform, pa, *[p for _, p in pa.named_parameters()])


def perception_train_forward(pa, waveform):
    """PerceptionAgent.forward (agents/perception.py:216-251) under autograd: (z_real, z_imag [B, D, T_pa], sigma).  sigma
    is evaluated on the inference kernels and carries no gradient (the uncertainty head is not part of any objective on
    the path)."""
    from . import functional as Fn
    if waveform.dim() == 3:
        waveform = waveform.squeeze(1)
    zcat = perception_latents_train(pa, waveform)
    with torch.no_grad():
        pk = pa._packed(lambda sd: Fn.pack_perception(sd, pa.sample_rate))
        _, sigma = Fn.perception_forward(waveform.detach().float(), pk, latents=False)
    D = zcat.shape[-1] // 2
    z = zcat.transpose(1, 2)
    return z[:, :D], z[:, D:], sigma.reshape(sigma.shape[0], 1, -1)


# ---------------------------------------------------------------------------
# MetacognitiveArbitrationAgent (agents/maa.py) and VectorQuantizer (models/vq.py) under autograd
# ---------------------------------------------------------------------------
def maa_pack_params(net):
    """decision_net (Linear(1,64), ReLU, Linear(64,64), ReLU, Linear(64,4)) -> the flat fp32 layout of routing.hip"""
    return torch.cat([_f32(net[0].weight).reshape(-1), _f32(net[0].bias), _f32(net[2].weight).reshape(-1), _f32(net[2].bias),
                      _f32(net[4].weight).reshape(-1), _f32(net[4].bias)]).contiguous()


class MaaFunction(torch.autograd.Function):
    """sigma [N], stats (running_mean, running_var), the six decision_net tensors -> logits, probs [N,4], confidence [N],
    decisions int64 [N] (not differentiable)"""

    @staticmethod
    def forward(ctx, sigma, stats, w1, b1, w2, b2, w3, b3):
        s = sigma.detach().float().contiguous()
        params = torch.cat([_f32(w1).reshape(-1), _f32(b1), _f32(w2).reshape(-1), _f32(b2), _f32(w3).reshape(-1), _f32(b3)]).contiguous()
        logits, probs, dec, conf = ops.maa_forward(s, stats, params)
        ctx.saved = (s, stats.clone(), params)
        ctx.meta = (sigma.dtype, [t.dtype for t in (w1, b1, w2, b2, w3, b3)], [tuple(t.shape) for t in (w1, b1, w2, b2, w3, b3)])
        ctx.mark_non_differentiable(dec)
        return logits, probs, conf, dec

    @staticmethod
    def backward(ctx, g_logits, g_probs, g_conf, _g_dec):
        s, stats, params = ctx.saved
        ctx.saved = None
        c = lambda g: None if g is None else g.detach().float().contiguous()
        dsig, dp = ops.maa_backward(s, stats, params, c(g_logits), c(g_probs), c(g_conf))
        sdt, pdts, shapes = ctx.meta
        outs, o = [], 0
        for dt_, shp in zip(pdts, shapes):
            n = int(torch.Size(shp).numel())
            outs.append(dp[o:o + n].view(shp).to(dt_))
            o += n
        return (dsig.to(sdt), None) + tuple(outs)


class VQFunction(torch.autograd.Function):
    """x, centroids -> quantized (straight-through gradient), indices, beta * commitment + codebook loss (models/vq.py:54-96)"""

    @staticmethod
    def forward(ctx, x, centroids, beta):
        xf = x.detach().float().contiguous()
        cf = centroids.detach().float().contiguous()
        q, idx, acc = ops.vq_forward(xf, cf)
        loss = (acc[0] * ((1.0 + beta) / xf.numel())).float()
        ctx.saved = (xf, idx, cf)
        ctx.meta = (float(beta), x.dtype, centroids.dtype)
        ctx.mark_non_differentiable(idx)
        return q.to(x.dtype), idx, loss

    @staticmethod
    def backward(ctx, g_q, _g_idx, g_loss):
        xf, idx, cf = ctx.saved
        ctx.saved = None
        beta, xdt, cdt = ctx.meta
        gq = None if g_q is None else g_q.detach().float().contiguous()
        gl = None if g_loss is None else g_loss.detach().float().reshape(1).contiguous()
        dx, dcent = ops.vq_backward(xf, idx, cf, gq, gl, beta)
        return dx.to(xdt), dcent.to(cdt), None
