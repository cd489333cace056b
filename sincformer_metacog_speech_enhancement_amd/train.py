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


class _Seeds:
    def __init__(self, base):
        self.base, self.i = int(base) & 0x7FFFFFFF, 0

    def next(self):
        self.i += 1
        return (self.base * 2654435761 + self.i * 40503) & 0xFFFFFFFF


# ---------------------------------------------------------------------------
# FeedForwardModule (models/conformer.py:41-49)
# ---------------------------------------------------------------------------
def _ffn_fwd(x, P, pre, p, seeds):
    dt = ops.compute_dtype()
    M, D = x.shape
    lw, lb = _f32(P[pre + "layer_norm.weight"]), _f32(P[pre + "layer_norm.bias"])
    f1, b1 = _lin_pack(_f32(P[pre + "linear1.weight"]), _f32(P[pre + "linear1.bias"]))
    f2, b2 = _lin_pack(_f32(P[pre + "linear2.weight"]), _f32(P[pre + "linear2.bias"]))
    h16 = _ln16(x, lw, lb)
    z1 = ops.linear16(h16, f1)                                            # [M, FF] 16-bit pre-activation
    s1, s2 = seeds.next(), seeds.next()
    u = torch.empty_like(z1)
    ops.ew_train(ops.EW_SWISH_FWD, u, z=z1, p=p, seed=s1)
    o = ops.linear16(u, f2, out_dtype=torch.float32)
    y = torch.empty_like(x)
    ops.ew_train(ops.EW_SCALE_DROP, y, z=x, g=o, alpha=0.5, p=p, seed=s2)
    return y, dict(x=x, lw=lw, h16=h16, z1=z1, u=u, b1=b1, b2=b2, s1=s1, s2=s2, p=p)


def _ffn_bwd(dy, c, G, pre):
    dt = ops.compute_dtype()
    M, D = dy.shape
    FF = c["z1"].shape[1]
    do = torch.empty(M, D, device=dy.device, dtype=dt)
    ops.ew_train(ops.EW_SCALE_DROP, do, g=dy, alpha=0.5, p=c["p"], seed=c["s2"])
    ops.gemm16_tn(do, c["u"], G[pre + "linear2.weight"])
    ops.colsum(do, G[pre + "linear2.bias"])
    du = ops.linear16(do, c["b2"], out_dtype=torch.float32)                 # [M, FF]
    dz = torch.empty(M, FF, device=dy.device, dtype=dt)
    ops.ew_train(ops.EW_SWISH_BWD, dz, z=c["z1"], g=du, p=c["p"], seed=c["s1"])
    ops.gemm16_tn(dz, c["h16"], G[pre + "linear1.weight"])
    ops.colsum(dz, G[pre + "linear1.bias"])
    dh = ops.linear16(dz, c["b1"], out_dtype=torch.float32)                 # [M, D]
    return ops.layernorm_bwd(c["x"], c["lw"], dh, dy, G[pre + "layer_norm.weight"], G[pre + "layer_norm.bias"])


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
    o = ops.linear16(O, fout, out_dtype=torch.float32)
    y = torch.empty_like(x)
    ops.ew_train(ops.EW_SCALE_DROP, y, z=x, g=o, alpha=1.0, p=p, seed=sd)
    return y, dict(x=x, lw=lw, h16=h16, qkv=qkv, O=O, lse=lse, bin=bin_, bout=bout, sa=sa, sd=sd, p=p, qs=qs, B=B, T=T, H=H)


def _mhsa_bwd(dy, c, G):
    dt = ops.compute_dtype()
    M, D = dy.shape
    do = torch.empty(M, D, device=dy.device, dtype=dt)
    ops.ew_train(ops.EW_SCALE_DROP, do, g=dy, alpha=1.0, p=c["p"], seed=c["sd"])
    ops.gemm16_tn(do, c["O"], G["mhsa.attention.out_proj.weight"])
    ops.colsum(do, G["mhsa.attention.out_proj.bias"])
    dO = ops.linear16(do, c["bout"])                                         # 16-bit [M, D]
    dqkv = ops.attention_bwd(c["qkv"], c["O"], dO, c["lse"], c["B"], c["T"], c["H"], D // c["H"], p_drop=c["p"], seed=c["sa"])
    gw, gb = G["mhsa.attention.in_proj_weight"], G["mhsa.attention.in_proj_bias"]
    # the Q rows of W were multiplied by qs in the forward: accumulate into scratch, then scale those rows
    tw = torch.zeros_like(gw)
    tb = torch.zeros_like(gb)
    ops.gemm16_tn(dqkv, c["h16"], tw)
    ops.colsum(dqkv, tb)
    tw[:D] *= c["qs"]
    tb[:D] *= c["qs"]
    gw += tw
    gb += tb
    dh = ops.linear16(dqkv, c["bin"], out_dtype=torch.float32)
    return ops.layernorm_bwd(c["x"], c["lw"], dh, dy, G["mhsa.layer_norm.weight"], G["mhsa.layer_norm.bias"])


# ---------------------------------------------------------------------------
# ConvolutionModule (models/conformer.py:101-128), BatchNorm1d in training mode
# ---------------------------------------------------------------------------
def _conv_fwd(x, P, B, T, p, seeds, bn_buffers, momentum=0.1, eps=1e-5):
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
    S = ops.col_stats(yc)
    mean = S[:, 0] / M
    var = (S[:, 1] / M - mean * mean).clamp_min(0.0)
    rstd = torch.rsqrt(var + eps)
    if bn_buffers is not None:                                               # running statistics (unbiased var), momentum 0.1
        rm, rv, nbt = bn_buffers
        with torch.no_grad():
            rm.mul_(1 - momentum).add_(momentum * mean.to(rm.dtype))
            rv.mul_(1 - momentum).add_(momentum * (var * (M / max(M - 1, 1))).to(rv.dtype))
            nbt += 1
    sc = (gam * rstd).reshape(1, D).contiguous()
    sh = (bet - mean * gam * rstd).reshape(1, D).contiguous()
    s16 = torch.empty(M, D, device=x.device, dtype=dt)
    ops.gn_apply(yc, sc, sh, s16, 1, M, D, act=2)                            # BatchNorm + Swish
    o = ops.linear16(s16, f2, out_dtype=torch.float32)
    sd = seeds.next()
    y = torch.empty_like(x)
    ops.ew_train(ops.EW_SCALE_DROP, y, z=x, g=o, alpha=1.0, p=p, seed=sd)
    return y, dict(x=x, lw=lw, h16=h16, pre=pre, g16=g16, yc=yc, mean=mean, rstd=rstd, gam=gam, bet=bet, s16=s16, b1=b1, b2=b2,
                   wdw=wdw, KS=KS, sd=sd, p=p, B=B, T=T)


def _conv_bwd(dy, c, G):
    dt = ops.compute_dtype()
    M, D = dy.shape
    B, T, KS = c["B"], c["T"], c["KS"]
    do = torch.empty(M, D, device=dy.device, dtype=dt)
    ops.ew_train(ops.EW_SCALE_DROP, do, g=dy, alpha=1.0, p=c["p"], seed=c["sd"])
    ops.gemm16_tn(do, c["s16"], G["conv.pointwise2.weight"].view(D, D))
    ops.colsum(do, G["conv.pointwise2.bias"])
    ds = ops.linear16(do, c["b2"], out_dtype=torch.float32)
    dyc, dgam, dbet = ops.bn_swish_bwd(ds, c["yc"], c["mean"], c["rstd"], c["gam"], c["bet"])
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
    ops.gemm16_tn(dpre, c["h16"], G["conv.pointwise1.weight"].view(2 * D, D))
    ops.colsum(dpre, G["conv.pointwise1.bias"])
    dh = ops.linear16(dpre, c["b1"], out_dtype=torch.float32)
    return ops.layernorm_bwd(c["x"], c["lw"], dh, dy, G["conv.layer_norm.weight"], G["conv.layer_norm.bias"])


# ---------------------------------------------------------------------------
# whole block
# ---------------------------------------------------------------------------
def block_train_forward(x32, P, B, T, H, p, seed, bn_buffers=None, momentum=0.1, eps=1e-5):
    seeds = _Seeds(seed)
    x1, c1 = _ffn_fwd(x32, P, "ff1.", p, seeds)
    x2, c2 = _mhsa_fwd(x1, P, B, T, H, p, seeds)
    x3, c3 = _conv_fwd(x2, P, B, T, p, seeds, bn_buffers, momentum, eps)
    x4, c4 = _ffn_fwd(x3, P, "ff2.", p, seeds)
    fw, fb = _f32(P["final_norm.weight"]), _f32(P["final_norm.bias"])
    y = torch.empty_like(x4)
    ops.layernorm(x4, fw, fb, out32=y)
    return y, dict(c1=c1, c2=c2, c3=c3, c4=c4, x4=x4, fw=fw)


def block_train_backward(dy32, ctx, P):
    G = {k: torch.zeros(P[k].shape, device=dy32.device, dtype=torch.float32) for k in PARAM_NAMES}
    d4 = ops.layernorm_bwd(ctx["x4"], ctx["fw"], dy32, None, G["final_norm.weight"], G["final_norm.bias"])
    d3 = _ffn_bwd(d4, ctx["c4"], G, "ff2.")
    d2 = _conv_bwd(d3, ctx["c3"], G)
    d1 = _mhsa_bwd(d2, ctx["c2"], G)
    dx = _ffn_bwd(d1, ctx["c1"], G, "ff1.")
    return dx, G


class ConformerBlockFunction(torch.autograd.Function):
    """y = ConformerBlock(x) in training mode; saved state lives on ctx as plain tensors (no autograd graph inside)."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        B, T, D = x.shape
        H, p, seed, bn_buffers, momentum, eps = meta
        P = dict(zip(PARAM_NAMES, params))
        x32 = x.detach().float().reshape(B * T, D).contiguous()
        y, saved = block_train_forward(x32, P, B, T, H, p, seed, bn_buffers, momentum, eps)
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
