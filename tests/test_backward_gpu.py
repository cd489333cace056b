"""Parity of the training-path kernels (backward of the ConformerBlock) against torch autograd on CPU."""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import arr, maxerr, rmse
from oracle import sfm_oracle as orc

pytestmark = pytest.mark.gpu
DTYPES = [torch.bfloat16, torch.float16]
EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from sincformer_metacog_speech_enhancement_amd import ops as _ops
    return _ops


def q16(x, dt):
    return x.to(dt).float()


def report(name, got, ref, tol):
    e, r = maxerr(got, ref), rmse(got, ref)
    print("%-44s max|err| %.3e rmse %.3e tol %.1e ref_rms %.3e" % (name, e, r, tol, float(torch.as_tensor(np.asarray(ref)).double().pow(2).mean().sqrt())))
    assert math.isfinite(e) and e <= tol, name


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (5000, 1024, 256), (333, 129, 128), (4096, 256, 1024), (70, 768, 256),
                                   (4500, 136, 72), (6001, 129, 128), (20011, 128, 448)])
def test_gemm16_tn_and_colsum(ops, dt, M, N, K):
    ops.set_compute_dtype(dt)
    g, x = arr("tg", (M, N), 1) * 0.1, arr("tx", (M, K), 2)
    dW = torch.zeros(N, K, device="cuda")
    gbuf = torch.zeros(M, (N + 7) // 8 * 8, device="cuda", dtype=dt)       # row stride must be a multiple of 8 elements
    gbuf[:, :N] = g.cuda().to(dt)
    dbf = torch.zeros(N, device="cuda")
    ops.gemm16_tn(gbuf[:, :N], x.cuda().to(dt).contiguous(), dW, dbf)
    ref = q16(g, dt).t() @ q16(x, dt)
    report("wgrad %dx%dx%d %s" % (M, N, K, dt), dW.cpu(), ref, 2e-3 * math.sqrt(M / 1000.0))
    report("wgrad fused bias grad", dbf.cpu(), q16(g, dt).sum(0), 1e-3)
    db = torch.zeros(N, device="cuda")
    ops.colsum(g.cuda(), db)
    report("colsum fp32", db.cpu(), g.sum(0), 1e-3)
    db.zero_()
    ops.colsum(g.cuda().to(dt), db)
    report("colsum 16", db.cpu(), q16(g, dt).sum(0), 1e-3)


@pytest.mark.parametrize("M,D", [(300, 256), (77, 64), (50, 258)])
def test_layernorm_bwd(ops, M, D):
    x = (arr("lx", (M, D), 3, 2.0) + 0.3).requires_grad_(True)
    w = (arr("lw", (D,), 4) * 0.1 + 1).requires_grad_(True)
    b = (arr("lb", (D,), 5) * 0.1).requires_grad_(True)
    dy, dres = arr("ldy", (M, D), 6), arr("ldr", (M, D), 7)
    y = orc.layer_norm(x, w, b)
    y.backward(dy)
    dg, dbt = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dx = ops.layernorm_bwd(x.detach().cuda(), w.detach().cuda(), dy.cuda(), dres.cuda(), dg, dbt)
    report("ln bwd dx", dx.cpu(), x.grad + dres, 2e-5)
    report("ln bwd dgamma", dg.cpu(), w.grad, 2e-4)
    report("ln bwd dbeta", dbt.cpu(), b.grad, 2e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,D,p", [(203, 256, 0.1), (64, 256, 0.0), (77, 144, 0.15), (50, 130, 0.1)])
def test_layernorm_bwd_also_writes_the_next_nodes_operand(ops, dt, M, D, p):
    """next_drop = (alpha, p, seed): the second output equals ew_train(EW_SCALE_DROP) of the returned dx BIT FOR BIT (same
    counters m * D + d, same order of the two multiplications) on the vector paths (D 256; D 144 with a ragged second half), to a
    rounding midpoint on the scalar path (D 130) - and dx / dgamma / dbeta are what the call without it returns"""
    ops.set_compute_dtype(dt)
    x, w = arr("nx", (M, D), 31, 2.0).cuda(), (arr("nw", (D,), 32) * 0.1 + 1).cuda()
    dy, dres = arr("ndy", (M, D), 33).cuda(), arr("ndr", (M, D), 34).cuda()
    g0, b0 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    g1, b1 = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dx0 = ops.layernorm_bwd(x, w, dy, dres, g0, b0)
    dx1, nxt = ops.layernorm_bwd(x, w, dy, dres, g1, b1, next_drop=(0.5, p, 12345))
    assert torch.equal(dx0, dx1)
    assert maxerr(g0.cpu(), g1.cpu()) < 1e-4 * float(g0.abs().max()) and maxerr(b0.cpu(), b1.cpu()) < 1e-4 * float(b0.abs().max() + 1)
    want = torch.empty(M, D, device="cuda", dtype=dt)
    ops.ew_train(ops.EW_SCALE_DROP, want, g=dx0, alpha=0.5, p=p, seed=12345)
    assert nxt.dtype == dt
    if D % 4 == 0:
        assert torch.equal(nxt, want)                                  # the paths the training step takes
    else:
        # scalar path: one element of 6500 at an exact rounding midpoint came out one fp16 ulp apart
        bad = nxt.float() != want.float()
        assert int(bad.sum()) <= 2 and maxerr(nxt.float().cpu(), want.float().cpu()) < 1e-3
    if p > 0:
        assert 0.5 * p < float((nxt == 0).float().mean()) < 1.5 * p + 0.02


@pytest.mark.parametrize("dt", DTYPES)
def test_elementwise_train(ops, dt):
    ops.set_compute_dtype(dt)
    M, N = 200, 256
    z, g = arr("ez", (M, N), 8, 2.0), arr("eg", (M, N), 9)
    zq = q16(z, dt).requires_grad_(True)
    u = orc.swish(zq)
    u.backward(g)
    z16 = z.cuda().to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    ops.ew_train(ops.EW_SWISH_FWD, out, z=z16)
    report("swish fwd", out.float().cpu(), u.detach(), 4 * EPS[dt] * 3)
    ops.ew_train(ops.EW_SWISH_BWD, out, z=z16, g=g.cuda())
    report("swish bwd", out.float().cpu(), zq.grad, 4 * EPS[dt] * 4)
    # GLU
    h = arr("eh", (M, 2 * N), 10, 1.5)
    hq = q16(h, dt).requires_grad_(True)
    y = hq[:, :N] * torch.sigmoid(hq[:, N:])
    y.backward(g)
    h16 = h.cuda().to(dt)
    ops.ew_train(ops.EW_GLU_FWD, out, z=h16)
    report("glu fwd", out.float().cpu(), y.detach(), 4 * EPS[dt] * 3)
    o2 = torch.empty(M, 2 * N, device="cuda", dtype=dt)
    ops.ew_train(ops.EW_GLU_BWD, o2, z=h16, g=g.cuda())
    report("glu bwd", o2.float().cpu(), hq.grad, 4 * EPS[dt] * 4)
    # dropout: keep fraction, scaling, determinism
    o32 = torch.empty(M, N, device="cuda")
    ones = torch.ones(M, N, device="cuda")
    ops.ew_train(ops.EW_SCALE_DROP, o32, g=ones, alpha=0.5, p=0.25, seed=123)
    vals = o32.cpu()
    kept = (vals != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.02 and abs(float(vals.max()) - 0.5 / 0.75) < 1e-6
    o33 = torch.empty(M, N, device="cuda")
    ops.ew_train(ops.EW_SCALE_DROP, o33, g=ones, alpha=0.5, p=0.25, seed=123)
    assert torch.equal(o32, o33)
    ops.ew_train(ops.EW_SCALE_DROP, o33, g=ones, alpha=0.5, p=0.25, seed=124)
    assert not torch.equal(o32, o33)


@pytest.mark.parametrize("dt", DTYPES)
def test_batchnorm_train_and_dwconv_wgrad(ops, dt):
    ops.set_compute_dtype(dt)
    B, T, C, KS = 3, 50, 256, 31
    M = B * T
    y = (arr("by", (M, C), 11, 1.5) + 0.2).requires_grad_(True)
    gamma = (arr("bg", (C,), 12) * 0.1 + 1).requires_grad_(True)
    beta = (arr("bb", (C,), 13) * 0.1).requires_grad_(True)
    g = arr("bgr", (M, C), 14)
    mu = y.mean(0)
    var = ((y - mu) ** 2).mean(0)
    t = (y - mu) / torch.sqrt(var + 1e-5) * gamma + beta
    out = orc.swish(t)
    out.backward(g)
    S = ops.col_stats(y.detach().cuda())
    mean = S[:, 0] / M
    varg = S[:, 1] / M - mean * mean
    report("bn batch mean", mean.cpu(), mu.detach(), 1e-5)
    report("bn batch var", varg.cpu(), var.detach(), 1e-4)
    rstd = torch.rsqrt(varg + 1e-5)
    dy, dgam, dbet = ops.bn_swish_bwd(g.cuda(), y.detach().cuda(), mean, rstd, gamma.detach().cuda(), beta.detach().cuda())
    report("bn+swish bwd dy", dy.cpu(), y.grad, 2e-4)
    report("bn dgamma", dgam.cpu(), gamma.grad, 2e-3)
    report("bn dbeta", dbet.cpu(), beta.grad, 2e-3)
    # depthwise conv weight gradient
    x = arr("wx", (B, T, C), 15)
    w = (arr("ww", (C, 1, KS), 16) / math.sqrt(KS)).requires_grad_(True)
    bb = (arr("wb", (C,), 17) * 0.1).requires_grad_(True)
    gy = arr("wg", (B, T, C), 18)
    o = F.conv1d(q16(x, dt).transpose(1, 2), w, bb, padding=15, groups=C).transpose(1, 2)
    o.backward(gy)
    dw, db = ops.dwconv_wgrad(x.cuda().to(dt).contiguous(), gy.cuda().contiguous(), B, T, C, KS)
    report("dwconv wgrad", dw.cpu(), w.grad.reshape(C, KS), 2e-4)
    report("dwconv bgrad", db.cpu(), bb.grad, 2e-4)


def _attn_ref(qkv, B, T, H, hd, keep=None):
    """q pre-scaled (log2 domain): P = softmax(ln2 * q'.k)"""
    D = H * hd
    q, k, v = qkv.reshape(B, T, 3 * D).split(D, dim=-1)
    q = q.reshape(B, T, H, hd).transpose(1, 2)
    k = k.reshape(B, T, H, hd).transpose(1, 2)
    v = v.reshape(B, T, H, hd).transpose(1, 2)
    s2 = q @ k.transpose(-1, -2)
    p = torch.softmax(s2 * math.log(2.0), dim=-1)
    lse2 = torch.logsumexp(s2 * math.log(2.0), dim=-1) / math.log(2.0)
    if keep is not None:
        p = p * keep
    return (p @ v).transpose(1, 2).reshape(B * T, D), lse2


def _keep_mask(seed, B, H, T, p):
    """numpy replica of the counter-based keep function (attention.hip / attention_bwd.hip): full hash of the probability row
    (b, h, q), then one multiply-add + xorshift-multiply round per key"""
    M32 = np.uint64(0xFFFFFFFF)
    row = np.arange(B * H * T, dtype=np.uint64)
    x = ((row & M32) * np.uint64(0x9E3779B1)) & M32
    x ^= ((row >> np.uint64(32)) * np.uint64(0x85EBCA77)) & M32
    x ^= np.uint64(seed)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    key = np.arange(T, dtype=np.uint64)
    y = (x[:, None] + ((key * np.uint64(0x9E3779B1)) & M32)[None, :]) & M32
    y ^= y >> np.uint64(15); y = (y * np.uint64(0x846CA68B)) & M32
    y ^= y >> np.uint64(16)
    thr = np.uint64(math.ceil(float(np.float32(p)) * 16777216.0))
    keep = ((y >> np.uint64(8)) >= thr).astype(np.float32) / (1.0 - p)
    return torch.from_numpy(keep.reshape(B, H, T, T))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,T,H,p,hd", [(2, 200, 4, 0.0, 64), (1, 64, 2, 0.0, 64), (2, 129, 1, 0.0, 64), (1, 150, 2, 0.15, 64),
                                        (2, 37, 4, 0.0, 16), (1, 50, 2, 0.2, 32)])       # hd != 64: generic kernels
def test_attention_train_fwd_bwd(ops, dt, B, T, H, p, hd):
    ops.set_compute_dtype(dt)
    seed = 77
    D = H * hd
    qkv = arr("aq", (B * T, 3 * D), 40 + T, 1.0)
    qkv[:, :D] *= 0.35                       # q' ~ pre-scaled magnitude
    dO = arr("ado", (B * T, D), 41 + T, 0.5)
    qq = q16(qkv, dt).requires_grad_(True)
    keep = _keep_mask(seed, B, H, T, p) if p > 0 else None
    ref_o, ref_lse = _attn_ref(qq, B, T, H, hd, keep)
    ref_o.backward(q16(dO, dt))
    q16d = qkv.cuda().to(dt).contiguous()
    O, lse = ops.attention_train(q16d, B, T, H, hd, p_drop=p, seed=seed)
    report("attn train fwd O p=%.2f" % p, O.float().cpu(), ref_o.detach(), 8 * EPS[dt])
    report("attn train lse", lse.cpu(), ref_lse.detach(), 6e-3 if dt is torch.bfloat16 else 1e-3)
    dqkv = ops.attention_bwd(q16d, O, dO.cuda().to(dt).contiguous(), lse, B, T, H, hd, p_drop=p, seed=seed)
    gref = qq.grad
    scale = float(gref.abs().max())
    report("attn bwd dq", dqkv.float().cpu()[:, :D], gref[:, :D], 0.03 * scale + 8 * EPS[dt] * scale)
    report("attn bwd dk", dqkv.float().cpu()[:, D:2 * D], gref[:, D:2 * D], 0.03 * scale + 8 * EPS[dt] * scale)
    report("attn bwd dv", dqkv.float().cpu()[:, 2 * D:], gref[:, 2 * D:], 0.03 * scale + 8 * EPS[dt] * scale)
    assert rmse(dqkv.float().cpu(), gref) < (0.02 if dt is torch.bfloat16 else 0.004) * float(gref.pow(2).mean().sqrt()) + 1e-6


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,L,C,G,act,two", [(2, 300, 64, 8, 1, False), (3, 257, 128, 16, 1, True), (1, 130, 256, 16, 0, False),
                                             (2, 200, 512, 32, 0, False), (2, 520, 64, 16, 1, True)])
def test_groupnorm_act_backward(ops, dt, B, L, C, G, act, two):
    """out = act(GN(x1) [+ GN(x2)]) (PerceptionAgent nodes, channels-last): dx, dgamma, dbeta vs F.group_norm autograd"""
    ops.set_compute_dtype(dt)
    xs = [(arr("gx%d" % i, (B, L, C), 500 + i + C, 1.5) + 0.3) for i in range(2 if two else 1)]
    gam = [(arr("gg%d" % i, (C,), 510 + i) * 0.2 + 1.0) for i in range(len(xs))]
    bet = [arr("gb%d" % i, (C,), 520 + i) * 0.2 for i in range(len(xs))]
    dout = arr("gdo", (B, L, C), 530 + C)
    xq = [q16(x, dt).requires_grad_(True) for x in xs]                       # the raw conv outputs are stored 16-bit
    gr = [g.clone().requires_grad_(True) for g in gam]
    br = [b_.clone().requires_grad_(True) for b_ in bet]
    p = sum(F.group_norm(x.transpose(1, 2), G, g, b_, 1e-5).transpose(1, 2) for x, g, b_ in zip(xq, gr, br))
    out = F.gelu(p) if act else p
    out.backward(dout)
    # forward statistics the way the producing GEMM + gn_finalize provide them
    args = []
    for x, g, b_ in zip(xq, gam, bet):
        xg = x.detach().reshape(B, L, G, C // G)
        mean = xg.mean(dim=(1, 3))
        rstd = torch.rsqrt(xg.var(dim=(1, 3), unbiased=False) + 1e-5)
        sc = rstd.repeat_interleave(C // G, dim=1) * g
        sh = b_ - mean.repeat_interleave(C // G, dim=1) * sc
        args += [x.detach().cuda().to(dt).contiguous(), sc.cuda(), sh.cuda(), mean.cuda(), rstd.cuda(), g.cuda()]
    res = ops.gn_act_backward(dout.cuda(), act, G, *args, dx_dtype=torch.float32)
    for i in range(len(xs)):
        dx, dg, db = res[3 * i:3 * i + 3]
        scale = float(xq[i].grad.abs().max())
        report("gn bwd dx%d C%d act%d" % (i, C, act), dx.cpu(), xq[i].grad, 2e-3 * scale + 1e-5)
        report("gn bwd dgamma%d" % i, dg.cpu(), gr[i].grad, 2e-3 * float(gr[i].grad.abs().max()) + 1e-4)
        report("gn bwd dbeta%d" % i, db.cpu(), br[i].grad, 2e-3 * float(br[i].grad.abs().max()) + 1e-4)
    # the storage formats of the training path: 16-bit incoming gradient and dx, fp32 raw input (latent heads)
    args32 = [a.float() if j % 6 == 0 else a for j, a in enumerate(args)]
    res16 = ops.gn_act_backward(dout.cuda().to(dt), act, G, *args32, dx_dtype=dt)
    for i in range(len(xs)):
        scale = float(xq[i].grad.abs().max())
        report("gn bwd dx%d (16-bit io)" % i, res16[3 * i].float().cpu(), xq[i].grad, (3e-2 if dt is torch.bfloat16 else 4e-3) * scale)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,Lin,Cin,N,k,s,p", [(2, 400, 64, 128, 7, 2, 3), (2, 301, 128, 128, 3, 1, 1), (3, 200, 64, 128, 1, 2, 0),
                                               (1, 333, 256, 256, 5, 2, 2), (2, 150, 256, 512, 1, 1, 0),
                                               # B * Lout >= 4096: the weight gradient runs on the LDS-DMA ring kernel (gemm16_tn2.hip)
                                               (3, 4001, 64, 128, 7, 2, 3), (2, 2500, 128, 128, 3, 1, 1), (5, 2000, 64, 128, 1, 2, 0),
                                               (2, 4099, 256, 256, 5, 2, 2), (1, 4500, 128, 256, 3, 1, 1)])
def test_conv1d_wgrad_and_dgrad(ops, dt, B, Lin, Cin, N, k, s, p):
    """weight / bias / input gradients of the PerceptionAgent's Conv1d shapes (channels-last, implicit GEMMs) vs F.conv1d"""
    ops.set_compute_dtype(dt)
    Lout = (Lin + 2 * p - k) // s + 1
    x = q16(arr("cvx", (B, Lin, Cin), 600 + Lin), dt)
    w = q16(arr("cvw", (N, Cin, k), 601 + k, 0.1), dt)
    dy = q16(arr("cvdy", (B, Lout, N), 602 + N, 0.5), dt)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = torch.zeros(N, requires_grad=True)
    y = F.conv1d(xr.transpose(1, 2), wr, br, stride=s, padding=p).transpose(1, 2)
    y.backward(dy)
    dy16 = dy.cuda().to(dt).contiguous()
    dW, db = ops.conv_wgrad16(dy16.reshape(B * Lout, N), x.cuda().to(dt).contiguous(), B, Lout, Lin, Cin, N, k, s, p)
    tolw = 3e-3 * float(wr.grad.abs().max()) + 1e-3
    report("conv wgrad k%d s%d" % (k, s), dW.cpu(), wr.grad, tolw)
    report("conv bias grad", db.cpu(), br.grad, 2e-3 * float(br.grad.abs().max()) + 1e-3)
    dx = ops.conv_dgrad16(dy16, w.cuda(), B, Lout, Lin, s, p)
    report("conv dgrad k%d s%d" % (k, s), dx.cpu(), xr.grad, 3e-3 * float(xr.grad.abs().max()) + 1e-3)
    # accumulation into an existing gradient (two consumers of one activation)
    base = arr("cvb", (B, Lin, Cin), 603).cuda()
    dx2 = ops.conv_dgrad16(dy16, w.cuda(), B, Lout, Lin, s, p, accumulate_into=base.clone())
    report("conv dgrad accumulate", (dx2 - base).cpu(), xr.grad, 3e-3 * float(xr.grad.abs().max()) + 1e-3)
    # the training path's form: 16-bit result written once (no zero-fill, no read-modify-write); for the stride-2 convs the
    # parallel k = 1 skip conv's input gradient (its own plain GEMM, fp32, even rows) rides in as the epilogue addend
    dx16 = ops.conv_dgrad16(dy16, w.cuda(), B, Lout, Lin, s, p, out_dtype=dt)
    assert dx16.dtype == dt
    gmax = float(xr.grad.abs().max())
    report("conv dgrad 16-bit out", dx16.float().cpu(), xr.grad, (3e-3 + (2 ** -8 if dt is torch.bfloat16 else 2 ** -11)) * gmax + 1e-3)
    if s == 2 and (Lin + 2 * p - k) // 2 + 1 == (Lin + 1) // 2:
        ws = q16(arr("cvws", (N, Cin, 1), 604, 0.1), dt)
        dys = q16(arr("cvdys", (B, Lout, N), 605, 0.5), dt)
        xs = x.clone().requires_grad_(True)
        ys = F.conv1d(xs.transpose(1, 2), ws, None, stride=2, padding=0).transpose(1, 2)
        assert ys.shape[1] == Lout
        ys.backward(dys)
        even = ops.conv_dgrad16(dys.cuda().to(dt).contiguous(), ws.cuda(), B, Lout, Lout, 1, 0)
        both = ops.conv_dgrad16(dy16, w.cuda(), B, Lout, Lin, s, p, out_dtype=dt, add_even=even)
        ref = xr.grad + xs.grad
        report("conv dgrad main + skip", both.float().cpu(), ref,
               (3e-3 + (2 ** -8 if dt is torch.bfloat16 else 2 ** -11)) * float(ref.abs().max()) + 1e-3)


@pytest.mark.parametrize("B,L", [(1, 800), (3, 2113), (2, 16000)])
@pytest.mark.parametrize("dt", DTYPES)
def test_sinc_fir_tap_gradient(B, L, dt):
    """dfilt[c, k] = sum_{b,t} dy[b, t, c] * wave[b, t + k - K//2]: the weight gradient of SincConv1d's F.conv1d
    (agents/perception.py:115-118), vs torch autograd in fp64."""
    from sincformer_metacog_speech_enhancement_amd import ops
    ops.set_compute_dtype(dt)
    C, K = 64, 251
    wave = arr("sw", (B, L), 31)
    dy = arr("sdy", (B, L, C), 32).to(dt)
    filt = torch.zeros(C, 1, K, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(wave.double().unsqueeze(1), filt, padding=K // 2)               # [B, C, L]
    (y * dy.double().transpose(1, 2)).sum().backward()
    got = ops.sinc_wgrad(wave.cuda(), dy.cuda(), C, K, exact=True).cpu().double()
    ref = filt.grad[:, 0]
    rel = float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    got16 = ops.sinc_wgrad(wave.cuda(), dy.cuda(), C, K).cpu().double()         # matrix cores, waveform rounded to 16 bits
    rel16 = float((got16 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print("sinc tap gradient B%d L%d %s: rel rmse %.2e (fp32 vector kernel), %.2e (MFMA, 16-bit waveform)" % (B, L, dt, rel, rel16))
    assert rel < 1e-5
    assert rel16 < 4 * EPS[dt]


@pytest.mark.parametrize("C", [48, 512, 50, 1028])
@pytest.mark.parametrize("Tin,Tout", [(200, 21), (21, 200), (150, 16), (7, 7), (1000, 101)])
def test_pool_time_adjoint(Tin, Tout, C):
    """glue G1 (adaptive average pooling over time) and its adjoint vs F.adaptive_avg_pool1d + autograd (C 512: the path's
    latents; 50: the scalar kernel; 1028: more than 256 channel vectors per row)"""
    from sincformer_metacog_speech_enhancement_amd import train
    B = 2
    x = arr("ptx", (B, Tin, C), 41)
    cot = arr("ptc", (B, Tout, C), 42)
    xr = x.clone().requires_grad_(True)
    yr = F.adaptive_avg_pool1d(xr.transpose(1, 2), Tout).transpose(1, 2)
    (yr * cot).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = train.PoolTimeFunction.apply(xg, Tout)
    (y * cot.cuda()).sum().backward()
    assert maxerr(y.detach().cpu(), yr.detach()) < 1e-6
    assert maxerr(xg.grad.cpu(), xr.grad) < 1e-6


@pytest.mark.parametrize("M,N,K", [(8192, 256, 256), (20001, 512, 256), (9000, 256, 1024)])
@pytest.mark.parametrize("dt", DTYPES)
def test_weight_gradient_gemm_wide_tiles(ops, dt, M, N, K):
    """dW[n,k] += sum_m G[m,n] X[m,k] and the fused bias gradient on the 256 x 256-tile kernel (N, K multiples of 256,
    M >= 8192), incl. a ragged M and accumulation into a non-zero dW, vs fp64"""
    ops.set_compute_dtype(dt)
    G = q16(arr("wg", (M, N), 700 + N, 0.5), dt)
    X = q16(arr("wx", (M, K), 701 + K, 0.5), dt)
    init = arr("wi", (N, K), 702)
    dW = init.clone().cuda()
    db = torch.zeros(N, device="cuda")
    ops.gemm16_tn(G.cuda().to(dt), X.cuda().to(dt), dW, db)
    ref = init.double() + G.double().t() @ X.double()
    assert maxerr(dW.cpu(), ref) < 2e-4 * float(ref.abs().max())
    assert maxerr(db.cpu(), G.double().sum(0)) < 2e-4 * float(G.double().sum(0).abs().max())


def test_ordered_reductions_repeat_bitwise_and_match_the_atomic_form(ops):
    """Every split reduction of the training step in its ordered form (per-workgroup partials folded by csrc/reduce.hip: the
    default) at shapes with hundreds to thousands of partials: two runs give the SAME BITS, and the result agrees with the
    fp32-atomics form (ops.set_deterministic(False)) to summation-order rounding."""
    dt = torch.float16
    ops.set_compute_dtype(dt)
    g = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)

    def cases():
        M = 40000
        G16, X16 = (rn(M, 256) * 0.1).to(dt), rn(M, 256).to(dt)

        def tn_wide():                                        # 256 x 256 tiles, 157 M-splits
            dW, db = torch.zeros(256, 256, device="cuda"), torch.zeros(256, device="cuda")
            ops.gemm16_tn(G16, X16, dW, db)
            return dW, db
        G2, X2 = (rn(M, 136) * 0.1).to(dt), rn(M, 72).to(dt)

        def tn_narrow():                                      # ragged 128 x 128 tiles
            dW, db = torch.zeros(136, 72, device="cuda"), torch.zeros(136, device="cuda")
            ops.gemm16_tn(G2, X2, dW, db)
            return dW, db
        B, Lin, Cin, N, k = 8, 4000, 64, 128, 7
        x16, dy16 = rn(B, Lin, Cin).to(dt), (rn(B * (Lin // 2), N) * 0.1).to(dt)

        def conv():                                           # LDS-DMA ring kernel (gemm16_tn2)
            return ops.conv_wgrad16(dy16, x16, B, Lin // 2, Lin, Cin, N, k, 2, 3)
        Gf = rn(M, 264)

        def colsum():
            out = torch.zeros(264, device="cuda")
            ops.colsum(Gf, out)
            return (out,)
        x32, dy32, gam = rn(M, 256), rn(M, 256), rn(256)

        def ln():                                             # 1024 partial rows -> two fold levels
            dg, dbt = torch.zeros(256, device="cuda"), torch.zeros(256, device="cuda")
            dx = ops.layernorm_bwd(x32, gam, dy32, None, dg, dbt)
            return dx, dg, dbt

        def colstats():                                       # 313 row blocks
            return (ops.col_stats(x32),)
        mean, rstd = x32.mean(0), 1.0 / x32.std(0)

        def bnbwd():
            return ops.bn_swish_bwd(dy32, x32, mean, rstd, gam, gam * 0.1)
        Bg, Lg, Cg, Gg = 4, 6000, 64, 8
        xg = rn(Bg, Lg, Cg).to(dt)
        sc, sh = rn(Bg, Cg) * 0.2 + 1.0, rn(Bg, Cg) * 0.1
        mu, rs = rn(Bg, Gg) * 0.1, rn(Bg, Gg).abs() + 0.5
        dout = rn(Bg, Lg, Cg).to(dt)

        def gn():
            return ops.gn_act_backward(dout, 1, Gg, xg, sc, sh, mu, rs, gam[:Cg])
        est, tgt = rn(6, 64000), rn(6, 64000)
        pr, pi, tr, ti = rn(100000), rn(100000), rn(100000), rn(100000)

        def moments():
            return ops.wave_moments(est, tgt), ops.spec_sums(pr, pi, tr, ti)
        Bd, Td, Cd = 12, 801, 256
        xd, dyd = rn(Bd, Td, Cd).to(dt), rn(Bd, Td, Cd)

        def dw():
            return ops.dwconv_wgrad(xd, dyd, Bd, Td, Cd, 31)
        return [("gemm16_tn wide", tn_wide), ("gemm16_tn narrow", tn_narrow), ("conv_wgrad16", conv), ("colsum", colsum),
                ("layernorm_bwd", ln), ("col_stats", colstats), ("bn_swish_bwd", bnbwd), ("gn_act_backward", gn),
                ("objective moments", moments), ("dwconv_wgrad", dw)]

    try:
        for name, fn in cases():
            ops.set_deterministic(True)
            a = [t.clone() for t in fn()]
            b = [t.clone() for t in fn()]
            for i, (u, v) in enumerate(zip(a, b)):
                assert torch.equal(u, v), (name, i, "ordered form is not bit-reproducible")
            ops.set_deterministic(False)
            c = fn()
            for i, (u, v) in enumerate(zip(a, c)):
                scale = float(v.double().abs().max()) + 1e-30
                err = float((u.double() - v.double()).abs().max()) / scale
                print("%-22s output %d: ordered vs atomics max|diff| / max|value| = %.2e" % (name, i, err))
                # (a 16-bit output may differ by one rounding step where the two fp32 sums straddle a rounding boundary - the atomic
                #  order changes from run to run, so this did show up once in many runs: 7e-5 of the tensor's maximum)
                assert err < (2e-5 if v.dtype in (torch.float32, torch.float64) else 1e-3), (name, i, err)
    finally:
        ops.set_deterministic(True)
