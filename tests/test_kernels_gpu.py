"""Kernel-level parity: every C-ABI entry point against an fp32 CPU evaluation
of the same operation (oracle primitives / plain torch) on seeded inputs.
Needs a real MI355X: `pytest -m gpu`."""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import arr, maxerr, rmse, synth_sd
from oracle import sfm_oracle as orc

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
# relative tolerance of one 16-bit rounding of an O(1) value
EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from sincformer_metacog_speech_enhancement_amd import ops as _ops
    return _ops


def dev(x):
    return x.cuda()


def q16(x, dt):
    """round to the 16-bit format and back (CPU)"""
    return x.to(dt).float()


def report(name, got, ref, tol):
    e = maxerr(got, ref)
    r = rmse(got, ref)
    print("%-40s max|err| %.3e  rmse %.3e  tol %.1e  ref_rms %.3e" % (name, e, r, tol, float(ref.double().pow(2).mean().sqrt())))
    assert math.isfinite(e) and e <= tol, "%s: max err %.3e > %.1e" % (name, e, tol)


# ---------------------------------------------------------------------------
def test_mfma_layout_identity(ops):
    """A = I with an asymmetric B catches transposed fragment maps."""
    ops.set_compute_dtype(torch.float16)
    K = 64
    A = torch.eye(128, K)
    W = (torch.arange(128 * K, dtype=torch.float32).reshape(128, K) % 61) / 8.0 - 3.0
    pw = ops.pack_linear(dev(W))
    out = ops.linear16(dev(A).half().contiguous(), pw, out_dtype=torch.float32)
    ref = A @ q16(W, torch.float16).t()
    report("mfma identity", out.cpu(), ref, 1e-6)


VARIANTS = [0, 2, 6, 9]       # gemm16 kernels: auto, 128-row tiles, persistent, wide tiles


@pytest.fixture(autouse=True)
def _reset_variant(ops):
    yield
    ops.set_gemm_variant(0)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,K,N", [(300, 256, 256), (129, 1024, 256), (77, 64, 129), (513, 256, 1024), (40, 128, 64), (5, 64, 1),
                                   (1000, 192, 384), (128, 64, 128)])
def test_gemm16_plain(ops, dt, M, K, N, variant):
    ops.set_compute_dtype(dt)
    ops.set_gemm_variant(variant)
    x = arr("gx", (M, K), 1)
    w = arr("gw", (N, K), 2) / math.sqrt(K)
    b = arr("gb", (N,), 3)
    pw = ops.pack_linear(dev(w), dev(b))
    out = ops.linear16(dev(x).to(dt).contiguous(), pw, out_dtype=torch.float32)
    ref = q16(x, dt) @ q16(w, dt).t() + b
    report("gemm16 %s %dx%dx%d" % (dt, M, K, N), out.cpu(), ref, 2e-4)


@pytest.mark.parametrize("M,K,N", [(300, 256, 768), (129, 1024, 256), (77, 64, 129), (2100, 256, 768)])
def test_gemm16_result_in_the_other_16bit_format(ops, M, K, N):
    """stage boundary of the precision policy: fp16 operands, result rounded once to bf16 (and the reverse), on the default,
    the wide (N % 256 == 0, many tiles) and the ragged-N tile paths"""
    x = arr("gx", (M, K), 1)
    w = arr("gw", (N, K), 2) / math.sqrt(K)
    b = arr("gb", (N,), 3)
    for dt, odt in ((torch.float16, torch.bfloat16), (torch.bfloat16, torch.float16)):
        ops.set_compute_dtype(dt)
        pw = ops.pack_linear(dev(w), dev(b))
        out = ops.linear16(dev(x).to(dt).contiguous(), pw, out_dtype=odt)
        assert out.dtype == odt
        ref = (q16(x, dt) @ q16(w, dt).t() + b).to(odt).float()
        report("gemm16 %s -> %s %dx%dx%d" % (dt, odt, M, K, N), out.float().cpu(), ref, 4 * EPS[odt] * float(ref.abs().max()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,glu", [(4096, 768, False), (4133, 768, False), (5000, 64, False), (4224, 512, True), (4101, 256, True),
                                     (4096, 1024, False)])
def test_lin256_resident_operand_kernel(ops, dt, M, N, glu):
    """K = 256 linears with a 16-bit result (Q | K | V projection, pointwise_conv1 + GLU) on csrc/lin256.hip: against the fp32
    reference on the rounded operands, against sfm_gemm16 on the same pack, into a strided output view, ragged last tile, result
    in the operands' and in the other 16-bit format"""
    ops.set_compute_dtype(dt)
    K = 256
    x = arr("l2x", (M, K), 21)
    w, b = arr("l2w", (N, K), 22) / 16.0, arr("l2b", (N,), 23)
    pw = ops.pack_linear(dev(w), dev(b), glu=glu)
    h = q16(x, dt) @ q16(w, dt).t() + b
    ref = h[:, :N // 2] * torch.sigmoid(h[:, N // 2:]) if glu else h
    epi = ops.EPI_GLU if glu else ops.EPI_NONE
    xd = dev(x).to(dt).contiguous()
    other = torch.bfloat16 if dt == torch.float16 else torch.float16
    for odt in (dt, other):
        ops.set_lin256(True)
        ncol = pw.N
        buf = torch.full((M, ncol + 24), 7.0, device="cuda", dtype=odt)           # a column slice of a wider buffer: ldo > N
        out = ops.linear16(xd, pw, epi=epi, out=buf[:, 8:8 + ncol])
        ops.set_lin256(False)
        old = ops.linear16(xd, pw, epi=epi, out_dtype=odt)
        ops.set_lin256(True)
        tol = 4 * EPS[odt] * float(ref.abs().max())
        report("lin256 %s -> %s M%d N%d glu%d" % (dt, odt, M, N, glu), out.float().cpu(), ref.to(odt).float(), tol)
        assert torch.equal(out, old), "lin256 and sfm_gemm16 differ in bits"   # same MFMA, k order and epilogue expressions
        assert float(buf[:, :8].float().min()) == 7.0 and float(buf[:, 8 + ncol:].float().max()) == 7.0
    if not glu:                                                                    # fp32 result, stored from the accumulator quads
        out32 = torch.full((M, N + 4), 7.0, device="cuda", dtype=torch.float32)
        ops.lin256(xd, pw, out32[:, :N])                                           # (linear16 keeps sfm_gemm16 for fp32 results)
        old32 = ops.linear16(xd, pw, out_dtype=torch.float32)
        report("lin256 fp32 out", out32[:, :N].cpu(), ref, 2e-4 * max(1.0, float(ref.abs().max())))
        assert torch.equal(out32[:, :N], old32) and float(out32[:, N:].min()) == 7.0
    # the kernel is the one that ran: below its row threshold linear16 keeps sfm_gemm16, and the results agree to rounding
    small = ops.linear16(xd[:100], pw, epi=epi)
    report("small-M path", small.float().cpu(), ref[:100].to(dt).float(), 4 * EPS[dt] * float(ref.abs().max()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,glu", [(4096, 512, True), (4133, 768, False), (300, 512, True)])
def test_ln_linear16_layernorm_as_the_gemm_prologue(ops, dt, M, N, glu):
    """LayerNorm -> Linear (conv.layer_norm -> pointwise_conv1 + GLU; mhsa.layer_norm -> Q | K | V) with the LayerNorm in the
    prologue of csrc/lin256.hip: the same bits as sfm_layernorm + sfm_lin256 / sfm_gemm16, against the fp32 reference, ragged last
    tile, and the unfused route below the row threshold"""
    ops.set_compute_dtype(dt)
    x = arr("lnx", (M, 256), 31) * 1.7 + 0.3
    lw, lb = arr("lnw", (256,), 32) * 0.2 + 1.0, arr("lnb", (256,), 33) * 0.1
    w, b = arr("l2w", (N, 256), 22) / 16.0, arr("l2b", (N,), 23)
    pw = ops.pack_linear(dev(w), dev(b), glu=glu)
    epi = ops.EPI_GLU if glu else ops.EPI_NONE
    xd = dev(x)
    other = torch.bfloat16 if dt == torch.float16 else torch.float16
    hn = torch.nn.functional.layer_norm(x, (256,), lw, lb, 1e-5)
    h = q16(hn, dt) @ q16(w, dt).t() + b
    ref = h[:, :N // 2] * torch.sigmoid(h[:, N // 2:]) if glu else h
    for odt in (dt, other):
        ops.set_lin256(True)
        out = ops.ln_linear16(xd, dev(lw), dev(lb), pw, epi=epi, out_dtype=odt)
        ops.set_lin256(False)
        old = ops.ln_linear16(xd, dev(lw), dev(lb), pw, epi=epi, out_dtype=odt)      # sfm_layernorm + sfm_gemm16
        ops.set_lin256(True)
        assert out.dtype == odt and torch.equal(out, old), "the fused prologue and sfm_layernorm + sfm_gemm16 differ in bits"
        # (the reference rounds LN(x) once more than the kernels see it differently: a rounding step of the operand format)
        report("ln + lin256 %s -> %s M%d N%d glu%d" % (dt, odt, M, N, glu), out.float().cpu(), ref.to(odt).float(),
               (6 * EPS[odt] + 16 * EPS[dt]) * float(ref.abs().max()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,Tin,Tout", [(3, 2555, 512), (2, 4000, 801), (2, 640, 129), (1, 300, 300)])
def test_headpool_heads_and_time_pooling_in_one_launch(ops, dt, B, Tin, Tout):
    """sfm_headpool: the PerceptionAgent's stacked latent heads (1x1 conv, 256 -> 512) + adaptive average pooling to the STFT frame
    count + GroupNorm partial sums of the full-rate outputs, against the fp32 reference on the rounded operands; ragged last tile,
    windows of 5 / 6 and of 1 frame, a row shared by two tiles counted once"""
    ops.set_compute_dtype(dt)
    ops.set_headpool(True)                                                           # (whatever SFM_HEADPOOL says)
    tiles = ops.headpool_tiles(Tin, Tout)
    assert tiles is not None and tiles[0] >= 1
    x = arr("hpx", (B, Tin, 256), 41)
    w, b = arr("hpw", (512, 256), 42) / 16.0, arr("hpb", (512,), 43)
    pw = ops.pack_linear(dev(w), dev(b))
    raw = q16(x, dt) @ q16(w, dt).t() + b                                            # [B, Tin, 512] fp32
    ref = torch.stack([raw[:, (i * Tin) // Tout: -((-(i + 1) * Tin) // Tout)].mean(1) for i in range(Tout)], dim=1)
    pooled = torch.full((B, Tout, 520), 5.0, device="cuda", dtype=dt)                # ldp > N
    part = torch.full((B, tiles[1], 32, 2), float("nan"), device="cuda")
    ops.headpool(dev(x).to(dt).contiguous(), pw, pooled[:, :, :512], part, B, Tin, Tout)
    report("headpool pooled %s T%d->%d" % (dt, Tin, Tout), pooled[:, :, :512].float().cpu(), ref, 3 * EPS[dt] * float(ref.abs().max()))
    assert float(pooled[:, :, 512:].float().min()) == 5.0
    sums = part.double().sum(1).cpu()                                                # [B, 32, 2]
    rg = raw.double().reshape(B, Tin, 32, 16)
    report("headpool sum", sums[..., 0].float(), rg.sum((1, 3)).float(), 2e-5 * float(rg.abs().sum((1, 3)).max()))
    report("headpool sum of squares", sums[..., 1].float(), (rg ** 2).sum((1, 3)).float(), 2e-5 * float((rg ** 2).sum((1, 3)).max()))
    again = torch.empty_like(part)
    ops.headpool(dev(x).to(dt).contiguous(), pw, torch.empty(B, Tout, 512, device="cuda", dtype=dt), again, B, Tin, Tout)
    assert torch.equal(again, part)                                                  # fixed summation order


@pytest.mark.parametrize("variant", VARIANTS)
def test_gemm16_strided_operand_views(ops, variant):
    """A is a column slice of a wider buffer (lda > K), output goes into a column slice (ldo > N)"""
    ops.set_compute_dtype(torch.float16)
    ops.set_gemm_variant(variant)
    M = 333
    buf = arr("sv", (M, 256), 15)
    w, b = arr("svw", (129, 128), 16) / 11.0, arr("svb", (129,), 17)
    pw = ops.pack_linear(dev(w), dev(b))
    x16 = dev(buf).half().contiguous()
    out = torch.full((M, 300), 9.0, device="cuda", dtype=torch.float32)
    ops.linear16(x16[:, 128:], pw, out=out[:, 100:229])
    ref = q16(buf[:, 128:], torch.float16) @ q16(w, torch.float16).t() + b
    report("gemm16 strided views v%d" % variant, out[:, 100:229].cpu(), ref, 2e-4)
    assert float(out[:, :100].min()) == 9.0 and float(out[:, 229:].min()) == 9.0


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dt", DTYPES)
def test_gemm16_epilogues(ops, dt, variant):
    ops.set_compute_dtype(dt)
    ops.set_gemm_variant(variant)
    M, K, N = 200, 256, 256
    x, w, b = arr("ex", (M, K), 4), arr("ew", (N, K), 5) / 16.0, arr("eb", (N,), 6)
    xq, wq = q16(x, dt), q16(w, dt)
    base = xq @ wq.t() + b
    pw = ops.pack_linear(dev(w), dev(b))
    xd = dev(x).to(dt).contiguous()
    tol16 = 4 * EPS[dt]
    report("epi swish", ops.linear16(xd, pw, epi=ops.EPI_SWISH, out_dtype=torch.float32).cpu(), base * torch.sigmoid(base), 5e-4)
    report("epi gelu", ops.linear16(xd, pw, epi=ops.EPI_GELU, out_dtype=torch.float32).cpu(), orc.gelu(base), 5e-4)
    report("epi gelu 16-bit out", ops.linear16(xd, pw, epi=ops.EPI_GELU).float().cpu(), orc.gelu(base), tol16 * 4)
    res = arr("er", (M, N), 7)
    report("epi resid", ops.linear16(xd, pw, epi=ops.EPI_RESID, resid=dev(res), alpha=0.5).cpu(), res + 0.5 * base, 5e-4)
    report("epi sigmoid", ops.linear16(xd, pw, epi=ops.EPI_SIGMOID, out_dtype=torch.float32).cpu(), torch.sigmoid(base), 2e-4)
    report("epi sigma", ops.linear16(xd, pw, epi=ops.EPI_SIGMA, out_dtype=torch.float32).cpu(),
           torch.exp(0.5 * torch.clamp(base, -10, 10)), 2e-3)
    ref = torch.cat([torch.sigmoid(base[:, :128]), math.pi * torch.tanh(base[:, 128:])], dim=1)
    report("epi cpea", ops.linear16(xd, pw, epi=ops.EPI_CPEA, out_dtype=torch.float32, alpha=math.pi, nsplit=128).cpu(), ref, 1e-3)
    # GLU: weight [2C, K]
    w2, b2 = arr("ew2", (512, K), 8) / 16.0, arr("eb2", (512,), 9)
    pg = ops.pack_linear(dev(w2), dev(b2), glu=True)
    h = xq @ q16(w2, dt).t() + b2
    report("epi glu", ops.linear16(xd, pg, epi=ops.EPI_GLU, out_dtype=torch.float32).cpu(), h[:, :256] * torch.sigmoid(h[:, 256:]), 5e-4)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin,cout,k,s,p,L", [(64, 128, 7, 2, 3, 301), (128, 128, 3, 1, 1, 150), (64, 128, 1, 2, 0, 301),
                                              (256, 256, 5, 2, 2, 77), (256, 64, 3, 1, 1, 40), (8, 16, 3, 1, 1, 50)])
def test_gemm16_conv_and_groupnorm(ops, dt, cin, cout, k, s, p, L, variant):
    ops.set_compute_dtype(dt)
    ops.set_gemm_variant(variant)
    B = 2
    x = arr("cx", (B, cin, L), 10)
    w = arr("cw", (cout, cin, k), 11) / math.sqrt(cin * k)
    b = arr("cb", (cout,), 12)
    Lout = (L + 2 * p - k) // s + 1
    ref = F.conv1d(q16(x, dt), q16(w, dt), b, stride=s, padding=p)              # [B, cout, Lout]
    xcl = dev(x).transpose(1, 2).contiguous().to(dt)                            # [B, L, cin]
    pw = ops.pack_linear(dev(w), dev(b))
    out = torch.empty(B, Lout, cout, device="cuda", dtype=torch.float32)
    G = 16 if cout >= 128 else cout // 8           # group width 8 / 16 / 32 columns (what the GEMM epilogue reduces)
    gs = cout // G
    P = 2 * ((Lout + 127) // 128)
    part = torch.zeros(B, P, G, 2, device="cuda", dtype=torch.float32)
    ops.gemm16(xcl, pw, out, B=B, Lout=Lout, Lin=L, a_batch_stride=L * cin, ldo=cout, o_batch_stride=Lout * cout,
               stride=s, pad=p, gn_partial=part, gn_group=gs)
    report("conv k%d s%d %d->%d" % (k, s, cin, cout), out.cpu().transpose(1, 2), ref, 3e-4)
    gw, gb = arr("gnw", (cout,), 13) * 0.1 + 1.0, arr("gnb", (cout,), 14) * 0.1
    sc, sh = ops.gn_finalize(part, dev(gw), dev(gb), B, P, G, cout, Lout)
    y = torch.empty(B, Lout, cout, device="cuda", dtype=torch.float32)
    ops.gn_apply(out, sc, sh, y, B, Lout, cout, act=1)
    refn = orc.gelu(orc.group_norm(ref, G, gw, gb))
    report("groupnorm+gelu", y.cpu().transpose(1, 2), refn, 5e-4)
    # two-branch residual form with 16-bit storage
    o16 = out.to(dt)
    y16 = torch.empty(B, Lout, cout, device="cuda", dtype=dt)
    ops.gn_apply(o16, sc, sh, y16, B, Lout, cout, act=1, x2=o16, sc2=sc, sh2=sh)
    ref2 = orc.gelu(2.0 * orc.group_norm(ref, G, gw, gb))
    report("groupnorm two-branch 16-bit", y16.float().cpu().transpose(1, 2), ref2, 16 * EPS[dt])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin,cout,k,s,p,two,skip,L,B", [
    (64, 128, 7, 2, 3, False, True, 301, 2),      # block 0: sinc output -> k7 s2 conv + fused 1x1 s2 skip conv
    (128, 128, 7, 2, 3, True, True, 777, 3),      # block 1: residual input (two raw tensors), two 64-channel slabs, 4 tiles
    (128, 256, 7, 2, 3, True, False, 300, 2),     # block 2 main conv: 256 outputs = two 128-column passes
    (128, 256, 1, 2, 0, True, False, 300, 2),     # block 2 skip conv on its own (only the even input rows are read)
    (128, 128, 3, 1, 1, False, False, 150, 2),    # second conv of a block
    (256, 256, 3, 1, 1, False, False, 129, 1),    # ... with 4 slabs and a 1-row second tile
    (256, 256, 5, 2, 2, True, False, 77, 2),      # downsample
])
def test_conv16p_normalises_its_input_while_staging(ops, dt, cin, cout, k, s, p, two, skip, L, B):
    """sfm_conv16p against F.conv1d(GELU(sc1 x1 + sh1 [+ sc2 x2 + sh2])) with the 16-bit roundings of the kernel (raw inputs,
    normalised operand, weights): outputs, GroupNorm partials of the outputs, the fused skip conv, ragged tile tails"""
    ops.set_compute_dtype(dt)
    x1 = arr("px1", (B, L, cin), 20)
    x2 = arr("px2", (B, L, cin), 21) if two else None
    sc1, sh1 = arr("ps1", (B, cin), 22) * 0.2 + 1.0, arr("ph1", (B, cin), 23) * 0.3
    sc2, sh2 = (arr("ps2", (B, cin), 24) * 0.2 + 0.7, arr("ph2", (B, cin), 25) * 0.3) if two else (None, None)
    w = arr("pw", (cout, cin, k), 26) / math.sqrt(cin * k)
    b = arr("pb", (cout,), 27)
    ws = arr("pws", (cout, cin, 1), 28) / math.sqrt(cin)
    bs = arr("pbs", (cout,), 29)
    z = q16(x1, dt) * sc1[:, None, :] + sh1[:, None, :]
    if two:
        z = z + q16(x2, dt) * sc2[:, None, :] + sh2[:, None, :]
    xn = q16(orc.gelu(z), dt).transpose(1, 2)                                   # [B, cin, L], operand rounding
    ref = F.conv1d(xn, q16(w, dt), b, stride=s, padding=p).transpose(1, 2)      # [B, Lout, cout]
    Lout = ref.shape[1]
    pw = ops.pack_linear(dev(w), dev(b))
    spw = ops.pack_linear(dev(ws), dev(bs)) if skip else None
    G = 16
    P = 2 * ((Lout + 127) // 128)
    out = torch.empty(B, Lout, cout, device="cuda", dtype=torch.float32)
    part = torch.zeros(B, P, G, 2, device="cuda", dtype=torch.float32)
    out_s = torch.empty(B, Lout, cout, device="cuda", dtype=torch.float32) if skip else None
    part_s = torch.zeros(B, P, G, 2, device="cuda", dtype=torch.float32) if skip else None
    d16 = lambda t: None if t is None else dev(t).to(dt).contiguous()
    dd = lambda t: None if t is None else dev(t).contiguous()
    ops.conv16p(d16(x1), dd(sc1), dd(sh1), pw, out, B=B, Lin=L, stride=s, pad=p, x2=d16(x2), sc2=dd(sc2), sh2=dd(sh2),
                gn_partial=part, gn_group=cout // G, skip_pw=spw, out_s=out_s, gn_partial_s=part_s)
    tol = 8 * EPS[dt]                      # one 16-bit rounding of the operand may differ (erf approximation 1.5e-7)
    report("conv16p k%d s%d %d->%d in%d" % (k, s, cin, cout, 2 if two else 1), out.cpu(), ref, tol)
    sums = part.cpu().double().sum(dim=1)                                       # [B, G, 2]
    rg = ref.double().reshape(B, Lout, G, cout // G)
    assert maxerr(sums[..., 0], rg.sum(dim=(1, 3))) < 2e-2 * Lout ** 0.5 and maxerr(sums[..., 1], (rg ** 2).sum(dim=(1, 3))) < 5e-2 * Lout ** 0.5
    if skip:
        refs = F.conv1d(xn, q16(ws, dt), bs, stride=2).transpose(1, 2)
        report("conv16p fused skip conv", out_s.cpu(), refs, tol)
        sums = part_s.cpu().double().sum(dim=1)
        assert maxerr(sums[..., 0], refs.double().reshape(B, Lout, G, cout // G).sum(dim=(1, 3))) < 2e-2 * Lout ** 0.5
    # 16-bit output in the stage's format
    o16 = torch.empty(B, Lout, cout, device="cuda", dtype=dt)
    ops.conv16p(d16(x1), dd(sc1), dd(sh1), pw, o16, B=B, Lin=L, stride=s, pad=p, x2=d16(x2), sc2=dd(sc2), sh2=dd(sh2),
                skip_pw=spw, out_s=torch.empty_like(o16) if skip else None)
    report("conv16p 16-bit out", o16.float().cpu(), ref, tol + 2 * EPS[dt] * float(ref.abs().max()))


@pytest.mark.parametrize("cin,cout,k,s,p,two,skip,L", [(64, 128, 7, 2, 3, False, True, 64000), (128, 128, 3, 1, 1, False, False, 32000),
                                                       (256, 256, 5, 2, 2, True, False, 8000)])
def test_conv16p_statistics_and_determinism_at_scale(ops, cin, cout, k, s, p, two, skip, L):
    """thousands of tiles (B 8 at the bench's sequence lengths): the GroupNorm partial sums must equal the sums of the rows
    the same launch wrote (fp32 output), in every slot, and two launches must agree bit for bit.  (A first version of the
    epilogue, whose sum / sum-of-squares updates the compiler had packed into v_pk_* pairs, lost part of the sum of squares in
    ~40 of 128 000 slots per launch, different ones each time: invisible at unit-test sizes.)"""
    ops.set_compute_dtype(torch.float16)
    dt, B, G = torch.float16, 8, 16
    g = torch.Generator(device="cuda").manual_seed(5)
    R = lambda *shape: torch.randn(*shape, device="cuda", generator=g)
    x1, x2 = R(B, L, cin).to(dt), (R(B, L, cin).to(dt) if two else None)
    sc1, sh1 = R(B, cin) * 0.1 + 1, R(B, cin) * 0.1
    sc2, sh2 = (R(B, cin) * 0.1 + 1, R(B, cin) * 0.1) if two else (None, None)
    pw = ops.pack_linear(R(cout, cin, k) / (cin * k) ** 0.5, R(cout))
    spw = ops.pack_linear(R(cout, cin, 1) / cin ** 0.5, R(cout)) if skip else None
    Lout = (L + 2 * p - k) // s + 1
    P = 2 * ((Lout + 127) // 128)

    def run():
        out = torch.empty(B, Lout, cout, device="cuda", dtype=torch.float32)
        part = torch.zeros(B, P, G, 2, device="cuda")
        outs = torch.empty(B, Lout, cout, device="cuda", dtype=torch.float32) if skip else None
        parts = torch.zeros(B, P, G, 2, device="cuda") if skip else None
        ops.conv16p(x1, sc1, sh1, pw, out, B=B, Lin=L, stride=s, pad=p, x2=x2, sc2=sc2, sh2=sh2, gn_partial=part,
                    gn_group=cout // G, skip_pw=spw, out_s=outs, gn_partial_s=parts)
        return [(out, part)] + ([(outs, parts)] if skip else [])
    first, second = run(), run()
    for (o1, p1), (o2, p2) in zip(first, second):
        assert torch.equal(o1, o2) and torch.equal(p1, p2)
        pad_rows = P * 64 - Lout
        rows = torch.cat([o1.double(), torch.zeros(B, pad_rows, cout, device="cuda", dtype=torch.float64)], dim=1)
        rows = rows.reshape(B, P, 64, G, cout // G)
        want = torch.stack([rows.sum(dim=(2, 4)), (rows ** 2).sum(dim=(2, 4))], dim=-1)
        err = (p1.double() - want).abs() / (1.0 + want.abs())
        assert float(err.max()) < 1e-4, float(err.max())


@pytest.mark.parametrize("cin,cout,k,s,p,L", [(64, 128, 7, 2, 3, 64000), (256, 256, 1, 1, 0, 8000)])
def test_gemm16_statistics_and_determinism_at_scale(ops, cin, cout, k, s, p, L):
    """the same at-scale check for sfm_gemm16's GroupNorm partial sums (the training-mode convs and the latent heads use
    them): exact against the fp32 rows of the same launch in every slot, bitwise equal between two launches"""
    ops.set_compute_dtype(torch.float16)
    dt, B, G = torch.float16, 8, 16
    g = torch.Generator(device="cuda").manual_seed(6)
    R = lambda *shape: torch.randn(*shape, device="cuda", generator=g)
    x = R(B, L, cin).to(dt)
    pw = ops.pack_linear(R(cout, cin, k) / (cin * k) ** 0.5, R(cout))
    Lout = (L + 2 * p - k) // s + 1
    P = 2 * ((Lout + 127) // 128)

    def run():
        out = torch.empty(B, Lout, cout, device="cuda", dtype=torch.float32)
        part = torch.zeros(B, P, G, 2, device="cuda")
        ops.gemm16(x, pw, out, B=B, Lout=Lout, Lin=L, a_batch_stride=L * cin, ldo=cout, o_batch_stride=Lout * cout, stride=s, pad=p,
                   gn_partial=part, gn_group=cout // G)
        return out, part
    (o1, p1), (o2, p2) = run(), run()
    assert torch.equal(o1, o2) and torch.equal(p1, p2)
    rows = torch.cat([o1.double(), torch.zeros(B, P * 64 - Lout, cout, device="cuda", dtype=torch.float64)], dim=1)
    rows = rows.reshape(B, P, 64, G, cout // G)
    want = torch.stack([rows.sum(dim=(2, 4)), (rows ** 2).sum(dim=(2, 4))], dim=-1)
    err = (p1.double() - want).abs() / (1.0 + want.abs())
    assert float(err.max()) < 1e-4, float(err.max())


def test_framed_gemm_statistics_and_determinism_at_scale(ops):
    """and for sfm_framed_gemm_f32 (the SincConv1d front-end of the module API and of training): B 8 x 4 s, 64 filters x 251
    taps, GroupNorm(8) partial sums per 32-row slot"""
    B, L, N, K = 8, 64000, 64, 251
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(B, L, device="cuda", generator=g) * 0.3
    Wt = torch.zeros(256, N, device="cuda")
    Wt[:K] = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
    P = 4 * ((L + 127) // 128)

    def run():
        out = torch.empty(B, L, N, device="cuda", dtype=torch.float32)
        part = torch.zeros(B, P, 8, 2, device="cuda")
        ops.framed_gemm(x, Wt, out, B=B, M=L, Ls=L, sig_batch_stride=L, hop=1, padl=125, K=K, N=N, o_batch_stride=L * N, ldm=N, ldn=1,
                        mode=0, gn_partial=part, gn_group=8)
        return out, part
    (o1, p1), (o2, p2) = run(), run()
    assert torch.equal(o1, o2) and torch.equal(p1, p2)
    rows = torch.cat([o1.double(), torch.zeros(B, P * 32 - L, N, device="cuda", dtype=torch.float64)], dim=1).reshape(B, P, 32, 8, 8)
    want = torch.stack([rows.sum(dim=(2, 4)), (rows ** 2).sum(dim=(2, 4))], dim=-1)
    err = (p1.double() - want).abs() / (1.0 + want.abs())
    assert float(err.max()) < 1e-4, float(err.max())


def test_conv16p_refuses_shapes_it_is_not_built_for(ops):
    ops.set_compute_dtype(torch.float16)
    x = torch.zeros(1, 64, 64, device="cuda", dtype=torch.float16)
    sc = torch.ones(1, 64, device="cuda")
    pw = ops.pack_linear(torch.zeros(128, 64, 3, device="cuda"))
    with pytest.raises(RuntimeError):
        ops.conv16p(x, sc, sc, pw, torch.empty(1, 32, 128, device="cuda"), B=1, Lin=64, stride=2, pad=1)


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("fs,scaled", [(16000, False), (16000, True), (8000, True)])
def test_sinc_filters_and_fir(ops, fs, scaled):
    ops.set_compute_dtype(torch.bfloat16)
    p = orc.sinc_init(64, 251, fs)
    if scaled:
        p["low_hz_"], p["band_hz_"] = p["low_hz_"] * (fs / 8.0), p["band_hz_"] * (fs / 8.0)
    ref_f = orc.sinc_filters(p["low_hz_"], p["band_hz_"], p["window"], p["n_"], fs)
    filt, Wt = ops.sinc_filters(dev(p["low_hz_"].reshape(-1)), dev(p["band_hz_"].reshape(-1)), dev(p["window"]),
                                dev(p["n_"].reshape(-1)), 64, 251, fs, 50.0, 50.0)
    report("sinc filters fs%d scaled=%s" % (fs, scaled), filt.cpu(), ref_f, 2e-6)
    B, L = 3, 1000
    x = arr("sx", (B, L), 20, 0.3)
    out = torch.empty(B, L, 64, device="cuda", dtype=torch.float32)
    part = torch.zeros(B, 4 * ((L + 127) // 128), 8, 2, device="cuda")
    ops.framed_gemm(dev(x), Wt, out, B=B, M=L, Ls=L, sig_batch_stride=L, hop=1, padl=125, K=251, N=64,
                    o_batch_stride=L * 64, ldm=64, ldn=1, mode=0, gn_partial=part, gn_group=8)
    ref = orc.sinc_conv(x, ref_f)
    report("sinc FIR (f32 mfma)", out.cpu().transpose(1, 2), ref, 2e-6)
    gw, gb = arr("sgw", (64,), 21) * 0.1 + 1.0, arr("sgb", (64,), 22) * 0.1
    sc, sh = ops.gn_finalize(part, dev(gw), dev(gb), B, part.shape[1], 8, 64, L)
    y = torch.empty(B, L, 64, device="cuda", dtype=torch.float32)
    ops.gn_apply(out, sc, sh, y, B, L, 64, act=1)
    report("sinc GN(8)+GELU", y.cpu().transpose(1, 2), orc.gelu(orc.group_norm(ref, 8, gw, gb)), 2e-4)
    # channels-first fp32 output (module API layout)
    out_cf = torch.empty(B, 64, L, device="cuda", dtype=torch.float32)
    ops.framed_gemm(dev(x), Wt, out_cf, B=B, M=L, Ls=L, sig_batch_stride=L, hop=1, padl=125, K=251, N=64,
                    o_batch_stride=L * 64, ldm=1, ldn=L, mode=0)
    report("sinc FIR channels-first", out_cf.cpu(), ref, 2e-6)


@pytest.mark.parametrize("L", [1600, 1637, 479, 200])
def test_stft_istft(ops, L):
    ops.set_compute_dtype(torch.bfloat16)
    B = 2
    x = arr("stx", (B, L), 30, 0.3)
    T = 1 + L // 80
    Wst = ops.stft_matrix(256, 160, "cuda")
    re = torch.empty(B, T, 129, device="cuda")
    im = torch.empty(B, T, 129, device="cuda")
    ops.framed_gemm(dev(x), Wst, re, B=B, M=T, Ls=L, sig_batch_stride=L, hop=80, padl=80, K=160, N=258,
                    o_batch_stride=T * 129, ldm=129, ldn=1, mode=1, out2=im, nsplit=129)
    rr, ri = orc.stft(x)
    report("stft real L%d" % L, re.cpu(), rr, 2e-5)
    report("stft imag L%d" % L, im.cpu(), ri, 2e-5)
    # istft of a modified spectrum
    pr, pi = rr * 0.7 - ri * 0.2, ri * 0.9 + rr * 0.1
    Wi, win2 = ops.istft_matrix(256, 160, "cuda")
    spec = torch.zeros(B * T, 264, device="cuda")
    ops.pack_spec(dev(pr), dev(pi), spec, B * T, 129, 264, 129)
    frames = torch.empty(B * T, 160, device="cuda")
    ops.framed_gemm(spec, Wi, frames, B=1, M=B * T, Ls=B * T * 264, sig_batch_stride=0, hop=264, padl=0, K=258,
                    N=160, o_batch_stride=0, ldm=160, ldn=1, mode=0)
    y = torch.empty(B, L, device="cuda")
    ops.istft_ola(frames, win2, y, B, T, L, 256, 80, 160, 160)
    report("istft L%d" % L, y.cpu(), orc.istft(pr, pi, L), 2e-5)


# ---------------------------------------------------------------------------
def _attn_ref(qkv, B, T, H, hd):
    D = H * hd
    q, k, v = qkv.reshape(B, T, 3 * D).split(D, dim=-1)
    q = q.reshape(B, T, H, hd).transpose(1, 2)
    k = k.reshape(B, T, H, hd).transpose(1, 2)
    v = v.reshape(B, T, H, hd).transpose(1, 2)
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    return (p @ v).transpose(1, 2).reshape(B * T, D)


@pytest.mark.parametrize("variant", [1, 3, 4, 5, 6])    # the head_dim-64 kernels: 32 query rows per wave / persistent ring / pipelined persistent (8 x 64, 4 x 64, 4 x 128 rows)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,T,H,hd", [(2, 200, 4, 64), (1, 64, 4, 64), (3, 1, 2, 64), (1, 801, 4, 64), (2, 129, 1, 64),
                                      (1, 1100, 2, 64), (2, 20, 4, 16), (1, 37, 2, 32)])
def test_attention(ops, dt, B, T, H, hd, variant):
    ops.set_compute_dtype(dt)
    ops.set_attention_variant(variant)
    try:
        qkv = arr("aq", (B * T, 3 * H * hd), 40 + T, 1.5)
        out = ops.attention(dev(qkv).to(dt).contiguous(), B, T, H, hd)
    finally:
        ops.set_attention_variant(0)
    ref = _attn_ref(q16(qkv, dt), B, T, H, hd)
    report("attention v%d %s B%d T%d H%d hd%d" % (variant, dt, B, T, H, hd), out.float().cpu(), ref, 6 * EPS[dt])


@pytest.mark.parametrize("variant", [3, 4, 5])
def test_attention_persistent_kernel_walks_several_items_per_workgroup(ops, variant):
    """more (batch, head, query tile) items than CUs: every workgroup of the ring kernel handles two items, the second one's
    Q / first K,V group prefetched under the first; ragged T (keys and query rows beyond T are zero-filled / dropped by the
    buffer descriptors), two query tiles per (batch, head)"""
    ops.set_compute_dtype(torch.bfloat16)
    B, T, H, hd = 40, 700, 4, 64                      # 2 query tiles x 4 heads x 40 = 320 items > 256 CUs
    qkv = arr("aq3", (B * T, 3 * H * hd), 91, 1.0)
    ops.set_attention_variant(variant)
    try:
        out = ops.attention(dev(qkv).to(torch.bfloat16).contiguous(), B, T, H, hd)
    finally:
        ops.set_attention_variant(0)
    ref = _attn_ref(q16(qkv, torch.bfloat16), B, T, H, hd)
    report("attention persistent v%d, 320 items" % variant, out.float().cpu(), ref, 6 * EPS[torch.bfloat16])


@pytest.mark.parametrize("variant", [1, 3, 4, 5])
def test_attention_result_in_the_other_16bit_format(ops, variant):
    """precision policy: bf16 attention core, O written as fp16 (and the reverse)"""
    B, T, H, hd = 2, 300, 4, 64
    qkv = arr("aq4", (B * T, 3 * H * hd), 92, 1.0)
    for dt, odt in ((torch.bfloat16, torch.float16), (torch.float16, torch.bfloat16)):
        ops.set_compute_dtype(dt)
        ops.set_attention_variant(variant)
        try:
            out = ops.attention(dev(qkv).to(dt).contiguous(), B, T, H, hd, out_dtype=odt)
        finally:
            ops.set_attention_variant(0)
        assert out.dtype == odt
        ref = _attn_ref(q16(qkv, dt), B, T, H, hd)
        report("attention v%d %s -> %s" % (variant, dt, odt), out.float().cpu(), ref, 6 * max(EPS[dt], EPS[odt]))


@pytest.mark.parametrize("variant", [1, 3, 4, 5])
def test_attention_online_softmax_rescale(ops, variant):
    """a spiked key late in the sequence forces the running-max rescale branch"""
    ops.set_compute_dtype(torch.float16)
    B, T, H, hd = 1, 300, 1, 64
    qkv = arr("aq2", (B * T, 3 * hd), 77, 0.5)
    qkv[5, :hd] = 3.0
    qkv[250, hd:2 * hd] = 3.0          # key 250 aligned with query 5: score jumps in the 4th tile
    ops.set_attention_variant(variant)
    try:
        out = ops.attention(dev(qkv).half().contiguous(), B, T, H, hd)
    finally:
        ops.set_attention_variant(0)
    ref = _attn_ref(q16(qkv, torch.float16), B, T, H, hd)
    report("attention rescale v%d" % variant, out.float().cpu(), ref, 3e-3)


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,D", [(100, 256), (101, 256), (7, 256), (33, 64), (17, 258)])   # D 256: the two-rows-per-wave kernel, odd M = its tail
def test_layernorm(ops, dt, M, D):
    ops.set_compute_dtype(dt)
    x, w, b = arr("lx", (M, D), 50, 2.0) + 0.3, arr("lw", (D,), 51) * 0.1 + 1, arr("lb", (D,), 52) * 0.1
    o16 = torch.empty(M, D, device="cuda", dtype=dt)
    o32 = torch.empty(M, D, device="cuda")
    ops.layernorm(dev(x), dev(w), dev(b), out16=o16, out32=o32)
    ref = orc.layer_norm(x, w, b)
    report("layernorm fp32 D%d" % D, o32.cpu(), ref, 2e-5)
    report("layernorm 16b D%d" % D, o16.float().cpu(), ref, 8 * EPS[dt])
    ops.layernorm(dev(x), dev(w), dev(b), out32=o32, act=1)
    report("layernorm+gelu", o32.cpu(), orc.gelu(ref), 2e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C,KS,T", [(256, 31, 200), (64, 7, 20), (256, 31, 64), (256, 31, 7)])
def test_dwconv_bn_swish(ops, dt, C, KS, T):
    ops.set_compute_dtype(dt)
    B = 2
    x = arr("dx", (B, T, C), 60)
    w, b = arr("dw", (C, 1, KS), 61) / math.sqrt(KS), arr("db", (C,), 62) * 0.1
    bw, bb = arr("dbw", (C,), 63) * 0.1 + 1, arr("dbb", (C,), 64) * 0.1
    rm, rv = arr("drm", (C,), 65) * 0.1, torch.rand(C, generator=torch.Generator().manual_seed(1)) + 0.5
    out = ops.dwconv_bn_swish(dev(x).to(dt).contiguous(), dev(w.reshape(C, KS).contiguous()), dev(b), dev(bw), dev(bb),
                              dev(rm), dev(rv), B, T, C)
    h = F.conv1d(q16(x, dt).transpose(1, 2), w, b, padding=(KS - 1) // 2, groups=C)
    ref = orc.swish(orc.batch_norm_eval(h, bw, bb, rm, rv)).transpose(1, 2)
    report("dwconv C%d k%d T%d" % (C, KS, T), out.float().cpu(), ref, 8 * EPS[dt])


def test_layout_kernels(ops):
    ops.set_compute_dtype(torch.float16)
    B, C, T = 2, 256, 37
    z = arr("tz", (B, C, T), 70)
    dst = torch.zeros(B * T, 1056, device="cuda", dtype=torch.float16)
    ops.transpose(dev(z), dst[:, 256:], B, C, T, C * T, T, T * 1056, 1056)      # [B,C,T] -> cols 256.. of [B*T, 1056]
    report("transpose cf->cl16", dst.float().cpu()[:, 256:512].reshape(B, T, C), z.transpose(1, 2), 2e-3)
    assert float(dst[:, :256].abs().max()) == 0.0 and float(dst[:, 512:].abs().max()) == 0.0
    back = torch.empty(B, C, T, device="cuda")
    ops.transpose(dst[:, 256:], back, B, T, C, T * 1056, 1056, C * T, T)
    report("transpose cl16->cf32", back.cpu(), z.half().float(), 1e-6)
    src = arr("cs", (50, 64), 71)
    d2 = torch.full((50, 100), 7.0, device="cuda", dtype=torch.float16)
    ops.convert_rows(dev(src), d2, 50, 64, 80, 64, 100)
    assert maxerr(d2[:, :64].float().cpu(), src) < 2e-3 and float(d2[:, 64:80].abs().max()) == 0 and float(d2[0, 80]) == 7.0
    # the 8-columns-per-thread form (16-byte aligned rows on both sides): same result, padding columns zero, beyond them untouched
    d3 = torch.full((50, 104), 7.0, device="cuda", dtype=torch.float16)
    ops.convert_rows(dev(src), d3, 50, 64, 80, 64, 104)
    assert torch.equal(d3[:, :80], d2[:, :80]) and float(d3[0, 80]) == 7.0 and float(d3[49, 103]) == 7.0
    zz = arr("pz", (B, 100, 512), 72)
    p16 = torch.empty(B, 21, 512, device="cuda", dtype=torch.float16)
    p32 = torch.empty(B, 21, 512, device="cuda")
    ops.pool_time(dev(zz), p16, p32, B, 100, 21, 512, 512, 512)
    refp = orc.pool_latents(zz.transpose(1, 2), 21).transpose(1, 2)
    report("pool_time", p32.cpu(), refp, 1e-6)
    report("pool_time 16", p16.float().cpu(), refp, 2e-3)
    # 16-bit source in EITHER format, whatever the destination's (a policy may run the PerceptionAgent in one format and the
    # front-end in the other), with the per-(utterance, channel) affine of a pooled GroupNorm
    sc, sh = arr("psc", (B, 512), 75, 0.2) + 1.0, arr("psh", (B, 512), 76, 0.2)
    for sdt in (torch.float16, torch.bfloat16):
        zq = zz.to(sdt)
        q32 = torch.empty(B, 21, 512, device="cuda")
        ops.pool_time(dev(zq), None, q32, B, 100, 21, 512, 512, 512, scale=dev(sc), shift=dev(sh))
        refq = orc.pool_latents(zq.float().transpose(1, 2), 21).transpose(1, 2) * sc[:, None, :] + sh[:, None, :]
        report("pool_time from %s + affine" % sdt, q32.cpu(), refq, 2e-6)
    re, im = arr("nr", (40, 129), 73, 0.5), arr("ni", (40, 129), 74, 0.5)
    pk = torch.full((40, 300), 3.0, device="cuda", dtype=torch.float16)
    ops.stft_lognorm_pack(dev(re), dev(im), pk, 40, 129, 30, 300)
    mag = torch.sqrt(re ** 2 + im ** 2 + 1e-8)
    nf = torch.log1p(mag) / mag
    report("stft lognorm pack", pk.float().cpu()[:, :258], torch.cat([re * nf, im * nf], dim=1), 2e-3)
    assert float(pk[:, 258:288].abs().max()) == 0 and float(pk[0, 288]) == 3.0


def test_polar_mask_and_complex_mul(ops):
    B, T, Fq = 2, 21, 129
    lm, lp = arr("plm", (B * T, Fq), 80), arr("plp", (B * T, Fq), 81)
    nr, ni = arr("pnr", (B * T, Fq), 82), arr("pni", (B * T, Fq), 83)
    bias = arr("pb", (B, Fq), 84) * 0.3
    mr, mi, er, ei, mm = (torch.empty(B * T, Fq, device="cuda") for _ in range(5))
    ops.polar_mask(dev(lm), dev(lp), B, T, Fq, 3.14159 / 8.0, Fq, mag_bias=dev(bias), nr=dev(nr), ni=dev(ni), mr=mr,
                   mi=mi, er=er, ei=ei, mmag=mm, ld_enh=Fq)
    mag = torch.sigmoid(lm + bias.repeat_interleave(T, dim=0))
    ph = torch.tanh(lp) * (3.14159 / 8.0)
    rmr, rmi = mag * torch.cos(ph), mag * torch.sin(ph)
    report("polar mask real", mr.cpu(), rmr, 2e-6)
    report("polar mask imag", mi.cpu(), rmi, 2e-6)
    report("polar enh real", er.cpu(), rmr * nr - rmi * ni, 5e-6)
    report("polar enh imag", ei.cpu(), rmr * ni + rmi * nr, 5e-6)
    report("polar mag", mm.cpu(), mag, 2e-6)
    cr, ci = ops.complex_mul(dev(nr), dev(ni), dev(rmr), dev(rmi))
    report("complex mul", cr.cpu(), rmr * nr - rmi * ni, 2e-6)
    report("complex mul i", ci.cpu(), rmr * ni + rmi * nr, 2e-6)


@pytest.mark.parametrize("B,T", [(2, 21), (3, 130)])
def test_bilstm_layer(ops, B, T):
    H, I = 128, 256
    sd = synth_sd("CorrelationPhaseEstimationAgent", 61)
    x = arr("lsx", (B, T, I), 90)
    lsd = orc.sub(sd, "lstm")
    xg = []
    whh = []
    for sfx in ("", "_reverse"):
        xg.append(x @ lsd["weight_ih_l0" + sfx].t() + lsd["bias_ih_l0" + sfx] + lsd["bias_hh_l0" + sfx])
        whh.append(lsd["weight_hh_l0" + sfx])
    xg = torch.stack(xg, dim=2).contiguous()            # [B, T, 2, 4H]
    whh = torch.stack(whh, dim=0).contiguous()          # [2, 4H, H]
    out = ops.bilstm_layer(dev(xg), dev(whh), B, T, H)
    f = orc._lstm_dir(x, lsd["weight_ih_l0"], lsd["weight_hh_l0"], lsd["bias_ih_l0"], lsd["bias_hh_l0"], False)
    r = orc._lstm_dir(x, lsd["weight_ih_l0_reverse"], lsd["weight_hh_l0_reverse"], lsd["bias_ih_l0_reverse"],
                      lsd["bias_hh_l0_reverse"], True)
    report("bilstm layer T%d" % T, out.cpu(), torch.cat([f, r], dim=-1), 2e-5)
    # the fused inference path's form: recurrent product on fp16 operands (W_hh and h rounded once per use, fp32 accumulation; the
    # error does not grow with T: h in (-1, 1), rounding <= 4.9e-4 relative per operand)
    out16 = ops.bilstm_layer(dev(xg), dev(whh), B, T, H, w16=True)
    report("bilstm layer fp16 recurrence T%d" % T, out16.cpu(), torch.cat([f, r], dim=-1), 2e-3)
    assert not torch.equal(out16, out)                     # the other kernel did run


def test_bilstm_layer_train_fp16_recurrence_saves_consistent_state(ops):
    """training forward on the fp16-operand recurrence (sfm_bilstm_layer_train_ex): output, saved gates and cell states against the
    fp32 kernel's; the saved state is what the BPTT consumes, so it must describe THIS forward (out = o * tanh(c) exactly)"""
    g = torch.Generator().manual_seed(6)
    H, B, T = 128, 3, 25
    xg = torch.randn(B, T, 2, 4 * H, generator=g)
    whh = torch.randn(2, 4 * H, H, generator=g) / H ** 0.5
    o32, s32 = ops.bilstm_layer_train(dev(xg), dev(whh), B, T, H, w16=False)
    o16, s16 = ops.bilstm_layer_train(dev(xg), dev(whh), B, T, H, w16=True)
    report("bilstm train fp16 recurrence: out", o16.cpu(), o32.cpu(), 2e-3)
    report("bilstm train fp16 recurrence: saved gates / cell", s16.cpu(), s32.cpu(), 4e-3)
    for d in range(2):
        h = s16[:, :, d, 3] * torch.tanh(s16[:, :, d, 4])
        report("saved state reproduces the output, dir %d" % d, h.cpu(), o16[:, :, d * H:(d + 1) * H].cpu(), 2e-6)
    assert not torch.equal(o16, o32)


def test_bilstm_layer_fp16_recurrence_other_sizes(ops):
    """sfm_bilstm_layer_ex: hidden 64 takes the fp16-operand kernel too, hidden 32 silently keeps the fp32 one (its 4-float slices
    are too short for the 16-byte reads); a batch larger than the CU count (two chains per CU)"""
    g = torch.Generator().manual_seed(5)
    for H, B, T in ((64, 3, 40), (32, 2, 17), (128, 300, 9)):
        xg = torch.randn(B, T, 2, 4 * H, generator=g)
        whh = torch.randn(2, 4 * H, H, generator=g) / H ** 0.5
        ref = ops.bilstm_layer(dev(xg), dev(whh), B, T, H).cpu()
        out = ops.bilstm_layer(dev(xg), dev(whh), B, T, H, w16=True).cpu()
        report("bilstm fp16 recurrence H%d B%d" % (H, B), out, ref, 2e-3)
        assert torch.equal(out, ref) == (H == 32)


def test_memory(ops):
    from sincformer_metacog_speech_enhancement_amd.functional import pack_memory_params
    sd = synth_sd("EpisodicMemory", 71)
    e = arr("g7_e", (3, 256), 72)
    params = pack_memory_params({k: dev(v) for k, v in sd.items()})
    bias, gate, top, sim = ops.memory_fwd(dev(e), params, 256, 129, 64, 1.0)
    ref = orc.memory_forward(sd, e)
    report("memory bias", bias.cpu(), ref["bias"], 2e-5)
    report("memory gate", gate.cpu(), ref["gate"], 2e-5)
    report("memory sim", sim.cpu(), ref["similarity"], 2e-5)
    assert np.array_equal(top.cpu().numpy().astype(np.int64), ref["top_indices"].numpy())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("L", [1000, 8192, 20011])
def test_sinc_fir16_split_mfma(ops, dt, L):
    """16-bit split-operand FIR (hi/lo x hi/lo, 3 MFMA passes) against the fp32 conv"""
    ops.set_compute_dtype(dt)
    fs = 16000
    p = orc.sinc_init(64, 251, fs)
    p["low_hz_"], p["band_hz_"] = p["low_hz_"] * (fs / 8.0), p["band_hz_"] * (fs / 8.0)
    ref_f = orc.sinc_filters(p["low_hz_"], p["band_hz_"], p["window"], p["n_"], fs)
    filt, _ = ops.sinc_filters(dev(p["low_hz_"].reshape(-1)), dev(p["band_hz_"].reshape(-1)), dev(p["window"]),
                               dev(p["n_"].reshape(-1)), 64, 251, fs, 50.0, 50.0)
    B = 2
    x = arr("fx", (B, L), 23, 0.3)
    out = torch.empty(B, L, 64, device="cuda", dtype=torch.float32)
    part, P = ops.sinc_fir16(dev(x), filt, out, B, L, 64, 251)
    ref = orc.sinc_conv(x, ref_f)
    report("sinc fir16 %s L%d" % (dt, L), out.cpu().transpose(1, 2), ref, 2e-6 if dt is torch.float16 else 2e-5)
    gw, gb = arr("sgw", (64,), 21) * 0.1 + 1.0, arr("sgb", (64,), 22) * 0.1
    sc, sh = ops.gn_finalize(part, dev(gw), dev(gb), B, P, 8, 64, L)
    y = torch.empty(B, L, 64, device="cuda", dtype=torch.float32)
    ops.gn_apply(out, sc, sh, y, B, L, 64, act=1)
    report("sinc fir16 GN(8)+GELU", y.cpu().transpose(1, 2), orc.gelu(orc.group_norm(ref, 8, gw, gb)), 3e-4)
    o16 = torch.empty(B, L, 64, device="cuda", dtype=dt)
    ops.sinc_fir16(dev(x), filt, o16, B, L, 64, 251, passes=3)
    report("sinc fir16 16-bit out, split operands", o16.float().cpu().transpose(1, 2), ref, 2 * EPS[dt] * 0.1)
    # one pass (the default for fp16 operands with a 16-bit result): waveform and taps rounded once to the operand format - 251
    # products with ~2^-11 (fp16) / 2^-8 (bf16) relative error each on top of the result's own rounding
    ops.sinc_fir16(dev(x), filt, o16, B, L, 64, 251, passes=1)
    report("sinc fir16 16-bit out, one pass", o16.float().cpu().transpose(1, 2), ref, 4 * EPS[dt] * 0.1)      # observed 1.1e-4 (fp16) / 8.4e-4 (bf16): 1.8 x the split form
    auto = torch.empty_like(o16)
    ops.sinc_fir16(dev(x), filt, auto, B, L, 64, 251)
    ops.sinc_fir16(dev(x), filt, o16, B, L, 64, 251, passes=1 if dt is torch.float16 else 3)
    assert torch.equal(auto, o16)                      # passes = 0: one pass for fp16 + 16-bit result, split operands otherwise


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,FF", [(300, 1024), (128, 64), (1000, 256)])
def test_ffn_fused(ops, dt, M, FF):
    """LN -> W1 -> swish -> W2 -> half-step residual in one launch vs fp32 math on 16-bit-rounded weights"""
    ops.set_compute_dtype(dt)
    D = 256
    x = arr("ffx", (M, D), 91, 1.5) + 0.2
    lw, lb = arr("fflw", (D,), 92) * 0.1 + 1.0, arr("fflb", (D,), 93) * 0.1
    w1, b1 = arr("ffw1", (FF, D), 94) / 16.0, arr("ffb1", (FF,), 95) * 0.1
    w2, b2 = arr("ffw2", (D, FF), 96) / math.sqrt(FF), arr("ffb2", (D,), 97) * 0.1
    out = ops.ffn_fused(dev(x), dev(lw), dev(lb), dev(w1).to(dt).contiguous(), dev(b1), dev(w2).to(dt).contiguous(), dev(b2))
    h = q16(orc.layer_norm(x, lw, lb), dt)
    u = q16(orc.swish(h @ q16(w1, dt).t() + b1), dt)
    ref = x + 0.5 * (u @ q16(w2, dt).t() + b2)
    report("ffn fused %s M%d FF%d" % (dt, M, FF), out.cpu(), ref, 6 * EPS[dt])
    assert rmse(out.cpu(), ref) < EPS[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C,KS,T", [(256, 31, 200), (64, 7, 20), (256, 31, 801), (128, 7, 3)])
def test_dwconv_folded(ops, dt, C, KS, T):
    ops.set_compute_dtype(dt)
    B = 2
    x = arr("dx", (B, T, C), 60)
    w, b = arr("dw", (C, 1, KS), 61) / math.sqrt(KS), arr("db", (C,), 62) * 0.1
    bw, bb = arr("dbw", (C,), 63) * 0.1 + 1, arr("dbb", (C,), 64) * 0.1
    rm, rv = arr("drm", (C,), 65) * 0.1, torch.rand(C, generator=torch.Generator().manual_seed(1)) + 0.5
    sc = bw / torch.sqrt(rv + 1e-5)
    sh = bb - rm * sc + b * sc
    out = ops.dwconv_folded(dev(x).to(dt).contiguous(), dev(w.reshape(C, KS).t().contiguous()), dev(sc), dev(sh), B, T, C)
    h = F.conv1d(q16(x, dt).transpose(1, 2), w, b, padding=(KS - 1) // 2, groups=C)
    ref = orc.swish(orc.batch_norm_eval(h, bw, bb, rm, rv)).transpose(1, 2)
    report("dwconv folded C%d k%d T%d" % (C, KS, T), out.float().cpu(), ref, 8 * EPS[dt])


@pytest.mark.parametrize("n_fft,hop,win,L", [(256, 64, 256, 4000), (512, 128, 512, 4321), (1024, 256, 1024, 9000),
                                             (256, 80, 160, 1637)])
def test_stft_split16_vs_oracle(ops, n_fft, hop, win, L):
    """STFT on the 16-bit matrix cores with split bf16 operands (the objective's STFTs): error budget 2e-5 of the rms"""
    from sincformer_metacog_speech_enhancement_amd import functional as Fn
    w = arr("sw", (3, L), 300 + L, 0.2)
    rr, ri = orc.stft(w, n_fft, hop, win)
    gr, gi = Fn.stft_split16(dev(w), n_fft, hop, win)
    rms = float(torch.cat([rr, ri], -1).pow(2).mean().sqrt())
    report("stft split16 re n%d" % n_fft, gr.cpu(), rr, 2e-5 * rms * 8)
    report("stft split16 im n%d" % n_fft, gi.cpu(), ri, 2e-5 * rms * 8)
    assert rmse(torch.cat([gr, gi], -1).cpu(), torch.cat([rr, ri], -1)) < 2e-5 * rms
