"""ComplexConformer at the configuration the reference's own plumbing test uses (n_freq 32, d_model 64, 2 blocks,
4 heads -> head_dim 16, d_ff 128, depthwise kernel 7: the small-shape attention / depthwise kernels), in the module's
default train() mode: dropout is active (`dropout=0.0` silently becomes 0.1 through the `x or config.X` defaulting of
models/conformer.py:179) and BatchNorm uses batch statistics.  The reference checks plumbing only; the
checks below are this build's own: plumbing properties of the HIP training-mode forward / backward, then a numeric
comparison against torch autograd of the oracle.  The eval-mode outputs of this configuration are pinned by the reference's
golden vector g4_cconf_small (tests/test_modules_gpu.py)."""
import numpy as np
import pytest
import torch

from helpers import arr, maxerr, rmse, synth_sd
from oracle import sfm_oracle as orc

pytestmark = pytest.mark.gpu
SMALL = dict(n_freq=32, d_model=64, num_blocks=2, num_heads=4, d_ff=128, kernel_size=7, dropout=0.0)


def _small_conformer(seed=0):
    from sincformer_metacog_speech_enhancement_amd.models.conformer import ComplexConformer
    torch.manual_seed(seed)
    return ComplexConformer(**SMALL).cuda()


def _spectrum(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).cuda(), torch.randn(*shape, generator=g).cuda()


@pytest.mark.parametrize("shape", [(2, 20, 32), (1, 10, 32), (3, 7, 32)])
def test_train_mode_masks_keep_the_spectrum_shape_and_stay_finite(shape):
    net = _small_conformer()
    assert net.training                                        # constructed modules are in train() mode, like the reference's
    re, im = _spectrum(shape, 1)
    m_re, m_im = net(re, im)
    for m in (m_re, m_im):
        assert tuple(m.shape) == shape and m.dtype == torch.float32 and bool(torch.isfinite(m).all())
    # dropout is live in this mode (the 0.0 -> 0.1 quirk): two passes over the same input differ
    m_re2, _ = net(re, im)
    assert not torch.equal(m_re, m_re2)


def test_apply_mask_is_the_complex_product_of_spectrum_and_mask():
    net = _small_conformer()
    re, im = _spectrum((2, 10, 32), 2)
    m_re, m_im = net(re, im)
    e_re, e_im = net.apply_mask(re, im, m_re, m_im)
    assert e_re.shape == re.shape and e_im.shape == im.shape
    want = torch.complex(re, im) * torch.complex(m_re.detach(), m_im.detach())
    assert maxerr(e_re.detach().cpu(), want.real.cpu()) < 1e-5 and maxerr(e_im.detach().cpu(), want.imag.cpu()) < 1e-5


def test_backward_reaches_the_inputs_and_every_parameter():
    net = _small_conformer()
    re, im = _spectrum((1, 10, 32), 3)
    re.requires_grad_(True)
    im.requires_grad_(True)
    m_re, m_im = net(re, im)
    (m_re.sum() + m_im.sum()).backward()
    for t in (re, im):
        assert t.grad is not None and bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().max()) > 0
    missing = [n for n, p in net.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
    assert not missing, missing


def test_count_parameters_equals_the_sum_over_the_state():
    net = _small_conformer()
    assert net.count_parameters() == sum(p.numel() for p in net.parameters()) == 135424


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_small_config_train_mode_matches_oracle_autograd(dt):
    """same architecture (head_dim 16: the generic attention forward / backward kernels), dropout forced to 0, against
    torch autograd of the oracle with BatchNorm batch statistics: masks, input gradients, parameter gradients."""
    from sincformer_metacog_speech_enhancement_amd import ops
    from sincformer_metacog_speech_enhancement_amd.models.conformer import ComplexConformer
    ops.set_compute_dtype(dt)
    m = ComplexConformer(n_freq=32, d_model=64, num_blocks=2, num_heads=4, d_ff=128, kernel_size=7, dropout=0.0)
    sd = synth_sd("ComplexConformerSmall", 41)
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():                      # the ctor turned 0.0 into 0.1 (reference quirk); the check needs p = 0
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    m.cuda().train()
    B, T, F = 2, 24, 32
    sr, si = arr("ccr", (B, T, F), 45), arr("cci", (B, T, F), 46)
    cr, ci = arr("ccg", (B, T, F), 47), arr("cch", (B, T, F), 48)
    ref_sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
              for k, v in sd.items()}
    xr, xi = sr.clone().requires_grad_(True), si.clone().requires_grad_(True)
    x = torch.cat([xr, xi], dim=-1)
    x = orc.linear(x, ref_sd["input_proj.weight"], ref_sd["input_proj.bias"])
    skip = x
    for i in range(2):
        x = orc.conformer_block(x, orc.sub(ref_sd, "blocks.%d" % i), 4, bn_train=True)
    y = orc.linear(x + skip, ref_sd["output_proj.weight"], ref_sd["output_proj.bias"])
    (y[..., :F] * cr + y[..., F:] * ci).sum().backward()
    gr, gi = sr.cuda().requires_grad_(True), si.cuda().requires_grad_(True)
    mr, mi = m(gr, gi)
    (mr * cr.cuda() + mi * ci.cuda()).sum().backward()
    tol = 2e-3 if dt is torch.float16 else 1.5e-2
    e = rmse(torch.cat([mr, mi], -1).detach().cpu(), y.detach())
    print("small config train-mode masks %s: rmse %.3e" % (dt, e))
    assert e < tol
    rel = lambda a, b: rmse(a, b) / (float(b.double().pow(2).mean().sqrt()) + 1e-12)
    tol_g = 0.02 if dt is torch.float16 else 0.1
    assert rel(gr.grad.cpu(), xr.grad) < tol_g and rel(gi.grad.cpu(), xi.grad) < tol_g
    worst = ("", 0.0)
    for k, p_ in m.named_parameters():
        if k.endswith("depthwise.bias"):
            continue
        r = rel(p_.grad.cpu(), ref_sd[k].grad)
        if r > worst[1]:
            worst = (k, r)
    print("  worst parameter-gradient rel rmse: %s %.3e" % worst)
    assert worst[1] < tol_g, worst
