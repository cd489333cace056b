"""Module-level parity on the MI355X: the host mirrors (reference class names,
signatures, state_dict keys) running the HIP path, against
  (a) the committed golden vectors = outputs of the reference itself, and
  (b) the oracle on the same seeded inputs.
Tolerances are stated per test; the north-star bound is mask RMSE <= 1e-3.
"""
import math
import numpy as np
import pytest
import torch

from helpers import gold, synth_sd, arr, maxerr, rmse
from oracle import sfm_oracle as orc
from sincformer_metacog_speech_enhancement_amd import synthetic as syn

pytestmark = pytest.mark.gpu
# Operand formats under test: "mixed" = the DEFAULT of inference and what bench.py / smoke() run (ops.POLICIES["mixed"]:
# fp16 GEMM / conv operands everywhere, bf16 only in the attention core), "fp16" / "bf16" = one format everywhere
# (ops.set_compute_dtype).
PRECISIONS = ["mixed", "fp16", "bf16"]
MASK_RMSE_BOUND = 1e-3
# The golden MSA / path vectors use DE-SATURATED heads (mask magnitude ~0.5): the hardest regime for the
# north-star bound (mask RMSE <= 1e-3 against the reference's CPU output).  Measured per-stage error budget:
# profiles/r02/precision_probe.json — default policy 2.2e-4 (MaskSynthesisAgent) / 2.0e-4 (whole path) / 3.3e-4
# (SpeechEnhancer), fp16 everywhere 1.6-1.8e-4, bf16 everywhere 1.2-1.6e-3.  The default and fp16 must meet 1e-3 in the hard
# regime; uniform bf16 is a non-default diagnostic mode that must meet it at the reference's own initialisation (masks
# ~0.993) and stay < 2e-3 in the hard regime.
HARD_BOUND = {"mixed": 1e-3, "fp16": 1e-3, "bf16": 2e-3}


def set_prec(ops, prec):
    if prec == "mixed":
        ops.reset_precision()
    else:
        ops.set_compute_dtype(prec)


@pytest.fixture(scope="module")
def pkg():
    assert torch.cuda.is_available()
    import sincformer_metacog_speech_enhancement_amd as p
    from sincformer_metacog_speech_enhancement_amd import ops, functional
    from sincformer_metacog_speech_enhancement_amd.agents import (PerceptionAgent, SincConv1d, MaskSynthesisAgent,
                                                                    CorrelationPhaseEstimationAgent, EpisodicMemory)
    from sincformer_metacog_speech_enhancement_amd.models.conformer import ComplexConformer, ConformerBlock
    from sincformer_metacog_speech_enhancement_amd.training import conformer_pipeline as cp

    class NS:
        pass
    ns = NS()
    ns.ops, ns.Fn, ns.cp = ops, functional, cp
    ns.PerceptionAgent, ns.SincConv1d, ns.MaskSynthesisAgent = PerceptionAgent, SincConv1d, MaskSynthesisAgent
    ns.CPEA, ns.EpisodicMemory = CorrelationPhaseEstimationAgent, EpisodicMemory
    ns.ComplexConformer, ns.ConformerBlock = ComplexConformer, ConformerBlock
    return ns


def load(module, table, seed, **kw):
    module.load_state_dict(synth_sd(table, seed, **kw), strict=True)
    return module.cuda().eval()


def rel(got, ref):
    ref = torch.as_tensor(np.asarray(ref)).double()
    return rmse(got, ref) / max(float(ref.pow(2).mean().sqrt()), 1e-30)


def show(name, got, ref):
    got = got.detach().float().cpu() if isinstance(got, torch.Tensor) else got
    print("%-44s rmse %.3e  max %.3e  rel_rmse %.3e" % (name, rmse(got, ref), maxerr(got, ref), rel(got, ref)))
    return rmse(got, ref), rel(got, ref)


def test_cpu_tensors_are_refused(pkg):
    m = pkg.ConformerBlock(64, 4, 128, 7, 0.1).eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 64))
    with pytest.raises(RuntimeError):
        pkg.cp.batch_stft(torch.zeros(1, 800), 256, 80, 160)


@pytest.mark.parametrize("L", [1600, 1637, 479])
def test_batch_stft_istft_vs_golden(pkg, L):
    g = gold("g3_stft_L%d" % L)
    noisy, _ = syn.synth_wave(2, L, 31)
    r, i = pkg.cp.batch_stft(torch.from_numpy(noisy).cuda(), 256, 80, 160)
    assert r.shape == (2, 1 + L // 80, 129)
    assert maxerr(r.cpu(), g["real"]) < 2e-5 and maxerr(i.cpu(), g["imag"]) < 2e-5
    y = pkg.cp.batch_istft(torch.from_numpy(g["mod_real"]).cuda(), torch.from_numpy(g["mod_imag"]).cuda(), 256, 80, 160, L)
    assert y.shape == (2, L) and maxerr(y.cpu(), g["istft"]) < 2e-5


@pytest.mark.parametrize("fs,scaled", [(8000, False), (16000, True)])
def test_sincconv_module_vs_golden(pkg, fs, scaled):
    g = gold("g1_sinc_%sfs%d" % ("scaled_" if scaled else "", fs))
    m = pkg.SincConv1d(64, 251, sample_rate=fs)
    if scaled:
        with torch.no_grad():
            m.low_hz_.mul_(fs / 8.0)
            m.band_hz_.mul_(fs / 8.0)
    m = m.cuda().eval()
    assert maxerr(m.filters().cpu(), g["filters"]) < 1e-6
    y = m(arr("g1_wave", (2, 1, 700), 11).cuda())
    assert y.shape == (2, 64, 700) and maxerr(y.cpu(), g["out"]) < 5e-6


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("tag,scale", [("default", None), ("scaled", 2000.0)])
def test_perception_agent_vs_golden(pkg, prec, tag, scale):
    set_prec(pkg.ops, prec)
    g = gold("g2_pa_%s" % tag)
    pa = load(pkg.PerceptionAgent(sample_rate=16000), "PerceptionAgent", 21, sinc_scale=scale)
    noisy, _ = syn.synth_wave(2, 1600, 22)
    zr, zi, sg = pa(torch.from_numpy(noisy).cuda())
    assert zr.shape == (2, 256, 100) and sg.shape == (2, 1, 100)
    # relative RMSE after 11 conv layers of 16-bit operands.  Measured (round 2): fp16 operands - which is what the default
    # "mixed" policy gives the PerceptionAgent - 1.2e-3 .. 1.5e-3 (sigma 3e-4); bf16 operands 0.95e-2 .. 1.2e-2.  Bounds = 2 x
    # / 1.7 x that: a defect in one of the 11 layers moves the figure by far more.
    tol = 2e-2 if prec == "bf16" else 3e-3
    for n, a, b in (("z_real", zr, g["z_real"]), ("z_imag", zi, g["z_imag"]), ("sigma", sg, g["sigma"])):
        _, rl = show("PA %s %s %s" % (tag, prec, n), a, b)
        assert rl < tol


@pytest.mark.parametrize("prec", PRECISIONS)
def test_complex_conformer_small_vs_golden(pkg, prec):
    """reference test config (tests/test_conformer.py:17-20): d64, 4 heads (hd 16), ff128, k7"""
    set_prec(pkg.ops, prec)
    g = gold("g4_cconf_small")
    cc = load(pkg.ComplexConformer(n_freq=32, d_model=64, num_blocks=2, num_heads=4, d_ff=128, kernel_size=7, dropout=0.0),
              "ComplexConformerSmall", 41)
    sr, si = arr("g4_sr", (2, 20, 32), 42).cuda(), arr("g4_si", (2, 20, 32), 42).cuda()
    mr, mi = cc(sr, si)
    assert mr.shape == (2, 20, 32) and mi.shape == (2, 20, 32)
    tol = 8e-3 if prec == "bf16" else 1.5e-3         # measured 3.8e-3 (bf16) / 4.8e-4 (fp16 and the default policy)
    assert show("cconf small %s mask_real" % prec, mr, g["mask_real"])[1] < tol
    assert show("cconf small %s mask_imag" % prec, mi, g["mask_imag"])[1] < tol
    er, ei = cc.apply_mask(sr, si, mr, mi)
    assert er.shape == sr.shape
    assert show("cconf small apply", er, g["enh_real"])[1] < tol
    assert cc.count_parameters() == 135424


@pytest.mark.parametrize("prec", PRECISIONS)
def test_conformer_block_full_vs_golden(pkg, prec):
    set_prec(pkg.ops, prec)
    g = gold("g4_block_full")
    blk = load(pkg.ConformerBlock(256, 4, 1024, 31, 0.1), "ConformerBlock", 43)
    x = arr("g4_xb", (2, 37, 256), 44).cuda()
    tol = 1.5e-2 if prec != "fp16" else 2e-3
    y1 = blk.ff1(x)
    assert show("block ff1 %s" % prec, y1, g["ff1"])[1] < tol
    y2 = blk.mhsa(torch.from_numpy(g["ff1"]).cuda())
    assert show("block mhsa %s" % prec, y2, g["mhsa"])[1] < tol
    y3 = blk.conv(torch.from_numpy(g["mhsa"]).cuda())
    assert show("block conv %s" % prec, y3, g["conv"])[1] < tol
    assert show("block out %s" % prec, blk(x), g["out"])[1] < tol


@pytest.mark.parametrize("prec", PRECISIONS)
def test_cpea_vs_golden(pkg, prec):
    set_prec(pkg.ops, prec)
    g = gold("g6_cpea")
    m = load(pkg.CPEA(), "CorrelationPhaseEstimationAgent", 61)
    out = m(arr("g6_z", (2, 256, 21), 62).cuda())
    tol = 2e-2 if prec != "fp16" else 3e-3
    for k in ("rho_s", "rho_n", "phi1", "phi2"):
        assert out[k].shape == (2, 21, 64)
        assert show("cpea %s %s" % (k, prec), out[k], g[k])[1] < tol


def test_memory_vs_golden(pkg):
    g = gold("g7_memory")
    m = load(pkg.EpisodicMemory(), "EpisodicMemory", 71)
    out = m(arr("g7_e", (3, 256), 72).cuda())
    assert maxerr(out["bias"].cpu(), g["bias"]) < 2e-5 and maxerr(out["gate"].cpu(), g["gate"]) < 2e-5
    assert np.array_equal(out["top_indices"].cpu().numpy(), g["top_indices"])
    assert maxerr(out["similarity"].cpu(), g["similarity"]) < 2e-5


@pytest.mark.parametrize("prec", PRECISIONS)
def test_msa_full_vs_golden(pkg, prec):
    """full default architecture, de-saturated heads: the north-star mask RMSE bound"""
    set_prec(pkg.ops, prec)
    g = gold("g5_msa")
    msa = load(pkg.MaskSynthesisAgent(), "MaskSynthesisAgent", 51)
    zr, zi = arr("g5_zr", (2, 256, 21), 52).cuda(), arr("g5_zi", (2, 256, 21), 52).cuda()
    nr, ni = arr("g5_nr", (2, 21, 129), 52, 0.5).cuda(), arr("g5_ni", (2, 21, 129), 52, 0.5).cuda()
    cpea = orc.cpea_forward(synth_sd("CorrelationPhaseEstimationAgent", 61), zr.cpu())
    cpea = {k: v.cuda() for k, v in cpea.items()}
    mr, mi = msa(zr, zi, cpea, nr, ni)
    assert mr.shape == (2, 21, 129)
    got = torch.cat([mr, mi], dim=-1)
    ref = np.concatenate([g["mask_real"], g["mask_imag"]], axis=-1)
    r, _ = show("MSA mask %s" % prec, got, ref)
    assert r <= HARD_BOUND[prec], "mask RMSE %.3e exceeds the bound %.1e" % (r, HARD_BOUND[prec])


@pytest.mark.parametrize("prec", PRECISIONS)
def test_msa_reference_init_regime(pkg, prec):
    """Same weights but the reference's own head initialisation (xavier gain 0.1, magnitude bias +5,
    agents/msa.py:78-104): masks sit near 0.993 and the 1e-3 bound must hold for both operand types."""
    set_prec(pkg.ops, prec)
    sd = synth_sd("MaskSynthesisAgent", 51)
    for k in list(sd):
        if k.startswith("mask_proj_") and k.endswith("weight"):
            sd[k] = sd[k] * 0.1
        if k.startswith("mask_proj_") and k.endswith("bias"):
            sd[k] = torch.zeros_like(sd[k])
    sd["mask_proj_real.2.bias"] = torch.full_like(sd["mask_proj_real.2.bias"], 5.0)
    msa = pkg.MaskSynthesisAgent()
    msa.load_state_dict(sd)
    msa = msa.cuda().eval()
    zr, zi = arr("g5_zr", (2, 256, 21), 52), arr("g5_zi", (2, 256, 21), 52)
    nr, ni = arr("g5_nr", (2, 21, 129), 52, 0.5), arr("g5_ni", (2, 21, 129), 52, 0.5)
    cpea = orc.cpea_forward(synth_sd("CorrelationPhaseEstimationAgent", 61), zr)
    er, ei = orc.msa_forward(sd, zr, zi, cpea, nr, ni)
    mr, mi = msa(zr.cuda(), zi.cuda(), {k: v.cuda() for k, v in cpea.items()}, nr.cuda(), ni.cuda())
    r, _ = show("MSA mask (reference init) %s" % prec, torch.cat([mr, mi], -1), torch.cat([er, ei], -1).numpy())
    assert float(er.mean()) > 0.98 and r <= MASK_RMSE_BOUND


@pytest.mark.parametrize("prec", PRECISIONS)
def test_speech_enhancer_vs_golden(pkg, prec):
    set_prec(pkg.ops, prec)
    g = gold("g8_enhancer")
    se = load(pkg.cp.SpeechEnhancer(n_freq=129), "SpeechEnhancer", 81)
    noisy, _ = syn.synth_wave(2, 2000, 82)
    w = torch.from_numpy(noisy).cuda()
    nr, ni = pkg.cp.batch_stft(w, 256, 80, 160)
    er, ei, mm = se(nr, ni)
    r, _ = show("SpeechEnhancer mask_mag %s" % prec, mm, g["mask_mag"])
    assert r <= HARD_BOUND[prec]
    show("SpeechEnhancer enh_real %s" % prec, er, g["enh_real"])
    y = pkg.cp.batch_istft(er, ei, 256, 80, 160, 2000)
    rw, rl = show("SpeechEnhancer wave %s" % prec, y, g["enh_wav"])
    assert rl < (2e-2 if prec != "fp16" else 3e-3)


@pytest.mark.parametrize("prec", PRECISIONS)
def test_end_to_end_path_vs_golden(pkg, prec):
    set_prec(pkg.ops, prec)
    g = gold("g9_path")
    path = pkg.cp.EnhancementPath(sample_rate=16000, use_memory=True)
    path.perception.load_state_dict(synth_sd("PerceptionAgent", 91, sinc_scale=2000.0))
    path.cpea.load_state_dict(synth_sd("CorrelationPhaseEstimationAgent", 92))
    path.msa.load_state_dict(synth_sd("MaskSynthesisAgent", 93))
    path.memory.load_state_dict(synth_sd("EpisodicMemory", 94))
    path = path.cuda().eval()
    noisy, _ = syn.synth_wave(2, 1600, 95)
    # the golden masks were produced WITHOUT the memory bias: run both ways
    out_m = path(torch.from_numpy(noisy).cuda())
    show("path mem_bias %s" % prec, out_m["mem_bias"], g["mem_bias"])
    assert rel(out_m["mem_bias"].cpu(), g["mem_bias"]) < (5e-2 if prec != "fp16" else 6e-3)
    path.memory = None
    path.__dict__.pop("_sfm_pack", None)
    out = path(torch.from_numpy(noisy).cuda())
    got = torch.cat([out["mask_real"], out["mask_imag"]], dim=-1)
    ref = np.concatenate([g["mask_real"], g["mask_imag"]], axis=-1)
    r, _ = show("PATH mask %s" % prec, got, ref)
    rw, rlw = show("PATH enhanced wave %s" % prec, out["enhanced"], g["enhanced"])
    assert r <= HARD_BOUND[prec], "mask RMSE %.3e" % r
    assert rlw < (2e-2 if prec != "fp16" else 3e-3)


@pytest.mark.parametrize("prec", PRECISIONS)
def test_enhancer_loss_forward_vs_golden(pkg, prec):
    """A21 forward: SI-SNR + 0.5 L1-mag + MR-STFT of ConformerPipeline._compute_loss on the golden enhancer outputs"""
    set_prec(pkg.ops, prec)
    g = gold("g8_enhancer")
    noisy, clean = syn.synth_wave(2, 2000, 82)
    cw = torch.from_numpy(clean).cuda()
    cr, ci = pkg.cp.batch_stft(cw, 256, 80, 160)
    # reference model outputs in, so that only the loss arithmetic is under test
    losses, enh = pkg.Fn.enhancer_loss(torch.from_numpy(g["enh_real"]).cuda(), torch.from_numpy(g["enh_imag"]).cuda(), cw, cr, ci)
    lv = losses.cpu().numpy()
    print("loss total %.6f (ref %.6f)  neg_sisnr %.6f (ref %.6f)" % (lv[0], float(g["loss"]), lv[1], float(g["neg_sisnr"])))
    assert maxerr(enh.cpu(), g["enh_wav"]) < 2e-5
    assert abs(lv[0] - float(g["loss"])) < 5e-4 and abs(lv[1] - float(g["neg_sisnr"])) < 5e-4
    # and end to end through the HIP SpeechEnhancer
    se = load(pkg.cp.SpeechEnhancer(n_freq=129), "SpeechEnhancer", 81)
    nr, ni = pkg.cp.batch_stft(torch.from_numpy(noisy).cuda(), 256, 80, 160)
    er, ei, _ = se(nr, ni)
    l2, _ = pkg.Fn.enhancer_loss(er, ei, cw, cr, ci)
    tol = 0.05 if prec != "fp16" else 0.01
    assert abs(float(l2[0]) - float(g["loss"])) < tol


def test_mrstft_sizes_vs_golden(pkg):
    g = gold("g3_mrstft")
    noisy, clean = syn.synth_wave(2, 2048, 32)
    w = torch.from_numpy(noisy).cuda()
    mr = pkg.cp.MultiResolutionSTFTLoss()
    for nf, hp in ((256, 64), (512, 128), (1024, 256)):
        mag = mr._stft_mag(w, nf, hp, nf)                                     # the reference's method name and [B, F, T] layout
        assert mag.shape == g["mag%d" % nf].shape and maxerr(mag.cpu(), g["mag%d" % nf]) < 2e-4


def test_loss_mirrors_by_their_reference_names_vs_golden(pkg):
    """si_snr_loss / MultiResolutionSTFTLoss (training/conformer_pipeline.py:52, :74) called exactly as the reference's
    training loop calls them: values against the reference's own outputs (g3), gradients against torch autograd of the
    oracle's restatement on the same inputs"""
    g = gold("g3_mrstft")
    noisy, clean = syn.synth_wave(2, 2048, 32)
    est = torch.from_numpy(noisy).cuda().requires_grad_(True)
    tgt = torch.from_numpy(clean).cuda()
    l_si = pkg.cp.si_snr_loss(est, tgt)
    mr = pkg.cp.MultiResolutionSTFTLoss()
    l_mr = mr(est, tgt)
    print("si_snr_loss %.6f (ref %.6f)   MultiResolutionSTFTLoss %.6f (ref %.6f)" %
          (float(l_si.detach()), float(g["sisnr"]), float(l_mr.detach()), float(g["loss"])))
    assert abs(float(l_si) - float(g["sisnr"])) < 2e-4 and abs(float(l_mr) - float(g["loss"])) < 2e-4
    g_si, = torch.autograd.grad(l_si, est, retain_graph=True)
    g_mr, = torch.autograd.grad(l_mr, est)
    e2 = torch.from_numpy(noisy).requires_grad_(True)
    r_si, = torch.autograd.grad(orc.si_snr_loss(e2, torch.from_numpy(clean)), e2)
    r_mr, = torch.autograd.grad(orc.mr_stft_loss(e2, torch.from_numpy(clean)), e2)
    assert rel(g_si.cpu(), r_si.numpy()) < 1e-4, rel(g_si.cpu(), r_si.numpy())
    # (sign(log|P| - log|T|) terms: bins whose two magnitudes agree to ~1e-6 flip with the 4e-6 error of the split-bf16 STFT;
    #  observed 6.2e-3, bound = 2 x that: profiles/README.md, "tolerances and what is achieved")
    print("MR-STFT gradient vs oracle autograd: rel rmse %.3e" % rel(g_mr.cpu(), r_mr.numpy()))
    assert rel(g_mr.cpu(), r_mr.numpy()) < 1.3e-2, rel(g_mr.cpu(), r_mr.numpy())
    # a non-default resolution list runs the exact-fp32 STFT path
    mr2 = pkg.cp.MultiResolutionSTFTLoss([128, 320], [32, 80], [128, 320])
    l2 = mr2(est, tgt)
    r2 = orc.mr_stft_loss(e2, torch.from_numpy(clean), (128, 320), (32, 80), (128, 320))
    assert abs(float(l2) - float(r2)) < 2e-4
    g2, = torch.autograd.grad(l2, est)
    rr2, = torch.autograd.grad(r2, e2)
    assert rel(g2.cpu(), rr2.numpy()) < 5e-3, rel(g2.cpu(), rr2.numpy())


@pytest.mark.parametrize("custom", [False, True])
def test_pipeline_compute_loss_by_its_reference_signature(pkg, custom, tmp_path, monkeypatch):
    """ConformerPipeline._compute_loss(noisy_real, noisy_imag, clean_wav, clean_real, clean_imag, mr_stft_fn) (:539), with
    the reference's default MultiResolutionSTFTLoss (fused objective node) and with a custom resolution list (composed from
    si_snr_loss / L1 magnitude / mr_stft_fn); save_model -> load_model(path=None) round trip of the checkpoint format"""
    pkg.ops.set_compute_dtype(torch.float16)
    g = gold("g8_enhancer")
    from sincformer_metacog_speech_enhancement_amd import config
    monkeypatch.setattr(config, "MODEL_DIR", str(tmp_path))
    pipe = pkg.cp.ConformerPipeline()
    pipe.model = load(pkg.cp.SpeechEnhancer(n_freq=129), "SpeechEnhancer", 81)
    noisy, clean = syn.synth_wave(2, 2000, 82)
    nw, cw = torch.from_numpy(noisy).cuda(), torch.from_numpy(clean).cuda()
    nr, ni = pkg.cp.batch_stft(nw, 256, 80, 160)
    cr, ci = pkg.cp.batch_stft(cw, 256, 80, 160)
    sizes = ([128, 320], [32, 80], [128, 320]) if custom else (None, None, None)
    mr = pkg.cp.MultiResolutionSTFTLoss(*sizes)
    with torch.no_grad():
        total, neg = pipe._compute_loss(nr, ni, cw, cr, ci, mr)
    if custom:
        sd = synth_sd("SpeechEnhancer", 81)
        onr, oni = orc.stft(torch.from_numpy(noisy))
        ocr, oci = orc.stft(torch.from_numpy(clean))
        er, ei, _ = orc.speech_enhancer_forward(sd, onr, oni, 4)
        wav = orc.istft(er, ei, 2000)
        mag = (torch.sqrt(er ** 2 + ei ** 2 + 1e-8) - torch.sqrt(ocr ** 2 + oci ** 2 + 1e-8)).abs().mean()
        want_neg = orc.si_snr_loss(wav, torch.from_numpy(clean))
        want = want_neg + 0.5 * mag + orc.mr_stft_loss(wav, torch.from_numpy(clean), (128, 320), (32, 80), (128, 320))
        want, want_neg = float(want), float(want_neg)
    else:
        want, want_neg = float(g["loss"]), float(g["neg_sisnr"])
    print("_compute_loss custom=%s: %.5f (ref %.5f), neg_sisnr %.5f (ref %.5f)" % (custom, float(total), want, float(neg), want_neg))
    assert abs(float(total) - want) < 1e-2 and abs(float(neg) - want_neg) < 1e-2
    # checkpoint round trip through the reference's file names and dict layout
    pipe.save_model()
    pipe._save_best()
    ck = torch.load(str(tmp_path / "conformer_final.pt"), map_location="cpu", weights_only=True)
    assert ck["model_class"] == "SpeechEnhancer" and set(ck["model_state"]) == set(pipe.model.state_dict())
    assert (tmp_path / "best_conformer.pt").exists()
    pipe2 = pkg.cp.ConformerPipeline()
    pipe2.load_model()                                                       # path=None -> MODEL_DIR/conformer_final.pt
    y1 = pipe.enhance_signal(noisy[0])
    y2 = pipe2.enhance_signal(noisy[0])
    assert np.array_equal(y1, y2)


# ---------------------------------------------------------------------------
# edge shapes of the whole path against the (golden-pinned) oracle: single utterance, odd batch, sample counts that are
# multiples of neither the hop (80) nor the encoder stride (16), the shortest signal the reflect padding admits
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("B,L", [(1, 1237), (3, 4001), (1, 400), (5, 2000), (1, 96080)])   # last: T 1202 -> 64-row attention kernel
def test_end_to_end_path_ragged_shapes_vs_oracle(pkg, B, L):
    prec = "mixed"                       # the default operand formats of inference (what bench.py runs)
    set_prec(pkg.ops, prec)
    sds = {"pa": synth_sd("PerceptionAgent", 191, sinc_scale=2000.0),
           "cpea": synth_sd("CorrelationPhaseEstimationAgent", 192),
           "msa": synth_sd("MaskSynthesisAgent", 193),
           "memory": synth_sd("EpisodicMemory", 194)}
    path = pkg.cp.EnhancementPath(sample_rate=16000, use_memory=True)
    path.perception.load_state_dict(sds["pa"])
    path.cpea.load_state_dict(sds["cpea"])
    path.msa.load_state_dict(sds["msa"])
    path.memory.load_state_dict(sds["memory"])
    path = path.cuda().eval()
    noisy, _ = syn.synth_wave(B, L, 195 + L)
    ref = orc.enhance_path(sds, noisy, 16000, use_memory=True)
    out = path(torch.from_numpy(noisy).cuda())
    T = 1 + L // 80
    assert tuple(out["mask_real"].shape) == (B, T, 129) and tuple(out["enhanced"].shape) == (B, L)
    got = torch.cat([out["mask_real"], out["mask_imag"]], dim=-1)
    want = torch.cat([ref["mask_real"], ref["mask_imag"]], dim=-1)
    r, _ = show("ragged path B%d L%d mask" % (B, L), got, want)
    _, rlw = show("ragged path B%d L%d wave" % (B, L), out["enhanced"], ref["enhanced"])
    assert r <= HARD_BOUND[prec], "mask RMSE %.3e" % r
    assert rlw < 3e-3


def test_path_rejects_signals_shorter_than_the_reflect_padding(pkg):
    """torch.stft(center=True) raises for L <= n_fft/2 (reflect padding); so does the HIP STFT"""
    path = pkg.cp.EnhancementPath(sample_rate=16000).cuda().eval()
    with pytest.raises(RuntimeError):
        path(torch.zeros(1, 100, device="cuda"))


def test_hipgraph_replay_equals_eager(pkg):
    """graph.GraphedForward: the whole path captured into one hipGraph gives the eager result bit for bit, also after the
    input changes (static input buffer) and for a second shape (second graph)."""
    from sincformer_metacog_speech_enhancement_amd.graph import GraphedForward
    pkg.ops.reset_precision()                 # the default operand formats (what bench.py runs)
    path = pkg.cp.EnhancementPath(sample_rate=16000, use_memory=True)
    path.perception.load_state_dict(synth_sd("PerceptionAgent", 191, sinc_scale=2000.0))
    path.cpea.load_state_dict(synth_sd("CorrelationPhaseEstimationAgent", 192))
    path.msa.load_state_dict(synth_sd("MaskSynthesisAgent", 193))
    path.memory.load_state_dict(synth_sd("EpisodicMemory", 194))
    path = path.cuda().eval()
    graphed = GraphedForward(lambda w: path(w))
    for seed, (B, L) in enumerate([(1, 16000), (1, 16000), (2, 4000)]):
        noisy, _ = syn.synth_wave(B, L, 400 + seed)
        w = torch.from_numpy(noisy).cuda()
        with torch.no_grad():
            eager = {k: v.clone() for k, v in path(w).items() if isinstance(v, torch.Tensor)}
        out = graphed(w)
        torch.cuda.synchronize()
        for k in ("mask_real", "mask_imag", "enhanced"):
            assert torch.equal(out[k], eager[k]), k
    assert len(graphed._cache) == 2


def test_empty_batch_is_a_loud_error():
    """no silent empty results: a zero-length batch or a zero-length waveform is refused by the entry points' shape checks"""
    from sincformer_metacog_speech_enhancement_amd.models.conformer import ConformerBlock
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import EnhancementPath
    blk = ConformerBlock(256, 4, 1024, 31, 0.1).cuda().eval()
    with torch.no_grad():
        with pytest.raises((RuntimeError, ValueError)):
            blk(torch.zeros(0, 16, 256, device="cuda"))
        path = EnhancementPath(sample_rate=16000).cuda().eval()
        with pytest.raises((RuntimeError, ValueError)):
            path(torch.zeros(0, 1600, device="cuda"))
        with pytest.raises((RuntimeError, ValueError)):
            path(torch.zeros(2, 0, device="cuda"))


def test_two_passes_in_flight_equal_sequential_passes():
    """bench.py keeps two forward passes in flight on two HIP streams: the results must be those of strictly sequential
    passes, bit for bit (no shared scratch between passes)"""
    from sincformer_metacog_speech_enhancement_amd import ops
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import EnhancementPath
    ops.reset_precision()                     # the default operand formats (what bench.py runs)
    path = EnhancementPath(sample_rate=16000, use_memory=True)
    sds = {"perception": synth_sd("PerceptionAgent", 991, sinc_scale=2000.0), "cpea": synth_sd("CorrelationPhaseEstimationAgent", 992),
           "msa": synth_sd("MaskSynthesisAgent", 993), "memory": synth_sd("EpisodicMemory", 994)}
    for n, sd in sds.items():
        getattr(path, n).load_state_dict(sd)
    path = path.cuda().eval()
    waves = [torch.from_numpy(syn.synth_wave(3, 6400 + 160 * i, 995 + i)[0]).cuda() for i in range(6)]
    with torch.no_grad():
        ref = [path(w) for w in waves]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        got = [None] * len(waves)
        for rep in range(3):
            for i, w in enumerate(waves):
                with torch.cuda.stream(streams[i % 2]):
                    got[i] = path(w)
        torch.cuda.synchronize()
    for r, g in zip(ref, got):
        for k in ("mask_real", "mask_imag", "enhanced"):
            assert torch.equal(r[k], g[k]), k
