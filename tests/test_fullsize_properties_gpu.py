"""Size-independent properties of the path at the BENCH workload's full size (BASELINE configs[1]: B 64 x 4 s, T 801),
where the CPU oracle is too slow to be the checker: batch independence and permutation equivariance (utterances do
not interact in eval mode), agreement with the oracle on ONE utterance of the big batch, the bounds of the polar mask
(agents/msa.py:166-172: |mask| <= 1, |phase| <= pi/8), STFT -> iSTFT identity, determinism."""
import math
import numpy as np
import pytest
import torch

from helpers import synth_sd, rmse
from oracle import sfm_oracle as orc
from sincformer_metacog_speech_enhancement_amd import synthetic as syn

pytestmark = pytest.mark.gpu
B, L = 64, 64000


@pytest.fixture(scope="module")
def path_and_out():
    from sincformer_metacog_speech_enhancement_amd import ops
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import EnhancementPath
    ops.reset_precision()                 # the bench's operand formats (default policy "mixed": ops.POLICIES)
    sds = {"pa": synth_sd("PerceptionAgent", 291, sinc_scale=2000.0), "cpea": synth_sd("CorrelationPhaseEstimationAgent", 292),
           "msa": synth_sd("MaskSynthesisAgent", 293)}
    path = EnhancementPath(sample_rate=16000)
    path.perception.load_state_dict(sds["pa"])
    path.cpea.load_state_dict(sds["cpea"])
    path.msa.load_state_dict(sds["msa"])
    path = path.cuda().eval()
    noisy, _ = syn.synth_wave(B, L, 295)
    wave = torch.from_numpy(noisy).cuda()
    with torch.no_grad():
        out = {k: v.clone() for k, v in path(wave).items() if isinstance(v, torch.Tensor)}
    return path, sds, noisy, wave, out


def test_shapes_bounds_and_determinism(path_and_out):
    path, sds, noisy, wave, out = path_and_out
    T = 1 + L // 80
    assert tuple(out["mask_real"].shape) == (B, T, 129) and tuple(out["enhanced"].shape) == (B, L)
    mag = torch.sqrt(out["mask_real"] ** 2 + out["mask_imag"] ** 2)
    ph = torch.atan2(out["mask_imag"], out["mask_real"])
    assert float(mag.max()) <= 1.0 + 1e-5 and float(mag.min()) >= 0.0
    assert float(ph.abs().max()) <= 3.14159 / 8 + 1e-4
    assert all(bool(torch.isfinite(v).all()) for v in out.values())
    with torch.no_grad():
        again = path(wave)
    assert torch.equal(again["mask_real"], out["mask_real"]) and torch.equal(again["enhanced"], out["enhanced"])


def test_batch_independence_and_permutation(path_and_out):
    """utterance i of the batch of 64 == the same utterance in a permuted batch of 64 (same kernels, different tile order:
    2e-5) and == the same utterance run in a batch of 3 / alone.  The attention kernel is chosen by shape AND by the amount of
    work (the persistent kernel needs >= 128 items to fill the chip): the small batches take the 32-rows-per-wave kernel,
    whose bf16 rounding of P differs (another reference maximum) - same math to 3e-4 of the 1e-3 budget; with the kernel
    pinned the small batches agree with the big one to 2e-5 as well."""
    from sincformer_metacog_speech_enhancement_amd import ops
    path, sds, noisy, wave, out = path_and_out
    with torch.no_grad():
        sub = path(wave[[5, 40, 63]].contiguous())
        one = path(wave[17:18].contiguous())
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
        shuf = path(wave[perm].contiguous())
        ops.set_attention_variant(1)
        try:
            full1 = path(wave)
            sub1 = path(wave[[5, 40, 63]].contiguous())
            one1 = path(wave[17:18].contiguous())
        finally:
            ops.set_attention_variant(0)
    for k in ("mask_real", "mask_imag", "enhanced"):
        assert rmse(shuf[k].cpu(), out[k][perm].cpu()) < 2e-5, k
        assert rmse(sub[k].cpu(), out[k][[5, 40, 63]].cpu()) < 3e-4, k
        assert rmse(one[k].cpu(), out[k][17:18].cpu()) < 3e-4, k
        assert rmse(sub1[k].cpu(), full1[k][[5, 40, 63]].cpu()) < 2e-5, k
        assert rmse(one1[k].cpu(), full1[k][17:18].cpu()) < 2e-5, k


def test_perception_agent_is_bitwise_repeatable_over_many_passes(path_and_out):
    """100 forward passes of the PerceptionAgent's fused path over the same B 64 x 4 s batch: every pass bit-identical to the
    first.  Pins the write-after-read race that round 3 found on the LDS-DMA weight rings (gemm16_epi.h `wait_ring`): before the
    fix 1-2 % of the passes had one 64 x 32 accumulator block of one conv tile wrong (tools/pa_determinism_probe.py names the
    stage), which is what the occasional failure of the permuted-batch comparison below was."""
    from sincformer_metacog_speech_enhancement_amd import functional as Fn
    path, sds, noisy, wave, out = path_and_out
    pk = path.perception._packed(lambda sd: Fn.pack_perception(sd, path.sample_rate))
    def fingerprint():
        (rz, sz, hz), sigma = Fn.perception_forward(wave, pk, latents=False)
        return [t.clone() for t in (rz, sz, hz, sigma)]
    with torch.no_grad():
        ref = fingerprint()
        bad = 0
        for _ in range(100):
            cur = fingerprint()
            bad += 0 if all(torch.equal(a, b) for a, b in zip(cur, ref)) else 1
    assert bad == 0, "%d of 100 passes differ from the first" % bad


def test_one_utterance_of_the_big_batch_vs_oracle(path_and_out):
    path, sds, noisy, wave, out = path_and_out
    ref = orc.enhance_path(sds, noisy[9:10], 16000)
    got = torch.cat([out["mask_real"][9:10], out["mask_imag"][9:10]], -1).cpu()
    want = torch.cat([ref["mask_real"], ref["mask_imag"]], -1)
    r = rmse(got, want)
    print("full-size batch, utterance 9 vs oracle: mask RMSE %.3e, wave RMSE %.3e" % (r, rmse(out["enhanced"][9:10].cpu(), ref["enhanced"])))
    assert r <= 1e-3


def test_stft_istft_identity_at_full_size(path_and_out):
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import batch_stft, batch_istft
    path, sds, noisy, wave, out = path_and_out
    re, im = batch_stft(wave, 256, 80, 160)
    back = batch_istft(re, im, 256, 80, 160, L)
    assert float((back - wave).abs().max()) < 5e-5
    # Parseval-type check of the STFT itself: energy of the windowed frames == energy of their spectra
    a = 2.0 * np.pi * np.arange(160) / 160
    w = torch.from_numpy((0.5 - 0.5 * np.cos(a)).astype(np.float32)).cuda()
    fr = wave[:, 8000:8160] * w                           # frame t = 101 covers samples [8000, 8160)
    spec_e = (re[:, 101] ** 2 + im[:, 101] ** 2)
    full = spec_e[:, 0] + spec_e[:, 128] + 2.0 * spec_e[:, 1:128].sum(-1)
    assert float(((full / 256.0) - (fr ** 2).sum(-1)).abs().max()) < 1e-3


from helpers import central_difference_along_gradient as _central_difference_along_gradient  # noqa: E402


def test_training_backward_directional_derivative_at_full_size():
    """BASELINE configs[2] size (B 256 x 4 s, SpeechEnhancer in train() mode), where autograd of the oracle is out of reach:
    for f(theta) = <cotangent, model(theta)> the HIP backward's gradient g must predict the change of f along its own
    direction, f(theta + e d) - f(theta - e d) = 2 e |g| with d = g / |g| (dropout off: f is a deterministic function).
    A fixed cotangent is used instead of the objective because the objective's own gradient (1 / |STFT bin| terms) turns the
    16-bit rounding noise of the forward into tens of per cent of gradient noise; the objective's
    backward is checked on identical inputs in tests/test_train_gpu.py."""
    from sincformer_metacog_speech_enhancement_amd import ops
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import SpeechEnhancer, batch_stft
    ops.reset_precision()                 # training runs in the base format (fp16), as bench.py --workload c3se does
    Bt, Lt = 256, 64000
    model = SpeechEnhancer(n_freq=129, d_model=256, num_blocks=4, num_heads=4, d_ff=1024, kernel_size=31, dropout=0.0)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, 4321).items()})
    model.cuda().train()
    noisy, _ = syn.synth_wave(Bt, Lt, 777)
    nr, ni = batch_stft(torch.from_numpy(noisy).cuda(), 256, 80, 160)
    g = torch.Generator(device="cuda").manual_seed(5)
    cot_r = torch.randn(nr.shape, device="cuda", generator=g) * 1e-3
    cot_i = torch.randn(nr.shape, device="cuda", generator=g) * 1e-3

    def objective():
        er, ei, _ = model(nr, ni)
        return (er * cot_r).sum() + (ei * cot_i).sum()

    val, gnorm, slope = _central_difference_along_gradient(list(model.parameters()), objective, eps=0.02)
    print("full-size SpeechEnhancer backward: f %.5f, |g| %.4f, central-difference slope along g %.4f" % (val, gnorm, slope))
    assert abs(slope - gnorm) < 0.05 * gnorm, (slope, gnorm)


@pytest.mark.parametrize("Bp", [64, 256])
def test_path_backward_directional_derivative_at_full_size(Bp):
    """the same check for the whole composition (PerceptionAgent + CPEA + MaskSynthesisAgent), cotangent on the enhanced
    spectrum: at B 64 x 4 s (BASELINE configs[1], bench workload c2t) and at B 256 x 4 s (configs[2] on the north-star
    composition, bench workload c3t), in the operand formats the bench runs"""
    from sincformer_metacog_speech_enhancement_amd import ops
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import EnhancementPath
    ops.reset_precision()                 # the bench's operand formats (default policy "mixed": ops.POLICIES)
    path = EnhancementPath(sample_rate=16000)
    path.perception.load_state_dict(synth_sd("PerceptionAgent", 291, sinc_scale=2000.0))
    path.cpea.load_state_dict(synth_sd("CorrelationPhaseEstimationAgent", 292))
    path.msa.load_state_dict(synth_sd("MaskSynthesisAgent", 293))
    for mod in path.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    path.cpea.lstm.dropout = 0.0
    path = path.cuda().train()
    noisy, _ = syn.synth_wave(Bp, L, 778)
    wave = torch.from_numpy(noisy).cuda()
    T = 1 + L // 80
    g = torch.Generator(device="cuda").manual_seed(6)
    cot_r = torch.randn(Bp, T, 129, device="cuda", generator=g) * 1e-3
    cot_i = torch.randn(Bp, T, 129, device="cuda", generator=g) * 1e-3

    def objective():
        out = path(wave, want=("mask", "spectrum"))
        return (out["enh_real"] * cot_r).sum() + (out["enh_imag"] * cot_i).sum()

    params = [p_ for n, p_ in path.named_parameters() if "uncertainty_head" not in n]
    val, gnorm, slope = _central_difference_along_gradient(params, objective, eps=0.01)
    print("full-size path backward (B %d): f %.5f, |g| %.4f, central-difference slope along g %.4f" % (Bp, val, gnorm, slope))
    assert abs(slope - gnorm) < 0.05 * gnorm, (slope, gnorm)


def test_long_utterances_with_memory_vs_oracle_and_batch_independence():
    """BASELINE configs[4] sequence length (30 s: L 480 000, T 6001, T_pa 30 000) with the episodic memory: the 64-rows-per-wave
    attention kernel, 94 key tiles per row, the two-pass memory key.  One utterance against the oracle (fp32 attention over the
    full 6001 x 6001 score matrix on the CPU), bounds, and independence of the other utterances of the batch."""
    from sincformer_metacog_speech_enhancement_amd import ops
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import EnhancementPath
    ops.reset_precision()                 # the bench's operand formats (default policy "mixed": ops.POLICIES)
    Bl, Ll = 3, 480000
    sds = {"pa": synth_sd("PerceptionAgent", 391, sinc_scale=2000.0), "cpea": synth_sd("CorrelationPhaseEstimationAgent", 392),
           "msa": synth_sd("MaskSynthesisAgent", 393), "memory": synth_sd("EpisodicMemory", 394)}
    path = EnhancementPath(sample_rate=16000, use_memory=True)
    path.perception.load_state_dict(sds["pa"])
    path.cpea.load_state_dict(sds["cpea"])
    path.msa.load_state_dict(sds["msa"])
    path.memory.load_state_dict(sds["memory"])
    path = path.cuda().eval()
    noisy, _ = syn.synth_wave(Bl, Ll, 396)
    wave = torch.from_numpy(noisy).cuda()
    # the attention kernel is chosen by shape (the persistent ring kernel needs enough (batch, head, query tile) items to
    # fill the chip): pin it, so that "one utterance alone" and "the same utterance in a batch" run the same arithmetic
    # and the independence check below can be held to fp32 rounding
    ops.set_attention_variant(3)
    try:
        with torch.no_grad():
            out = path(wave)
            one = path(wave[1:2].contiguous())
    finally:
        ops.set_attention_variant(0)
    with torch.no_grad():
        auto_one = path(wave[1:2].contiguous())               # default selection at batch 1: the 32-rows-per-wave kernel
    T = 1 + Ll // 80
    assert tuple(out["mask_real"].shape) == (Bl, T, 129) and tuple(out["enhanced"].shape) == (Bl, Ll)
    assert all(bool(torch.isfinite(v).all()) for v in out.values() if isinstance(v, torch.Tensor) and v.dtype.is_floating_point)
    mag = torch.sqrt(out["mask_real"] ** 2 + out["mask_imag"] ** 2)
    assert float(mag.max()) <= 1.0 + 1e-5
    for k in ("mask_real", "mask_imag", "enhanced"):
        assert rmse(one[k].cpu(), out[k][1:2].cpu()) < 2e-5, k
        assert rmse(auto_one[k].cpu(), out[k][1:2].cpu()) < 3e-4, k      # other kernel: same result up to 16-bit rounding of P
    ref = orc.enhance_path(sds, noisy[1:2], 16000, use_memory=True)
    got = torch.cat([out["mask_real"][1:2], out["mask_imag"][1:2]], -1).cpu()
    want = torch.cat([ref["mask_real"], ref["mask_imag"]], -1)
    r = rmse(got, want)
    print("30 s utterance vs oracle: mask RMSE %.3e, wave RMSE %.3e, memory slot %d (oracle %d)" %
          (r, rmse(out["enhanced"][1:2].cpu(), ref["enhanced"]), int(out["mem_top"][1]), int(ref["memory"]["top_indices"][0])))
    assert r <= 1e-3
    assert int(out["mem_top"][1]) == int(ref["memory"]["top_indices"][0])


def test_full_size_objective_training_steps_on_the_north_star_composition():
    """BASELINE configs[2] on the north-star composition WITH the real objective (compute_path_loss = the objective of
    training/conformer_pipeline.py:539-572) at B 256 x 4 s, dropout off so that a step is a deterministic function of the
    state: finite loss and gradients, bit-identical loss / gradient buffer / gradient norm from a second run started from the
    same state, and the objective going down over a few FlatAdamW steps under the dynamic loss scale.  (The oracle's autograd is out of reach at
    this size; the objective's backward is pinned on identical inputs in tests/test_train_gpu.py.)"""
    from sincformer_metacog_speech_enhancement_amd import ops
    from sincformer_metacog_speech_enhancement_amd.optim import FlatAdamW, DynamicLossScale
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import EnhancementPath, compute_path_loss
    ops.reset_precision()                 # the recipe bench.py --workload c3t trains with: fp16 operands + dynamic loss scale
    assert ops.compute_dtype() is torch.float16
    Bt = 256
    noisy, clean = syn.synth_wave(Bt, L, 779)
    noisy, clean = torch.from_numpy(noisy).cuda(), torch.from_numpy(clean).cuda()

    def build():
        path = EnhancementPath(sample_rate=16000)
        path.perception.load_state_dict(synth_sd("PerceptionAgent", 291, sinc_scale=2000.0))
        path.cpea.load_state_dict(synth_sd("CorrelationPhaseEstimationAgent", 292))
        path.msa.load_state_dict(synth_sd("MaskSynthesisAgent", 293))
        for mod in path.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        path.cpea.lstm.dropout = 0.0
        path = path.cuda().train()
        params = [p_ for n, p_ in path.named_parameters() if "uncertainty_head" not in n]
        return path, FlatAdamW(params, lr=5e-4, betas=(0.9, 0.98), weight_decay=0.01, max_norm=5.0), DynamicLossScale("cuda")

    def run(steps):
        """`steps` APPLIED steps (steps the loss scale is still too large for are skipped and halve it, as under GradScaler);
        returns the loss and the unscaled gradient norm of every applied step, the number of skipped ones, and a checksum of the
        flat gradient buffer of the first applied step"""
        path, opt, scaler = build()
        hist, norms, skipped, first_grad = [], [], 0, None
        while len(hist) < steps:
            assert skipped < 24, scaler.stats()
            opt.zero_grad()
            total, neg = compute_path_loss(path, noisy, clean)
            scaler.scale(total).backward()
            scaler.unscale_(opt)
            g_copy = opt.sync.flat.clone() if first_grad is None else None
            scaler.step(opt, loss=total)
            scaler.update()
            st = opt.stats()
            if st["skipped"]:
                skipped += 1
                continue
            if first_grad is None:
                first_grad = g_copy
            assert math.isfinite(st["grad_norm"]) and st["grad_norm"] > 0
            hist.append(float(total.detach()))
            norms.append(st["grad_norm"])
        assert all(math.isfinite(h) for h in hist)
        scale = scaler.get_scale()
        del path, opt
        torch.cuda.empty_cache()
        return hist, norms, skipped, scale, first_grad

    h1, n1, k1, s1, g1 = run(5)
    h2, n2, k2, s2, g2 = run(2)
    print("full-size objective (B 256 x 4 s, fp16 + dynamic loss scale): losses %s, gradient norms %s, %d steps skipped on the way "
          "to S = %g; second run %s / %s, %d skipped" %
          (["%.4f" % h for h in h1], ["%.3f" % n for n in n1], k1, s1, ["%.4f" % h for h in h2], ["%.3f" % n for n in n2], k2))
    # same state, same data, no dropout, and every reduction of the step in a fixed order (per-split partials folded by a second
    # pass instead of fp32 atomics): the two runs are the SAME computation - the scale search takes the same path, the first
    # applied step has bit-identical loss, gradient buffer and norm, and the second step (whose inputs are the first AdamW update)
    # stays within 1 %
    assert k1 == k2
    assert h1[0] == h2[0] and n1[0] == n2[0], (h1[0], h2[0], n1[0], n2[0])
    assert torch.equal(g1, g2)
    assert abs(h1[1] - h2[1]) <= 0.01 * abs(h1[1]), (h1[1], h2[1])
    assert abs(n1[1] - n2[1]) <= 0.01 * n1[1], (n1[1], n2[1])
    assert h1[-1] < h1[0], h1


def test_configs4_batch_of_32_thirty_second_utterances_with_memory():
    """BASELINE configs[4] AT ITS BATCH (B 32 x 30 s, T 6001, episodic memory on): shapes, mask bounds, finiteness, and
    independence of the batch - utterance 7 of the 32 equals the same utterance run inside a batch of 3 (the size
    test_long_utterances_with_memory_vs_oracle_and_batch_independence compares with the oracle) and in a permuted batch."""
    from sincformer_metacog_speech_enhancement_amd import ops
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import EnhancementPath
    ops.reset_precision()
    Bl, Ll = 32, 480000
    path = EnhancementPath(sample_rate=16000, use_memory=True)
    path.perception.load_state_dict(synth_sd("PerceptionAgent", 391, sinc_scale=2000.0))
    path.cpea.load_state_dict(synth_sd("CorrelationPhaseEstimationAgent", 392))
    path.msa.load_state_dict(synth_sd("MaskSynthesisAgent", 393))
    path.memory.load_state_dict(synth_sd("EpisodicMemory", 394))
    path = path.cuda().eval()
    noisy, _ = syn.synth_wave(Bl, Ll, 397)
    wave = torch.from_numpy(noisy).cuda()
    T = 1 + Ll // 80
    ops.set_attention_variant(4)          # one attention kernel for every batch size below (the default picks it at B 32)
    try:
        with torch.no_grad():
            out = path(wave)
            keep = {k: out[k][[7, 20, 31]].clone() for k in ("mask_real", "mask_imag", "enhanced")}
            top = out["mem_top"].clone()
            shape_mask, shape_enh = tuple(out["mask_real"].shape), tuple(out["enhanced"].shape)
            mag_max = float(torch.sqrt(out["mask_real"] ** 2 + out["mask_imag"] ** 2).max())
            ph_max = float(torch.atan2(out["mask_imag"], out["mask_real"]).abs().max())
            finite = all(bool(torch.isfinite(v).all()) for v in out.values() if isinstance(v, torch.Tensor) and v.dtype.is_floating_point)
            del out
            sub = path(wave[[7, 20, 31]].contiguous())
    finally:
        ops.set_attention_variant(0)
    assert shape_mask == (Bl, T, 129) and shape_enh == (Bl, Ll) and finite
    assert mag_max <= 1.0 + 1e-5 and ph_max <= 3.14159 / 8 + 1e-4
    for k in ("mask_real", "mask_imag", "enhanced"):
        assert rmse(sub[k].cpu(), keep[k].cpu()) < 2e-5, k
    assert torch.equal(sub["mem_top"].cpu(), top[[7, 20, 31]].cpu())
