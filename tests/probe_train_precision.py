#!/usr/bin/env python3
"""Training error budget per operand recipe (not a test: run on the GPU box,
`python tests/probe_train_precision.py > gpurun_out/r04/train_precision_probe.json`).

The SpeechEnhancer step (forward, SI-SNR + L1 + multi-resolution STFT objective, backward; B 2 x 0.25 s, 4 blocks, dropout 0)
and the whole north-star composition (EnhancementPath, B 2 x 0.2 s) against torch autograd of the fp32 oracle: loss error and the
relative RMSE of every parameter gradient, for
  bf16            uniform bf16 operands, no loss scale              (round 3's training format)
  fp16            uniform fp16 operands, no loss scale
  fp16+S=2^k      fp16 operands, the loss multiplied by a static S  (what the dynamic scale settles at, swept)
  amp16           fp16 operands + optim.DynamicLossScale through FlatAdamW (lr 0): THE training format (bench.py, smoke())
Also a 1-ulp sensitivity draw: the bf16 step twice with the Swish epilogue perturbed is NOT repeated here (tools/swish_sensitivity.py).
Lives under tests/ because it uses the oracle (test infrastructure)."""
import json
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from helpers import synth_sd, arr                                  # noqa: E402
from oracle import sfm_oracle as orc                               # noqa: E402
from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn      # noqa: E402
from sincformer_metacog_speech_enhancement_amd.optim import FlatAdamW, DynamicLossScale   # noqa: E402
from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import (SpeechEnhancer, batch_stft, compute_loss,      # noqa: E402
                                                                                    EnhancementPath, compute_path_loss)


def rel(g, r):
    g, r = g.double().cpu(), r.double()
    return float(((g - r) ** 2).mean().sqrt() / (r ** 2).mean().sqrt().clamp_min(1e-300))


def summarise(pairs):
    """pairs: {name: (grad, ref)} -> worst / median / BatchNorm-projected worst relative RMSE"""
    rows = {k: rel(g, r) for k, (g, r) in pairs.items() if not k.endswith("depthwise.bias")}
    bn = {k: v for k, v in rows.items() if "conv.layer_norm" in k or "conv.pointwise1" in k}
    rest = {k: v for k, v in rows.items() if k not in bn}
    srt = sorted(rest.values())
    tot_g = math.sqrt(sum(float(g.double().pow(2).sum()) for g, _ in pairs.values()))
    tot_r = math.sqrt(sum(float(r.double().pow(2).sum()) for _, r in pairs.values()))
    dot = sum(float((g.double().cpu() * r.double()).sum()) for g, r in pairs.values())
    return {"worst": max(rest.items(), key=lambda kv: kv[1]), "median": srt[len(srt) // 2],
            "worst_behind_batchnorm": max(bn.items(), key=lambda kv: kv[1]) if bn else None,
            "cosine_of_whole_gradient": dot / (tot_g * tot_r), "norm_ratio": tot_g / tot_r}


def waves(kind, B, L, seed):
    """'test': the pair of tests/test_train_gpu.py (clean N(0, 0.1^2) + N(0, 0.05^2): 6 dB SNR); 'bench': synthetic.synth_wave, the
    pairs bench.py trains on (SNR cycled over -5 / 0 / 5 / 10 dB: many enhanced STFT bins near zero or near the target's magnitude,
    where the objective's 1 / |bin| slopes and sign() terms make its gradient discontinuous)"""
    if kind == "test":
        clean = arr("cw", (B, L), seed, 0.1)
        return clean + arr("nw", (B, L), seed + 1, 0.05), clean
    noisy, clean = syn.synth_wave(B, L, seed)
    return torch.from_numpy(noisy), torch.from_numpy(clean)


def enhancer_case(recipe, scale=1.0, kind="test"):
    B, L = 2, 4000
    sd = synth_sd("SpeechEnhancer", 23)
    if recipe == "bf16":
        ops.set_compute_dtype(torch.bfloat16)
    else:
        ops.set_compute_dtype(torch.float16)
    m = SpeechEnhancer(n_freq=129, d_model=256, num_blocks=4, num_heads=4, d_ff=1024, kernel_size=31, dropout=0.0)
    m.load_state_dict(sd, strict=True)
    m.cuda().train()
    noisy, clean = waves(kind, B, L, 80)
    ref_sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    ref_total, _, _ = orc.enhancer_loss(ref_sd, noisy, clean, 4, bn_train=True)
    ref_total.backward()
    nr, ni = batch_stft(noisy.cuda(), 256, 80, 160)
    cr, ci = batch_stft(clean.cuda(), 256, 80, 160)
    names = [k for k, _ in m.named_parameters()]
    out = {}
    if recipe == "amp16":
        opt = FlatAdamW(m.parameters(), lr=0.0, betas=(0.9, 0.98), weight_decay=0.01, max_norm=5.0)
        scaler = DynamicLossScale("cuda")
        bn0 = {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}
        for attempt in range(24):
            m.load_state_dict(bn0, strict=False)
            opt.zero_grad()
            total, _ = compute_loss(m, nr, ni, clean.cuda(), cr, ci)
            S = scaler.get_scale()
            scaler.scale(total).backward()
            scaler.step(opt, loss=total)
            if not opt.stats()["skipped"]:
                break
        grads = {k: (p_.grad.detach() / S, ref_sd[k].grad) for k, p_ in zip(names, opt.params)}
        out.update({"loss_scale": S, "skipped_steps": scaler.stats()["skipped_inf"]})
    else:
        total, _ = compute_loss(m, nr, ni, clean.cuda(), cr, ci)
        (total * scale).backward()
        grads = {k: (p_.grad.detach() / scale, ref_sd[k].grad) for k, p_ in m.named_parameters()}
        out["static_scale"] = scale
    finite = all(bool(torch.isfinite(g).all()) for g, _ in grads.values())
    out.update({"loss_rel_err": abs(float(total) - float(ref_total)) / abs(float(ref_total)), "finite": finite})
    if finite:
        out.update(summarise(grads))
    return out


def path_case(recipe, kind="test"):
    B, L = 2, 3200
    if recipe == "bf16":
        ops.set_compute_dtype(torch.bfloat16)
    else:
        ops.set_compute_dtype(torch.float16)
    sds = {"pa": synth_sd("PerceptionAgent", 291, sinc_scale=2000.0), "cpea": synth_sd("CorrelationPhaseEstimationAgent", 292),
           "msa": synth_sd("MaskSynthesisAgent", 293)}
    path = EnhancementPath(sample_rate=16000)
    path.perception.load_state_dict(sds["pa"])
    path.cpea.load_state_dict(sds["cpea"])
    path.msa.load_state_dict(sds["msa"])
    for mod in path.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    path.cpea.lstm.dropout = 0.0
    path = path.cuda().train()
    noisy, clean = waves(kind, B, L, 81)
    ref = {n: {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k and
                   k.split(".")[-1] not in ("window", "n_") else v.clone()) for k, v in sd.items()} for n, sd in sds.items()}
    ref_total, _, _ = orc.path_loss(ref, noisy, clean, 16000, bn_train=True)
    ref_total.backward()
    total, _ = compute_path_loss(path, noisy.cuda(), clean.cuda())
    scale = 1.0 if recipe != "amp16" else 65536.0
    (total * scale).backward()
    pref = {"perception": "pa", "cpea": "cpea", "msa": "msa"}
    grads = {}
    for k, p_ in path.named_parameters():
        top, rest = k.split(".", 1)
        rg = ref[pref[top]][rest].grad
        if p_.grad is None or rg is None or float(rg.abs().max()) == 0.0 or rest.startswith("sinc_conv."):
            continue
        grads[k] = (p_.grad.detach() / scale, rg)
    finite = all(bool(torch.isfinite(g).all()) for g, _ in grads.values())
    out = {"loss_rel_err": abs(float(total) - float(ref_total)) / abs(float(ref_total)), "finite": finite, "static_scale": scale}
    if finite:
        out.update(summarise(grads))
    return out


def main():
    res = {}
    for kind in ("test", "bench"):
        a, b = res.setdefault("speech_enhancer_step / %s waves" % kind, {}), res.setdefault("path_step / %s waves" % kind, {})
        for recipe, scale in (("bf16", 1.0), ("fp16", 1.0), ("fp16", 4096.0), ("fp16", 65536.0), ("amp16", None)):
            key = recipe if scale in (None, 1.0) else "%s x S=%g" % (recipe, scale)
            a[key] = enhancer_case(recipe, scale or 1.0, kind)
            print(kind, key, json.dumps(a[key]), file=sys.stderr, flush=True)
        for recipe in ("bf16", "fp16", "amp16"):
            key = recipe if recipe != "amp16" else "fp16 x S=65536"
            b[key] = path_case(recipe, kind)
            print(kind, "path", key, json.dumps(b[key]), file=sys.stderr, flush=True)
    ops.reset_precision()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
