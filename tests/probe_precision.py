#!/usr/bin/env python3
"""Per-stage error budget of the 16-bit operand formats (not a test: run on the GPU box,
`python tests/probe_precision.py > gpurun_out/precision_probe.json`).

For each golden case (outputs of the reference: MaskSynthesisAgent g5, end-to-end path g9, SpeechEnhancer g8) and one
larger case against the oracle, the mask RMSE under a list of precision policies (ops.set_precision_policy): everything
bf16, everything fp16, ONE stage fp16 with the rest bf16 (what that stage contributes to the bf16 error), ONE stage bf16
with the rest fp16, and the candidate recipes.  Lives under tests/ because it uses the oracle (test infrastructure)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from helpers import gold, synth_sd, arr, rmse                      # noqa: E402
from oracle import sfm_oracle as orc                               # noqa: E402
from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn      # noqa: E402
from sincformer_metacog_speech_enhancement_amd.agents import MaskSynthesisAgent   # noqa: E402
from sincformer_metacog_speech_enhancement_amd.training import conformer_pipeline as cp   # noqa: E402

BF, FP = torch.bfloat16, torch.float16


def policies():
    out = [("all_bf16", {s: BF for s in ops.STAGES}), ("all_fp16", {s: FP for s in ops.STAGES})]
    for s in ops.STAGES:
        p = {t: BF for t in ops.STAGES}
        p[s] = FP
        out.append(("only_%s_fp16" % s, p))
    for s in ops.STAGES:
        p = {t: FP for t in ops.STAGES}
        p[s] = BF
        out.append(("only_%s_bf16" % s, p))
    for i in range(6):
        p = {t: BF for t in ops.STAGES}
        p["block%d" % i] = FP
        p["attn%d" % i] = FP
        out.append(("only_block%d_fp16" % i, p))
    out.append(("mixed = the default policy (attn bf16, rest fp16)", dict(ops.POLICIES["mixed"])))
    out.append(("mixed2(pa bf16)", {"pa": BF, "front": FP, "block": FP, "attn": FP, "tail": FP}))
    out.append(("mixed3(attn bf16)", {"pa": FP, "front": FP, "block": FP, "attn": BF, "tail": FP}))
    out.append(("mixed4(pa,attn,front bf16)", {"pa": BF, "front": BF, "block": FP, "attn": BF, "tail": FP}))
    return out


def case_msa():
    g = gold("g5_msa")
    msa = MaskSynthesisAgent()
    msa.load_state_dict(synth_sd("MaskSynthesisAgent", 51), strict=True)
    msa = msa.cuda().eval()
    zr, zi = arr("g5_zr", (2, 256, 21), 52).cuda(), arr("g5_zi", (2, 256, 21), 52).cuda()
    nr, ni = arr("g5_nr", (2, 21, 129), 52, 0.5).cuda(), arr("g5_ni", (2, 21, 129), 52, 0.5).cuda()
    cpea = orc.cpea_forward(synth_sd("CorrelationPhaseEstimationAgent", 61), zr.cpu())
    cpea = {k: v.cuda() for k, v in cpea.items()}
    ref = np.concatenate([g["mask_real"], g["mask_imag"]], axis=-1)

    def run():
        mr, mi = msa(zr, zi, cpea, nr, ni)
        return rmse(torch.cat([mr, mi], -1).cpu(), ref)
    return run


def _path(seeds, use_memory=False):
    path = cp.EnhancementPath(sample_rate=16000, use_memory=use_memory)
    path.perception.load_state_dict(synth_sd("PerceptionAgent", seeds[0], sinc_scale=2000.0))
    path.cpea.load_state_dict(synth_sd("CorrelationPhaseEstimationAgent", seeds[1]))
    path.msa.load_state_dict(synth_sd("MaskSynthesisAgent", seeds[2]))
    return path.cuda().eval()


def case_path_golden():
    g = gold("g9_path")
    path = _path((91, 92, 93))
    noisy, _ = syn.synth_wave(2, 1600, 95)
    w = torch.from_numpy(noisy).cuda()
    ref = np.concatenate([g["mask_real"], g["mask_imag"]], axis=-1)

    def run():
        out = path(w)
        return rmse(torch.cat([out["mask_real"], out["mask_imag"]], -1).cpu(), ref)
    return run


def case_path_oracle(B=2, L=16000):
    sds = {"pa": synth_sd("PerceptionAgent", 191, sinc_scale=2000.0), "cpea": synth_sd("CorrelationPhaseEstimationAgent", 192),
           "msa": synth_sd("MaskSynthesisAgent", 193)}
    path = _path((191, 192, 193))
    noisy, _ = syn.synth_wave(B, L, 777)
    o = orc.enhance_path(sds, noisy, 16000, use_memory=False)
    ref = torch.cat([o["mask_real"], o["mask_imag"]], -1).numpy()
    w = torch.from_numpy(noisy).cuda()

    def run():
        out = path(w)
        return rmse(torch.cat([out["mask_real"], out["mask_imag"]], -1).cpu(), ref)
    return run


def case_enhancer():
    g = gold("g8_enhancer")
    se = cp.SpeechEnhancer(n_freq=129)
    se.load_state_dict(synth_sd("SpeechEnhancer", 81), strict=True)
    se = se.cuda().eval()
    noisy, _ = syn.synth_wave(2, 2000, 82)
    w = torch.from_numpy(noisy).cuda()

    def run():
        nr, ni = cp.batch_stft(w, 256, 80, 160)
        _, _, mm = se(nr, ni)
        return rmse(mm.cpu(), g["mask_mag"])
    return run


def main():
    cases = {"msa_g5": case_msa(), "path_g9": case_path_golden(), "path_oracle_B2_L16000": case_path_oracle(),
             "enhancer_g8": case_enhancer()}
    table = {}
    with torch.no_grad():
        for name, pol in policies():
            ops.set_precision_policy(pol)
            row = {c: run() for c, run in cases.items()}
            table[name] = row
            print("%-28s " % name + "  ".join("%s %.3e" % (c, v) for c, v in row.items()), file=sys.stderr, flush=True)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
