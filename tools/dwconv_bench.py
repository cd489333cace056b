#!/usr/bin/env python3
"""depthwise conv (k 31, C 256) + folded BatchNorm + Swish: the fp16 dot-product kernel against the multiply-add kernel
(SFM_DWCONV_DOT=0), one subprocess each, at the shapes of the bench workloads"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype(torch.float16)
g = torch.Generator(device="cuda").manual_seed(0)
C, KS = 256, 31
r = {}
for B, T in ((64, 801), (256, 512), (256, 801)):
    x = torch.randn(B * T, C, device="cuda", generator=g).half()
    wT = (torch.randn(KS, C, device="cuda", generator=g) / KS ** 0.5).contiguous()
    sc, sh = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g)
    for _ in range(3): out = ops.dwconv_folded(x, wT, sc, sh, B, T, C)
    torch.cuda.synchronize()
    ts = []
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): out = ops.dwconv_folded(x, wT, sc, sh, B, T, C)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    r["B%%d_T%%d_us" %% (B, T)] = round(sorted(ts)[1] * 1e3, 1)
    r["B%%d_T%%d_sum" %% (B, T)] = float(out.float().double().abs().sum())
print(json.dumps(r))
''' % ROOT
for rnd in range(2):
    for dot, name in (("0", "fma"), ("1", "dot")):
        env = dict(os.environ, SFM_DWCONV_DOT=dot)
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        print(name, line[-1] if line else o.stderr[-400:])
