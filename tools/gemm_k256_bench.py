#!/usr/bin/env python3
"""The four K = 256 linears of a Conformer block at the bench shape (M = 64 x 801 rows): ms per launch with the default kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype("f16")
M, K = 64 * 801, 256
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(M, K, device="cuda", generator=g).half()
resid = torch.randn(M, 256, device="cuda", generator=g)
res = {}
for name, N, epi, kw in (("qkv N768", 768, ops.EPI_NONE, {}), ("out_proj N256 resid f32", 256, ops.EPI_RESID, {"resid": resid}),
                         ("pw1 N512 glu", 512, ops.EPI_GLU, {}), ("pw2 N256 resid f32", 256, ops.EPI_RESID, {"resid": resid})):
    pw = ops.pack_linear(torch.randn(N, K, device="cuda", generator=g) / 16, torch.zeros(N, device="cuda"), glu=(epi == ops.EPI_GLU))
    run = lambda: ops.linear16(x, pw, epi=epi, **kw)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    res[name] = {"us": round(1e3 * ms, 1), "tflops": round(2.0 * M * N * K / ms / 1e9)}
print(json.dumps(res))
