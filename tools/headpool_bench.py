#!/usr/bin/env python3
"""PerceptionAgent latent heads + time pooling: one launch (sfm_headpool) + the affine pass against heads GEMM + pooling pass"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
from sincformer_metacog_speech_enhancement_amd import functional as Fn
ops.set_compute_dtype(torch.float16)
dt = torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[1] * 1e3
for B, Tin, Tout in ((256, 2555, 512), (64, 4000, 801)):
    xd = torch.randn(B, Tin, 256, device="cuda", generator=g).to(dt)
    pw = ops.pack_linear(torch.randn(512, 256, 1, device="cuda", generator=g) / 16, torch.randn(512, device="cuda", generator=g))
    sc, sh = torch.rand(B, 512, device="cuda", generator=g) + 0.5, torch.randn(B, 512, device="cuda", generator=g)
    fused = torch.empty(B * Tout, 1344, device="cuda", dtype=dt)
    tiles = ops.headpool_tiles(Tin, Tout)
    pooled = torch.empty(B, Tout, 512, device="cuda", dtype=dt)
    part = torch.empty(B, tiles[1], 32, 2, device="cuda")
    r = {"B": B, "Tin": Tin, "Tout": Tout, "frames_per_tile": tiles[0]}
    r["heads_gemm_us"] = round(timeit(lambda: Fn._conv_gn(xd, pw, B, Tin, 1, 0, 32, dt)), 1)
    raw = Fn._conv_gn(xd, pw, B, Tin, 1, 0, 32, dt)[0]
    r["pool_pass_us"] = round(timeit(lambda: ops.pool_time(raw, fused, None, B, Tin, Tout, 512, 512, 1344, scale=sc, shift=sh)), 1)
    r["headpool_us"] = round(timeit(lambda: ops.headpool(xd, pw, pooled, part, B, Tin, Tout)), 1)
    r["affine_pass_us"] = round(timeit(lambda: ops.pool_time(pooled, fused, None, B, Tout, Tout, 512, 512, 1344, scale=sc, shift=sh)), 1)
    print(json.dumps(r))
