#!/usr/bin/env python3
"""FeedForwardModule microbenchmark: fused kernel vs LayerNorm + two gemm16 launches."""
import argparse, json, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops, functional as Fn

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=51264)
ap.add_argument("--ff", type=int, default=1024)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtype", default="f16")
a = ap.parse_args()
ops.set_compute_dtype(a.dtype)
M, D, FF = a.rows, 256, a.ff
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(M, D, device="cuda", generator=g)
sd = {"layer_norm.weight": torch.ones(D, device="cuda"), "layer_norm.bias": torch.zeros(D, device="cuda"),
      "linear1.weight": torch.randn(FF, D, device="cuda", generator=g) / 16, "linear1.bias": torch.zeros(FF, device="cuda"),
      "linear2.weight": torch.randn(D, FF, device="cuda", generator=g) / math.sqrt(FF), "linear2.bias": torch.zeros(D, device="cuda")}
pk = Fn.pack_ffn(sd)
res = {}
for fused in (True, False):
    Fn.FUSED_FFN = fused
    for _ in range(3):
        Fn.ffn_forward(x, pk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        Fn.ffn_forward(x, pk)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    res["fused" if fused else "unfused"] = {"ms": ms, "tflops": 4.0 * M * D * FF / ms / 1e9}
print(json.dumps({"rows": M, "ff": FF, **res}))
