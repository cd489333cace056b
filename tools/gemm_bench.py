#!/usr/bin/env python3
"""gemm16 micro-benchmark over the GEMM / implicit-conv shapes of the bench workload, per kernel variant
(2: 128-row tiles, 2-stage ring; 3: 3 stages; 4 / 5: 256-row tiles with 2 / 3 stages)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops

# (B, Lout, Cin, ksize, stride, N, out_f32, gn)   conv: Lin = Lout*stride ; linear: ksize 1
SHAPES = [
    (64, 32000, 64, 7, 2, 128, 0, 1), (64, 32000, 128, 3, 1, 128, 0, 1), (64, 16000, 128, 7, 2, 128, 0, 1),
    (64, 8000, 128, 7, 2, 256, 0, 1), (64, 8000, 256, 3, 1, 256, 0, 1), (64, 4000, 256, 5, 2, 256, 0, 1),
    (64, 32000, 64, 1, 2, 128, 0, 1), (1, 51264, 256, 1, 1, 768, 0, 0), (1, 51264, 256, 1, 1, 256, 1, 0),
    (1, 51264, 1024, 1, 1, 256, 1, 0), (1, 51264, 256, 1, 1, 1024, 0, 0), (1, 205056, 256, 1, 1, 1024, 0, 0),
    (1, 205056, 1024, 1, 1, 256, 1, 0),
]
ops.set_compute_dtype("bf16")
res = []
for (B, Lout, Cin, k, s, N, of32, gn) in SHAPES:
    Lin = Lout * s
    w = torch.randn(N, Cin, k, device="cuda") * 0.05 if k > 1 else torch.randn(N, Cin, device="cuda") * 0.05
    pw = ops.pack_linear(w, torch.zeros(N, device="cuda"))
    x = torch.randn(B, Lin, Cin, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, Lout, N, device="cuda", dtype=torch.float32 if of32 else torch.bfloat16)
    part = torch.empty(B, 2 * ((Lout + 127) // 128), 8, 2, device="cuda") if gn else None
    row = {"shape": "B%d Lout%d Cin%d k%d s%d N%d %s" % (B, Lout, Cin, k, s, N, "f32" if of32 else "16b")}
    ref = None
    for v in (2, 9, 10):
        ops.set_gemm_variant(v)
        def run():
            ops.gemm16(x, pw, out, B=B, Lout=Lout, Lin=Lin, a_batch_stride=Lin * Cin, ldo=N, o_batch_stride=Lout * N,
                       stride=s, pad=(k - 1) // 2, gn_partial=part, gn_group=(N // 8 if gn else 0))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.float().clone()
            pref = part.clone() if gn else None
        else:
            assert torch.equal(out.float(), ref), "variant %d differs" % v
            if gn:
                assert torch.allclose(part, pref, rtol=1e-4, atol=1e-3), "variant %d gn partials differ" % v
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        row["v%d" % v] = "%.3f ms %4.0f TF" % (ms, 2.0 * B * Lout * N * Cin * k / ms / 1e9)
    ops.set_gemm_variant(0)
    print(json.dumps(row))
