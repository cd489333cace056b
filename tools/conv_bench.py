#!/usr/bin/env python3
"""conv16p micro-benchmark on the PerceptionAgent layer shapes of the bench workload (B 64 x 4 s)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype("f16")
dt = torch.float16
B = 64
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)
LAYERS = [("b0.c1+cs", 64, 128, 7, 2, 3, False, True, 64000), ("b0.c2", 128, 128, 3, 1, 1, False, False, 32000),
          ("b1.c1+cs", 128, 128, 7, 2, 3, True, True, 32000), ("b1.c2", 128, 128, 3, 1, 1, False, False, 16000),
          ("b2.c1", 128, 256, 7, 2, 3, True, False, 16000), ("b2.cs", 128, 256, 1, 2, 0, True, False, 16000),
          ("b2.c2", 256, 256, 3, 1, 1, False, False, 8000), ("down", 256, 256, 5, 2, 2, True, False, 8000)]
tot = 0.0
res = {}
for name, cin, cout, k, s, p, two, skip, L in LAYERS:
    x1, x2 = R(B, L, cin).to(dt), (R(B, L, cin).to(dt) if two else None)
    sc1, sh1 = R(B, cin) * 0.1 + 1, R(B, cin) * 0.1
    sc2, sh2 = (R(B, cin) * 0.1 + 1, R(B, cin) * 0.1) if two else (None, None)
    pw = ops.pack_linear(R(cout, cin, k) / (cin * k) ** 0.5, R(cout))
    spw = ops.pack_linear(R(cout, cin, 1) / cin ** 0.5, R(cout)) if skip else None
    Lout = (L + 2 * p - k) // s + 1
    P = 2 * ((Lout + 127) // 128)
    out = torch.empty(B, Lout, cout, device="cuda", dtype=dt)
    part = torch.zeros(B, P, 16, 2, device="cuda")
    outs = torch.empty(B, Lout, cout, device="cuda", dtype=dt) if skip else None
    parts = torch.zeros(B, P, 16, 2, device="cuda") if skip else None
    run = lambda: ops.conv16p(x1, sc1, sh1, pw, out, B=B, Lin=L, stride=s, pad=p, x2=x2, sc2=sc2, sh2=sh2, gn_partial=part,
                              gn_group=cout // 16, skip_pw=spw, out_s=outs, gn_partial_s=parts)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * Lout * cout * (k * cin + (cin if skip else 0))
    res[name] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
    tot += ms
res["total_ms"] = round(tot, 4)
print(json.dumps(res))
