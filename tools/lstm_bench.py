#!/usr/bin/env python3
"""bilstm_layer micro-benchmark at the bench workload's shape (B 64 chains x 2 directions, T 801, H 128)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
B, T, H = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 801, 128
g = torch.Generator(device="cuda").manual_seed(0)
xg = torch.randn(B, T, 2, 4 * H, device="cuda", generator=g)
whh = torch.randn(2, 4 * H, H, device="cuda", generator=g) / H ** 0.5
run = lambda: ops.bilstm_layer(xg, whh, B, T, H)
for _ in range(2):
    out = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(json.dumps({"ms": ms, "us_per_step": 1e3 * ms / T, "checksum": float(out.double().abs().sum())}))
# training pair at the c3t batch (B 256): forward with the saved gates, then the BPTT kernel
Bt = 256
xg = torch.randn(Bt, T, 2, 4 * H, device="cuda", generator=g)
out, save = ops.bilstm_layer_train(xg, whh, Bt, T, H)
dout = torch.randn(Bt, T, 2 * H, device="cuda", generator=g)
for name, fn in (("fwd_train_b256", lambda: ops.bilstm_layer_train(xg, whh, Bt, T, H)), ("bwd_b256", lambda: ops.bilstm_layer_bwd(save, whh, dout, Bt, T, H))):
    for _ in range(2):
        r = fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    r0 = r[0] if isinstance(r, tuple) else r
    print(json.dumps({"what": name, "ms": ms, "us_per_step": 1e3 * ms / T, "checksum": float(r0.double().abs().sum())}))
