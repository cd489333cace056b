#!/usr/bin/env python3
"""Diagnostic (library built by `tools/variant_lib.sh attention_pipe stamps -DSFM_ATTNP_STAMPS`, loaded through SFM_LIB_PATH):
where a wave of attn_fwd_hd64p spends its cycles at the headline shape - shares of [wait + barrier], [item prologue],
[step loops], [drain + normalise] - for waves 0-3 and 4-7 (the static-priority half)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sincformer_metacog_speech_enhancement_amd import ops, lib

B, T, H, hd = 256, int(os.environ.get("T", 512)), 4, 64
ops.set_compute_dtype("bf16")
ops.set_attention_variant(4)
L = lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B * T, 3 * H * hd, device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty(B * T, H * hd, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda", dtype=torch.float32)
D = H * hd
for _ in range(6):
    rc = L.sfm_attention_fwd_train(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T, H, hd, 3 * D, D, D, 2 * D, T * 3 * D, T * D,
                                   1.0 / 8.0, 0.0, 0, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
torch.cuda.synchronize()
st = lse.view(torch.int64).reshape(-1)[: 256 * 8 * 8].reshape(256, 8, 8).cpu().numpy().astype(np.float64)
tot = st[:, :, 1] - st[:, :, 0]
res = {}
for name, sl in (("waves0-3", slice(0, 4)), ("waves4-7", slice(4, 8))):
    t = tot[:, sl]
    res[name] = {"total_cycles_p50": float(np.median(t)),
                 "barrier_share": float(np.median(st[:, sl, 2] / t)), "prologue_share": float(np.median(st[:, sl, 3] / t)),
                 "steps_share": float(np.median(st[:, sl, 4] / t)), "drain_share": float(np.median(st[:, sl, 5] / t)),
                 "qfrag_share": float(np.median((st[:, sl, 6] // 2 ** 32) / t)), "store_o_share": float(np.median((st[:, sl, 6] % 2 ** 32) / t)),
                 "drain_mfma_share": float(np.median(st[:, sl, 7] / t)),
                 "steps_cycles_per_ritem": float(np.median(st[:, sl, 4])) / (4 * 2 * ((T + 31) // 32))}
print(json.dumps(res, indent=1))
