#!/usr/bin/env python3
"""Diagnostic (needs a library built with -DSFM_ABL=9, tools/attn_ablate.sh 9): per-workgroup start / end stamps of the
persistent attention kernel at the headline shape -> dispatch stagger, per-item times, spread over CUs."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sincformer_metacog_speech_enhancement_amd import ops, lib

B, T, H, hd = 256, 512, 4, 64
ops.set_compute_dtype("bf16")
ops.set_attention_variant(3)
L = lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B * T, 3 * H * hd, device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty(B * T, H * hd, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(B, H, T, device="cuda", dtype=torch.float32)
D = H * hd


def run():
    rc = L.sfm_attention_fwd_train(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T, H, hd, 3 * D, D, D, 2 * D, T * 3 * D, T * D,
                                   1.0 / 8.0, 0.0, 0, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for _ in range(5):
    run()
torch.cuda.synchronize()
lse.zero_()
lse2 = torch.zeros_like(lse)
run()
lse, lse2 = lse2, lse
run()                                        # second launch right behind the first: the gap between them
lse, lse2 = lse2, lse
torch.cuda.synchronize()
s1 = lse.view(torch.int64).reshape(-1)[: 256 * 8].reshape(256, 8).cpu().numpy().astype(np.float64)
s2 = lse2.view(torch.int64).reshape(-1)[: 256 * 8].reshape(256, 8).cpu().numpy().astype(np.float64)
print("launch 1: first start 0, last end %.2f us; launch 2: first start %.2f us, last end %.2f us -> gap %.2f us, period %.2f us" %
      ((s1[:, 1].max() - s1[:, 0].min()) / 100, (s2[:, 0].min() - s1[:, 0].min()) / 100, (s2[:, 1].max() - s1[:, 0].min()) / 100,
       (s2[:, 0].min() - s1[:, 1].max()) / 100, (s2[:, 0].min() - s1[:, 0].min()) / 100), file=sys.stderr)
st = lse.view(torch.int64).reshape(-1)[: 256 * 8].reshape(256, 8).cpu().numpy().astype(np.float64)
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0          # microseconds
items = (st[:, 2:6] - t0) / 100.0
dur = end - start
print(json.dumps({"wg_start_us": {"min": start.min(), "p50": float(np.median(start)), "max": start.max()},
                  "wg_end_us": {"min": end.min(), "p50": float(np.median(end)), "max": end.max()},
                  "wg_duration_us": {"min": dur.min(), "p50": float(np.median(dur)), "max": dur.max()},
                  "item_us_p50": [float(np.median(items[:, 0] - start))] + [float(np.median(items[:, i] - items[:, i - 1])) for i in range(1, 4)],
                  "item_us_max": [float((items[:, 0] - start).max())] + [float((items[:, i] - items[:, i - 1]).max()) for i in range(1, 4)],
                  "by_xcd_duration_p50": [float(np.median(dur[x::8])) for x in range(8)]}, indent=1))
