#!/usr/bin/env python3
"""Attention-only microbenchmark (headline of BASELINE.json: Conformer attention at batch 256 x 512-frame
utterances, 4 heads x 64): HIP-event timing of sfm_attention_fwd, TFLOP/s = 4*B*H*T^2*hd / t."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--heads", type=int, default=4)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--variant", type=int, default=0, help="0: auto, 1: 32 q rows/wave, 2: 64 q rows/wave, 3: persistent ring, 4 / 5: pipelined persistent kernel with 8 / 4 waves (attention_pipe.hip)")
ap.add_argument("--check", action="store_true", help="compare with the 32-rows-per-wave kernel (variant 1) and fp32 torch on one (b, h)")
a = ap.parse_args()
ops.set_compute_dtype(a.dtype)
ops.set_attention_variant(a.variant)
B, T, H, hd = a.batch, a.frames, a.heads, 64
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B * T, 3 * H * hd, device="cuda", generator=g)
qkv[:, :H * hd] *= 1.4426950408889634 / hd ** 0.5            # q pre-scaled as functional.pack_mhsa folds it into W_q: the path's call
qkv = qkv.to(ops.compute_dtype())
out = torch.empty(B * T, H * hd, device="cuda", dtype=ops.compute_dtype())
for _ in range(3):
    ops.attention(qkv, B, T, H, hd, out=out, prescaled=True)
torch.cuda.synchronize()
if a.check:
    ops.set_attention_variant(1)
    ref = ops.attention(qkv, B, T, H, hd, prescaled=True)
    ops.set_attention_variant(a.variant)
    torch.cuda.synchronize()
    d = (out.float() - ref.float()).abs()
    q, k, v = [t.float().reshape(B, T, H, hd) for t in qkv.split(H * hd, dim=1)]
    bi, hi = B - 1, H - 1
    p = torch.softmax(q[bi, :, hi] @ k[bi, :, hi].t() * 0.6931471805599453, dim=-1) @ v[bi, :, hi]   # q carries log2(e)/sqrt(hd)
    e = (out.float().reshape(B, T, H, hd)[bi, :, hi] - p).abs().max()
    print("check: max |variant %d - variant 1| = %.3e (mean %.3e); max |out - fp32 softmax| on (b=%d, h=%d) = %.3e, finite %s" %
          (a.variant, float(d.max()), float(d.mean()), bi, hi, float(e), bool(torch.isfinite(out.float()).all())), file=sys.stderr)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    ops.attention(qkv, B, T, H, hd, out=out, prescaled=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
fl = 4.0 * B * H * T * T * hd
print(json.dumps({"kernel": ops.attention_kernel_name(B, T, H), "B": B, "T": T, "H": H, "dtype": a.dtype, "ms": ms,
                  "tflops": fl / ms / 1e9, "frac_of_2.5PF": fl / ms / 1e9 / 2500.0}))
