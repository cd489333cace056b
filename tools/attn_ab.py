#!/usr/bin/env python3
"""A/B of the head_dim-64 attention kernels in ONE process, interleaved rounds (cdna guide rule 24): per variant the median
and minimum of `rounds` x `iters` launches at the headline shape (B 256 x T 512 x 4 heads) or any other; also checks each
variant against fp32 torch on one (batch, head)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--heads", type=int, default=4)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--variants", default="3,4")
ap.add_argument("--prescaled", action="store_true", help="q already carries log2(e)/sqrt(hd), as functional.pack_mhsa folds it into W_q: the call the path makes")
a = ap.parse_args()
ops.set_compute_dtype(a.dtype)
B, T, H, hd = a.batch, a.frames, a.heads, 64
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B * T, 3 * H * hd, device="cuda", generator=g).to(ops.compute_dtype())
out = torch.empty(B * T, H * hd, device="cuda", dtype=ops.compute_dtype())
variants = [int(v) for v in a.variants.split(",")]
q, k, v = [t.float().reshape(B, T, H, hd) for t in qkv.split(H * hd, dim=1)]
bi, hi = B - 1, H - 1
ref = torch.softmax(q[bi, :, hi] @ k[bi, :, hi].t() / hd ** 0.5, dim=-1) @ v[bi, :, hi]
if a.prescaled:
    qkv[:, :H * hd] = (qkv[:, :H * hd].float() * (1.4426950408889634 / hd ** 0.5)).to(qkv.dtype)
res = {}
for var in variants:
    ops.set_attention_variant(var)
    out.zero_()
    for _ in range(3):
        ops.attention(qkv, B, T, H, hd, out=out, prescaled=a.prescaled)
    torch.cuda.synchronize()
    err = float((out.float().reshape(B, T, H, hd)[bi, :, hi] - ref).abs().max())
    res[var] = {"max_err_vs_fp32": err, "finite": bool(torch.isfinite(out.float()).all()), "ms": []}
for r in range(a.rounds):
    for var in variants:
        ops.set_attention_variant(var)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.attention(qkv, B, T, H, hd, out=out, prescaled=a.prescaled)
        e1.record()
        torch.cuda.synchronize()
        res[var]["ms"].append(e0.elapsed_time(e1) / a.iters)
ops.set_attention_variant(0)
fl = 4.0 * B * H * T * T * hd
for var in variants:
    ms = sorted(res[var].pop("ms"))
    med, mn = ms[len(ms) // 2], ms[0]
    res[var].update({"ms_median": med, "ms_min": mn, "tflops_median": fl / med / 1e9, "frac_2.5PF_median": fl / med / 1e9 / 2500,
                     "frac_2.5PF_best": fl / mn / 1e9 / 2500})
print(json.dumps({"B": B, "T": T, "H": H, "dtype": a.dtype, "variants": res}))
