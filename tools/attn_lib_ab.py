#!/usr/bin/env python3
"""A/B of several BUILDS of the library on the attention kernel (or any `--cmd`), one subprocess per (round, build), interleaved, so
that the builds see the same box and the same clock drift:  tools/attn_lib_ab.py base=<lib.so> exp1=<lib.so> ... [--rounds 5]
Each subprocess prints the average launch time of `iters` back-to-back launches at B 256 x T 512 x 4 heads x 64 (bf16, prescaled q)."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
from sincformer_metacog_speech_enhancement_amd import ops
B, T, H, hd = int(os.environ.get("AB_B", 256)), int(os.environ.get("AB_T", 512)), 4, 64
ops.set_compute_dtype(os.environ.get("AB_DT", "bf16"))
v = int(os.environ.get("AB_VARIANT", "0"))
if v: ops.set_attention_variant(v)
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B * T, 3 * H * hd, device="cuda", generator=g)
qkv[:, :H * hd] *= 1.4426950408889634 / hd ** 0.5
qkv = qkv.to(ops.compute_dtype())
out = torch.empty(B * T, H * hd, device="cuda", dtype=ops.compute_dtype())
for _ in range(5): ops.attention(qkv, B, T, H, hd, out=out, prescaled=True)
torch.cuda.synchronize()
ms = []
for w in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.attention(qkv, B, T, H, hd, out=out, prescaled=True)
    e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1) / 20)
print(json.dumps({"ms": sorted(ms)[2], "min": min(ms), "finite": bool(torch.isfinite(out.float()).all())}))
''' % ROOT


def main():
    rounds = 5
    builds = []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--rounds":
            rounds = int(args.pop(0))
        else:
            name, path = a.split("=", 1)
            builds.append((name, path))
    res = {n: [] for n, _ in builds}
    for r in range(rounds):
        for n, path in builds:
            env = dict(os.environ)
            if path != "default":
                env["SFM_LIB_PATH"] = path
            o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", o.stderr[-600:])
                continue
            res[n].append(json.loads(line[-1])["ms"])
    B, T = int(os.environ.get("AB_B", 256)), int(os.environ.get("AB_T", 512))
    fl = 4.0 * B * 4 * T * T * 64
    for n, _ in builds:
        ms = sorted(res[n])
        if ms:
            med = ms[len(ms) // 2]
            print("%-14s median %.4f ms  min %.4f  max %.4f   %.1f TFLOP/s = %.2f %% of 2.5 PF   (%s)" %
                  (n, med, ms[0], ms[-1], fl / med / 1e9, fl / med / 1e9 / 25.0, " ".join("%.4f" % m for m in res[n])))


if __name__ == "__main__":
    main()
