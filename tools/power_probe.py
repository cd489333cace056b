#!/usr/bin/env python3
"""Socket power and shader clock (rocm-smi, one sample per second) while ONE kind of kernel runs back to back for ~8 s: a device copy,
the attention kernel (B 256 x T 512), conv16p (b0.c1 + skip at B 64), ffn_fused, the whole forward pass (B 256 x 512 frames).
Evidence for what the chip's power management does under the path's mixed MFMA + VALU + HBM kernels."""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops

def smi():
    o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Package Power \(W\): ([0-9.]+)", o)
    s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
    return (float(p.group(1)) if p else None, int(s.group(1)) if s else None)

def run(name, fn, seconds=8.0, per_sync=200):
    samples, stop = [], [False]
    def sampler():
        time.sleep(2.0)
        while not stop[0]:
            samples.append(smi())
            time.sleep(1.0)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(per_sync): fn()                     # (deep enough that the host never starves the queue)
        torch.cuda.synchronize(); n += per_sync
    dt = time.time() - t0
    stop[0] = True; th.join()
    pw = [a for a, _ in samples if a]; ck = [b for _, b in samples if b]
    print(json.dumps({"load": name, "launch_us": round(1e6 * dt / n, 1), "power_W": pw, "sclk_MHz": ck}))

g = torch.Generator(device="cuda").manual_seed(0)
x = torch.empty(1 << 30, device="cuda", dtype=torch.uint8)
run("copy 1 GiB", lambda: x.clone(), per_sync=50)
ops.set_compute_dtype("bf16")
B, T, H, hd = 256, 512, 4, 64
qkv = torch.randn(B * T, 3 * H * hd, device="cuda", generator=g)
qkv[:, :H * hd] *= 1.4426950408889634 / hd ** 0.5       # q pre-scaled as the path's W_q pack does (bench.py headline_attention)
qkv = qkv.to(torch.bfloat16)
out = torch.empty(B * T, H * hd, device="cuda", dtype=torch.bfloat16)
run("attention B256 T512", lambda: ops.attention(qkv, B, T, H, hd, out=out, prescaled=True))
ops.set_compute_dtype("f16")
dt = torch.float16
Bc, L, cin, cout = 64, 64000, 64, 128
R = lambda *s: torch.randn(*s, device="cuda", generator=g)
x1 = R(Bc, L, cin).to(dt); sc1, sh1 = R(Bc, cin) * 0.1 + 1, R(Bc, cin) * 0.1
pw = ops.pack_linear(R(cout, cin, 7) / (cin * 7) ** 0.5, R(cout)); spw = ops.pack_linear(R(cout, cin, 1) / cin ** 0.5, R(cout))
Lout = 32000; P = 2 * ((Lout + 127) // 128)
o1 = torch.empty(Bc, Lout, cout, device="cuda", dtype=dt); o2 = torch.empty_like(o1)
p1 = torch.zeros(Bc, P, 16, 2, device="cuda"); p2 = torch.zeros_like(p1)
run("conv16p b0.c1+skip B64", lambda: ops.conv16p(x1, sc1, sh1, pw, o1, B=Bc, Lin=L, stride=2, pad=3, gn_partial=p1, gn_group=cout // 16,
                                                  skip_pw=spw, out_s=o2, gn_partial_s=p2))
# the whole forward pass at the bench shape (one pass at a time)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _b
from sincformer_metacog_speech_enhancement_amd import synthetic as syn
path, _w = _b.build_path("mixed")
path = path.cuda().eval()
noisy, _ = syn.synth_wave(256, 40880, 1234)
wave = torch.from_numpy(noisy).cuda()
def fwd():
    with torch.no_grad():
        path(wave)
run("forward pass B256 x 512 frames", fwd, seconds=10.0, per_sync=10)
