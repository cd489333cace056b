import sys, torch
sys.path.insert(0, ".")
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype("bf16")
T = int(sys.argv[1]); B, H, hd = 1, 1, 64
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B * T, 3 * H * hd, device="cuda", generator=g).to(torch.bfloat16)
out = ops.attention(qkv, B, T, H, hd)
q, k, v = qkv.float().split(64, dim=1)
ref = torch.softmax(q @ k.t() / 8.0, dim=-1) @ v
print("T", T, "max err", float((out.float() - ref).abs().max()))
r = (out.float() / ref)
for row in range(min(T, 70)):
    if row in (0, 1, 2, 31, 32, 33, 63, 64, 65):
        print(row, "ratio first 6:", [round(float(x), 3) for x in r[row, :6]], "err", float((out.float()[row] - ref[row]).abs().max()))
