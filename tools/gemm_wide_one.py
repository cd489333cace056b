"""a few launches of the wide-tile GEMM kernels (for rocprofv3 --pmc passes): forward 256 x 256 tiles, 512 x 128 tiles, and the
256 x 256 weight-gradient kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype("bf16")
M, N, K = 205056, 1024, 256
x = torch.randn(M, K, device="cuda").bfloat16()
pw = ops.pack_linear(torch.randn(N, K, device="cuda") * 0.05, torch.zeros(N, device="cuda"))
for _ in range(6):
    ops.linear16(x, pw)                                                    # gemm16w 256 x 256
G = torch.randn(M, N, device="cuda").bfloat16()
dW = torch.zeros(N, K, device="cuda")
for _ in range(6):
    ops.gemm16_tn(G, x, dW)                                                # gemm16_tn_wide
B, Lout, Cin, k, s, Nn = 64, 32000, 64, 7, 2, 128
w = torch.randn(Nn, Cin, k, device="cuda") * 0.05
pc = ops.pack_linear(w, torch.zeros(Nn, device="cuda"))
xc = torch.randn(B, Lout * s, Cin, device="cuda").to(torch.bfloat16)
out = torch.empty(B, Lout, Nn, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm16(xc, pc, out, B=B, Lout=Lout, Lin=Lout * s, a_batch_stride=Lout * s * Cin, ldo=Nn, o_batch_stride=Lout * Nn,
               stride=s, pad=3)                                            # gemm16w 512 x 128
torch.cuda.synchronize()
