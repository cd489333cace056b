"""one-off probes of gemm16 on the large PerceptionAgent conv shape: what bounds it?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype("bf16")


def run(B, Lout, Cin, k, s, N, of32=0, gn=0, label=""):
    Lin = Lout * s
    w = torch.randn(N, Cin, k, device="cuda") * 0.05 if k > 1 else torch.randn(N, Cin, device="cuda") * 0.05
    pw = ops.pack_linear(w, torch.zeros(N, device="cuda"))
    x = torch.randn(B, Lin, Cin, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, Lout, N, device="cuda", dtype=torch.float32 if of32 else torch.bfloat16)
    part = torch.empty(B, 2 * ((Lout + 127) // 128), 8, 2, device="cuda") if gn else None

    def go():
        ops.gemm16(x, pw, out, B=B, Lout=Lout, Lin=Lin, a_batch_stride=Lin * Cin, ldo=N, o_batch_stride=Lout * N,
                   stride=s, pad=(k - 1) // 2, gn_partial=part, gn_group=(N // 8 if gn else 0))
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * Lout * N * Cin * k
    by = 2.0 * B * Lin * Cin + (4 if of32 else 2) * B * Lout * N
    print("%-34s B%d Lout%d Cin%d k%d s%d N%d: %.3f ms %5.0f TF/s, %5.0f GB/s unique" % (label, B, Lout, Cin, k, s, N, ms, fl / ms / 1e9, by / ms / 1e6))


run(64, 32000, 64, 7, 2, 128, gn=1, label="conv k7 s2 (as on the path)")
run(64, 32000, 64, 7, 2, 128, gn=0, label="conv k7 s2, no GN partials")
run(1, 2048000, 448, 1, 1, 128, label="plain GEMM, same M N K")
run(64, 32000, 64, 7, 2, 64, label="conv, N 64")
run(64, 32000, 64, 7, 2, 256, label="conv, N 256")
run(64, 32000, 64, 3, 2, 128, label="conv k3 (K 192)")
run(64, 32000, 64, 15, 2, 128, label="conv k15 (K 960)")
run(1, 2048000, 64, 1, 1, 128, label="plain GEMM K 64")
run(1, 2048000, 1024, 1, 1, 128, label="plain GEMM K 1024")
run(1, 2048000, 1024, 1, 1, 256, label="plain GEMM K 1024 N 256")
run(1, 262144, 4096, 1, 1, 256, label="plain GEMM K 4096 N 256")
run(1, 65536, 4096, 1, 1, 4096, label="plain GEMM 64k x 4096 x 4096")
