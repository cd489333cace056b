"""Weight-gradient GEMM (dW[n,k] += sum_m G[m,n] X[m,k]) timings on the shapes of the training step.
    python tools/gemm_tn_bench.py [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(205056, 1024, 256), (205056, 256, 1024), (205056, 256, 256), (205056, 768, 256), (205056, 512, 256),
          (51264, 1024, 256), (51264, 256, 1024), (51264, 256, 256)]
CONVS = [(64, 32000, 64000, 64, 128, 7, 2, 3), (64, 16000, 32000, 128, 128, 7, 2, 3), (64, 8000, 16000, 128, 256, 7, 2, 3),
         (64, 32000, 32000, 128, 128, 3, 1, 1), (64, 32000, 64000, 64, 128, 1, 2, 0),
         # the training step at B 256 x 4 s (bench.py --workload c3t)
         (256, 32000, 64000, 64, 128, 7, 2, 3), (256, 32000, 32000, 128, 128, 3, 1, 1), (256, 16000, 32000, 128, 128, 7, 2, 3),
         (256, 8000, 16000, 128, 256, 7, 2, 3), (256, 8000, 8000, 256, 256, 3, 1, 1), (256, 4000, 8000, 256, 256, 5, 2, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import torch
    from sincformer_metacog_speech_enhancement_amd import ops
    ops.set_compute_dtype("bf16")

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    for M, N, K in SHAPES:
        G = torch.randn(M, N, device="cuda").bfloat16()
        X = torch.randn(M, K, device="cuda").bfloat16()
        dW = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        dW.zero_(); db.zero_()
        ops.gemm16_tn(G, X, dW, db)
        ref = G.float().t() @ X.float()
        err = float((dW - ref).abs().max() / ref.abs().max())
        berr = float((db - G.float().sum(0)).abs().max() / G.float().sum(0).abs().max())
        assert err < 2e-3 and berr < 2e-3, (M, N, K, err, berr)
        ms = timeit(lambda: ops.gemm16_tn(G, X, dW, db))
        print("tn M%-7d N%-5d K%-5d %7.3f ms %7.1f TF/s  %6.0f GB/s (unique bytes)" %
              (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * (N + K) / ms / 1e6))
    for B, Lout, Lin, Cin, N, k, s, p in CONVS:
        dy = torch.randn(B * Lout, N, device="cuda").bfloat16()
        x = torch.randn(B, Lin, Cin, device="cuda").bfloat16()
        ms = timeit(lambda: ops.conv_wgrad16(dy, x, B, Lout, Lin, Cin, N, k, s, p))
        print("conv M%-7d N%-4d K%-5d s%d %7.3f ms %7.1f TF/s  %6.0f GB/s (unique bytes)" %
              (B * Lout, N, k * Cin, s, ms, 2.0 * B * Lout * N * k * Cin / ms / 1e9,
               2.0 * (B * Lout * N + B * Lin * Cin) / ms / 1e6))


if __name__ == "__main__":
    main()
