"""Weight-gradient GEMMs and the LayerNorm backward of the training step timed COLD (a 2 GB fill between launches evicts L2 and
the 256 MB MALL), the state they run in inside the step - tools/gemm_tn_bench.py loops over the same operands, which flatters
them.    python tools/tn_cold_probe.py [--reps 6]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    import torch
    from sincformer_metacog_speech_enhancement_amd import ops
    ops.set_compute_dtype("bf16")
    flush = torch.empty(1 << 29, device="cuda", dtype=torch.float32)

    def cold(fn):
        ts = []
        for _ in range(a.reps):
            flush.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]

    def warm(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10

    M = 205056
    for N, K in ((1024, 256), (256, 1024), (256, 256), (768, 256), (512, 256)):
        G = torch.randn(M, N, device="cuda").bfloat16()
        X = torch.randn(M, K, device="cuda").bfloat16()
        dW = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        fn = lambda: ops.gemm16_tn(G, X, dW, db)
        print("tn M%d N%-5d K%-5d cold %7.3f ms  warm %7.3f ms   (%.0f MB of operands)" % (M, N, K, cold(fn), warm(fn), 2.0 * M * (N + K) / 1e6))
        del G, X
    D = 256
    x = torch.randn(M, D, device="cuda")
    dres = torch.randn(M, D, device="cuda")
    dy = torch.randn(M, D, device="cuda").bfloat16()
    gam = torch.randn(D, device="cuda")
    dg, dbt = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    fn = lambda: ops.layernorm_bwd(x, gam, dy, dres, dg, dbt)
    print("layernorm_bwd M%d D%d 16-bit dy: cold %7.3f ms  warm %7.3f ms  (%.0f MB)" % (M, D, cold(fn), warm(fn), M * D * 14.0 / 1e6))


if __name__ == "__main__":
    main()
