"""Forward-type GEMMs of the training step at M = B x T rows (fused-Swish FFN Linear forward / backward, residual and plain epilogues):
    python tools/gemm_train_bench.py [--rows 205056] [--iters 10]
A/B knobs are process-wide environment variables (SFM_SWISH_VARIANT), so run the tool once per setting."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=205056)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--variant", type=int, default=0, help="ops.set_gemm_variant for the non-Swish GEMMs")
    a = ap.parse_args()
    import torch
    from sincformer_metacog_speech_enhancement_amd import ops
    ops.set_compute_dtype("bf16")
    ops.set_gemm_variant(a.variant)
    M = a.rows
    dt = torch.bfloat16

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

    def line(name, ms, flops, nbytes):
        print("%-44s %7.3f ms %7.1f TF/s %6.0f GB/s" % (name, ms, flops / ms / 1e9, nbytes / ms / 1e6))

    g = torch.Generator(device="cuda").manual_seed(3)
    x256 = (torch.randn(M, 256, device="cuda", generator=g)).to(dt)
    x1024 = (torch.randn(M, 1024, device="cuda", generator=g) * 0.5).to(dt)
    w1 = torch.randn(1024, 256, device="cuda", generator=g) * 0.06
    w2 = torch.randn(256, 1024, device="cuda", generator=g) * 0.03
    b1 = torch.randn(1024, device="cuda", generator=g) * 0.1
    p1 = ops.pack_linear(w1, b1)
    p1t = ops.pack_linear(w2.t().contiguous())          # [1024, 256]: backward of the second Linear
    # fused Swish forward / backward, with and without hidden dropout
    for pd in (0.0, 0.1):
        d, u = ops.linear16_swish(x256, p1, p_drop=pd, seed=5)
        z = x256.float() @ w1.t().to(dt).float() + b1
        if pd == 0.0:
            ref = z * torch.sigmoid(z)
            err = float((u.float() - ref).abs().max() / ref.abs().max())
            dref = torch.sigmoid(z) * (1 + z * (1 - torch.sigmoid(z)))
            derr = float((d.float() - dref).abs().max())
            assert err < 2e-2 and derr < 2e-2, (err, derr)
            del ref, dref
        del z
        ms = timeit(lambda: ops.linear16_swish(x256, p1, p_drop=pd, seed=5))
        line("swish fwd N1024 K256 p_drop %.1f" % pd, ms, 2.0 * M * 1024 * 256, M * (256 + 2 * 1024) * 2.0)
        gy = ops.linear16_swish(x256, p1t, aux=d)
        ref = (x256.float() @ w2.to(dt).float()) * d.float()
        err = float((gy.float() - ref).abs().max() / ref.abs().max())
        assert err < 2e-2, err
        del ref
        ms = timeit(lambda: ops.linear16_swish(x256, p1t, aux=d))
        line("swish bwd N1024 K256", ms, 2.0 * M * 1024 * 256, M * (256 + 2 * 1024) * 2.0)
        del d, u, gy
    # second FFN Linear: fp32 residual in and out (+ dropout), and the plain fp32 / 16-bit outputs of the input gradients
    p2 = ops.pack_linear(w2, torch.zeros(256, device="cuda"))
    res = torch.randn(M, 256, device="cuda", generator=g)
    out32 = torch.empty(M, 256, device="cuda")
    for pd in (0.0, 0.1):
        ms = timeit(lambda: ops.linear16(x1024, p2, epi=ops.EPI_RESID, resid=res, alpha=0.5, out=out32, p_drop=pd, seed=9))
        line("resid N256 K1024 fp32 io p_drop %.1f" % pd, ms, 2.0 * M * 256 * 1024, M * (1024 * 2 + 256 * 8.0))
    ms = timeit(lambda: ops.linear16(x1024, p2, out=out32))
    line("plain N256 K1024 fp32 out", ms, 2.0 * M * 256 * 1024, M * (1024 * 2 + 256 * 4.0))
    for N in (256, 512, 768):
        pw = ops.pack_linear(torch.randn(N, 256, device="cuda", generator=g) * 0.06, torch.zeros(N, device="cuda"))
        o16 = torch.empty(M, N, device="cuda", dtype=dt)
        ms = timeit(lambda: ops.linear16(x256, pw, out=o16))
        line("plain N%d K256 16-bit out" % N, ms, 2.0 * M * N * 256, M * (256 + N) * 2.0)
        o32 = torch.empty(M, N, device="cuda")
        ms = timeit(lambda: ops.linear16(x256, pw, out=o32))
        line("plain N%d K256 fp32 out" % N, ms, 2.0 * M * N * 256, M * (256 * 2 + N * 4.0))


if __name__ == "__main__":
    main()
