import torch, sys
sys.path.insert(0, "/root/repo")
M = 205056
def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (N, K) in [(768, 256), (512, 256), (1024, 256), (256, 256), (256, 1024), (256, 768)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ms = t(lambda: torch.matmul(a, w.t(), out=out))
    print("NT  M%d N%d K%d bf16 out: %.3f ms %.0f TF/s %.0f GB/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, (M * K + M * N) * 2 / ms / 1e6))
for (N, K) in [(1024, 256), (256, 1024), (256, 256), (768, 256)]:
    g = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, K, device="cuda").bfloat16()
    ms = t(lambda: torch.matmul(g.t(), x))
    print("TN  M%d N%d K%d: %.3f ms %.0f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
