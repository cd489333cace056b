"""is the weight-gradient GEMM reproducible and right at the training sizes?  (same inputs, several launches, vs fp32 matmul)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype(torch.float16)
torch.manual_seed(0)
for (M, N, K) in [(205056, 256, 1024), (205056, 1024, 256), (205056, 512, 256), (205056, 256, 264), (205056, 129, 256), (205056, 256, 256)]:
    G = (torch.randn(M, N, device="cuda") * 0.01).half()
    Kp = (K + 7) // 8 * 8
    X = torch.zeros(M, Kp, device="cuda", dtype=torch.float16)
    X[:, :K] = torch.randn(M, K, device="cuda").half()
    Gp = torch.zeros(M, (N + 7) // 8 * 8, device="cuda", dtype=torch.float16)
    Gp[:, :N] = G
    ref = Gp[:, :N].float().t() @ X[:, :K].float()
    outs = []
    for it in range(5):
        dW = torch.zeros(N, K, device="cuda")
        db = torch.zeros(N, device="cuda")
        ops.gemm16_tn(Gp[:, :N], X[:, :K], dW, db)
        torch.cuda.synchronize()
        outs.append((dW.clone(), db.clone()))
    errs = [float((o[0] - ref).abs().max() / ref.abs().max()) for o in outs]
    berrs = [float((o[1] - Gp[:, :N].float().sum(0)).abs().max() / Gp[:, :N].float().sum(0).abs().max()) for o in outs]
    print("M%d N%d K%d: max rel err vs fp32 matmul per launch %s ; bias %s" % (M, N, K, ["%.1e" % e for e in errs], ["%.1e" % e for e in berrs]))
