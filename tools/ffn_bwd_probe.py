"""FeedForwardModule backward at the c3t row count (M 205 056) run the way the step runs it (forward of several modules first, so
the saved tensors are cold), every launch timed: where do the weight-gradient GEMMs lose time inside the step?
    python tools/ffn_bwd_probe.py [--p 0.1] [--layers 4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=float, default=0.1)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--scale", type=float, default=1.0, help="scale of the incoming gradient")
    ap.add_argument("--variant", default="step", help="step | nobias | late (dW2 after the Swish-backward GEMM) | clone (u copied first) | "
                    "flush (2 GB fill before dW2) | sync (device sync before dW2)")
    a = ap.parse_args()
    import torch
    from sincformer_metacog_speech_enhancement_amd import ops, train
    ops.set_compute_dtype("bf16")
    M, D, FF = 205056, 256, 1024
    pre = "ff1."
    g = torch.Generator(device="cuda").manual_seed(1)
    P = {pre + "layer_norm.weight": torch.ones(D, device="cuda"), pre + "layer_norm.bias": torch.zeros(D, device="cuda"),
         pre + "linear1.weight": torch.randn(FF, D, device="cuda", generator=g) / 16, pre + "linear1.bias": torch.zeros(FF, device="cuda"),
         pre + "linear2.weight": torch.randn(D, FF, device="cuda", generator=g) / 32, pre + "linear2.bias": torch.zeros(D, device="cuda")}
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    x = torch.randn(M, D, device="cuda", generator=g)
    seeds = train._Seeds(7)
    caches = []
    for _ in range(a.layers):
        y, c = train._ffn_fwd(x, P, pre, a.p, seeds)
        caches.append(c)
        x = y
    dy = torch.randn(M, D, device="cuda", generator=g) * a.scale
    torch.cuda.synchronize()
    ops.profiler.enable(None, tags=True)
    flush = torch.empty(1 << 29, device="cuda") if a.variant == "flush" else None
    for c in reversed(caches):
        if a.variant == "step":
            dy = train._ffn_bwd(dy, c, G, pre)
            continue
        do = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        ops.ew_train(ops.EW_SCALE_DROP, do, g=dy, alpha=0.5, p=c["p"], seed=c["s2"])
        u = c["u"].clone() if a.variant == "clone" else c["u"]
        if a.variant == "flush":
            flush.fill_(1.0)
        if a.variant == "sync":
            torch.cuda.synchronize()
        if a.variant != "late":
            ops.gemm16_tn(do, u, G[pre + "linear2.weight"], None if a.variant == "nobias" else G[pre + "linear2.bias"])
        dz = ops.linear16_swish(do, c["b2"], p_drop=c["p"], seed=c["s1"], aux=c["z1"])
        if a.variant == "late":
            ops.gemm16_tn(do, u, G[pre + "linear2.weight"], G[pre + "linear2.bias"])
        ops.gemm16_tn(dz, c["h16"], G[pre + "linear1.weight"], None if a.variant == "nobias" else G[pre + "linear1.bias"])
        dh = ops.linear16(dz, c["b1"])
        dy = ops.layernorm_bwd(c["x"], c["lw"], dh, dy, G[pre + "layer_norm.weight"], G[pre + "layer_norm.bias"])
    summ = ops.profiler.summary()
    ops.profiler.disable()
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms_total"]):
        if "gemm16_tn[" in k:
            print("  %-56s n %3d  %8.3f ms each" % (k, v["n"], v["ms_total"] / v["n"]))


if __name__ == "__main__":
    main()
