"""GroupNorm(+GELU) of the PerceptionAgent nodes at the c3t batch (B 256 x 4 s): the forward pass that materialises the activated
tensor (gn_apply) and the backward's reduce and apply pass, per node.
    python tools/gn_bwd_bench.py [--batch 256]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (L, C, groups, two inputs)
NODES = [(64000, 64, 8, False), (32000, 128, 8, True), (32000, 128, 8, False), (16000, 128, 8, True), (16000, 128, 8, False),
         (8000, 256, 16, True), (8000, 256, 16, False), (4000, 256, 16, False)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    import torch
    from sincformer_metacog_speech_enhancement_amd import ops
    ops.set_compute_dtype("bf16")
    B = a.batch
    dt = torch.bfloat16
    tot = [0.0, 0.0, 0.0]
    for L, C, G, two in NODES:
        g = torch.Generator(device="cuda").manual_seed(L + C)
        mk = lambda: torch.randn(B, L, C, device="cuda", generator=g).to(dt)
        x1, dout = mk(), mk()
        x2 = mk() if two else None
        tab = lambda: (torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda") * 0.1, torch.randn(B, G, device="cuda") * 0.1,
                       torch.rand(B, G, device="cuda") + 0.5)
        t1, t2 = tab(), tab()
        gam = torch.ones(C, device="cuda")
        args = [x1, t1[0], t1[1], t1[2], t1[3], gam] + ([x2, t2[0], t2[1], t2[2], t2[3], gam] if two else [])
        line = "L %6d C %4d %s:" % (L, C, "two" if two else "one")
        out = torch.empty_like(x1)
        fwd = lambda: ops.gn_apply(x1, t1[0], t1[1], out, B, L, C, act=1, x2=x2, sc2=t2[0] if two else None, sh2=t2[1] if two else None)
        for _ in range(2):
            fwd()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fwd()
        e1.record()
        torch.cuda.synchronize()
        f_ms = e0.elapsed_time(e1) / 5
        tot[2] += f_ms
        del out
        for _ in range(2):
            ops.gn_act_backward(dout, 1, G, *args, dx_dtype=dt)
        torch.cuda.synchronize()
        ops.profiler.enable(None, tags=False)
        for _ in range(5):
            ops.gn_act_backward(dout, 1, G, *args, dx_dtype=dt)
        s = ops.profiler.summary()
        ops.profiler.disable()
        r, ap_ = s["gn_bwd_reduce"]["ms_total"] / 5, s["gn_bwd_apply"]["ms_total"] / 5
        tot[0] += r
        tot[1] += ap_
        nel = float(B) * L * C * 2
        line += "   forward %6.3f ms (%5.0f GB/s)" % (f_ms, float(B) * L * C * 2 * (2 + two) / f_ms / 1e6)
        line += "   reduce %6.3f ms (%5.0f GB/s)  apply %6.3f ms (%5.0f GB/s)" % (r, nel * (2 + two) / r / 1e6, ap_, nel * (3 + 2 * two) / ap_ / 1e6)
        print(line, flush=True)
        del x1, x2, dout
    print("all nodes: forward %.3f ms; backward reduce %.3f + apply %.3f = %.3f ms" % (tot[2], tot[0], tot[1], tot[0] + tot[1]))


if __name__ == "__main__":
    main()
