#!/usr/bin/env python3
"""Diagnostic: run kernels twice on identical inputs and compare bit for bit (races show up as differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops

ops.set_compute_dtype("f16")
dt = torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)


def twice(name, fn):
    outs = []
    for _ in range(3):
        o = fn()
        torch.cuda.synchronize()
        outs.append([t.clone() for t in o])
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])) and all(torch.equal(a, b) for a, b in zip(outs[0], outs[2]))
    d = max(float((a.float() - b.float()).abs().max()) for a, b in zip(outs[0], outs[1]))
    print("%-50s %s  (max diff %.3e)" % (name, "deterministic" if same else "DIFFERS", d))


B = 8
for (cin, cout, k, s, p, two, skip, L) in [(64, 128, 7, 2, 3, False, True, 64000), (128, 128, 7, 2, 3, True, True, 32000),
                                           (128, 256, 7, 2, 3, True, False, 16000), (128, 256, 1, 2, 0, True, False, 16000),
                                           (128, 128, 3, 1, 1, False, False, 32000), (256, 256, 3, 1, 1, False, False, 8000),
                                           (256, 256, 5, 2, 2, True, False, 8000)]:
    x1, x2 = R(B, L, cin).to(dt), (R(B, L, cin).to(dt) if two else None)
    sc1, sh1 = R(B, cin) * 0.1 + 1, R(B, cin) * 0.1
    sc2, sh2 = (R(B, cin) * 0.1 + 1, R(B, cin) * 0.1) if two else (None, None)
    pw = ops.pack_linear(R(cout, cin, k) / (cin * k) ** 0.5, R(cout))
    spw = ops.pack_linear(R(cout, cin, 1) / cin ** 0.5, R(cout)) if skip else None
    Lout = (L + 2 * p - k) // s + 1
    P = 2 * ((Lout + 127) // 128)

    def run():
        out = torch.empty(B, Lout, cout, device="cuda", dtype=dt)
        part = torch.zeros(B, P, 16, 2, device="cuda")
        outs = torch.empty(B, Lout, cout, device="cuda", dtype=dt) if skip else None
        parts = torch.zeros(B, P, 16, 2, device="cuda") if skip else None
        ops.conv16p(x1, sc1, sh1, pw, out, B=B, Lin=L, stride=s, pad=p, x2=x2, sc2=sc2, sh2=sh2, gn_partial=part, gn_group=cout // 16,
                    skip_pw=spw, out_s=outs, gn_partial_s=parts)
        return [out, part] + ([outs, parts] if skip else [])
    twice("conv16p cin%d cout%d k%d s%d two%d skip%d" % (cin, cout, k, s, two, skip), run)

for (M, K, N, epi, odt, gn) in [(32000, 256, 512, ops.EPI_NONE, torch.float32, True), (51264, 256, 768, ops.EPI_NONE, dt, False),
                                (51264, 256, 256, ops.EPI_RESID, torch.float32, False), (51264, 256, 512, ops.EPI_GLU, dt, False),
                                (51264, 128, 129, ops.EPI_NONE, torch.float32, False), (51264, 256, 1024, ops.EPI_NONE, torch.float32, False)]:
    x = R(M, K).to(dt)
    pw = ops.pack_linear(R(N, K) / K ** 0.5, R(N), glu=(epi == ops.EPI_GLU))
    res = R(M, pw.N) if epi == ops.EPI_RESID else None

    def run():
        if gn:
            out = torch.empty(1, M, N, device="cuda", dtype=odt)
            P = 2 * ((M + 127) // 128)
            part = torch.zeros(1, P, 32, 2, device="cuda")
            ops.gemm16(x, pw, out, B=1, Lout=M, Lin=M, a_batch_stride=0, ldo=N, o_batch_stride=0, gn_partial=part, gn_group=N // 32)
            return [out, part]
        return [ops.linear16(x, pw, epi=epi, out_dtype=odt, resid=res)]
    twice("gemm16 M%d K%d N%d epi%d" % (M, K, N, epi), run)
