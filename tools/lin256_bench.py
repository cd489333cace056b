#!/usr/bin/env python3
"""K = 256 linears with a 16-bit result: csrc/lin256.hip against sfm_gemm16 (gemm16v2) on the same pack, interleaved, at the row
counts of the bench workloads (M = 51 264: B 64 x 801 frames; M = 205 056: B 256 x 801; M = 131 072: B 256 x 512)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype(torch.float16)
g = torch.Generator(device="cuda").manual_seed(0)
for M in (51264, 131072, 205056):
    x = (torch.randn(M, 256, device="cuda", generator=g)).half()
    for name, N, glu, odt in (("qkv", 768, False, torch.bfloat16), ("pw1_glu", 512, True, torch.float16), ("n256", 256, False, torch.float16), ("lstm_in_f32", 1024, False, torch.float32)):
        w = torch.randn(N, 256, device="cuda", generator=g) / 16
        b = torch.randn(N, device="cuda", generator=g)
        pw = ops.pack_linear(w, b, glu=glu)
        epi = ops.EPI_GLU if glu else ops.EPI_NONE
        out = torch.empty(M, pw.N, device="cuda", dtype=odt)
        res = {}
        for rnd in range(3):
            for on in (False, True):
                ops.set_lin256(on)
                run = (lambda: ops.lin256(x, pw, out)) if (on and odt == torch.float32) else (lambda: ops.linear16(x, pw, epi=epi, out=out))
                for _ in range(3): run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): run()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(on, []).append(e0.elapsed_time(e1) / 20)
        a, c = sorted(res[False])[1], sorted(res[True])[1]
        fl = 2.0 * M * N * 256
        by = M * 512 + M * pw.N * out.element_size()
        print(json.dumps({"M": M, "what": name, "gemm16_us": round(a * 1e3, 1), "lin256_us": round(c * 1e3, 1),
                          "lin256_TFLOPs": round(fl / c / 1e9, 1), "lin256_GBs": round(by / c / 1e6, 1)}))
ops.set_lin256(True)
