#!/usr/bin/env python3
"""time the lin256 kernel of the library named by SFM_LIB_PATH (diagnostic builds: tools/variant_lib.sh lin256 <tag> -DSFM_L2_ABL=n)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype(torch.float16)
g = torch.Generator(device="cuda").manual_seed(0)
M = int(os.environ.get("AB_M", 131072))
x = (torch.randn(M, 256, device="cuda", generator=g)).half()
r = {}
for name, N, glu, odt in (("qkv", 768, False, torch.bfloat16), ("pw1_glu", 512, True, torch.float16)):
    w = torch.randn(N, 256, device="cuda", generator=g) / 16
    pw = ops.pack_linear(w, torch.randn(N, device="cuda", generator=g), glu=glu)
    epi = ops.EPI_GLU if glu else ops.EPI_NONE
    out = torch.empty(M, pw.N, device="cuda", dtype=odt)
    ts = []
    for rnd in range(3):
        for _ in range(3): ops.linear16(x, pw, epi=epi, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.linear16(x, pw, epi=epi, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    r[name] = round(sorted(ts)[1] * 1e3, 1)
print(os.environ.get("AB_TAG", "?"), json.dumps(r))
