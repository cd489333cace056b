#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sincformer_metacog_speech_enhancement_amd import ops
ops.set_compute_dtype("f16")
dt = torch.float16
g = torch.Generator(device="cuda").manual_seed(0)
R = lambda *s: torch.randn(*s, device="cuda", generator=g)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cin, cout, k, s, p, L = 128, 128, 3, 1, 1, int(sys.argv[2]) if len(sys.argv) > 2 else 32000
x1 = R(B, L, cin).to(dt)
sc1, sh1 = R(B, cin) * 0.1 + 1, R(B, cin) * 0.1
pw = ops.pack_linear(R(cout, cin, k) / (cin * k) ** 0.5, R(cout))
Lout = L
P = 2 * ((Lout + 127) // 128)
outs = []
for it in range(4):
    out = torch.empty(B, Lout, cout, device="cuda", dtype=dt)
    part = torch.zeros(B, P, 16, 2, device="cuda")
    ops.conv16p(x1, sc1, sh1, pw, out, B=B, Lin=L, stride=s, pad=p, gn_partial=part, gn_group=cout // 16)
    torch.cuda.synchronize()
    outs.append((out.float().cpu(), part.cpu()))
xn = torch.nn.functional.gelu(x1.float() * sc1[:, None, :] + sh1[:, None, :]).to(dt).float().transpose(1, 2)
ref = (torch.nn.functional.conv1d(xn, pw.w.float().reshape(cout, k, cin).permute(0, 2, 1).contiguous(), pw.bias, padding=p).transpose(1, 2)).cpu()
for it in range(4):
    o, pt = outs[it]
    bad = ((o - ref).abs() > 0.05) | ~torch.isfinite(o)
    nb = int(bad.sum())
    print("run %d: out vs ref: bad elements %d, nan %d" % (it, nb, int((~torch.isfinite(o)).sum())))
    if nb:
        idx = bad.nonzero()
        rows = idx[:, 1]
        print("   batches", sorted(set(idx[:, 0].tolist()))[:10], "rows%128 hist", torch.bincount(rows % 128, minlength=128).nonzero().flatten().tolist()[:40],
              "tiles", sorted(set((rows // 128).tolist()))[:20], "cols", sorted(set((idx[:, 2] // 8).tolist()))[:20])
G = 16
rg = ref.double().reshape(B, Lout // 64 if Lout % 64 == 0 else -1, 64, G, cout // G) if Lout % 64 == 0 else None
for it in range(4):
    pt = outs[it][1].double()
    d01 = (pt - outs[0][1].double()).abs()
    msg = "run %d: part vs run 0: max diff %.3e at %s" % (it, float(d01.max()), (d01 == d01.max()).nonzero()[0].tolist())
    if rg is not None:
        want = torch.stack([rg.sum(dim=(2, 4)), (rg ** 2).sum(dim=(2, 4))], dim=-1)        # [B, P, G, 2]
        e = (pt[:, :want.shape[1]] - want).abs()
        msg += "; vs ref max err %.3e (bad slots %d of %d)" % (float(e.max()), int((e > 1.0).sum()), e.numel())
        if int((e > 1.0).sum()):
            bad = (e > 1.0).nonzero()
            msg += " first bad %s: got %s want %s" % (bad[0].tolist(), pt[tuple(bad[0][:3])].tolist(), want[tuple(bad[0][:3])].tolist())
    print(msg)
if rg is not None:
    pt = outs[1][1].double()
    want = torch.stack([rg.sum(dim=(2, 4)), (rg ** 2).sum(dim=(2, 4))], dim=-1)
    e = (pt[:, :want.shape[1]] - want).abs()
    bad = (e > 1.0).nonzero()
    for bi in bad[:8]:
        b_, s_, g_, _ = bi.tolist()
        rows = (rg[b_, s_, :, g_, :] ** 2).sum(dim=-1)                    # [64] per-row sum of squares of this group
        got, tot = float(pt[b_, s_, g_, 1]), float(rows.sum())
        miss = tot - got
        per_pass = rows.reshape(8, 8).sum(dim=1)
        # which row subsets explain the deficit?
        cand = [(float(abs(miss - per_pass[p_])), "pass %d" % p_) for p_ in range(8)]
        cand += [(float(abs(miss - rows[r_])), "row %d" % r_) for r_ in range(64)]
        for r8 in range(8):
            cand.append((float(abs(miss - rows[r8::8].sum())), "rows = %d mod 8" % r8))
        cand.sort()
        print("slot", (b_, s_, g_), "got %.3f total %.3f missing %.3f best explanations:" % (got, tot, miss), cand[:3])
