"""Hunt for rare run-to-run differences in the PerceptionAgent's fused inference path: N passes over the same batch, every kernel
output of a pass reduced to per-utterance checksums (and the GroupNorm partial tensors kept whole), compared with the first pass.
Prints the first stage of a pass that differs and which slots / utterances.
    python tools/pa_determinism_probe.py [passes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import synth_sd
from sincformer_metacog_speech_enhancement_amd import ops, functional as Fn, synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B, L = 64, 64000
ops.reset_precision()
pk = Fn.pack_perception({k: v.cuda() for k, v in synth_sd("PerceptionAgent", 291, sinc_scale=2000.0).items()}, 16000)
noisy, _ = syn.synth_wave(B, L, 295)
wave = torch.from_numpy(noisy).cuda()
log = []
def rec(name, t, whole=False):
    if t is None:
        return
    if whole:
        log.append((name, t.detach().clone()))
    else:
        x = t.detach().reshape(t.shape[0], -1)
        # exact integer checksum of the bit patterns per utterance (order independent, no rounding)
        bits = x.view(torch.int16 if x.element_size() == 2 else torch.int32).to(torch.int64)
        log.append((name, bits.sum(dim=1)))
orig = {n: getattr(ops, n) for n in ("sinc_fir16", "conv16p", "gn_finalize", "gemm16", "gn_apply")}
def w_sinc(wave_, filt, out, *a, **k):
    r = orig["sinc_fir16"](wave_, filt, out, *a, **k); rec("sinc_fir16.raw", out); rec("sinc_fir16.part", r[0], True); return r
FULL = os.environ.get("PROBE_FULL", "")            # e.g. "k3 N128": keep that stage's whole output and print where it differs
def w_conv(x1, sc1, sh1, pw, out, **k):
    r = orig["conv16p"](x1, sc1, sh1, pw, out, **k)
    if FULL and FULL == "k%d N%d" % (pw.ksize, pw.N) and not any(n == "FULL" for n, _ in log):
        log.append(("FULL", out.detach().clone()))
    rec("conv16p.raw k%d N%d" % (pw.ksize, pw.N), out); rec("conv16p.part", k.get("gn_partial"), True)
    rec("conv16p.raw_skip", k.get("out_s")); rec("conv16p.part_skip", k.get("gn_partial_s"), True); return r
def w_fin(part, *a, **k):
    r = orig["gn_finalize"](part, *a, **k); rec("gn_finalize.sc", r[0], True); rec("gn_finalize.sh", r[1], True); return r
def w_gemm(A, pw, out, **k):
    r = orig["gemm16"](A, pw, out, **k); rec("gemm16.out N%d" % pw.N, out.reshape(k["B"], -1)); rec("gemm16.part", k.get("gn_partial"), True); return r
def w_apply(raw, sc, sh, out, *a, **k):
    r = orig["gn_apply"](raw, sc, sh, out, *a, **k); rec("gn_apply.out", out); return r
ops.sinc_fir16, ops.conv16p, ops.gn_finalize, ops.gemm16, ops.gn_apply = w_sinc, w_conv, w_fin, w_gemm, w_apply
ref = None
nbad = 0
with torch.no_grad():
    for p in range(N):
        log.clear()
        Fn.perception_forward(wave, pk, latents=False)
        torch.cuda.synchronize()
        cur = [(n, t.clone()) for n, t in log]
        if ref is None:
            ref = cur
            print("stages per pass:", [n for n, _ in ref])
            continue
        for (n, a), (_, b) in zip(cur, ref):
            if not torch.equal(a, b):
                nbad += 1
                d = (a != b)
                idx = d.nonzero()
                print("pass %d: FIRST difference at stage '%s' (shape %s): %d entries differ; first indices %s" %
                      (p, n, tuple(a.shape), int(d.sum()), idx[:6].tolist()))
                if n == "FULL":
                    bb, ll, cc = idx[:, 0], idx[:, 1], idx[:, 2]
                    print("    utterances %s rows %d..%d (tiles of 128: %s) channels %d..%d; differing rows: %d" %
                          (sorted(set(bb.tolist())), int(ll.min()), int(ll.max()), sorted(set((ll // 128).tolist()))[:10], int(cc.min()), int(cc.max()),
                           len(set(ll.tolist()))))
                    rows = sorted(set(ll.tolist()))[:3]
                    for r_ in rows:
                        print("      row", r_, "got", a[int(bb[0]), r_, :8].float().tolist(), "ref", b[int(bb[0]), r_, :8].float().tolist())
                if a.dtype.is_floating_point and n != "FULL":
                    for i in idx[:6].tolist():
                        print("    ", i, "got %.9g ref %.9g" % (float(a[tuple(i)]), float(b[tuple(i)])))
                break
print("passes with a difference: %d of %d" % (nbad, N - 1))
