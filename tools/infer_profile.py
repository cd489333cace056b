"""Per-shape kernel breakdown of ONE inference pass of the bench workload (HIP events around every launch, the pass alone on the
device): which GEMM shapes make up the `gemm16` family of `bench.py`'s breakdown.
    python tools/infer_profile.py [--workload c2] [--dtype mixed]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--dtype", default="mixed")
    a = ap.parse_args()
    import torch
    import bench
    from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
    B, L, desc = bench.WORKLOADS[a.workload]
    path, _ = bench.build_path(a.dtype, seed=4321)
    path = path.cuda().eval()
    noisy, _ = syn.synth_wave(B, L, 1234)
    wave = torch.from_numpy(noisy).cuda()
    with torch.no_grad():
        for _ in range(3):
            path(wave)
        torch.cuda.synchronize()
        ops.profiler.enable(None, tags=True)
        path(wave)
        summ = ops.profiler.summary()
        ops.profiler.disable()
    fam = {k: v for k, v in summ.items() if "[" not in k}
    tagged = {k: v for k, v in summ.items() if "[" in k}
    print("%s: one pass, kernels sum to %.3f ms (%d launches)" % (desc, sum(v["ms_total"] for v in fam.values()), sum(v["n"] for v in fam.values())))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_total"]):
        print("  %-24s n %4d  %8.3f ms" % (k, v["n"], v["ms_total"]))
    print("by shape:")
    for k, v in sorted(tagged.items(), key=lambda kv: -kv[1]["ms_total"])[:40]:
        tf = v["flops"] / max(v["ms_total"], 1e-9) / 1e9
        gb = v["bytes"] / max(v["ms_total"], 1e-9) / 1e6
        print("  %-64s n %3d  %8.3f ms  %6.1f us each  %7.1f TF/s %7.0f GB/s" % (k, v["n"], v["ms_total"], 1e3 * v["ms_total"] / v["n"], tf, gb))


if __name__ == "__main__":
    main()
