"""Per-shape kernel breakdown of one training step (HIP events around every launch, grouped by family and shape tag) and the
host-side gap: wall time of a step vs the sum of the instrumented kernels.
    python tools/train_profile.py [--workload c2t|c3se] [--batch B] [--out file.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2t")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--out", default=None)
    ap.add_argument("--families", default="", help="comma list: print EVERY tagged shape of these kernel families (e.g. convert_rows)")
    a = ap.parse_args()
    import torch
    import bench
    from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
    from sincformer_metacog_speech_enhancement_amd.optim import FlatAdamW
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import (SpeechEnhancer, batch_stft, compute_loss,
                                                                                      compute_path_loss)
    ops.set_compute_dtype(a.dtype)
    B, L, desc = bench.WORKLOADS[a.workload]
    B = a.batch or B
    if a.workload in ("c2t", "c3t"):
        model, _ = bench.build_path(a.dtype, seed=4321)
    else:
        model = SpeechEnhancer(n_freq=129, d_model=256, num_blocks=4, num_heads=4, d_ff=1024, kernel_size=31, dropout=0.15)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(shapes, 4321).items()})
    model.cuda().train()
    opt = FlatAdamW(model.parameters(), lr=5e-4, betas=(0.9, 0.98), weight_decay=0.01, max_norm=5.0)
    noisy, clean = syn.synth_wave(B, L, 1234)
    noisy, clean = torch.from_numpy(noisy).cuda(), torch.from_numpy(clean).cuda()

    def step():
        opt.zero_grad()
        if a.workload in ("c2t", "c3t"):
            total, _ = compute_path_loss(model, noisy, clean)
        else:
            total, _ = compute_loss(model, *batch_stft(noisy, 256, 80, 160), clean, *batch_stft(clean, 256, 80, 160))
        total.backward()
        opt.step(loss=total)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3 * 1e3
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    host = (time.perf_counter() - t0) / 3 * 1e3          # enqueue time only (no sync): is the step launch-bound?
    torch.cuda.synchronize()
    ops.profiler.enable(None, tags=True)
    step()
    summ = ops.profiler.summary()
    ops.profiler.disable()
    fam = {k: v for k, v in summ.items() if "[" not in k}
    tagged = {k: v for k, v in summ.items() if "[" in k}
    ksum = sum(v["ms_total"] for v in fam.values())
    nlaunch = sum(v["n"] for v in fam.values())
    print("%s batch %d: wall %.2f ms/step, host enqueue %.2f ms/step, instrumented kernels %.2f ms (%d launches)" %
          (a.workload, B, wall, host, ksum, nlaunch))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms_total"]):
        print("  %-24s n %4d  %8.3f ms" % (k, v["n"], v["ms_total"]))
    print("by shape:")
    for k, v in sorted(tagged.items(), key=lambda kv: -kv[1]["ms_total"])[:40]:
        tf = v["flops"] / max(v["ms_total"], 1e-9) / 1e9
        gb = v["bytes"] / max(v["ms_total"], 1e-9) / 1e6
        print("  %-64s n %3d  %8.3f ms  %7.1f TF/s %7.0f GB/s" % (k, v["n"], v["ms_total"], tf, gb))
    for famname in [f for f in a.families.split(",") if f]:
        print("all shapes of %s:" % famname)
        for k, v in sorted(tagged.items(), key=lambda kv: -kv[1]["ms_total"]):
            if k.startswith(famname + "["):
                print("  %-64s n %3d  %8.3f ms  %7.0f GB/s" % (k, v["n"], v["ms_total"], v["bytes"] / max(v["ms_total"], 1e-9) / 1e6))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump({"wall_ms": wall, "host_enqueue_ms": host, "kernel_ms": ksum, "launches": nlaunch, "families": fam,
                       "shapes": tagged}, fh, indent=1)


if __name__ == "__main__":
    main()
