"""Where does the HOST spend the enqueue time of one training step (cProfile over 3 steps, top functions by own time and by
cumulative time)?  c2t (B 64) is bound by it: 500 launches, 50-56 ms.
    python tools/host_profile.py [--workload c2t] [--batch B]"""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2t")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    import torch
    import bench
    from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
    from sincformer_metacog_speech_enhancement_amd.optim import FlatAdamW
    from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import compute_path_loss
    ops.set_compute_dtype("bf16")
    B, L, _ = bench.WORKLOADS[a.workload]
    B = a.batch or B
    model, _ = bench.build_path("bf16", seed=4321)
    model.cuda().train()
    opt = FlatAdamW(model.parameters(), lr=5e-4, betas=(0.9, 0.98), weight_decay=0.01, max_norm=5.0)
    noisy, clean = syn.synth_wave(B, L, 1234)
    noisy, clean = torch.from_numpy(noisy).cuda(), torch.from_numpy(clean).cuda()

    def step():
        opt.zero_grad()
        total, _ = compute_path_loss(model, noisy, clean)
        total.backward()
        opt.step(loss=total)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        print("==== by %s (3 steps)" % key)
        st = pstats.Stats(pr, stream=sys.stdout)
        st.strip_dirs().sort_stats(key).print_stats(a.top)


if __name__ == "__main__":
    main()
