"""Which host call sites issue the small torch kernels of a training step (fills, adds, copies)?  Counts calls of
torch.zeros / zeros_like / Tensor.to / contiguous / clone / float per source line over one c2t step."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
from sincformer_metacog_speech_enhancement_amd.optim import FlatAdamW
from sincformer_metacog_speech_enhancement_amd.training.conformer_pipeline import compute_path_loss

ops.set_compute_dtype("bf16")
model, _ = bench.build_path("bf16", seed=4321)
model.cuda().train()
opt = FlatAdamW(model.parameters(), lr=5e-4, betas=(0.9, 0.98), weight_decay=0.01, max_norm=5.0)
noisy, clean = syn.synth_wave(8, 16000, 1234)
noisy, clean = torch.from_numpy(noisy).cuda(), torch.from_numpy(clean).cuda()


def step():
    opt.zero_grad()
    total, _ = compute_path_loss(model, noisy, clean)
    total.backward()
    opt.step(loss=total)


step()
counts = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "sincformer_metacog" in fr.filename and "count_torch_calls" not in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"


def wrap(obj, name, tag):
    orig = getattr(obj, name)

    def f(*a, **k):
        counts[(tag, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


wrap(torch, "zeros", "zeros")
wrap(torch, "zeros_like", "zeros_like")
wrap(torch, "cat", "cat")
wrap(torch, "stack", "stack")
wrap(torch, "ones", "ones")
step()
torch.cuda.synchronize()
for (tag, s), n in counts.most_common(40):
    print("%5d  %-12s %s" % (n, tag, s))
print("total", sum(counts.values()))
