"""call sites of the small torch kernels (copies, fills, cats) in one inference pass of the path"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from sincformer_metacog_speech_enhancement_amd import ops, synthetic as syn
path, _ = bench.build_path("bf16")
path = path.cuda().eval()
wave = torch.from_numpy(syn.synth_wave(8, 16000, 1)[0]).cuda()
with torch.no_grad():
    path(wave)
counts = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "sincformer_metacog" in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"


def wrap(obj, name, tag):
    orig = getattr(obj, name)

    def f(*a, **k):
        counts[(tag, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


for n in ("zeros", "cat", "stack", "ones", "zeros_like", "empty_like", "flip", "arange", "tensor", "where"):
    wrap(torch, n, n)
for n in ("contiguous", "clone", "copy_", "to", "float", "repeat_interleave", "sum", "double", "mean", "reshape", "__getitem__", "__setitem__", "mul_", "add_", "zero_", "fill_"):
    wrap(torch.Tensor, n, "T." + n)
with torch.no_grad():
    path(wave)
torch.cuda.synchronize()
for (tag, s), n in counts.most_common(45):
    print("%5d  %-20s %s" % (n, tag, s))
