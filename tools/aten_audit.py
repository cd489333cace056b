"""Which Python lines issue the aten ops of one training step, by bytes touched (TorchDispatchMode + the Python stack; ops issued
by the autograd engine itself - gradient accumulation, slice / view backward - have no frame in this package and show as <autograd>):
    python tools/aten_audit.py [--workload c3t] [--batch 64]"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3t")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--top", type=int, default=70)
    a = ap.parse_args()
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode
    import bench
    step, opt, sd, B, L, T, desc, whole = bench.build_train_step(a.workload, "bf16", 0, a.batch)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0])
    skip = ("aten.view", "aten.detach", "aten._unsafe_view", "aten.transpose", "aten.t.", "aten.slice.", "aten.select", "aten.alias",
            "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.permute", "aten.reshape", "aten.empty",
            "aten.narrow", "aten.unbind", "aten.split", "aten._local_scalar", "aten.is_", "aten.stride", "aten.size")

    def nbytes(x):
        if isinstance(x, torch.Tensor):
            return x.numel() * x.element_size() if x.is_cuda else 0
        if isinstance(x, (list, tuple)):
            return sum(nbytes(y) for y in x)
        return 0

    class Audit(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            if any(name.startswith(s) for s in skip):
                return out
            by = nbytes(args) + nbytes(out)
            if by == 0:
                return out
            where = "<autograd>"
            for fr in reversed(traceback.extract_stack(limit=40)):
                fn = fr.filename
                if ("sincformer_metacog_speech_enhancement_amd" in fn or fn.endswith("bench.py")) and "aten_audit" not in fn:
                    where = "%s:%d %s" % (os.path.relpath(fn, ROOT), fr.lineno, fr.name)
                    break
            agg[(name, where)][0] += by
            agg[(name, where)][1] += 1
            return out

    with Audit():
        step()
    torch.cuda.synchronize()
    total = sum(v[0] for v in agg.values())
    print("aten ops in one step (B %d): %.1f MB touched (%.2f ms at 4 TB/s)" % (B, total / 1e6, total / 4e9))
    for (name, where), (by, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print("%9.1f MB  n %4d  %-26s %s" % (by / 1e6, n, name.replace("aten.", ""), where))


if __name__ == "__main__":
    main()
