"""Data-parallel plumbing for the path (new functionality — the reference
is single-process): one process per GPU, utterances sharded over ranks, ONE exchange per
training step = all-reduce(sum)/world of a flat fp32 gradient buffer over RCCL/xGMI
(backend "nccl" on ROCm), issued in buckets so early buckets overlap the rest of the
backward; a 1-element MAX all-reduce keeps the reference's NaN/Inf step-skip
(training/conformer_pipeline.py:509) rank-consistent; the global-norm clip (:514) is
computed on the already reduced gradients, so it needs no further collective.
Inference shards utterances and needs no collective at all.
"""
import math
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """contiguous, balanced [start, end) of `n_items` utterances for `rank`."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class FlatGradSynchronizer:
    def __init__(self, params, bucket_bytes=16 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:                      # gradients become views of the flat buffer
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        per = max(1, bucket_bytes // 4)
        self.buckets = [(s, min(s + per, n)) for s in range(0, n, per)]
        self.flag = torch.zeros(1, device=dev, dtype=torch.float32)

    def zero(self):
        self.flat.zero_()

    def sync(self, loss=None, max_norm=None):
        """all-reduce the gradients (mean over ranks).  Returns dict(skip, grad_norm, clip_coef)."""
        bad = 0.0
        if loss is not None and not bool(torch.isfinite(loss.detach()).all()):
            bad = 1.0
        self.flag.fill_(bad)
        works = []
        if self.world > 1:
            for s, e in self.buckets:
                works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            works.append(dist.all_reduce(self.flag, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
            for w in works:
                w.wait()
            self.flat.div_(self.world)
        skip = bool(self.flag.item() > 0)
        norm = float(torch.linalg.vector_norm(self.flat))
        coef = 1.0
        if not math.isfinite(norm):
            skip = True
        elif max_norm is not None and norm > max_norm:
            coef = max_norm / (norm + 1e-6)
            self.flat.mul_(coef)
        return {"skip": skip, "grad_norm": norm, "clip_coef": coef}
