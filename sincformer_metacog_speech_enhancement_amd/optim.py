"""AdamW + global-norm clip + NaN/Inf step-skip of the reference's training loop
(training/conformer_pipeline.py:424-429, 509, 514) as three HIP launches on flat buffers (csrc/optim.hip),
with no host synchronisation: the clip coefficient, the skip decision and the bias corrections are computed
on the device.  Data parallel: pass a dp.FlatGradSynchronizer; its buffer holds the SUM over ranks and 1/world
is folded into the unscale factor."""
import os
import torch

from . import ops, dp as _dp


class FlatAdamW:
    def __init__(self, params, lr=5e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, max_norm=5.0, sync=None,
                 bucket_bytes=16 << 20, overlap=True, steal_grads=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not self.params[0].is_cuda:
            raise RuntimeError("FlatAdamW: parameters must live on the MI355X (no CPU fallback)")
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        if steal_grads is None:
            steal_grads = os.environ.get("SFM_STEAL_GRADS", "1") != "0"          # A/B knob
        # steal_grads: autograd keeps the gradient tensors of the backward nodes and the synchronizer gathers them into the flat
        # buffer with a few cat launches (dp.FlatGradSynchronizer) instead of one aten add per parameter; `p.grad` is then None
        # after zero_grad() (torch's set_to_none) - code that writes gradients into the flat views by hand passes False
        self.sync = sync or _dp.FlatGradSynchronizer(self.params, bucket_bytes=bucket_bytes, overlap=overlap, steal=steal_grads)
        dev = self.params[0].device
        n = self.sync.n
        self.flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:                     # parameters become views of one flat buffer
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().float().reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)
                off += k
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.ctl = torch.zeros(8, device=dev, dtype=torch.float64)
        offs = [0]
        for _, k in self.sync._spans:
            offs.append(offs[-1] + k)
        self.spans = torch.tensor(offs, device=dev, dtype=torch.int64)

    def zero_grad(self):
        self.sync.zero()

    def step(self, loss=None, grad_scale=1.0, lr=None):
        """exchange (if data parallel) -> ||g|| -> clip/skip decision -> AdamW, all enqueued on the current stream."""
        self.sync.finish(loss)
        L = ops._lib.load()
        n = self.sync.n
        ops._call("optim", L.sfm_sumsq, (ops._p(self.sync.flat), n, self.ctl[1:].data_ptr(), ops._stream()), 0.0, 4.0 * n)
        self.ctl[2:3].copy_(self.sync.flag.double())
        inv = 1.0 / (float(grad_scale) * self.sync.world)
        # parameters no rank's backward pass reached in this step (`p.grad is None` for torch.optim.AdamW: no decay, no update,
        # moments unchanged) are skipped ON THE DEVICE from the mask that was all-reduced with the gradients
        # (dp.FlatGradSynchronizer): p, m and v stay as they are, and every replica takes the same decision.  (One global step
        # count feeds the bias corrections; torch keeps one per parameter, which only differs for a parameter that is
        # trained in some steps and idle in others.)
        ops._call("optim", L.sfm_adamw_step_masked,
                  (ops._p(self.flat_p), ops._p(self.sync.flat), ops._p(self.m), ops._p(self.v), n, ops._p(self.ctl),
                   float(lr if lr is not None else self.lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, inv,
                   float(self.max_norm or 0.0), 0, ops._p(self.spans), ops._p(self.sync.touched_dev), len(self.sync.params),
                   ops._stream()), 0.0, 28.0 * n)
        self._bump_versions()                         # packed-weight caches key on the parameters' version counter

    def _bump_versions(self):
        # The parameters were rebound to views of flat_p (`p.data = ...`), which gives each of them its OWN version
        # counter: an in-place op on flat_p does not bump it.  The packed-weight caches therefore also key on a global
        # weights generation (ops.policy_key), advanced here; the in-place no-op on ONE element keeps flat_p's own counter
        # honest (views share their base's version counter).
        self.flat_p[:1].add_(0.0)
        ops.bump_weights_generation()

    def stats(self):
        """host sync: {'step', 'grad_norm', 'skipped'} of the last step."""
        c = self.ctl.cpu()
        return {"step": int(c[0]), "grad_norm": float(c[7]), "skipped": bool(c[4] > 0)}
