"""AdamW + global-norm clip + NaN/Inf step-skip of the reference's training loop
(training/conformer_pipeline.py:424-429, 509, 514) as three HIP launches on flat buffers (csrc/optim.hip),
with no host synchronisation: the clip coefficient, the skip decision and the bias corrections are computed
on the device.  Data parallel: pass a dp.FlatGradSynchronizer; its buffer holds the SUM over ranks and 1/world
is folded into the unscale factor.  DynamicLossScale = torch.amp.GradScaler of the reference's AMP branch
(:442, 504, 512-517) with the scale, its growth / backoff and the Inf check kept on the device."""
import os
import torch

from . import ops, dp as _dp


class DynamicLossScale:
    """torch.amp.GradScaler('cuda') of training/conformer_pipeline.py:442 under its own method names - scale(loss), unscale_,
    step(optimizer), update() - for optim.FlatAdamW.  The scale S is a device float: scale() multiplies the loss by it (so the
    whole backward pass carries S: the 16-bit gradient tensors of an fp16 step stay above the subnormal range), FlatAdamW's
    1-thread prepare kernel divides it out again, skips the step when a gradient is Inf / NaN, halves S after such a step and
    doubles it after `growth_interval` clean ones (csrc/optim.hip, adamw_prepare_scaled_kernel) - GradScaler's defaults, no host
    synchronisation anywhere.  unscale_ and update() therefore have nothing left to do and exist for the reference's call order."""

    def __init__(self, device=None, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.enabled = bool(enabled)
        self._init = float(init_scale) if enabled else 1.0
        self.state = None
        if device is not None:
            self._alloc(torch.device(device))

    def _alloc(self, device):
        if self.state is None:
            if device.type != "cuda":
                raise RuntimeError("DynamicLossScale lives on the MI355X (no CPU fallback)")
            self.state = torch.tensor([self._init, 0.0, 0.0, 0.0], device=device, dtype=torch.float32)
        return self.state

    def scale(self, loss):
        if not self.enabled:
            return loss
        return loss * self._alloc(loss.device)[0]

    def unscale_(self, optimizer):
        """folded into FlatAdamW.step (the unscale factor is applied by the prepare kernel / the update itself)"""

    def step(self, optimizer, loss=None, lr=None):
        return optimizer.step(loss=loss, lr=lr, scaler=self)

    def update(self):
        """folded into FlatAdamW.step (growth / backoff happen in the same 1-thread kernel that takes the skip decision)"""

    def get_scale(self):
        """host sync"""
        return float(self.state[0]) if self.state is not None else self._init

    def stats(self):
        """host sync: {'scale', 'clean_steps', 'skipped_inf', 'skipped_loss'}"""
        c = (self.state.cpu() if self.state is not None else torch.tensor([self._init, 0.0, 0.0, 0.0])).tolist()
        return {"scale": c[0], "clean_steps": int(c[1]), "skipped_inf": int(c[2]), "skipped_loss": int(c[3])}

    def state_dict(self):
        s = self.stats()
        return {"scale": s["scale"], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": s["clean_steps"]}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self._init = float(sd["scale"])
        if self.state is not None:
            self.state.copy_(torch.tensor([self._init, float(sd.get("_growth_tracker", 0)), 0.0, 0.0]))


class FlatAdamW:
    def __init__(self, params, lr=5e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, max_norm=5.0, sync=None,
                 bucket_bytes=16 << 20, overlap=True, steal_grads=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not self.params[0].is_cuda:
            raise RuntimeError("FlatAdamW: parameters must live on the MI355X (no CPU fallback)")
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        if steal_grads is None:
            steal_grads = os.environ.get("SFM_STEAL_GRADS", "0") != "0"          # A/B knob
        # steal_grads (opt-in: its A/B, profiles/r03/steal_grads_ab.txt, is inside the run-to-run noise, and it changes what
        # `p.grad` is between backward and step): autograd keeps the gradient tensors of the backward nodes and the synchronizer
        # gathers them into the flat buffer with a few cat launches (dp.FlatGradSynchronizer) instead of one aten add per
        # parameter; `p.grad` is then None after zero_grad() (torch's set_to_none)
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

    def step(self, loss=None, grad_scale=1.0, lr=None, scaler=None):
        """exchange (if data parallel) -> ||g|| -> clip/skip decision -> AdamW, all enqueued on the current stream.
        scaler: a DynamicLossScale whose scale() multiplied the loss of this step (the reference's AMP branch); grad_scale: a
        static factor the caller multiplied the loss with."""
        self.sync.finish(loss)
        L = ops._lib.load()
        n = self.sync.n
        ops._call("optim", L.sfm_sumsq, (ops._p(self.sync.flat), n, self.ctl[1:].data_ptr(),
                                         ops._p(ops._ws(2048, self.flat_p.device, torch.float64)), ops._stream()), 0.0, 4.0 * n)
        self.ctl[2:3].copy_(self.sync.flag.double())
        inv = 1.0 / (float(grad_scale) * self.sync.world)
        if scaler is not None and scaler.enabled:
            ls = scaler._alloc(self.flat_p.device)
            ops._call("optim", L.sfm_adamw_step_scaled,
                      (ops._p(self.flat_p), ops._p(self.sync.flat), ops._p(self.m), ops._p(self.v), n, ops._p(self.ctl),
                       float(lr if lr is not None else self.lr), self.betas[0], self.betas[1], self.eps, self.weight_decay, inv,
                       float(self.max_norm or 0.0), 0, ops._p(self.spans), ops._p(self.sync.touched_dev), len(self.sync.params),
                       ops._p(ls), scaler.growth_factor, scaler.backoff_factor, scaler.growth_interval, ops._stream()),
                      0.0, 28.0 * n)
            self._bump_versions()
            return
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
