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
    """Flat fp32 gradient buffer, preceded by a header of 64 + n_params floats: slot 0 is the NaN/Inf flag, slots 64.. are
    the per-parameter "received a gradient in this step" mask (0 / 1 per rank; after the all-reduce: the number of ranks whose
    backward pass reached the parameter).  The header rides in bucket 0 (the first parameters, whose gradients are the LAST
    that backward produces) and costs no collective of its own; optim.FlatAdamW steps exactly the parameters whose reduced
    mask is > 0, so replicas whose graphs differ (data-dependent routing) still take the same decision.  Gradients of the parameters are views of the buffer; buckets are runs of whole
    parameters of about `bucket_bytes`.  With overlap=True every bucket is all-reduced (async, RCCL's own stream) as
    soon as autograd has accumulated the gradient of its last parameter, so the exchange overlaps the rest of the
    backward pass; finish() launches what is left and waits.

    steal=True: zero() sets every `p.grad` to None, so autograd KEEPS the tensor a backward node returns instead of adding it
    into the flat view (one aten add launch per parameter: 322 launches, 2 ms of GPU time per step of the whole-path
    training loop); the kept tensors are gathered into the flat buffer with one `torch.cat` per run of parameters - per bucket,
    right before the bucket goes out - and `p.grad` is bound to the flat views again (they hold the sum over ranks after
    finish(), as without steal)."""

    HEADER = 64                                   # fixed part of the header; the touched mask follows it

    def __init__(self, params, bucket_bytes=16 << 20, group=None, overlap=False, steal=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = sum(p.numel() for p in self.params)
        self.n = n
        dev = self.params[0].device
        H = self.HEADER + ((len(self.params) + 63) // 64) * 64
        self.header = H
        self.buf = torch.zeros(H + n, device=dev, dtype=torch.float32)
        self.flat = self.buf[H:]
        self.flag = self.buf[:1]
        self.touched_dev = self.buf[self.HEADER:self.HEADER + len(self.params)]
        self._mask_dev = torch.zeros(len(self.params), device=self.buf.device, dtype=torch.float32)
        self._mask_host = ()                          # nothing uploaded yet
        off = 0
        per = max(1, bucket_bytes // 4)
        self.buckets, self._bucket_of, start = [], {}, 0
        self.steal = bool(steal)
        self._views, self._bucket_params, first = [], [], 0
        for i, p in enumerate(self.params):        # gradients become views of the flat buffer
            self._views.append(self.flat[off:off + p.numel()].view_as(p))
            p.grad = self._views[-1]
            off += p.numel()
            self._bucket_of[id(p)] = len(self.buckets)
            if off - start >= per or i == len(self.params) - 1:
                self.buckets.append((start, off))
                self._bucket_params.append((first, i + 1))
                start, first = off, i + 1
        self.buckets = [(s + H, e + H) for s, e in self.buckets]
        self.buckets[0] = (0, self.buckets[0][1])  # the header (flag) travels with bucket 0
        self._members = [0] * len(self.buckets)
        for p in self.params:
            self._members[self._bucket_of[id(p)]] += 1
        self._pending = list(self._members)
        self._launched = [False] * len(self.buckets)
        self._next = len(self.buckets) - 1
        self._works = []
        self.overlap = overlap and self.world > 1
        # which parameters autograd accumulated into since zero(): a parameter off the loss's graph keeps its (zero) flat
        # gradient view but must not be stepped - torch.optim skips `p.grad is None` (optim.FlatAdamW.step)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._touched = [False] * len(self.params)
        self._spans, off = [], 0
        for p in self.params:
            self._spans.append((off, p.numel()))
            off += p.numel()
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._on_grad)
        # optional measurement (bench.py): HIP events around the wait in finish() = the time the compute stream is held up by
        # the exchange, i.e. the part of the all-reduce that did NOT overlap the backward pass
        self.time_exposed = False
        self._exposed_events = []

    # -- overlap machinery ---------------------------------------------------------------------------------------
    def _launch(self, b):
        s, e = self.buckets[b]
        self._launched[b] = True
        self._gather(b)
        self._works.append(dist.all_reduce(self.buf[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _gather(self, b):
        """steal mode: the gradients autograd kept for bucket b's parameters -> their spans of the flat buffer (one cat per run
        of consecutive parameters that have one; the spans of the others stay zero), then `p.grad` = the flat view again"""
        if not self.steal:
            return
        lo, hi = self._bucket_params[b]
        run, first = [], lo

        def flush():
            if run:
                a = self._spans[first][0]
                torch.cat(run, out=self.flat[a:a + sum(t.numel() for t in run)])

        with torch.no_grad():
            for i in range(lo, hi):
                p, v = self.params[i], self._views[i]
                g = p.grad
                if g is None or g.data_ptr() == v.data_ptr():          # idle, or written into the view by hand
                    flush()
                    run, first = [], i + 1
                else:
                    run.append(g.detach().reshape(-1) if g.dtype == torch.float32 else g.detach().float().reshape(-1))
                p.grad = v
            flush()

    def _on_grad(self, p):
        self._touched[self._index[id(p)]] = True
        if not self.overlap:
            return
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] < 0 or self._launched[b]:
            # a second backward() before step(): the buckets of the first one are already on the wire (or reduced in place), a
            # second micro-batch accumulated into them would be summed over ranks twice or not at all
            raise RuntimeError("FlatGradSynchronizer(overlap=True): a gradient arrived for a bucket that has already been "
                               "all-reduced in this step - gradient accumulation over several backward() calls needs overlap=False")
        # Buckets go out in ONE fixed order on every rank - last bucket first, bucket 0 (flag + touched mask) last, from
        # finish() - whatever the order in which their gradients complete: a rank whose graph never reaches some parameter
        # (data-dependent routing) simply stops launching from hooks at that bucket and finish() sends the rest, in the same
        # order.  (Launching "whichever bucket is complete" pairs different buckets across such ranks.)
        while self._next >= 1 and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next -= 1

    def untouched(self):
        """(offset, numel) in the flat buffers of every parameter that received no gradient since zero().  When autograd
        accumulated into none of them the gradients were written into the flat views by hand: nothing is idle then."""
        if not any(self._touched):
            return []
        return [sp for sp, t in zip(self._spans, self._touched) if not t]

    def zero(self):
        self.buf.zero_()
        if self.steal:
            for p in self.params:
                p.grad = None
        self._touched = [False] * len(self.params)
        self._pending = list(self._members)
        self._launched = [False] * len(self.buckets)
        self._next = len(self.buckets) - 1
        self._works = []

    def set_flag(self, loss):
        """device-side: flag = 1 when the loss is NaN/Inf (no host sync)."""
        if loss is not None:
            self.flag.copy_((~torch.isfinite(loss.detach().float())).float().reshape(1))

    def _publish_touched(self):
        """this rank's 0 / 1 mask -> header, before bucket 0 goes out.  The mask lives on the device (`_mask_dev`) and is copied
        into the (zeroed, all-reduced) header with one device-to-device copy per step; it is uploaded again only when it
        CHANGES, from a fresh pinned tensor (the host never rewrites pinned memory a pending copy may still read).  When autograd
        accumulated into NO parameter the gradients were written into the flat views by hand: all ones then."""
        mask = tuple(self._touched) if any(self._touched) else None
        if mask != self._mask_host:
            self._mask_host = mask
            src = torch.tensor(mask if mask is not None else [True] * len(self.params), dtype=torch.float32)
            if self.buf.is_cuda:
                src = src.pin_memory()
            self._mask_dev.copy_(src, non_blocking=True)
        self.touched_dev.copy_(self._mask_dev)

    def finish(self, loss=None):
        """launch the remaining buckets and wait; the buffer then holds the SUM over ranks (callers fold 1/world into
        their unscale factor, see optim.FlatAdamW), the header the flag and the summed touched mask."""
        self.set_flag(loss)
        self._publish_touched()
        if self.world == 1:
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self._launched[b] = True
                    self._gather(b)
        if self.world > 1:
            ev = None
            if self.time_exposed and self.buf.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            for b in range(len(self.buckets) - 1, -1, -1):         # the fixed order: descending, bucket 0 last
                if not self._launched[b]:
                    self._launch(b)
            for w in self._works:
                w.wait()
            if ev is not None:
                ev[1].record()
                self._exposed_events.append(ev)
        self._works = []

    def exposed_ms(self, reset=True):
        """average milliseconds per step the compute stream waited for the exchange (time_exposed must be set; host sync)."""
        if not self._exposed_events:
            return None
        torch.cuda.synchronize()
        v = sum(a.elapsed_time(b) for a, b in self._exposed_events) / len(self._exposed_events)
        if reset:
            self._exposed_events = []
        return v

    def bytes_per_step(self):
        """payload of the per-step all-reduce (flat fp32 gradient + the 64-float header that carries the NaN/Inf flag)."""
        return 4 * int(self.buf.numel())

    # -- host-synchronising convenience (tests, simple loops) --------------------------------------------------------
    def sync(self, loss=None, max_norm=None):
        """all-reduce the gradients (mean over ranks).  Returns dict(skip, grad_norm, clip_coef)."""
        self.finish(loss)
        if self.world > 1:
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
