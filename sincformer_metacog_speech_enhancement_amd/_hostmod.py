"""Shared plumbing of the nn.Module mirrors: device checks and the packed-weight cache."""
import torch

from . import ops


class HipModule(torch.nn.Module):
    """Base of every mirror.  Parameters/buffers live in ordinary torch submodules
    (so state_dict keys, shapes and default init equal the reference's), but
    forward() never calls them: it packs the tensors for the HIP kernels (cached
    until a parameter changes) and enqueues kernels through the C ABI."""

    def _pack_key(self):
        sig = [ops.policy_key()]
        for t in list(self.parameters()) + list(self.buffers()):
            sig.append((t.data_ptr(), t._version))
        return tuple(sig)

    def _packed(self, builder):
        key = self._pack_key()
        cache = self.__dict__.get("_sfm_pack")
        if cache is None or cache[0] != key:
            sd = {k: v.detach() for k, v in self.state_dict(keep_vars=True).items()}
            cache = (key, builder(sd))
            self.__dict__["_sfm_pack"] = cache
        return cache[1]

    def _require_device(self, *tensors):
        p = next(self.parameters(), None)
        for t in tensors:
            if not t.is_cuda:
                raise RuntimeError(
                    "%s: the sincformer HIP path only runs on an MI355X device tensor; got a CPU tensor "
                    "(there is deliberately no CPU fallback)." % type(self).__name__)
        if p is not None and not p.is_cuda:
            raise RuntimeError("%s: parameters are on CPU; call .cuda() first" % type(self).__name__)

    def enable_eval_autograd(self, flag=True):
        """eval() normally runs the fused inference kernels and records no autograd graph (also when parameters require
        grad, which they do by default: inference code that forgets torch.no_grad() must not pay for a training
        forward).  With this flag - or whenever an INPUT requires grad - eval() runs the autograd nodes instead
        (dropout off, BatchNorm on running statistics), as the reference's modules would under autograd."""
        for m in self.modules():
            if isinstance(m, HipModule):
                m.__dict__["_sfm_eval_autograd"] = bool(flag)
        return self

    def _wants_autograd(self, *inputs):
        if not torch.is_grad_enabled():
            return False
        return self.__dict__.get("_sfm_eval_autograd", False) or any(t.requires_grad for t in inputs)

    def count_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
