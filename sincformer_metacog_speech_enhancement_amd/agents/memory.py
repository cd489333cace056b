"""Host-side mirror of agents/memory.py (EpisodicMemory)."""
import torch
from torch import nn

from .. import config, functional as Fn, ops
from .._hostmod import HipModule


class EpisodicMemory(HipModule):
    """agents/memory.py:24-158: cosine-similarity key/value read-out -> gated tanh bias."""

    def __init__(self, key_dim=None, value_dim=None, num_slots=64, temperature=1.0):
        super().__init__()
        self.key_dim = key_dim or config.PA_ENCODER_CHANNELS
        self.value_dim = value_dim or (config.FFT_SIZE // 2 + 1)
        self.num_slots, self.temperature = num_slots, temperature
        self.keys = nn.Parameter(torch.randn(num_slots, self.key_dim) * 0.01)
        self.values = nn.Parameter(torch.randn(num_slots, self.value_dim) * 0.01)
        self.key_proj = nn.Sequential(nn.Linear(self.key_dim, self.key_dim), nn.LayerNorm(self.key_dim), nn.GELU(),
                                      nn.Linear(self.key_dim, self.key_dim))
        self.value_proj = nn.Sequential(nn.Linear(self.value_dim, self.value_dim), nn.Tanh())
        nn.init.xavier_uniform_(self.value_proj[0].weight, gain=0.01)
        nn.init.zeros_(self.value_proj[0].bias)
        self.gate = nn.Sequential(nn.Linear(self.key_dim + self.value_dim, 1), nn.Sigmoid())
        self.register_buffer("usage_count", torch.zeros(num_slots))
        self.register_buffer("num_queries", torch.tensor(0))

    def _pack_key(self):      # usage counters must not invalidate the packed parameters
        return tuple([ops.policy_key()] + [(p.data_ptr(), p._version) for p in self.parameters()])

    def forward(self, environment_embedding):
        self._require_device(environment_embedding)
        if (self.training and torch.is_grad_enabled()) or self._wants_autograd(environment_embedding):
            from .. import train
            out = train.memory_train_forward(self, environment_embedding)
            if self.training:
                with torch.no_grad():
                    top = out["top_indices"]
                    self.usage_count.index_add_(0, top, torch.ones_like(top, dtype=self.usage_count.dtype))
                    self.num_queries += environment_embedding.shape[0]
            return out
        params = self._packed(Fn.pack_memory_params)
        e = environment_embedding.float().contiguous()
        bias, gate, top, sim = ops.memory_fwd(e, params, self.key_dim, self.value_dim, self.num_slots, self.temperature)
        top = top.long()
        if self.training:
            # agents/memory.py:136-141 without the per-sample Python loop / host sync
            with torch.no_grad():
                self.usage_count.index_add_(0, top, torch.ones_like(top, dtype=self.usage_count.dtype))
                self.num_queries += e.shape[0]
        return {"bias": bias, "gate": gate, "top_indices": top, "similarity": sim}

    def get_usage_stats(self):
        total = self.num_queries.item()
        if total == 0:
            return torch.zeros(self.num_slots)
        return self.usage_count / total
