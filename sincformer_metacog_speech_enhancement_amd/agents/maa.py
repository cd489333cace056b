"""Host-side mirror of agents/maa.py (MetacognitiveArbitrationAgent, :26-143): same constructor, forward dict and
state_dict keys; the arithmetic runs in routing.hip (one thread per time step, weights in LDS).  SURVEY 8f N4."""
import torch
from torch import nn

from .. import config, ops
from .._hostmod import HipModule


class MetacognitiveArbitrationAgent(HipModule):
    SOFT_MASK = 0
    RESAMPLE = 1
    HARD_MASK = 2
    ESCALATE = 3

    def __init__(self, input_dim=1, hidden_dim=64, num_classes=4, initial_threshold=None):
        super().__init__()
        if (input_dim, hidden_dim, num_classes) != (1, 64, 4):
            raise NotImplementedError("MetacognitiveArbitrationAgent (HIP build): input_dim 1, hidden_dim 64, num_classes 4 "
                                      "(the reference's configuration) only")
        self.threshold_init = initial_threshold or config.MAA_THRESHOLD_INIT
        self.threshold = nn.Parameter(torch.tensor([self.threshold_init]))
        self.decision_net = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim),
                                          nn.ReLU(), nn.Linear(hidden_dim, num_classes))
        self.register_buffer("running_mean", torch.tensor(0.0))
        self.register_buffer("running_var", torch.tensor(1.0))
        self.register_buffer("num_updates", torch.tensor(0))

    def forward(self, sigma):
        """sigma [B, 1, T] or [B, T] -> dict(decisions [B,T] int64, probs, logits [B,T,4], threshold, confidence [B,T])"""
        from .. import train
        self._require_device(sigma)
        if sigma.dim() == 3:
            sigma = sigma.squeeze(1)
        B, T = sigma.shape
        flat = sigma.reshape(-1)
        stats = torch.stack([self.running_mean.float(), self.running_var.float()])
        if self.training:
            ops.maa_update_stats(flat.detach().float().contiguous(), stats, self.num_updates)     # agents/maa.py:126-135
            self.running_mean.copy_(stats[0])
            self.running_var.copy_(stats[1])
        net = self.decision_net
        logits, probs, conf, dec = train.MaaFunction.apply(flat, stats, net[0].weight, net[0].bias, net[2].weight, net[2].bias,
                                                           net[4].weight, net[4].bias)
        return {"decisions": dec.reshape(B, T), "probs": probs.reshape(B, T, 4), "logits": logits.reshape(B, T, 4),
                "threshold": self.threshold, "confidence": conf.reshape(B, T)}

    def get_strategy_name(self, decision_idx):
        names = {0: "SOFT_MASK (high confidence)", 1: "RESAMPLE (ensemble averaging)", 2: "HARD_MASK (quantized fallback)",
                 3: "ESCALATE (human review)"}
        return names.get(decision_idx, "UNKNOWN")
