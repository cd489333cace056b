"""Host-side mirror of agents/cpea.py (CorrelationPhaseEstimationAgent)."""
import torch
from torch import nn

from .. import config, functional as Fn, ops
from .._hostmod import HipModule


class CorrelationPhaseEstimationAgent(HipModule):
    """agents/cpea.py:22-115: 2-layer BiLSTM + four heads (sigmoid, sigmoid, pi*tanh, pi*tanh).
    The recurrence runs in the persistent HIP kernel csrc/lstm.hip."""

    def __init__(self, input_dim=None, hidden_size=None, num_layers=None, output_channels=None):
        super().__init__()
        self.input_dim = input_dim or config.PA_ENCODER_CHANNELS
        self.hidden_size = hidden_size or config.CPEA_HIDDEN_SIZE
        self.num_layers = num_layers or config.CPEA_NUM_LAYERS
        self.output_channels = output_channels or config.NUM_CHANNELS
        self.lstm = nn.LSTM(input_size=self.input_dim, hidden_size=self.hidden_size, num_layers=self.num_layers,
                            batch_first=True, bidirectional=True, dropout=0.1 if self.num_layers > 1 else 0.0)
        width = 2 * self.hidden_size
        self.rho_s_head = nn.Sequential(nn.Linear(width, self.output_channels), nn.Sigmoid())
        self.rho_n_head = nn.Sequential(nn.Linear(width, self.output_channels), nn.Sigmoid())
        self.phi1_head = nn.Sequential(nn.Linear(width, self.output_channels), nn.Tanh())
        self.phi2_head = nn.Sequential(nn.Linear(width, self.output_channels), nn.Tanh())

    def forward(self, z_t):
        self._require_device(z_t)
        if self.training or self._wants_autograd(z_t):
            from .. import train                   # train() (or eval() under autograd): BPTT on the HIP kernels
            return train.cpea_train_forward(self, z_t)
        pk = self._packed(lambda sd: Fn.pack_cpea(sd, self.num_layers))
        z = z_t.float()
        dt = ops.stage_dtype("front")
        if z.dim() == 3 and z.shape[-1] != self.input_dim:        # agents/cpea.py:94-96
            B, D, T = z.shape
            z16 = torch.empty(B * T, D, device=z.device, dtype=dt)
            with ops.stage("front"):
                ops.transpose(z.contiguous(), z16, B, D, T, D * T, T, T * D, D)
        else:
            B, T, D = z.shape
            z16 = torch.empty(B * T, D, device=z.device, dtype=dt)
            with ops.stage("front"):
                ops.convert_rows(z.contiguous(), z16, B * T, D, D, D, D)
        out = Fn.cpea_forward(z16, pk, B, T).reshape(B, T, -1)
        oc = self.output_channels
        return {"rho_s": out[..., :oc], "rho_n": out[..., oc:2 * oc], "phi1": out[..., 2 * oc:3 * oc],
                "phi2": out[..., 3 * oc:4 * oc]}
