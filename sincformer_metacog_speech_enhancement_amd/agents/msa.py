"""Host-side mirror of agents/msa.py (MaskSynthesisAgent)."""
import torch
from torch import nn

from .. import config, functional as Fn
from .._hostmod import HipModule
from ..models.conformer import ComplexConformer


class MaskSynthesisAgent(HipModule):
    """agents/msa.py:20-177: 8-way feature fusion -> ComplexConformer(n_freq=d/2) -> two 2-layer
    heads -> bounded polar mask (|mask| <= 1, |phase| <= 3.14159/8)."""

    def __init__(self, latent_dim=None, cpea_dim=None, d_model=None):
        super().__init__()
        latent_dim = latent_dim or config.PA_ENCODER_CHANNELS
        cpea_dim = cpea_dim or config.NUM_CHANNELS
        d_model = d_model or config.CONFORMER_D_MODEL
        n_freq = config.FFT_SIZE // 2 + 1
        fusion_in = 2 * latent_dim + 4 * cpea_dim + 2 * n_freq
        if fusion_in > Fn.FUSE_LD:
            raise NotImplementedError("fusion width %d exceeds the packed operand width %d" % (fusion_in, Fn.FUSE_LD))
        self._dims = (latent_dim, cpea_dim, n_freq)
        self.fusion = nn.Sequential(nn.Linear(fusion_in, d_model), nn.LayerNorm(d_model), nn.GELU(),
                                    nn.Linear(d_model, d_model), nn.LayerNorm(d_model))
        self.conformer = ComplexConformer(n_freq=d_model // 2, d_model=d_model)
        half = d_model // 2
        self.mask_proj_real = nn.Sequential(nn.Linear(half, half), nn.GELU(), nn.Linear(half, n_freq))
        self.mask_proj_imag = nn.Sequential(nn.Linear(half, half), nn.GELU(), nn.Linear(half, n_freq))
        # agents/msa.py:78-104: xavier(gain 0.1) heads, magnitude bias +5 (sigmoid ~ 0.993), phase bias 0
        for head in (self.mask_proj_real, self.mask_proj_imag):
            for m in head:
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight, gain=0.1)
                    nn.init.zeros_(m.bias)
        nn.init.constant_(self.mask_proj_real[-1].bias, 5.0)

    def forward(self, z_real, z_imag, cpea_outputs, noisy_stft_real, noisy_stft_imag, mag_logit_bias=None, latents_cl=None):
        """Returns (mask_real, mask_imag), each [B, T, n_freq] fp32.  `mag_logit_bias` ([B, n_freq],
        optional, build-defined glue G3) is added to the magnitude logit before the sigmoid.  `latents_cl` (optional, train()
        mode): z_real | z_imag as ONE channels-last [B, T, 2 latent_dim] tensor - the layout the path holds them in."""
        self._require_device(z_real, z_imag, noisy_stft_real, noisy_stft_imag)
        if z_real.shape[-1] != noisy_stft_real.shape[1]:
            raise RuntimeError("Sizes of tensors must match except in dimension 2. Expected size %d but got size %d "
                               "(latents must be at the STFT frame rate)" % (z_real.shape[-1], noisy_stft_real.shape[1]))
        if self.training or self._wants_autograd(z_real, z_imag, *cpea_outputs.values()):
            from .. import train                   # train() (or eval() under autograd): HIP autograd nodes
            return train.msa_train_forward(self, z_real, z_imag, cpea_outputs, noisy_stft_real, noisy_stft_imag,
                                           mag_logit_bias, latents_cl=latents_cl)
        pk = self._packed(lambda sd: Fn.pack_msa(sd, self.conformer.num_blocks, self.conformer.num_heads))
        return Fn.msa_forward(z_real, z_imag, cpea_outputs, noisy_stft_real, noisy_stft_imag, pk,
                              self.conformer.num_heads, mag_bias=mag_logit_bias)
