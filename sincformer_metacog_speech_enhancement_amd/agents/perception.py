"""Host-side mirror of agents/perception.py (SincConv1d, PerceptionAgent).

Parameter containers reproduce the reference's state_dict layout
(sinc_conv.{low_hz_,band_hz_,window,n_}, sinc_norm, conv_blocks.k.{main,skip},
downsample, real_proj, imag_proj, uncertainty_head); the arithmetic is the HIP
path of functional.perception_forward.
"""
import math
import numpy as np
import torch
from torch import nn

from .. import config, functional as Fn, ops
from .._hostmod import HipModule


class SincConv1d(HipModule):
    """agents/perception.py:23-118.  128 learnable cut-off scalars -> 64 Hamming-windowed
    band-pass FIR filters (incl. the double division by sample_rate)."""

    def __init__(self, out_channels, kernel_size, sample_rate=None, min_low_hz=50, min_band_hz=50):
        super().__init__()
        self.out_channels = out_channels
        self.sample_rate = sample_rate or config.SAMPLE_RATE
        self.kernel_size = kernel_size + 1 if kernel_size % 2 == 0 else kernel_size
        self.min_low_hz, self.min_band_hz = min_low_hz, min_band_hz
        # ERB-spaced initial cut-offs (agents/perception.py:50-65)
        lo, hi = min_low_hz, self.sample_rate / 2 - min_band_hz
        erb = np.linspace(21.4 * math.log10(1 + lo / 228.7), 21.4 * math.log10(1 + hi / 228.7), out_channels + 1)
        hz = 228.7 * (10 ** (erb / 21.4) - 1)
        self.low_hz_ = nn.Parameter(torch.Tensor(hz[:-1]).view(-1, 1))
        self.band_hz_ = nn.Parameter(torch.Tensor(np.diff(hz)).view(-1, 1))
        k = self.kernel_size
        n = torch.linspace(0, k - 1, k)
        self.register_buffer("window", 0.54 - 0.46 * torch.cos(2 * math.pi * n / k))
        self.register_buffer("n_", 2 * math.pi * torch.arange(-(k - 1) / 2.0, 0).view(1, -1) / self.sample_rate)

    def filters(self):
        """[out_channels, kernel_size] fp32 filter bank synthesised on device."""
        self._require_device(self.low_hz_)
        f, _ = ops.sinc_filters(self.low_hz_.detach().reshape(-1).contiguous(), self.band_hz_.detach().reshape(-1).contiguous(),
                                self.window.contiguous(), self.n_.reshape(-1).contiguous(), self.out_channels,
                                self.kernel_size, self.sample_rate, self.min_low_hz, self.min_band_hz)
        return f

    def forward(self, waveform):
        """(batch, 1, samples) -> (batch, out_channels, samples), fp32."""
        self._require_device(waveform)
        if self.training and torch.is_grad_enabled() and (self.low_hz_.requires_grad or self.band_hz_.requires_grad):
            from .. import train                                    # train() mode: tap gradient + chain rule to the cut-offs
            return train.SincConvFunction.apply(waveform, self, self.low_hz_, self.band_hz_)
        B, _, L = waveform.shape
        C, K = self.out_channels, self.kernel_size
        _, Wt = ops.sinc_filters(self.low_hz_.detach().reshape(-1).contiguous(), self.band_hz_.detach().reshape(-1).contiguous(),
                                 self.window.contiguous(), self.n_.reshape(-1).contiguous(), C, K, self.sample_rate,
                                 self.min_low_hz, self.min_band_hz, want_filt=False)
        out = torch.empty(B, C, L, device=waveform.device, dtype=torch.float32)
        w = waveform.float().reshape(B, L).contiguous()
        ops.framed_gemm(w, Wt, out, B=B, M=L, Ls=L, sig_batch_stride=L, hop=1, padl=K // 2, K=K, N=C,
                        o_batch_stride=C * L, ldm=1, ldn=L, mode=0)
        return out


class _ResidualBlock(nn.Module):
    """container only: main = [Conv k7 s2, GN, (GELU), Conv k3, GN], skip = [Conv k1 s2, GN]."""

    def __init__(self, main, skip):
        super().__init__()
        self.main, self.skip = main, skip


def _gn(ch):
    return nn.GroupNorm(min(16, ch), ch)


class PerceptionAgent(HipModule):
    """agents/perception.py:132-254: waveform -> (z_real, z_imag [B, D, L/16], sigma [B, 1, L/16])."""

    def __init__(self, encoder_channels=None, sample_rate=None):
        super().__init__()
        self.encoder_channels = encoder_channels or config.PA_ENCODER_CHANNELS
        self.sample_rate = sample_rate or config.SAMPLE_RATE
        D = self.encoder_channels
        if D != 256:
            raise NotImplementedError("HIP PerceptionAgent is built for encoder_channels=256 (config.PA_ENCODER_CHANNELS)")
        self.sinc_conv = SincConv1d(out_channels=D // 4, kernel_size=251, sample_rate=self.sample_rate)
        self.sinc_norm = nn.GroupNorm(8, D // 4)
        widths = [D // 4, D // 2, D // 2, D]
        self.conv_blocks = nn.ModuleList()
        for cin, cout in zip(widths[:-1], widths[1:]):
            main = nn.Sequential(nn.Conv1d(cin, cout, 7, stride=2, padding=3), _gn(cout), nn.GELU(),
                                 nn.Conv1d(cout, cout, 3, padding=1), _gn(cout))
            skip = nn.Sequential(nn.Conv1d(cin, cout, 1, stride=2), _gn(cout))
            self.conv_blocks.append(_ResidualBlock(main, skip))
        self.downsample = nn.Sequential(nn.Conv1d(D, D, kernel_size=5, stride=2, padding=2), nn.GroupNorm(16, D), nn.GELU())
        self.real_proj = nn.Sequential(nn.Conv1d(D, D, 1), nn.GroupNorm(16, D))
        self.imag_proj = nn.Sequential(nn.Conv1d(D, D, 1), nn.GroupNorm(16, D))
        self.uncertainty_head = nn.Sequential(nn.Conv1d(D, D // 4, 3, padding=1), nn.GELU(), nn.Conv1d(D // 4, 1, 1))
        # agents/perception.py:208-214: kaiming-normal (linear gain) conv weights, zero biases
        for m in self.modules():
            if isinstance(m, nn.Conv1d):
                nn.init.kaiming_normal_(m.weight, nonlinearity="linear")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_channels_last(self, waveform):
        """Internal layout: zcat [B, T_pa, 2D] fp32 (z_real | z_imag), sigma [B, T_pa]."""
        if waveform.dim() == 3:
            waveform = waveform.squeeze(1)
        self._require_device(waveform)
        pk = self._packed(lambda sd: Fn.pack_perception(sd, self.sample_rate))
        return Fn.perception_forward(waveform.float(), pk)

    def forward(self, waveform):
        if (self.training and torch.is_grad_enabled()) or self._wants_autograd(waveform):
            from .. import train                   # train() (or eval() under autograd): HIP backward kernels
            self._require_device(waveform)
            return train.perception_train_forward(self, waveform)
        zcat, sigma = self.forward_channels_last(waveform)
        B, Tpa, D2 = zcat.shape
        D = D2 // 2
        z = torch.empty(B, D2, Tpa, device=zcat.device, dtype=torch.float32)
        ops.transpose(zcat, z, B, Tpa, D2, Tpa * D2, D2, D2 * Tpa, Tpa)
        return z[:, :D], z[:, D:], sigma.reshape(B, 1, Tpa)
