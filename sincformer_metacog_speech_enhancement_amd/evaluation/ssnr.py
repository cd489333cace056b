"""evaluation/ssnr.py (compute_ssnr :26-92, compute_ssnr_improvement :95-111), batched on the device."""
import torch

from .. import config, ops
from ._common import to_device_batch


def compute_ssnr(clean_signal, enhanced_signal, fs=None, frame_size=None, hop_size=None, upper_bound=35.0,
                 lower_bound=-10.0):
    """1-D inputs -> float (the reference's signature); [B, L] inputs -> tensor [B] (stays on the device)."""
    frame_size = frame_size or config.FRAME_SIZE
    hop_size = hop_size or config.HOP_SIZE
    (c, e), one_d = to_device_batch(clean_signal, enhanced_signal)
    B, L = c.shape
    acc = torch.zeros(B, 2, device=c.device, dtype=torch.float64)
    Lb = ops._lib.load()
    ops._call("metrics", Lb.sfm_ssnr_frames, (ops._p(c), ops._p(e), ops._p(acc), B, L, int(frame_size), int(hop_size),
                                              float(upper_bound), float(lower_bound), ops._stream()), 0.0, 8.0 * B * L)
    out = torch.where(acc[:, 1] > 0, acc[:, 0] / acc[:, 1].clamp_min(1.0), torch.zeros_like(acc[:, 0]))
    return float(out[0]) if one_d else out


def compute_ssnr_improvement(clean_signal, noisy_signal, enhanced_signal, fs=None):
    return compute_ssnr(clean_signal, enhanced_signal, fs) - compute_ssnr(clean_signal, noisy_signal, fs)
