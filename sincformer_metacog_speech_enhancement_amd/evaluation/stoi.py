"""evaluation/stoi.py: the reference calls pystoi when it is installed and otherwise its own simplified measure
(_stoi_simplified :53-99).  pystoi exists neither in the reference's environment here nor on the GPU box, so the
simplified measure is what the reference computes; that one is batched here on the device (`extended` is accepted and
ignored, as in the fallback)."""
import numpy as np
import torch

from .. import config, ops
from ._common import to_device_batch

_mat = {}


def _dft_operand(frame_len, dev):
    """[frame_len, 2F] windowed rfft as a real matrix: np.hanning (symmetric) window, F = frame_len // 2 + 1."""
    key = (frame_len, str(dev))
    m = _mat.get(key)
    if m is None:
        F = frame_len // 2 + 1
        w = np.hanning(frame_len)
        n = np.arange(frame_len, dtype=np.float64)[:, None]
        f = np.arange(F, dtype=np.float64)[None, :]
        ang = 2.0 * np.pi * ((n * f) % frame_len) / frame_len
        M = np.concatenate([w[:, None] * np.cos(ang), -w[:, None] * np.sin(ang)], axis=1)
        m = ops.pack_f32_matrix(torch.from_numpy(M.astype(np.float32)).to(dev))
        _mat[key] = m
    return m


def compute_stoi(clean_signal, enhanced_signal, fs=None, extended=False):
    """1-D inputs -> float; [B, L] inputs -> tensor [B]."""
    fs = fs or config.SAMPLE_RATE
    (c, e), one_d = to_device_batch(clean_signal, enhanced_signal)
    B, L = c.shape
    frame_len = int(0.0256 * fs)
    hop = frame_len // 2
    nframes = (L - frame_len) // hop + 1
    if L < frame_len or nframes < 1:
        out = torch.zeros(B, device=c.device, dtype=torch.float64)
        return 0.0 if one_d else out
    F = frame_len // 2 + 1
    W = _dft_operand(frame_len, c.device)
    spec = []
    for sig in (c, e):
        re = torch.empty(B, nframes, F, device=c.device, dtype=torch.float32)
        im = torch.empty(B, nframes, F, device=c.device, dtype=torch.float32)
        ops.framed_gemm(sig, W, re, B=B, M=nframes, Ls=L, sig_batch_stride=L, hop=hop, padl=0, K=frame_len, N=2 * F,
                        o_batch_stride=nframes * F, ldm=F, ldn=1, mode=0, out2=im, nsplit=F)
        spec += [re, im]
    S = ops.wave_moments(e, c)                                        # {sum e, sum c, sum e^2, sum c^2, sum e c}
    sc = (1.0 / (torch.sqrt(S[:, 3] / L) + 1e-10)).contiguous()
    se = (1.0 / (torch.sqrt(S[:, 2] / L) + 1e-10)).contiguous()
    acc = torch.zeros(B, device=c.device, dtype=torch.float64)
    Lb = ops._lib.load()
    ops._call("metrics", Lb.sfm_stoi_frames, (ops._p(spec[0]), ops._p(spec[1]), ops._p(spec[2]), ops._p(spec[3]), ops._p(sc),
                                              ops._p(se), ops._p(acc), B, nframes, F, ops._stream()), 0.0,
              16.0 * B * nframes * F)
    out = (acc / nframes).clamp(0.0, 1.0)
    return float(out[0]) if one_d else out
