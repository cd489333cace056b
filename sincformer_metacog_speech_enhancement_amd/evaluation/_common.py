import numpy as np
import torch


def to_device_batch(*signals):
    """1-D numpy / tensor (the reference's call style) or [B, L] -> contiguous fp32 [B, L] device tensors, trimmed to the
    shortest (evaluation/ssnr.py:50-52, evaluation/stoi.py:42-44); returns (tensors, was_1d)."""
    ts = []
    for s in signals:
        t = torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32)) if isinstance(s, np.ndarray) else s.detach().float()
        ts.append(t)
    one_d = ts[0].dim() == 1
    ts = [t.unsqueeze(0) if t.dim() == 1 else t for t in ts]
    n = min(t.shape[-1] for t in ts)
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP metrics need an MI355X (no CPU fallback)")
    return [t[:, :n].cuda().contiguous() for t in ts], one_d
