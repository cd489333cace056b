"""Host-side mirror of the reference's models/conformer.py public classes.

Same class names, constructor signatures (including the `x or config.X`
defaulting of models/conformer.py:173-179, under which dropout=0.0 silently
becomes 0.1) and state_dict keys; forward() runs on the HIP kernels
(functional.py) instead of aten.  Shapes and semantics per class cite the
reference lines they replace.
"""
import torch
from torch import nn

from .. import config, functional as Fn, ops
from .._hostmod import HipModule


def _submodule_autograd(mod, kind, x, heads, p, bn):
    """train() mode (or eval() under autograd) of a stand-alone Conformer sub-module: one HIP autograd node."""
    from .. import train
    if x.dim() == 2:
        x = x.unsqueeze(0)
    named = dict(mod.named_parameters())
    params = [named[k] for k in train.submodule_param_names(kind)]
    p = float(p) if mod.training else 0.0
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0
    buffers = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if (bn is not None and bn.track_running_stats) else None
    meta = (kind, heads, p, seed, buffers, float(bn.momentum or 0.1) if bn is not None else 0.1,
            float(bn.eps) if bn is not None else 1e-5, not mod.training)
    return train.SubmoduleFunction.apply(x, meta, *params)


class FeedForwardModule(HipModule):
    """models/conformer.py:28-49 — LN -> Linear -> Swish -> Linear, half-step residual."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, x):
        self._require_device(x)
        if self.training or self._wants_autograd(x):
            return _submodule_autograd(self, "ffn", x, 1, self.dropout.p, None)
        with ops.stage("block"):
            pk = self._packed(Fn.pack_ffn)
            shp = x.shape
            y = Fn.ffn_forward(x.float().reshape(-1, shp[-1]).contiguous(), pk)
        return y.reshape(shp)


class MultiHeadSelfAttention(HipModule):
    """models/conformer.py:52-71 — pre-LN MHSA (nn.MultiheadAttention parameters), residual."""

    def __init__(self, d_model, num_heads, dropout=0.1):
        super().__init__()
        self.attention = nn.MultiheadAttention(d_model, num_heads, dropout=dropout, batch_first=True)
        self.layer_norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.num_heads = num_heads

    def forward(self, x):
        self._require_device(x)
        if self.training or self._wants_autograd(x):
            return _submodule_autograd(self, "mhsa", x, self.num_heads, self.dropout.p, None)
        with ops.stage("block"):
            pk = self._packed(lambda sd: Fn.pack_mhsa(sd, self.num_heads))
            B, T, D = x.shape
            y = Fn.mhsa_forward(x.float().reshape(B * T, D).contiguous(), pk, B, T, self.num_heads)
        return y.reshape(B, T, D)


class ConvolutionModule(HipModule):
    """models/conformer.py:74-128 — LN, pointwise+GLU, depthwise, BatchNorm, Swish, pointwise, residual."""

    def __init__(self, d_model, kernel_size=31, dropout=0.1):
        super().__init__()
        self.layer_norm = nn.LayerNorm(d_model)
        self.pointwise1 = nn.Conv1d(d_model, 2 * d_model, 1)
        self.depthwise = nn.Conv1d(d_model, d_model, kernel_size, padding=(kernel_size - 1) // 2, groups=d_model)
        self.batch_norm = nn.BatchNorm1d(d_model)
        self.pointwise2 = nn.Conv1d(d_model, d_model, 1)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        self._require_device(x)
        if self.training or self._wants_autograd(x):
            return _submodule_autograd(self, "conv", x, 1, self.dropout.p, self.batch_norm)
        with ops.stage("block"):
            pk = self._packed(Fn.pack_convmod)
            B, T, D = x.shape
            y = Fn.convmod_forward(x.float().reshape(B * T, D).contiguous(), pk, B, T)
        return y.reshape(B, T, D)


class ConformerBlock(HipModule):
    """models/conformer.py:131-151 — ff1 -> mhsa -> conv -> ff2 -> LayerNorm.
    This is the class training/conformer_pipeline.py:45 imports."""

    def __init__(self, d_model, num_heads, d_ff, kernel_size, dropout):
        super().__init__()
        self.ff1 = FeedForwardModule(d_model, d_ff, dropout)
        self.mhsa = MultiHeadSelfAttention(d_model, num_heads, dropout)
        self.conv = ConvolutionModule(d_model, kernel_size, dropout)
        self.ff2 = FeedForwardModule(d_model, d_ff, dropout)
        self.final_norm = nn.LayerNorm(d_model)
        self.num_heads = num_heads

    def forward(self, x):
        self._require_device(x)
        if self.training or self._wants_autograd(x):
            return self._train_forward(x)
        pk = self._packed(lambda sd: Fn.pack_block(sd, self.num_heads))
        B, T, D = x.shape
        y = Fn.block_forward(x.float().reshape(B * T, D).contiguous(), pk, B, T, self.num_heads)
        return y.reshape(B, T, D)


    def _train_forward(self, x):
        """Training mode (dropout active, BatchNorm batch statistics + running-stat update): one autograd node
        whose forward and backward run on the HIP kernels (train.ConformerBlockFunction).  The dropout seed is
        drawn from torch's default CPU generator, so torch.manual_seed() makes a step reproducible."""
        from .. import train
        named = dict(self.named_parameters())
        params = [named[k] for k in train.PARAM_NAMES]
        bn = self.conv.batch_norm
        buffers = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if bn.track_running_stats else None
        # eval() with autograd (gradient-based analysis, fine-tuning with frozen statistics): no dropout, BatchNorm on its
        # running statistics, same backward kernels
        p = float(self.ff1.dropout.p) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0
        meta = (self.num_heads, p, seed, buffers, float(bn.momentum or 0.1), float(bn.eps), not self.training)
        return train.ConformerBlockFunction.apply(x, meta, *params)


class ComplexConformer(HipModule):
    """models/conformer.py:154-249 — cat(real, imag) -> input_proj -> N blocks -> +skip -> output_proj -> split."""

    def __init__(self, n_freq=None, d_model=None, num_blocks=None, num_heads=None, d_ff=None, kernel_size=None,
                 dropout=None):
        super().__init__()
        self.n_freq = n_freq or (config.FFT_SIZE // 2 + 1)
        self.d_model = d_model or config.CONFORMER_D_MODEL
        num_blocks = num_blocks or config.CONFORMER_NUM_BLOCKS
        num_heads = num_heads or config.CONFORMER_NUM_HEADS
        d_ff = d_ff or config.CONFORMER_FF_DIM
        kernel_size = kernel_size or config.CONFORMER_KERNEL_SIZE
        dropout = dropout or config.CONFORMER_DROPOUT          # falsy 0.0 -> 0.1, as the reference
        self.num_heads, self.num_blocks = num_heads, num_blocks
        self.input_proj = nn.Linear(2 * self.n_freq, self.d_model)
        self.blocks = nn.ModuleList(
            [ConformerBlock(self.d_model, num_heads, d_ff, kernel_size, dropout) for _ in range(num_blocks)])
        self.output_proj = nn.Linear(self.d_model, 2 * self.n_freq)

    def train_core(self, x, B, T):
        """[M, 2F] (real | imag side by side) -> [M, 2F]: Linear(2F -> d) -> N x ConformerBlock (dropout, BatchNorm batch
        statistics) -> + skip -> Linear(d -> 2F), HIP autograd nodes throughout"""
        from .. import train
        M = B * T
        x = train.LNLinearFunction.apply(x, None, None, self.input_proj.weight, self.input_proj.bias)
        skip = x
        h = x.reshape(B, T, -1)
        for block in self.blocks:
            h = block(h)
        x = h.reshape(M, -1) + skip
        return train.LNLinearFunction.apply(x, None, None, self.output_proj.weight, self.output_proj.bias)

    def _train_forward(self, stft_real, stft_imag):
        """train() mode (the mode the reference's tests/test_conformer.py runs in)"""
        B, T, F = stft_real.shape
        x = torch.cat([stft_real.float(), stft_imag.float()], dim=-1).reshape(B * T, 2 * F)
        y = self.train_core(x, B, T).reshape(B, T, 2 * F)
        return y[..., :F], y[..., F:]

    def forward(self, stft_real, stft_imag):
        self._require_device(stft_real, stft_imag)
        if self.training or self._wants_autograd(stft_real, stft_imag):
            return self._train_forward(stft_real, stft_imag)
        pk = self._packed(lambda sd: Fn.pack_complex_conformer(sd, self.num_blocks, self.num_heads))
        return Fn.complex_conformer_forward(stft_real.float(), stft_imag.float(), pk, self.num_heads)

    def apply_mask(self, stft_real, stft_imag, mask_real, mask_imag):
        """models/conformer.py:230-245 complex multiply."""
        self._require_device(stft_real, stft_imag, mask_real, mask_imag)
        from .. import ops
        if torch.is_grad_enabled() and any(t.requires_grad for t in (stft_real, stft_imag, mask_real, mask_imag)):
            from .. import train
            return train.ComplexMulFunction.apply(stft_real, stft_imag, mask_real, mask_imag)
        shp = stft_real.shape
        er, ei = ops.complex_mul(stft_real.float().contiguous(), stft_imag.float().contiguous(),
                                 mask_real.float().contiguous(), mask_imag.float().contiguous())
        return er.reshape(shp), ei.reshape(shp)
