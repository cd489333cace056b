"""Host-side mirror of the model / STFT / objective part of training/conformer_pipeline.py
(si_snr_loss :52, MultiResolutionSTFTLoss :74, batch_stft :196, batch_istft :205, SpeechEnhancer :218-301,
ConformerPipeline._compute_loss :539, _save_best :611, save_model :618, load_model :628, enhance_signal :653) and the
north-star agent composition (EnhancementPath; SURVEY.md §3.3 with the glue of DESIGN.md).
The data loaders / optimiser loop of the reference are out of scope.
"""
import math
import os
import numpy as np
import torch
from torch import nn

from .. import config, functional as Fn, ops
from .._hostmod import HipModule
from ..models.conformer import ConformerBlock
from ..agents.perception import PerceptionAgent
from ..agents.cpea import CorrelationPhaseEstimationAgent
from ..agents.msa import MaskSynthesisAgent
from ..agents.memory import EpisodicMemory


def batch_stft(waveform, fft_size, hop_size, frame_size):
    """training/conformer_pipeline.py:196-202 -> (real, imag) each [B, T, fft_size/2+1] fp32
    (contiguous; the reference returns transposed views of [B, F, T])."""
    if not waveform.is_cuda:
        raise RuntimeError("batch_stft: HIP path needs a device tensor (no CPU fallback)")
    return Fn.stft(waveform.float(), fft_size, hop_size, frame_size)


def batch_istft(stft_real, stft_imag, fft_size, hop_size, frame_size, length):
    """training/conformer_pipeline.py:205-211 -> [B, length] fp32."""
    if not stft_real.is_cuda:
        raise RuntimeError("batch_istft: HIP path needs a device tensor (no CPU fallback)")
    return Fn.istft(stft_real.float(), stft_imag.float(), length, fft_size, hop_size, frame_size)


def si_snr_loss(estimated, target):
    """training/conformer_pipeline.py:52-71 - negative mean scale-invariant SNR of `estimated` [..., L] against `target`
    (to minimise).  Differentiable w.r.t. `estimated` (HIP backward kernel)."""
    if not estimated.is_cuda:
        raise RuntimeError("si_snr_loss: HIP path needs device tensors (no CPU fallback)")
    from .. import train
    return train.SiSnrFunction.apply(estimated, target)


class MultiResolutionSTFTLoss(nn.Module):
    """training/conformer_pipeline.py:74-108 - spectral convergence + L1 log-magnitude at three STFT resolutions (Hann
    windows, centre reflect padding), averaged over the resolutions.  Same constructor and method names as the reference;
    the STFTs, the reductions and the backward run on the HIP kernels."""

    def __init__(self, fft_sizes=None, hop_sizes=None, win_sizes=None):
        super().__init__()
        self.fft_sizes = fft_sizes or [256, 512, 1024]
        self.hop_sizes = hop_sizes or [64, 128, 256]
        self.win_sizes = win_sizes or [256, 512, 1024]

    def _sizes(self):
        return tuple(zip(self.fft_sizes, self.hop_sizes, self.win_sizes))

    def _stft_mag(self, x, fft_size, hop_size, win_size):
        """|STFT| laid out [B, F, T] like torch.stft (monitoring / tests; forward() does not materialise it)."""
        if not x.is_cuda:
            raise RuntimeError("MultiResolutionSTFTLoss: HIP path needs device tensors (no CPU fallback)")
        r, i = Fn.stft(x.float(), fft_size, hop_size, win_size)
        return torch.sqrt(r * r + i * i).transpose(1, 2)

    def forward(self, predicted, target):
        if not predicted.is_cuda:
            raise RuntimeError("MultiResolutionSTFTLoss: HIP path needs device tensors (no CPU fallback)")
        from .. import train
        return train.MrStftFunction.apply(predicted, target, self._sizes())


class SpeechEnhancer(HipModule):
    """training/conformer_pipeline.py:218-301: LN(2F) -> Linear -> N ConformerBlocks -> LN ->
    magnitude (sigmoid) / phase (tanh * pi/6) heads -> polar mask applied to the noisy STFT."""

    def __init__(self, n_freq=None, d_model=256, num_blocks=4, num_heads=4, d_ff=1024, kernel_size=31, dropout=0.15):
        super().__init__()
        self.n_freq = n_freq or (config.FFT_SIZE // 2 + 1)
        self.num_heads, self.num_blocks = num_heads, num_blocks
        self.input_norm = nn.LayerNorm(2 * self.n_freq)
        self.input_proj = nn.Linear(2 * self.n_freq, d_model)
        self.blocks = nn.ModuleList([ConformerBlock(d_model, num_heads, d_ff, kernel_size, dropout)
                                     for _ in range(num_blocks)])
        self.output_norm = nn.LayerNorm(d_model)
        self.mag_head = nn.Linear(d_model, self.n_freq)
        self.phase_head = nn.Linear(d_model, self.n_freq)

    def _pack(self, sd):
        F2 = 2 * self.n_freq
        heads_w = torch.cat([sd["mag_head.weight"], sd["phase_head.weight"]], dim=0)
        heads_b = torch.cat([sd["mag_head.bias"], sd["phase_head.bias"]], dim=0)
        with ops.stage("front"):
            proj = ops.pack_linear(sd["input_proj.weight"], sd["input_proj.bias"], k_pad_to=ops.round_up(F2, 64))
        with ops.stage("tail"):
            heads = ops.pack_linear(heads_w, heads_b)
        return {"in_w": sd["input_norm.weight"].float().contiguous(), "in_b": sd["input_norm.bias"].float().contiguous(),
                "proj": proj,
                "blocks": [Fn.pack_block(Fn.sub(sd, "blocks.%d" % i), self.num_heads, i) for i in range(self.num_blocks)],
                "on_w": sd["output_norm.weight"].float().contiguous(), "on_b": sd["output_norm.bias"].float().contiguous(),
                "heads": heads}

    def _train_forward(self, noisy_real, noisy_imag):
        """train() mode: same graph built from autograd nodes whose forward AND backward are HIP kernels
        (train.LNLinearFunction, ConformerBlockFunction via ConformerBlock, PolarMaskFunction)."""
        from .. import train
        nr, ni = noisy_real.float().contiguous(), noisy_imag.float().contiguous()
        B, T, F = nr.shape
        M = B * T
        ldc = ops.round_up(2 * F, 8)
        cat = torch.empty(M, ldc, device=nr.device, dtype=torch.float32)
        ops.pack_spec(nr, ni, cat, M, F, ldc, F)
        x = train.LNLinearFunction.apply(cat, self.input_norm.weight, self.input_norm.bias, self.input_proj.weight,
                                         self.input_proj.bias)
        x = x.reshape(B, T, -1)
        for block in self.blocks:
            x = block(x)
        heads_w = torch.cat([self.mag_head.weight, self.phase_head.weight], dim=0)
        heads_b = torch.cat([self.mag_head.bias, self.phase_head.bias], dim=0)
        logits = train.LNLinearFunction.apply(x.reshape(M, -1), self.output_norm.weight, self.output_norm.bias, heads_w,
                                              heads_b)
        return train.PolarMaskFunction.apply(logits, nr, ni, math.pi / 6)

    def forward(self, noisy_real, noisy_imag):
        self._require_device(noisy_real, noisy_imag)
        if self.training or self._wants_autograd(noisy_real, noisy_imag):
            return self._train_forward(noisy_real, noisy_imag)
        pk = self._packed(self._pack)
        nr, ni = noisy_real.float().contiguous(), noisy_imag.float().contiguous()
        B, T, F = nr.shape
        M = B * T
        dev = nr.device
        ldc = ops.round_up(2 * F, 8)
        cat = torch.empty(M, ldc, device=dev, dtype=torch.float32)
        ops.pack_spec(nr, ni, cat, M, F, ldc, F)
        ld16 = pk["proj"].Kpad
        with ops.stage("front"):
            x16 = torch.zeros(M, ld16, device=dev, dtype=ops.compute_dtype())
            ops.layernorm(cat, pk["in_w"], pk["in_b"], out16=x16)            # D = 2F columns, pad stays zero
            x = ops.linear16(x16, pk["proj"], out_dtype=torch.float32)
        for bp in pk["blocks"]:
            x = Fn.block_forward(x, bp, B, T, self.num_heads)
        with ops.stage("tail"):
            h16 = Fn._ln16(x, pk["on_w"], pk["on_b"])
            logits = ops.linear16(h16, pk["heads"], out_dtype=torch.float32)   # [M, 2F] = mag | phase
        er = torch.empty(B, T, F, device=dev, dtype=torch.float32)
        ei = torch.empty(B, T, F, device=dev, dtype=torch.float32)
        mm = torch.empty(B, T, F, device=dev, dtype=torch.float32)
        ops.polar_mask(logits, logits[:, F:], B, T, F, math.pi / 6, logits.stride(0), nr=nr, ni=ni, er=er, ei=ei,
                       mmag=mm, ld_enh=F)
        return er, ei, mm

    # LayerNorm over 2F needs D passed explicitly: ops.layernorm infers D from x32.shape[1]
    # (cat has ldc >= 2F columns), so view the valid columns.


def compute_loss(model, noisy_real, noisy_imag, clean_wav, clean_real, clean_imag, fft_size=None, hop_size=None,
                 frame_size=None):
    """ConformerPipeline._compute_loss (training/conformer_pipeline.py:539-572): model forward -> iSTFT ->
    SI-SNR + 0.5 * L1 magnitude + multi-resolution STFT.  Returns (total, neg_sisnr) like the reference; `total` carries
    the autograd graph (HIP backward kernels) when the model is in train() mode."""
    from .. import train
    fft_size, hop_size = fft_size or config.FFT_SIZE, hop_size or config.HOP_SIZE
    frame_size = frame_size or config.FRAME_SIZE
    enh_real, enh_imag, _ = model(noisy_real, noisy_imag)
    T = min(enh_real.shape[1], clean_real.shape[1])
    total, aux, _ = train.EnhancerLossFunction.apply(enh_real[:, :T], enh_imag[:, :T], clean_wav, clean_real[:, :T],
                                                     clean_imag[:, :T], fft_size, hop_size, frame_size)
    return total, aux[0]


def compute_path_loss(path, noisy_wav, clean_wav):
    """The objective of ConformerPipeline._compute_loss (training/conformer_pipeline.py:539-572) on the north-star
    composition (EnhancementPath in train() mode): SI-SNR + 0.5 * L1 magnitude + multi-resolution STFT of the enhanced
    spectrum against the clean utterance.  Returns (total, neg_sisnr)."""
    from .. import train
    out = path(noisy_wav, want=("mask", "spectrum"))
    with torch.no_grad():
        # split-bf16 operands on the 16-bit matrix cores (4e-6 relative error): the form the objective's own STFTs use
        cr, ci = Fn.stft_split16(clean_wav.float().contiguous())
    total, aux, _ = train.EnhancerLossFunction.apply(out["enh_real"], out["enh_imag"], clean_wav, cr, ci, Fn.N_FFT, Fn.HOP, Fn.WIN)
    return total, aux[0]


class EnhancementPath(HipModule):
    """North-star composition: PerceptionAgent -> (pool to STFT frames) -> CPEA -> STFT ->
    [EpisodicMemory] -> MaskSynthesisAgent -> apply_mask -> iSTFT, fused in the internal
    channels-last layouts (functional.enhance_path).  Glue G1-G3: DESIGN.md."""

    def __init__(self, sample_rate=16000, use_memory=False):
        super().__init__()
        self.perception = PerceptionAgent(sample_rate=sample_rate)
        self.cpea = CorrelationPhaseEstimationAgent()
        self.msa = MaskSynthesisAgent()
        self.memory = EpisodicMemory() if use_memory else None
        self.sample_rate = sample_rate

    def _pack(self, sd):
        packs = {"pa": Fn.pack_perception(Fn.sub(sd, "perception"), self.sample_rate),
                 "cpea": Fn.pack_cpea(Fn.sub(sd, "cpea"), self.cpea.num_layers),
                 "msa": Fn.pack_msa(Fn.sub(sd, "msa"), self.msa.conformer.num_blocks, self.msa.conformer.num_heads)}
        if self.memory is not None:
            m = self.memory
            packs["memory"] = (Fn.pack_memory_params(Fn.sub(sd, "memory")), m.key_dim, m.value_dim, m.num_slots,
                               m.temperature)
        return packs

    def _pack_key(self):
        return tuple([ops.policy_key()] + [(p.data_ptr(), p._version) for p in self.parameters()] +
                     [(b.data_ptr(), b._version) for n, b in self.named_buffers() if "usage" not in n and "num_queries" not in n])

    def freeze_perception(self, flag=True):
        """Train CPEA / memory / MaskSynthesisAgent on a frozen front-end: the PerceptionAgent then runs on the inference
        kernels (its GroupNorm-only stack has no train/eval difference, so the forward is the same either way) and keeps
        none of its activations."""
        self.__dict__["_sfm_pa_frozen"] = bool(flag)
        for p in self.perception.parameters():
            p.requires_grad_(not flag)
        return self

    def _train_forward(self, waveform, want=("mask", "wave")):
        """train() mode: PerceptionAgent, pooling (glue G1), CPEA (BPTT), EpisodicMemory, MaskSynthesisAgent, apply_mask and
        iSTFT as HIP autograd nodes; glue G1-G3 as in eval().  With freeze_perception() the front-end runs without autograd."""
        from .. import train
        wave = waveform.float().contiguous()
        B, L = wave.shape
        T = 1 + L // Fn.HOP
        D = self.perception.encoder_channels if hasattr(self.perception, "encoder_channels") else 256
        with torch.no_grad():
            nr, ni = Fn.stft_split16(wave)                   # as the fused inference path: split-bf16 operands, 4e-6 relative error
        if self.__dict__.get("_sfm_pa_frozen", False):
            with torch.no_grad():
                pa = self.perception._packed(lambda sd: Fn.pack_perception(sd, self.sample_rate))     # frozen: packed once
                (rz, sz, hz), _sigma = Fn.perception_forward(wave, pa, latents=False)
                zp = torch.empty(B, T, 2 * D, device=wave.device, dtype=torch.float32)
                ops.pool_time(rz, None, zp, B, rz.shape[1], T, 2 * D, 2 * D, 2 * D, scale=sz, shift=hz)      # glue G1
        else:
            zp = train.PoolTimeFunction.apply(train.perception_latents_train(self.perception, wave), T)     # glue G1
        if zp.requires_grad:
            zp_all, z_half = train.LatentFanoutFunction.apply(zp, D)             # one fused gradient fan-in for the two consumers
        else:
            zp_all, z_half = zp, zp[..., :D]
        z_real, z_imag = zp_all[..., :D].transpose(1, 2), zp_all[..., D:].transpose(1, 2)
        cpea = self.cpea(z_half)
        bias = None
        out = {}
        if self.memory is not None:
            mem = self.memory(train.MeanTimeFunction.apply(zp[..., :self.memory.key_dim]))                # glue G2
            bias = mem["bias"]                                                                            # glue G3
            out.update(mem_bias=mem["bias"], mem_gate=mem["gate"], mem_top=mem["top_indices"], mem_sim=mem["similarity"])
        mr, mi = self.msa(z_real, z_imag, cpea, nr, ni, mag_logit_bias=bias, latents_cl=zp_all)
        er, ei = train.ComplexMulFunction.apply(nr, ni, mr, mi)
        out.update(mask_real=mr, mask_imag=mi, noisy_real=nr, noisy_imag=ni, enh_real=er, enh_imag=ei)
        if "wave" in want:
            out["enhanced"] = train.IstftFunction.apply(er, ei, L, Fn.N_FFT, Fn.HOP, Fn.WIN)
        return out

    def forward(self, waveform, want=("mask", "wave")):
        self._require_device(waveform)
        if self.training and torch.is_grad_enabled():
            return self._train_forward(waveform, want)
        packs = self._packed(self._pack)
        return Fn.enhance_path(waveform.float(), packs, H=self.msa.conformer.num_heads,
                               use_memory=self.memory is not None, want=want)


class ConformerPipeline:
    """training/conformer_pipeline.py:308-685 without its data loading: the training step (_train_epoch :484, _validate :574,
    _compute_loss :539), model I/O (:611-649) and enhance_signal (:653)."""

    def __init__(self, fs=None, device=None):
        self.fs = fs or config.SAMPLE_RATE
        self.fft_size, self.hop_size, self.frame_size = config.FFT_SIZE, config.HOP_SIZE, config.FRAME_SIZE
        if not torch.cuda.is_available():
            raise RuntimeError("ConformerPipeline (HIP build) needs an MI355X; there is no CPU fallback")
        self.device = torch.device(device or "cuda")
        self.model = None
        self.use_amp = True           # :334 `use_amp = device.type == 'cuda'`: fp16 operands + dynamic loss scale (make_optimizer)
        self.use_graph = False        # True: enhance_signal replays one hipGraph per signal length (graph.GraphedForward)
        self._graphed = None

    def _compute_loss(self, noisy_real, noisy_imag, clean_wav, clean_real, clean_imag, mr_stft_fn):
        """training/conformer_pipeline.py:539-572: model forward -> iSTFT -> SI-SNR + 0.5 * L1 magnitude +
        multi-resolution STFT; returns (total, neg_sisnr).  With the reference's default resolutions the whole objective is
        one fused autograd node (compute_loss); any other `mr_stft_fn` is composed from the stand-alone nodes."""
        default = isinstance(mr_stft_fn, MultiResolutionSTFTLoss) and mr_stft_fn._sizes() == tuple(Fn.MR_STFT)
        if default:
            return compute_loss(self.model, noisy_real, noisy_imag, clean_wav, clean_real, clean_imag, self.fft_size,
                                self.hop_size, self.frame_size)
        from .. import train
        enh_real, enh_imag, _ = self.model(noisy_real, noisy_imag)
        T = min(enh_real.shape[1], clean_real.shape[1])
        enh_real, enh_imag = enh_real[:, :T], enh_imag[:, :T]
        enh_wav = train.IstftFunction.apply(enh_real, enh_imag, clean_wav.shape[-1], self.fft_size, self.hop_size, self.frame_size)
        loss_sisnr = si_snr_loss(enh_wav, clean_wav)
        loss_mag = train.L1MagnitudeFunction.apply(enh_real, enh_imag, clean_real[:, :T], clean_imag[:, :T])
        loss_stft = mr_stft_fn(enh_wav, clean_wav)
        return loss_sisnr + 0.5 * loss_mag + loss_stft, loss_sisnr

    # -- the training step (training/conformer_pipeline.py:424-429, 442, 484-537) -----------------------------------------
    def make_optimizer(self, params=None, sync=None):
        """(optimizer, scaler) of ConformerPipeline.train: AdamW(lr 5e-4, betas (0.9, 0.98), weight_decay 0.01) with the
        gradient clip 5.0 of :514 folded in (optim.FlatAdamW), and torch.amp.GradScaler('cuda') of :442 when `use_amp`
        (optim.DynamicLossScale: the same scale / growth / backoff rule, kept on the device)."""
        from ..optim import FlatAdamW, DynamicLossScale
        opt = FlatAdamW(list(params if params is not None else self.model.parameters()), lr=5e-4, betas=(0.9, 0.98),
                        weight_decay=0.01, max_norm=5.0, sync=sync)
        return opt, (DynamicLossScale(self.device) if self.use_amp else None)

    def _train_epoch(self, loader, optimizer, mr_stft_fn, scaler):
        """training/conformer_pipeline.py:484-537 with the reference's call order; `optimizer` = optim.FlatAdamW, `scaler` =
        optim.DynamicLossScale or None.  What the reference does on the host per step happens on the device here: the
        `torch.isnan(loss) or torch.isinf(loss): continue` of :509 is the optimiser's skip flag, clip_grad_norm_(5.0) its clip
        coefficient, `loss.item()` a device-side accumulation that is read ONCE at the end of the epoch."""
        self.model.train()
        acc = torch.zeros(3, device=self.device, dtype=torch.float64)           # sum loss, sum SI-SNR, batches counted
        for noisy, clean in loader:
            noisy = noisy.to(self.device, non_blocking=True)
            clean = clean.to(self.device, non_blocking=True)
            optimizer.zero_grad()
            noisy_real, noisy_imag = batch_stft(noisy, self.fft_size, self.hop_size, self.frame_size)
            clean_real, clean_imag = batch_stft(clean, self.fft_size, self.hop_size, self.frame_size)
            loss, neg_sisnr = self._compute_loss(noisy_real, noisy_imag, clean, clean_real, clean_imag, mr_stft_fn)
            if scaler is not None:
                scaler.scale(loss).backward()
                scaler.unscale_(optimizer)
                scaler.step(optimizer, loss=loss)            # NaN / Inf loss -> skipped on the device (:509), clip 5.0 (:514)
                scaler.update()
            else:
                loss.backward()
                optimizer.step(loss=loss)
            with torch.no_grad():
                ok = torch.isfinite(loss.detach()).double()
                acc += torch.stack([torch.nan_to_num(loss.detach().double()) * ok,
                                    -torch.nan_to_num(neg_sisnr.detach().double()) * ok, ok])
        a = acc.cpu()
        n = max(float(a[2]), 1.0)
        return float(a[0]) / n, float(a[1]) / n

    @torch.no_grad()
    def _validate(self, loader, mr_stft_fn):
        """training/conformer_pipeline.py:574-609: the objective in eval() mode, averaged over the batches."""
        self.model.eval()
        acc = torch.zeros(3, device=self.device, dtype=torch.float64)
        for noisy, clean in loader:
            noisy, clean = noisy.to(self.device, non_blocking=True), clean.to(self.device, non_blocking=True)
            noisy_real, noisy_imag = batch_stft(noisy, self.fft_size, self.hop_size, self.frame_size)
            clean_real, clean_imag = batch_stft(clean, self.fft_size, self.hop_size, self.frame_size)
            loss, neg_sisnr = self._compute_loss(noisy_real, noisy_imag, clean, clean_real, clean_imag, mr_stft_fn)
            ok = torch.isfinite(loss).double()                                  # :601 non-finite batches are not counted
            acc += torch.stack([torch.nan_to_num(loss.double()) * ok, -torch.nan_to_num(neg_sisnr.double()) * ok, ok])
        a = acc.cpu()
        n = max(float(a[2]), 1.0)
        return float(a[0]) / n, float(a[1]) / n

    # -- model I/O (training/conformer_pipeline.py:611-649): {'model_state', 'model_class'} checkpoints ----------------
    def _checkpoint(self):
        return {"model_state": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}, "model_class": "SpeechEnhancer"}

    def _save_best(self):
        os.makedirs(config.MODEL_DIR, exist_ok=True)
        torch.save(self._checkpoint(), os.path.join(config.MODEL_DIR, "best_conformer.pt"))

    def save_model(self, filename="conformer_final.pt"):
        if self.model is None:
            return
        os.makedirs(config.MODEL_DIR, exist_ok=True)
        path = os.path.join(config.MODEL_DIR, filename)
        torch.save(self._checkpoint(), path)
        print(f"  + Model saved: {path}")

    def load_model(self, path=None):
        if path is None:
            path = os.path.join(config.MODEL_DIR, "conformer_final.pt")
            if not os.path.exists(path):
                path = os.path.join(config.MODEL_DIR, "best_conformer.pt")
        # tensors only: a checkpoint is {'model_state': state_dict, 'model_class': str} (the reference unpickles with
        # weights_only=False; nothing in its checkpoints needs that)
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        self.model = SpeechEnhancer(n_freq=self.fft_size // 2 + 1, d_model=256, num_blocks=4, num_heads=4, d_ff=1024,
                                    kernel_size=31, dropout=0.15)
        self.model.load_state_dict(ckpt["model_state"])
        self.model.to(self.device).eval()
        self._graphed = None          # a captured hipGraph replays the OLD model's packed weights
        print(f"  + Conformer loaded: {path}")

    @torch.no_grad()
    def enhance_signal(self, noisy_signal):
        if self.model is None:
            raise RuntimeError("No model loaded.")
        self.model.eval()
        x = torch.from_numpy(np.asarray(noisy_signal, dtype=np.float32)).unsqueeze(0).to(self.device)
        if self.use_graph:
            if self._graphed is None:
                from ..graph import GraphedForward
                self._graphed = GraphedForward(self._enhance_device)
            y = self._graphed(x)
        else:
            y = self._enhance_device(x)
        return y.squeeze(0).cpu().numpy()

    def _enhance_device(self, x):
        nr, ni = batch_stft(x, self.fft_size, self.hop_size, self.frame_size)
        er, ei, _ = self.model(nr, ni)
        return batch_istft(er, ei, self.fft_size, self.hop_size, self.frame_size, length=x.shape[-1])
