"""ORACLE — test infrastructure only.  NOT part of the product path.

CPU fp32 restatement of the reference's hot path (SincNet front-end ->
Complex-Conformer mask synthesis over framed STFT bins), written from the
reference's algorithm as plain functions over a state_dict of CPU tensors.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product (sincformer_metacog_speech_enhancement_amd) never does
and raises if its HIP extension is missing.

The arithmetic of the reference lives in a third-party dependency (PyTorch
aten, `torch>=2.0.0`, reference requirements.txt:1; oracle run = torch
2.10.0 CPU).  High-level modules the reference calls (nn.MultiheadAttention,
nn.LSTM, nn.GroupNorm/LayerNorm/BatchNorm1d, F.glu, torch.stft/istft,
adaptive pooling) are restated here from primitives (matmul, conv1d,
elementwise); STFT/iSTFT are restated as explicit DFT matrix products.

Pinning: the reference's own tests hold no numeric vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself run in the build container: tests/golden/*.npz, produced by
tests/golden/make_golden.py (which imports /root/reference there) and checked
by tests/test_oracle_golden.py to <= 2e-5.

Every function cites the reference file:line it follows
(paths relative to the reference root).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# constants mirrored from config.py:17-21,93-98,105-106
# ----------------------------------------------------------------------------
FFT_SIZE = 256
HOP_SIZE = 80
FRAME_SIZE = 160
N_FREQ = FFT_SIZE // 2 + 1


def _t(x):
    if isinstance(x, torch.Tensor):
        if x.requires_grad:                      # keep the autograd graph (gradient references in tests/)
            return x
        return x.detach().to(torch.float32).cpu()
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.float32)


def sub(sd, prefix):
    """state_dict view with `prefix.` stripped."""
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def gelu(x):
    """exact (erf) GELU — F.gelu default, agents/perception.py:234."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def swish(x):
    """x * sigmoid(x) — models/conformer.py:45,119."""
    return x * torch.sigmoid(x)


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim (biased variance)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def group_norm(x, groups, w, b, eps=1e-5):
    """nn.GroupNorm on [B, C, L]: stats over (C/groups x L) per (b, group)."""
    B, C, L = x.shape
    xg = x.reshape(B, groups, (C // groups) * L)
    mu = xg.mean(dim=-1, keepdim=True)
    var = ((xg - mu) ** 2).mean(dim=-1, keepdim=True)
    xn = ((xg - mu) / torch.sqrt(var + eps)).reshape(B, C, L)
    return xn * w.view(1, C, 1) + b.view(1, C, 1)


def batch_norm_eval(x, w, b, rm, rv, eps=1e-5):
    """nn.BatchNorm1d in eval mode on [B, C, T]."""
    return (x - rm.view(1, -1, 1)) / torch.sqrt(rv.view(1, -1, 1) + eps) * w.view(1, -1, 1) + b.view(1, -1, 1)


def linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


# ----------------------------------------------------------------------------
# SincConv1d  (agents/perception.py:23-118)
# ----------------------------------------------------------------------------
def sinc_init(out_channels, kernel_size, sample_rate, min_low_hz=50, min_band_hz=50):
    """Parameters/buffers as SincConv1d.__init__ builds them
    (agents/perception.py:35-77).  Returns dict low_hz_, band_hz_, window, n_."""
    if kernel_size % 2 == 0:
        kernel_size += 1
    low_hz = min_low_hz
    high_hz = sample_rate / 2 - min_band_hz
    erb_low = 21.4 * math.log10(1 + low_hz / 228.7)
    erb_high = 21.4 * math.log10(1 + high_hz / 228.7)
    erb_points = np.linspace(erb_low, erb_high, out_channels + 1)
    hz_points = 228.7 * (10 ** (erb_points / 21.4) - 1)
    low = torch.Tensor(hz_points[:-1]).view(-1, 1)
    band = torch.Tensor(np.diff(hz_points)).view(-1, 1)
    n = torch.linspace(0, kernel_size - 1, kernel_size)
    window = 0.54 - 0.46 * torch.cos(2 * math.pi * n / kernel_size)
    half = (kernel_size - 1) / 2.0
    n_ = 2 * math.pi * torch.arange(-half, 0).view(1, -1) / sample_rate
    return {"low_hz_": low, "band_hz_": band, "window": window, "n_": n_}


def sinc_filters(low_hz_, band_hz_, window, n_, sample_rate, min_low_hz=50, min_band_hz=50):
    """[C, K] filter bank — agents/perception.py:88-112 (including the
    double division by sample_rate)."""
    low_hz_, band_hz_, window, n_ = _t(low_hz_), _t(band_hz_), _t(window), _t(n_)
    low = min_low_hz + torch.abs(low_hz_)
    high = torch.clamp(low + min_band_hz + torch.abs(band_hz_), max=sample_rate / 2.0)
    f_low = low / sample_rate
    f_high = high / sample_rate
    left = (torch.sin(f_high * n_) - torch.sin(f_low * n_)) / (n_ / 2.0 + 1e-8)
    center = 2 * (f_high - f_low)
    right = torch.flip(left, dims=[1])
    bp = torch.cat([left, center, right], dim=1)
    bp = bp * window
    bp = bp / (bp.abs().sum(dim=1, keepdim=True) + 1e-8)
    return bp


def sinc_conv(wave, filters):
    """F.conv1d(stride 1, pad K//2) — agents/perception.py:115-118.
    wave [B, L] or [B,1,L] -> [B, C, L]."""
    wave = _t(wave)
    if wave.dim() == 2:
        wave = wave.unsqueeze(1)
    K = filters.shape[1]
    return F.conv1d(wave, filters.unsqueeze(1), stride=1, padding=K // 2)


# ----------------------------------------------------------------------------
# PerceptionAgent (agents/perception.py:132-251)
# ----------------------------------------------------------------------------
def _res_block(x, sd, groups):
    """_make_block/_ResidualBlock — agents/perception.py:121-129,192-206."""
    m = F.conv1d(x, sd["main.0.weight"], sd["main.0.bias"], stride=2, padding=3)
    m = gelu(group_norm(m, groups, sd["main.1.weight"], sd["main.1.bias"]))
    m = F.conv1d(m, sd["main.3.weight"], sd["main.3.bias"], stride=1, padding=1)
    m = group_norm(m, groups, sd["main.4.weight"], sd["main.4.bias"])
    s = F.conv1d(x, sd["skip.0.weight"], sd["skip.0.bias"], stride=2)
    s = group_norm(s, groups, sd["skip.1.weight"], sd["skip.1.bias"])
    return gelu(m + s)


def perception_forward(sd, wave, sample_rate, return_stages=False):
    """PerceptionAgent.forward — agents/perception.py:216-251.
    Returns z_real, z_imag [B, D, L/16], sigma [B, 1, L/16]."""
    sd = {k: _t(v) for k, v in sd.items()}
    stages = {}
    filt = sinc_filters(sd["sinc_conv.low_hz_"], sd["sinc_conv.band_hz_"],
                        sd["sinc_conv.window"], sd["sinc_conv.n_"], sample_rate)
    x = sinc_conv(wave, filt)
    stages["sinc"] = x
    x = gelu(group_norm(x, 8, sd["sinc_norm.weight"], sd["sinc_norm.bias"]))
    stages["sinc_act"] = x
    for i in range(3):
        bsd = sub(sd, "conv_blocks.%d" % i)
        out_ch = bsd["main.0.weight"].shape[0]
        x = _res_block(x, bsd, min(16, out_ch))
        stages["block%d" % i] = x
    x = F.conv1d(x, sd["downsample.0.weight"], sd["downsample.0.bias"], stride=2, padding=2)
    x = gelu(group_norm(x, 16, sd["downsample.1.weight"], sd["downsample.1.bias"]))
    stages["down"] = x
    zr = F.conv1d(x, sd["real_proj.0.weight"], sd["real_proj.0.bias"])
    zr = group_norm(zr, 16, sd["real_proj.1.weight"], sd["real_proj.1.bias"])
    zi = F.conv1d(x, sd["imag_proj.0.weight"], sd["imag_proj.0.bias"])
    zi = group_norm(zi, 16, sd["imag_proj.1.weight"], sd["imag_proj.1.bias"])
    u = gelu(F.conv1d(x, sd["uncertainty_head.0.weight"], sd["uncertainty_head.0.bias"], padding=1))
    lv = F.conv1d(u, sd["uncertainty_head.2.weight"], sd["uncertainty_head.2.bias"])
    sigma = torch.exp(0.5 * torch.clamp(lv, -10, 10))
    if return_stages:
        return zr, zi, sigma, stages
    return zr, zi, sigma


# ----------------------------------------------------------------------------
# STFT / iSTFT  (training/conformer_pipeline.py:196-211, torch.stft semantics:
# center=True reflect pad n_fft/2, periodic Hann(win) zero-padded to n_fft
# centred, onesided, not normalised)
# ----------------------------------------------------------------------------
def hann_periodic(n):
    k = torch.arange(n, dtype=torch.float64)
    return 0.5 - 0.5 * torch.cos(2.0 * math.pi * k / n)


def _padded_window(n_fft, win_length):
    w = torch.zeros(n_fft, dtype=torch.float64)
    left = (n_fft - win_length) // 2
    w[left:left + win_length] = hann_periodic(win_length)
    return w


def stft(wave, n_fft=FFT_SIZE, hop=HOP_SIZE, win_length=FRAME_SIZE):
    """batch_stft — returns real, imag [B, T, n_fft/2+1] (fp32).
    DFT evaluated as a float64-twiddle matrix product applied in fp32."""
    wave = _t(wave)
    B, L = wave.shape
    pad = n_fft // 2
    xp = F.pad(wave.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    T = 1 + L // hop
    idx = (torch.arange(T).unsqueeze(1) * hop + torch.arange(n_fft).unsqueeze(0))
    frames = xp[:, idx]                                     # [B, T, n_fft]
    w = _padded_window(n_fft, win_length)
    n = torch.arange(n_fft, dtype=torch.float64).unsqueeze(1)
    f = torch.arange(n_fft // 2 + 1, dtype=torch.float64).unsqueeze(0)
    ang = 2.0 * math.pi * ((n * f) % n_fft) / n_fft
    cr = (w.unsqueeze(1) * torch.cos(ang)).to(torch.float32)
    ci = (-w.unsqueeze(1) * torch.sin(ang)).to(torch.float32)
    return frames @ cr, frames @ ci


def istft(real, imag, length, n_fft=FFT_SIZE, hop=HOP_SIZE, win_length=FRAME_SIZE):
    """batch_istft — torch.istft(center=True, length=length): irfft per frame,
    x window, overlap-add, / sum(window^2), trim n_fft/2.  [B,T,F] -> [B,length]."""
    real, imag = _t(real), _t(imag)
    B, T, Fq = real.shape
    w = _padded_window(n_fft, win_length)
    n = torch.arange(n_fft, dtype=torch.float64).unsqueeze(0)
    f = torch.arange(Fq, dtype=torch.float64).unsqueeze(1)
    ang = 2.0 * math.pi * ((f * n) % n_fft) / n_fft
    coef = torch.full((Fq, 1), 2.0, dtype=torch.float64)
    coef[0, 0] = 1.0
    if n_fft % 2 == 0:
        coef[Fq - 1, 0] = 1.0
    br = (coef * torch.cos(ang) / n_fft * w.unsqueeze(0))          # [F, n_fft]
    bi = (-coef * torch.sin(ang) / n_fft * w.unsqueeze(0))
    bi[0, :] = 0.0                    # irfft ignores imag of DC / Nyquist
    if n_fft % 2 == 0:
        bi[Fq - 1, :] = 0.0
    frames = real @ br.to(torch.float32) + imag @ bi.to(torch.float32)   # [B,T,n_fft]
    full = n_fft + hop * (T - 1)
    y = torch.zeros(B, full, dtype=torch.float32)
    env = torch.zeros(full, dtype=torch.float64)
    w2 = w * w
    for t in range(T):
        y[:, t * hop:t * hop + n_fft] += frames[:, t]
        env[t * hop:t * hop + n_fft] += w2
    start = n_fft // 2
    y = y[:, start:start + length]
    e = env[start:start + length].to(torch.float32)
    if y.shape[1] < length:
        y = F.pad(y, (0, length - y.shape[1]))
        e = F.pad(e, (0, length - e.shape[0]), value=1.0)
    return y / e


# ----------------------------------------------------------------------------
# Conformer (models/conformer.py)
# ----------------------------------------------------------------------------
def ffn(x, sd):
    """FeedForwardModule.forward eval — models/conformer.py:41-49."""
    h = layer_norm(x, sd["layer_norm.weight"], sd["layer_norm.bias"])
    h = swish(linear(h, sd["linear1.weight"], sd["linear1.bias"]))
    h = linear(h, sd["linear2.weight"], sd["linear2.bias"])
    return x + 0.5 * h


def mhsa(x, sd, num_heads):
    """MultiHeadSelfAttention.forward eval — models/conformer.py:66-71
    (nn.MultiheadAttention batch_first, no mask, scale 1/sqrt(head_dim))."""
    B, T, D = x.shape
    hd = D // num_heads
    h = layer_norm(x, sd["layer_norm.weight"], sd["layer_norm.bias"])
    qkv = linear(h, sd["attention.in_proj_weight"], sd["attention.in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(B, T, num_heads, hd).transpose(1, 2)
    k = k.reshape(B, T, num_heads, hd).transpose(1, 2)
    v = v.reshape(B, T, num_heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, D)
    o = linear(o, sd["attention.out_proj.weight"], sd["attention.out_proj.bias"])
    return x + o


def batch_norm_train(x, w, b, eps=1e-5):
    """nn.BatchNorm1d in training mode on [B, C, T]: biased batch statistics over (B, T)."""
    mu = x.mean(dim=(0, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(0, 2), keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w.view(1, -1, 1) + b.view(1, -1, 1)


def conv_module(x, sd, bn_train=False):
    """ConvolutionModule.forward — models/conformer.py:101-128 (bn_train: BatchNorm batch statistics,
    i.e. module.train() with dropout p = 0)."""
    B, T, D = x.shape
    h = layer_norm(x, sd["layer_norm.weight"], sd["layer_norm.bias"]).transpose(1, 2)
    h = F.conv1d(h, sd["pointwise1.weight"], sd["pointwise1.bias"])
    a, g = h.split(D, dim=1)
    h = a * torch.sigmoid(g)
    ksz = sd["depthwise.weight"].shape[-1]
    h = F.conv1d(h, sd["depthwise.weight"], sd["depthwise.bias"], padding=(ksz - 1) // 2, groups=D)
    if bn_train:
        h = batch_norm_train(h, sd["batch_norm.weight"], sd["batch_norm.bias"])
    else:
        h = batch_norm_eval(h, sd["batch_norm.weight"], sd["batch_norm.bias"],
                            sd["batch_norm.running_mean"], sd["batch_norm.running_var"])
    h = swish(h)
    h = F.conv1d(h, sd["pointwise2.weight"], sd["pointwise2.bias"]).transpose(1, 2)
    return x + h


def conformer_block(x, sd, num_heads, bn_train=False):
    """ConformerBlock.forward — models/conformer.py:145-151."""
    x = ffn(x, sub(sd, "ff1"))
    x = mhsa(x, sub(sd, "mhsa"), num_heads)
    x = conv_module(x, sub(sd, "conv"), bn_train)
    x = ffn(x, sub(sd, "ff2"))
    return layer_norm(x, sd["final_norm.weight"], sd["final_norm.bias"])


def _num_blocks(sd, prefix="blocks."):
    ids = {int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)}
    return (max(ids) + 1) if ids else 0


def complex_conformer_forward(sd, stft_real, stft_imag, num_heads, bn_train=False):
    """ComplexConformer.forward — models/conformer.py:193-228."""
    sd = {k: _t(v) for k, v in sd.items()}
    x = torch.cat([_t(stft_real), _t(stft_imag)], dim=-1)
    n_freq = x.shape[-1] // 2
    x = linear(x, sd["input_proj.weight"], sd["input_proj.bias"])
    skip = x
    for i in range(_num_blocks(sd)):
        x = conformer_block(x, sub(sd, "blocks.%d" % i), num_heads, bn_train)
    x = x + skip
    x = linear(x, sd["output_proj.weight"], sd["output_proj.bias"])
    return x[..., :n_freq], x[..., n_freq:]


def apply_mask(sr, si, mr, mi):
    """ComplexConformer.apply_mask — models/conformer.py:230-245."""
    sr, si, mr, mi = _t(sr), _t(si), _t(mr), _t(mi)
    return mr * sr - mi * si, mr * si + mi * sr


# ----------------------------------------------------------------------------
# MaskSynthesisAgent (agents/msa.py:106-174)
# ----------------------------------------------------------------------------
def msa_forward(sd, z_real, z_imag, cpea, noisy_real, noisy_imag, num_heads=4,
                mag_logit_bias=None, return_logits=False, bn_train=False):
    """MaskSynthesisAgent.forward.  z_* [B, D, T]; cpea dict of [B, T, 64];
    noisy_* [B, T, 129].  mag_logit_bias (optional [B,129]) is the build-defined
    injection point of the episodic-memory bias (DESIGN.md glue G3): added to
    the magnitude logit before the sigmoid (agents/msa.py:166)."""
    sd = {k: _t(v) for k, v in sd.items()}
    z_r = _t(z_real).transpose(1, 2)
    z_i = _t(z_imag).transpose(1, 2)
    nr, ni = _t(noisy_real), _t(noisy_imag)
    B, T, _ = z_r.shape
    mag = torch.sqrt(nr ** 2 + ni ** 2 + 1e-8)
    nf = torch.log1p(mag) / mag
    fused = torch.cat([z_r, z_i, _t(cpea["rho_s"]), _t(cpea["rho_n"]), _t(cpea["phi1"]),
                       _t(cpea["phi2"]), nr * nf, ni * nf], dim=-1)
    h = linear(fused, sd["fusion.0.weight"], sd["fusion.0.bias"])
    h = gelu(layer_norm(h, sd["fusion.1.weight"], sd["fusion.1.bias"]))
    h = linear(h, sd["fusion.3.weight"], sd["fusion.3.bias"])
    h = layer_norm(h, sd["fusion.4.weight"], sd["fusion.4.bias"])
    d_half = h.shape[-1] // 2
    mr, mi = complex_conformer_forward(sub(sd, "conformer"), h[..., :d_half], h[..., d_half:], num_heads, bn_train)
    lm = linear(gelu(linear(mr, sd["mask_proj_real.0.weight"], sd["mask_proj_real.0.bias"])),
                sd["mask_proj_real.2.weight"], sd["mask_proj_real.2.bias"])
    lp = linear(gelu(linear(mi, sd["mask_proj_imag.0.weight"], sd["mask_proj_imag.0.bias"])),
                sd["mask_proj_imag.2.weight"], sd["mask_proj_imag.2.bias"])
    if mag_logit_bias is not None:
        lm = lm + _t(mag_logit_bias).unsqueeze(1)
    mmag = torch.sigmoid(lm)
    mph = torch.tanh(lp) * (3.14159 / 8.0)          # literal constant, agents/msa.py:168
    out = (mmag * torch.cos(mph), mmag * torch.sin(mph))
    if return_logits:
        return out + (lm, lp)
    return out


# ----------------------------------------------------------------------------
# CPEA (agents/cpea.py:79-112) — nn.LSTM restated (gate order i,f,g,o)
# ----------------------------------------------------------------------------
def _lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    B, T, _ = x.shape
    H = w_hh.shape[1]
    xg = x @ w_ih.t() + b_ih + b_hh
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    out = torch.zeros(B, T, H)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xg[:, t] + h @ w_hh.t()
        i, f, gg, o = g.split(H, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


def bilstm(x, sd, num_layers):
    for l in range(num_layers):
        f = _lstm_dir(x, sd["weight_ih_l%d" % l], sd["weight_hh_l%d" % l],
                      sd["bias_ih_l%d" % l], sd["bias_hh_l%d" % l], False)
        r = _lstm_dir(x, sd["weight_ih_l%d_reverse" % l], sd["weight_hh_l%d_reverse" % l],
                      sd["bias_ih_l%d_reverse" % l], sd["bias_hh_l%d_reverse" % l], True)
        x = torch.cat([f, r], dim=-1)
    return x


def cpea_forward(sd, z_t, input_dim=256, num_layers=2):
    """CorrelationPhaseEstimationAgent.forward eval — agents/cpea.py:79-112
    (including the last-dim != input_dim auto-transpose, :94-96)."""
    sd = {k: _t(v) for k, v in sd.items()}
    z_t = _t(z_t)
    if z_t.dim() == 3 and z_t.shape[-1] != input_dim:
        z_t = z_t.transpose(1, 2)
    h = bilstm(z_t, sub(sd, "lstm"), num_layers)
    return {
        "rho_s": torch.sigmoid(linear(h, sd["rho_s_head.0.weight"], sd["rho_s_head.0.bias"])),
        "rho_n": torch.sigmoid(linear(h, sd["rho_n_head.0.weight"], sd["rho_n_head.0.bias"])),
        "phi1": torch.tanh(linear(h, sd["phi1_head.0.weight"], sd["phi1_head.0.bias"])) * math.pi,
        "phi2": torch.tanh(linear(h, sd["phi2_head.0.weight"], sd["phi2_head.0.bias"])) * math.pi,
    }


# ----------------------------------------------------------------------------
# EpisodicMemory (agents/memory.py:95-148), eval mode (no usage side effects)
# ----------------------------------------------------------------------------
def memory_forward(sd, emb, temperature=1.0):
    sd = {k: _t(v) for k, v in sd.items()}
    emb = _t(emb)
    q = linear(emb, sd["key_proj.0.weight"], sd["key_proj.0.bias"])
    q = gelu(layer_norm(q, sd["key_proj.1.weight"], sd["key_proj.1.bias"]))
    q = linear(q, sd["key_proj.3.weight"], sd["key_proj.3.bias"])
    qn = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    kn = sd["keys"] / sd["keys"].norm(dim=-1, keepdim=True).clamp_min(1e-12)
    sim = (qn @ kn.t()) / temperature
    att = torch.softmax(sim, dim=-1)
    ret = att @ sd["values"]
    bias = torch.tanh(linear(ret, sd["value_proj.0.weight"], sd["value_proj.0.bias"]))
    gate = torch.sigmoid(linear(torch.cat([q, ret], dim=-1), sd["gate.0.weight"], sd["gate.0.bias"]))
    return {"bias": bias * gate, "gate": gate, "top_indices": sim.argmax(dim=-1),
            "similarity": sim.max(dim=-1)[0]}


# ----------------------------------------------------------------------------
# SpeechEnhancer (training/conformer_pipeline.py:260-298)
# ----------------------------------------------------------------------------
def speech_enhancer_forward(sd, noisy_real, noisy_imag, num_heads=4, bn_train=False):
    sd = {k: _t(v) for k, v in sd.items()}
    nr, ni = _t(noisy_real), _t(noisy_imag)
    x = torch.cat([nr, ni], dim=-1)
    x = layer_norm(x, sd["input_norm.weight"], sd["input_norm.bias"])
    x = linear(x, sd["input_proj.weight"], sd["input_proj.bias"])
    for i in range(_num_blocks(sd)):
        x = conformer_block(x, sub(sd, "blocks.%d" % i), num_heads, bn_train)
    x = layer_norm(x, sd["output_norm.weight"], sd["output_norm.bias"])
    mmag = torch.sigmoid(linear(x, sd["mag_head.weight"], sd["mag_head.bias"]))
    mph = torch.tanh(linear(x, sd["phase_head.weight"], sd["phase_head.bias"])) * (math.pi / 6)
    mr, mi = mmag * torch.cos(mph), mmag * torch.sin(mph)
    return mr * nr - mi * ni, mr * ni + mi * nr, mmag


# ----------------------------------------------------------------------------
# Losses (training/conformer_pipeline.py:52-108, 539-572)
# ----------------------------------------------------------------------------
def si_snr_loss(est, tgt):
    est, tgt = _t(est), _t(tgt)
    tgt = tgt - tgt.mean(dim=-1, keepdim=True)
    est = est - est.mean(dim=-1, keepdim=True)
    dot = (est * tgt).sum(dim=-1, keepdim=True)
    s_energy = (tgt ** 2).sum(dim=-1, keepdim=True) + 1e-8
    s_target = dot * tgt / s_energy
    e_noise = est - s_target
    si = 10 * torch.log10((s_target ** 2).sum(dim=-1) / ((e_noise ** 2).sum(dim=-1) + 1e-8) + 1e-8)
    return -si.mean()


def mr_stft_loss(pred, tgt, fft_sizes=(256, 512, 1024), hop_sizes=(64, 128, 256), win_sizes=(256, 512, 1024)):
    loss = torch.tensor(0.0)
    for nf, hp, wn in zip(fft_sizes, hop_sizes, win_sizes):
        pr, pi = stft(pred, nf, hp, wn)
        tr, ti = stft(tgt, nf, hp, wn)
        pm = torch.sqrt(pr ** 2 + pi ** 2)
        tm = torch.sqrt(tr ** 2 + ti ** 2)
        sc = torch.linalg.norm((tm - pm).reshape(-1)) / (torch.linalg.norm(tm.reshape(-1)) + 1e-8)
        lm = (torch.log(pm + 1e-8) - torch.log(tm + 1e-8)).abs().mean()
        loss = loss + sc + lm
    return loss / len(fft_sizes)


def enhancer_loss(sd, noisy_wav, clean_wav, num_heads=4, bn_train=False):
    """ConformerPipeline._compute_loss — training/conformer_pipeline.py:539-572.
    Returns total, neg_sisnr, enh_wav."""
    nr, ni = stft(noisy_wav)
    cr, ci = stft(clean_wav)
    er, ei, _ = speech_enhancer_forward(sd, nr, ni, num_heads, bn_train)
    return spectrum_objective(er, ei, clean_wav, cr, ci)


def spectrum_objective(er, ei, clean_wav, cr, ci):
    """the three terms of training/conformer_pipeline.py:553-572 on an enhanced spectrum: total, neg_sisnr, enh_wav"""
    enh = istft(er, ei, _t(clean_wav).shape[-1])
    l_si = si_snr_loss(enh, clean_wav)
    l_mag = (torch.sqrt(er ** 2 + ei ** 2 + 1e-8) - torch.sqrt(cr ** 2 + ci ** 2 + 1e-8)).abs().mean()
    l_st = mr_stft_loss(enh, clean_wav)
    return l_si + 0.5 * l_mag + l_st, l_si, enh


def path_loss(sds, noisy_wav, clean_wav, sample_rate, use_memory=False, num_heads=4, bn_train=False):
    """the same objective on the north-star composition (enhance_path): total, neg_sisnr, enh_wav"""
    out = enhance_path(sds, noisy_wav, sample_rate, use_memory, num_heads, bn_train)
    cr, ci = stft(clean_wav)
    return spectrum_objective(out["enh_real"], out["enh_imag"], clean_wav, cr, ci)


# ----------------------------------------------------------------------------
# Glue (build-defined; SURVEY.md F3/H1, DESIGN.md "Glue")
# ----------------------------------------------------------------------------
def pool_latents(z, T):
    """G1: adaptive average pool [B, D, T_pa] -> [B, D, T]
    (window i = [floor(i*Tin/T), ceil((i+1)*Tin/T)) )."""
    z = _t(z)
    Tin = z.shape[-1]
    out = torch.empty(z.shape[0], z.shape[1], T)
    for i in range(T):
        s = (i * Tin) // T
        e = -((-(i + 1) * Tin) // T)
        out[:, :, i] = z[:, :, s:e].mean(dim=-1)
    return out


def enhance_path(sds, wave, sample_rate, use_memory=False, num_heads=4, bn_train=False):
    """North-star composition with the glue G1-G3:
      PA -> pool to T frames -> CPEA(z_real) -> STFT -> [memory] -> MSA ->
      apply_mask -> iSTFT.  sds: dict of state_dicts pa/cpea/msa[/memory]."""
    wave = _t(wave)
    L = wave.shape[-1]
    T = 1 + L // HOP_SIZE
    zr, zi, sigma = perception_forward(sds["pa"], wave, sample_rate)
    zr_t, zi_t = pool_latents(zr, T), pool_latents(zi, T)
    cpea = cpea_forward(sds["cpea"], zr_t.transpose(1, 2))
    nr, ni = stft(wave)
    bias = None
    mem = None
    if use_memory:
        mem = memory_forward(sds["memory"], zr_t.mean(dim=-1))     # G2: key = mean over frames
        bias = mem["bias"]
    mr, mi = msa_forward(sds["msa"], zr_t, zi_t, cpea, nr, ni, num_heads, mag_logit_bias=bias, bn_train=bn_train)
    er, ei = apply_mask(nr, ni, mr, mi)
    wav = istft(er, ei, L)
    return {"mask_real": mr, "mask_imag": mi, "enh_real": er, "enh_imag": ei, "enhanced": wav,
            "z_real": zr, "z_imag": zi, "sigma": sigma, "noisy_real": nr, "noisy_imag": ni,
            "cpea": cpea, "memory": mem}


# ----------------------------------------------------------------------------
# Quality metrics right after the path: evaluation/ssnr.py:26-92, evaluation/stoi.py:53-99
# (numpy float64, one utterance at a time, like the reference)
# ----------------------------------------------------------------------------
def ssnr(clean, enhanced, frame_size=FRAME_SIZE, hop_size=HOP_SIZE, upper_bound=35.0, lower_bound=-10.0):
    n = min(len(clean), len(enhanced))
    c = np.asarray(clean[:n], dtype=np.float64)
    e = np.asarray(enhanced[:n], dtype=np.float64)
    num_frames = (n - frame_size) // hop_size + 1
    if num_frames < 1:
        return 0.0
    vals = []
    for k in range(num_frames):
        cf = c[k * hop_size:k * hop_size + frame_size]
        ef = e[k * hop_size:k * hop_size + frame_size]
        sp = float(np.sum(cf ** 2))
        ep = float(np.sum((cf - ef) ** 2))
        if sp < 1e-10:
            continue
        v = upper_bound if ep < 1e-10 else 10.0 * math.log10(sp / ep)
        vals.append(min(max(v, lower_bound), upper_bound))
    return float(np.mean(vals)) if vals else 0.0


def stoi_simplified(clean, enhanced, fs):
    n = min(len(clean), len(enhanced))
    c = np.asarray(clean[:n], dtype=np.float64)
    e = np.asarray(enhanced[:n], dtype=np.float64)
    frame_len = int(0.0256 * fs)
    hop = frame_len // 2
    c = c / (np.sqrt(np.mean(c ** 2)) + 1e-10)
    e = e / (np.sqrt(np.mean(e ** 2)) + 1e-10)
    num_frames = (len(c) - frame_len) // hop + 1
    if num_frames < 1:
        return 0.0
    w = np.hanning(frame_len)
    corrs = []
    for k in range(num_frames):
        cs = np.abs(np.fft.rfft(c[k * hop:k * hop + frame_len] * w))
        es = np.abs(np.fft.rfft(e[k * hop:k * hop + frame_len] * w))
        ce = np.sqrt(np.sum(cs ** 2) + 1e-10)
        en = es / (np.sqrt(np.sum(es ** 2)) + 1e-10) * ce
        corr = np.sum(cs * en) / (np.sqrt(np.sum(cs ** 2) * np.sum(en ** 2)) + 1e-10)
        corrs.append(min(max(corr, -1.0), 1.0))
    return float(min(max(np.mean(corrs), 0.0), 1.0))


# ----------------------------------------------------------------------------
# MetacognitiveArbitrationAgent (agents/maa.py:26-143) and VectorQuantizer (models/vq.py:28-122)
# ----------------------------------------------------------------------------
def maa_forward(sd, sigma, training=False, momentum=0.1):
    """agents/maa.py:70-124.  sigma [B,1,T] or [B,T].  Returns (dict, new_stats) where new_stats = (running_mean,
    running_var, num_updates) after the train-mode EMA update of :126-135 (unbiased batch variance), else the inputs."""
    sigma = _t(sigma)
    if sigma.dim() == 3:
        sigma = sigma.squeeze(1)
    rm, rv, nu = _t(sd["running_mean"]), _t(sd["running_var"]), sd["num_updates"]
    if training:
        with torch.no_grad():
            rm = (1 - momentum) * rm + momentum * sigma.mean()
            rv = (1 - momentum) * rv + momentum * sigma.var()
            nu = nu + 1
    norm = (sigma - rm) / (torch.sqrt(rv) + 1e-8)
    h = norm.unsqueeze(-1)
    h = torch.relu(linear(h, sd["decision_net.0.weight"], sd["decision_net.0.bias"]))
    h = torch.relu(linear(h, sd["decision_net.2.weight"], sd["decision_net.2.bias"]))
    logits = linear(h, sd["decision_net.4.weight"], sd["decision_net.4.bias"])
    probs = torch.softmax(logits, dim=-1)
    decisions = (probs if training else logits).argmax(dim=-1)
    out = {"decisions": decisions, "probs": probs, "logits": logits, "threshold": sd["threshold"],
           "confidence": torch.sigmoid(-norm)}
    return out, (rm, rv, nu)


def vq_forward(centroids, x, beta=0.25):
    """models/vq.py:54-96: nearest-centroid quantisation with the straight-through estimator; returns (quantized,
    indices, commitment + codebook loss)."""
    centroids, x = _t(centroids), _t(x)
    d = (x.reshape(-1, 1) - centroids.reshape(1, -1)) ** 2
    idx = torch.argmin(d, dim=-1)
    q = centroids[idx].reshape(x.shape)
    loss = beta * F.mse_loss(x, q.detach()) + F.mse_loss(x.detach(), q)
    return x + (q - x).detach(), idx.reshape(x.shape), loss
