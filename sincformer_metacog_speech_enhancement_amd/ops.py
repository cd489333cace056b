"""Thin Python wrappers over the C ABI (one function per entry point) plus the
host-side weight packing the kernels expect.  Tensors must live on the HIP
device; there is no CPU fallback (lib.load() raises if the .so is missing)."""
import ctypes
import math
import numpy as np
import torch

from . import lib as _lib

EPI_NONE, EPI_SWISH, EPI_GELU, EPI_RESID, EPI_GLU, EPI_SIGMOID, EPI_TANH_SCALE, EPI_SIGMA, EPI_CPEA = range(9)

_DT_ID = {torch.bfloat16: 0, torch.float16: 1}
import os as _os
TRAIN_DTYPE = torch.float16         # base (training) operand format: the reference's own AMP recipe, see reset_precision
_state = {"dtype": TRAIN_DTYPE, "gemm_variant": int(_os.environ.get("SFM_GEMM_VARIANT", "0"))}


def set_gemm_variant(v):
    """0 auto, 2 = 128-row tiles, 6 = persistent, 9 = 256-row wide tiles, 10 = 512 x 128 tiles (A/B testing)."""
    _state["gemm_variant"] = int(v)



_DT_NAMES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16,
             "float16": torch.float16}
# Inference stages of the path whose 16-bit operand format can be chosen separately (the boundaries between them are
# fp32 tensors, so no stage reads another stage's 16-bit data in the wrong format):
#   pa    PerceptionAgent convs (inputs are GroupNorm + GELU outputs, the sinc FIR output is bounded by the waveform)
#   front CPEA, the fused 1026-column operand, fusion MLP, input projections
#   block the Linear / Conv1d GEMMs of the Conformer blocks (operands are LayerNorm outputs or bounded activations)
#   attn  the Q | K | V buffer and the attention core (QK^T, softmax, PV), head_dim 64 kernels
#   tail  output projection and the mask heads
STAGES = ("pa", "front", "block", "attn", "tail")
POLICIES = {
    "bf16": {s: torch.bfloat16 for s in STAGES},
    "fp16": {s: torch.float16 for s in STAGES},
    # default of inference: measured per-stage error budget in profiles/r02/precision_probe.json
    # (bf16 everywhere misses the 1e-3 mask bound: 1.3-1.6e-3; a bf16 PerceptionAgent alone costs 6.5e-4 of it; with the
    # attention core in bf16 and fp16 elsewhere the whole path is at 3e-4)
    "mixed": {"pa": torch.float16, "front": torch.float16, "block": torch.float16, "attn": torch.bfloat16,
              "tail": torch.float16},
}
_state["policy"] = dict(POLICIES["mixed"])          # inference default; training uses the base dtype (fp16 + loss scale)
_state["policy_name"] = "mixed"


def reset_precision():
    """the defaults: inference stages per POLICIES["mixed"]; base (training) format fp16 - what the reference trains in
    (fp16 autocast + torch.amp.GradScaler, training/conformer_pipeline.py:442, 504, 512-517), used with optim.DynamicLossScale.
    A uniform bf16 step stays available (set_compute_dtype) as a diagnostic: its 8-bit mantissas make this objective's
    gradient a quarter noise (training error budget)."""
    _state["dtype"] = TRAIN_DTYPE
    _state["policy"] = dict(POLICIES["mixed"])
    _state["policy_name"] = "mixed"


def _as_dtype(dt):
    if isinstance(dt, str):
        dt = _DT_NAMES[dt]
    if dt not in _DT_ID:
        raise ValueError("compute dtype must be bfloat16 or float16")
    return dt


def set_compute_dtype(dt):
    """16-bit MFMA operand/activation format for EVERYTHING (training and every inference stage):
    torch.bfloat16 or torch.float16.  See set_precision_policy for the per-stage recipe of inference."""
    dt = _as_dtype(dt)
    _state["dtype"] = dt
    _state["policy"] = {s: dt for s in STAGES}
    _state["policy_name"] = "bf16" if dt is torch.bfloat16 else "fp16"


def set_precision_policy(policy):
    """Operand format per inference stage: a name of POLICIES ("bf16", "fp16", "mixed") or a dict stage -> dtype
    (keys of STAGES; "block3" / "attn3" override one Conformer block).  Training keeps the base dtype."""
    if isinstance(policy, str):
        name, pol = policy, dict(POLICIES[policy])
    else:
        pol = dict(POLICIES["bf16"])
        pol.update({k: _as_dtype(v) for k, v in policy.items()})
        name = "custom:" + ",".join("%s=%s" % (k, "bf16" if v is torch.bfloat16 else "fp16") for k, v in sorted(pol.items()))
    _state["policy"] = pol
    _state["policy_name"] = name


def policy_name():
    return _state["policy_name"]


_state["weights_gen"] = 0


def bump_weights_generation():
    """Called by optimisers that update parameters through storage the parameter's own version counter does not see
    (optim.FlatAdamW steps a flat buffer the parameters are views of): invalidates every packed-weight cache."""
    _state["weights_gen"] += 1


def policy_key():
    """hashable signature of (base dtype, per-stage formats, weights generation): part of every packed-weight cache key"""
    return (_state["dtype"], _state["weights_gen"]) + tuple(sorted((k, str(v)) for k, v in _state["policy"].items()))


def stage_dtype(name, index=None):
    pol = _state["policy"]
    if index is not None and (name + str(index)) in pol:
        return pol[name + str(index)]
    return pol[name]


class stage:
    """with ops.stage("block", 3): ... — kernels and packs inside use that stage's operand format"""

    def __init__(self, name, index=None):
        self.dt = stage_dtype(name, index)

    def __enter__(self):
        self.prev = _state["dtype"]
        _state["dtype"] = self.dt
        return self

    def __exit__(self, *exc):
        _state["dtype"] = self.prev
        return False


def compute_dtype():
    return _state["dtype"]


def dtype_name():
    return "bf16" if _state["dtype"] is torch.bfloat16 else "f16"


def _dt():
    return _DT_ID[_state["dtype"]]


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sincformer HIP ops need device tensors (no CPU fallback); got a CPU tensor")


def round_up(x, m):
    return (x + m - 1) // m * m


class KernelProfiler:
    """Optional HIP-event timing of individual launches (bench.py / profiling only).
    Events are recorded on the current torch stream, i.e. the stream the kernels are
    enqueued on.  enable(None) instruments every family, enable({"attention"}) one."""

    def __init__(self):
        self.families = set()
        self.all = False
        self.records = []
        self.tags = False

    def enable(self, families=None, tags=False):
        self.tags = tags
        self.all = families is None
        self.families = set(families or ())
        self.records = []

    def disable(self):
        self.all = False
        self.families = set()

    def active(self, name):
        return self.all or name in self.families

    def summary(self):
        """{family: dict(n, ms_total, ms_avg, flops, bytes)} — synchronises."""
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, fl, by in self.records:
            d = out.setdefault(name, {"n": 0, "ms_total": 0.0, "flops": 0.0, "bytes": 0.0})
            d["n"] += 1
            d["ms_total"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        for d in out.values():
            d["ms_avg"] = d["ms_total"] / max(d["n"], 1)
        return out


    def launches(self, name):
        """[(ms, flops, bytes)] of every recorded launch of family `name`, in launch order — synchronises."""
        torch.cuda.synchronize()
        return [(e0.elapsed_time(e1), fl, by) for n, e0, e1, fl, by in self.records if n == name]


profiler = KernelProfiler()

# ---------------------------------------------------------------------------
# ordered reductions: scratch for the per-workgroup partials
# ---------------------------------------------------------------------------
# Every split reduction of the training step (weight-gradient M-splits, bias column sums, LayerNorm / GroupNorm dgamma / dbeta,
# BatchNorm statistics, the objective's moments, ||g||^2) writes its per-workgroup partials into a workspace and folds them in a
# fixed order (csrc/reduce.hip) instead of issuing atomics: a step is then bit-reproducible, like the reference's CPU step.
# SFM_DETERMINISTIC=0 / set_deterministic(False) passes NULL workspaces = the fp32-atomics form (A/B of the cost).
_DET = {"on": _os.environ.get("SFM_DETERMINISTIC", "1") != "0", "bufs": {}}


def set_deterministic(flag):
    _DET["on"] = bool(flag)


def is_deterministic():
    return _DET["on"]


def _ws(n, device, dtype=torch.float32):
    """scratch of >= n elements for the kernel about to be enqueued on the CURRENT stream (one grow-only buffer per device,
    stream and dtype: launches of one stream run in order, so the fold of one launch has read the partials before the next
    launch overwrites them; a replaced buffer goes back to the caching allocator, which hands memory freed on a stream only
    to later work of that stream).  None when the atomics form is selected."""
    if not _DET["on"]:
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, dtype)
    t = _DET["bufs"].get(key)
    if t is None or t.numel() < n:
        t = _DET["bufs"][key] = torch.empty(max(int(n), 1 << 18), device=device, dtype=dtype)
    return t


def _call(name, fn, args, flops=0.0, nbytes=0.0, tag=None):
    if profiler.active(name):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        profiler.records.append((name, e0, e1, float(flops), float(nbytes)))
        if tag is not None and profiler.tags:
            profiler.records.append(("%s[%s]" % (name, tag), e0, e1, float(flops), float(nbytes)))
    else:
        rc = fn(*args)
    _lib.check(rc, name)



def _cost_of(name, v):
    """algorithmic (flops, bytes) of one launch, from the wrapper's local variables
    (bytes = each operand read once + each result written once)."""
    try:
        if name == "gemm16":
            pw = v["pw"]
            ncols = pw.Npad if pw.glu else pw.N
            rows = v["B"] * v["Lout"]
            fl = 2.0 * rows * ncols * pw.K
            osz = 4 if v["out"].dtype == torch.float32 else 2
            by = v["B"] * v["Lin"] * pw.cin * 2 + pw.Npad * pw.Kpad * 2 + rows * pw.N * osz
            if v.get("resid") is not None:
                by += rows * pw.N * 4
            return fl, by, "M%d N%d K%d k%d s%d epi%d o%d" % (rows, ncols, pw.K, pw.ksize, v["stride"], v["epi"], osz)
        if name == "framed_gemm_f32":
            rows = v["B"] * v["M"]
            osz = 4 if v["out"].dtype == torch.float32 else 2
            return (2.0 * rows * v["N"] * v["K"], v["B"] * v["Ls"] * 4 + rows * v["N"] * osz,
                    "M%d N%d K%d hop%d" % (rows, v["N"], v["K"], v["hop"]))
        if name == "attention_fwd":
            B, T, H, hd = v["B"], v["T"], v["H"], v["hd"]
            return 4.0 * B * H * T * T * hd, 4.0 * B * T * H * hd * 2
        if name == "layernorm":
            M, D = v["M"], v["D"]
            return 8.0 * M * D, M * D * (4 + (2 if v["out16"] is not None else 0) + (4 if v["out32"] is not None else 0))
        if name == "gn_apply":
            n = v["Bn"] * v["rows"] * v["C"]
            isz = 4 if v["in_f32"] else 2
            osz = 4 if v["out_f32"] else 2
            return 10.0 * n, n * (isz * (2 if v["x2"] is not None else 1) + osz)
        if name == "dwconv_bn_swish":
            n = v["B"] * v["T"] * v["C"]
            return 2.0 * n * v["KS"], n * 4
        if name == "bilstm_layer":
            B, T, H = v["B"], v["T"], v["H"]
            return 2.0 * B * T * 2 * 4 * H * H, B * T * (8 * H + 2 * H) * 4
        if name == "pool_time":
            return 0.0, v["B"] * (v["Tin"] * v["src_bytes"] + v["Tout"] * 4) * v["C"]
        if name == "polar_mask":
            n = v["B"] * v["rows"] * v["F"]
            return 20.0 * n, n * 4 * 8
        if name == "stft_lognorm_pack":
            return 0.0, v["M"] * v["F"] * (8 + 4)
        if name == "istft_ola":
            return 0.0, v["B"] * v["Ln"] * 12
    except Exception:
        pass
    return 0.0, 0.0


# ---------------------------------------------------------------------------
# weight packing
# ---------------------------------------------------------------------------
class PackedWeight:
    """W [Npad, Kpad] 16-bit (+ fp32 bias [Npad]) for sfm_gemm16."""
    __slots__ = ("w", "bias", "N", "K", "Npad", "Kpad", "ksize", "cin", "glu")

    def __init__(self, w, bias, N, K, ksize, cin, glu=False):
        self.w, self.bias, self.N, self.K = w, bias, N, K
        self.Npad, self.Kpad = w.shape
        self.ksize, self.cin, self.glu = ksize, cin, glu


_ZERO_BIAS = {}


def _zero_bias(n, device):
    """read-only fp32 zeros [n], cached per device"""
    key = (int(n), str(device))
    t = _ZERO_BIAS.get(key)
    if t is None:
        t = _ZERO_BIAS[key] = torch.zeros(n, device=device, dtype=torch.float32)
    return t


def pack_linear(weight, bias=None, glu=False, k_pad_to=None, dtype=None):
    """weight [N, K] (nn.Linear) or [N, Cin, k] (nn.Conv1d; repacked to tap-major K).
    glu=True interleaves the two halves in 32-row groups for the GLU epilogue."""
    dtype = dtype or _state["dtype"]
    w = weight.detach().float()
    ksize, cin = 1, w.shape[1]
    if w.dim() == 3:
        ksize = w.shape[2]
        w = w.permute(0, 2, 1).reshape(w.shape[0], -1)       # [N, k*Cin], tap-major
    N, K = w.shape
    Kpad = round_up(k_pad_to or K, 64)
    if glu or (N > 64 and round_up(N, 128) - N < 64):
        Npad = round_up(N, 128)          # 128-column tiles
    else:
        Npad = round_up(N, 64)           # 64-column tiles (kernel picks BN from Npad % 128)
    if not glu and N == Npad and K == Kpad:
        # already tile-aligned (every Linear / Conv1d of the path but the heads): one convert launch, the bias is used in
        # place (the training step packs ~150 weights per step; zero-fill + slice-copy per operand was ~800 tiny launches)
        W = w.to(dtype).contiguous()
        Bp = bias.detach().float().contiguous() if bias is not None else _zero_bias(Npad, w.device)
    else:
        b = bias.detach().float() if bias is not None else _zero_bias(N, w.device)
        if glu:
            C = N // 2
            idx = torch.arange(N, device=w.device)
            blk, t = idx // 64, idx % 64
            src = torch.where(t < 32, blk * 32 + t, C + blk * 32 + (t - 32))
            w, b = w[src], b[src]
        W = torch.zeros(Npad, Kpad, device=w.device, dtype=dtype)
        W[:N, :K] = w.to(dtype)
        Bp = torch.zeros(Npad, device=w.device, dtype=torch.float32)
        Bp[:N] = b
    if ksize == 1:
        cin = Kpad if k_pad_to else K    # operand rows must be zero padded up to cin
        if cin % 8 != 0:
            raise ValueError("K=%d is not a multiple of 8: pass k_pad_to and zero-pad the operand" % K)
    return PackedWeight(W.contiguous(), Bp, (N // 2) if glu else N, K, ksize, cin, glu)


def gemm16(A, pw, out, *, B, Lout, Lin, a_batch_stride, ldo, o_batch_stride, lda=None, stride=1, pad=0,
           epi=EPI_NONE, resid=None, ldr=0, r_batch_stride=0, alpha=1.0, gn_partial=None, gn_group=0, nsplit=0,
           p_drop=0.0, seed=0):
    """A: [B, Lin, cin] 16-bit with position stride lda (default cin) -> out rows (b, l)."""
    _need_dev(A, out)
    L = _lib.load()
    cin = pw.cin
    lda = cin if lda is None else lda
    # out_f32: 0 = 16-bit in the operands' format, 1 = fp32, 2 = the OTHER 16-bit format (a stage boundary of the precision
    # policy, e.g. the Q | K | V buffer of a bf16 attention core behind fp16 projections): one rounding of the fp32 accumulators
    out_f32 = 1 if out.dtype == torch.float32 else (0 if out.dtype == _state["dtype"] else 2)
    if A.dtype != _state["dtype"] or pw.w.dtype != _state["dtype"]:
        raise RuntimeError("gemm16: operand formats %s / %s do not match the stage's %s (precision policy)" %
                           (A.dtype, pw.w.dtype, _state["dtype"]))
    if p_drop > 0.0:                                   # training forward: residual-branch dropout in the epilogue
        _call("gemm16", L.sfm_gemm16_train, (_p(A), _p(pw.w), _p(pw.bias), _p(out), _p(resid), _p(gn_partial), B, Lout, Lin, cin,
                                             lda, pw.ksize, stride, pad, a_batch_stride, pw.Kpad, pw.N, pw.Npad, ldo,
                                             o_batch_stride, ldr, r_batch_stride, float(alpha), epi, out_f32, gn_group, nsplit,
                                             _dt(), _state["gemm_variant"], float(p_drop), int(seed) & 0xffffffff, _stream()),
              *_cost_of("gemm16", locals()))
        return out
    _call("gemm16", L.sfm_gemm16_ex, (_p(A), _p(pw.w), _p(pw.bias), _p(out), _p(resid), _p(gn_partial), B, Lout, Lin, cin, lda,
                      pw.ksize, stride, pad, a_batch_stride, pw.Kpad, pw.N, pw.Npad, ldo, o_batch_stride, ldr, r_batch_stride,
                      float(alpha), epi, out_f32, gn_group, nsplit, _dt(), _state["gemm_variant"], _stream()),
          *_cost_of("gemm16", locals()))
    return out


def conv16p_supported(cin, n, ksize, stride, pad, two_inputs, skip):
    """layer shapes sfm_conv16p is built for (the PerceptionAgent's: include/sincformer_hip.h)"""
    if cin % 64:
        return False
    key = (ksize, stride, pad, n)
    if skip:
        return key == (7, 2, 3, 128)
    return (key == (7, 2, 3, 256) and two_inputs) or (key in ((3, 1, 1, 128), (3, 1, 1, 256)) and not two_inputs) or \
        (key in ((5, 2, 2, 256), (1, 2, 0, 256)) and two_inputs)


def conv16p(x1, sc1, sh1, pw, out, *, B, Lin, stride, pad, x2=None, sc2=None, sh2=None, gn_partial=None, gn_group=0,
            skip_pw=None, out_s=None, gn_partial_s=None):
    """out = Conv1d(GELU(sc1 * x1 + sh1 [+ sc2 * x2 + sh2]); pw) on channels-last raw conv outputs x1 / x2 [B, Lin, Cin] with
    per-(batch, channel) GroupNorm scale / shift [B, Cin]: the normalised activation is produced while the operand is staged
    and never written to HBM.  skip_pw / out_s / gn_partial_s: the residual block's 1x1 stride-2 skip conv on the same input."""
    _need_dev(x1, out)
    L = _lib.load()
    cin, N, ks = pw.cin, pw.N, pw.ksize
    dt = _state["dtype"]
    if x1.dtype != dt or pw.w.dtype != dt or (x2 is not None and x2.dtype != dt):
        raise RuntimeError("conv16p: operand formats do not match the stage's %s (precision policy)" % dt)
    if pw.Npad != N or pw.Kpad != ks * cin or (skip_pw is not None and (skip_pw.Npad != N or skip_pw.Kpad != cin)):
        raise RuntimeError("conv16p: weights must be packed without padding")
    out_f32 = 1 if out.dtype == torch.float32 else (0 if out.dtype == dt else 2)
    Lout = (Lin + 2 * pad - ks) // stride + 1
    nin = 2 if x2 is not None else 1
    flops = 2.0 * B * Lout * N * (ks * cin + (cin if skip_pw is not None else 0))
    nbytes = B * Lin * cin * 2.0 * nin + B * Lout * N * out.element_size() * (2 if skip_pw is not None else 1) + N * ks * cin * 2.0
    _call("conv16p", L.sfm_conv16p, (_p(x1), _p(sc1), _p(sh1), _p(x2), _p(sc2), _p(sh2), _p(pw.w), _p(pw.bias), _p(out),
                                     _p(gn_partial), _p(skip_pw.w if skip_pw is not None else None),
                                     _p(skip_pw.bias if skip_pw is not None else None), _p(out_s), _p(gn_partial_s), B, Lin, cin, N,
                                     ks, stride, pad, out_f32, gn_group, _dt(), _stream()),
          flops, nbytes, tag="Lout%d Cin%d N%d k%d s%d in%d%s" % (B * Lout, cin, N, ks, stride, nin, " +skip" if skip_pw is not None else ""))
    return out


_LIN256 = {"on": _os.environ.get("SFM_LIN256", "1") != "0"}


def set_lin256(flag):
    """K = 256 linears with a plain / GLU epilogue and a 16-bit result on the resident-operand kernel (csrc/lin256.hip; default on,
    SFM_LIN256=0 / False = sfm_gemm16 for every shape: the A/B)"""
    _LIN256["on"] = bool(flag)


def _lin256_ok(x16, pw, epi, out, resid, nsplit, p_drop):
    return (_LIN256["on"] and _state["gemm_variant"] == 0 and epi in (EPI_NONE, EPI_GLU) and (epi == EPI_GLU) == bool(pw.glu) and
            pw.K == 256 and pw.Kpad == 256 and pw.ksize == 1 and pw.Npad % (128 if pw.glu else 64) == 0 and
            (pw.Npad == (2 * pw.N if pw.glu else pw.N)) and pw.Npad <= 2048 and
            resid is None and nsplit == 0 and p_drop == 0.0 and
            (out.dtype in (torch.float16, torch.bfloat16) or (out.dtype == torch.float32 and not pw.glu and out.stride(0) % 4 == 0)) and
            x16.shape[0] >= 4096 and x16.stride(1) == 1 and out.stride(1) == 1 and x16.stride(0) % 8 == 0 and
            (out.stride(0) % 8 == 0 or out.dtype == torch.float32) and
            x16.dtype == _state["dtype"] and pw.w.dtype == _state["dtype"])


def lin256(x16, pw, out):
    """sfm_lin256 (csrc/lin256.hip): out = x16[:, :256] @ W^T + b, or its GLU when pw was packed with glu=True; out 16-bit (either
    format) or, plain epilogue only, fp32.  linear16 routes the shapes of the path here (see _lin256_ok)."""
    _need_dev(x16, out)
    L = _lib.load()
    M, ld = x16.shape[0], x16.stride(0)
    odt_id = 2 if out.dtype == torch.float32 else (1 if out.dtype == torch.float16 else 0)
    osz = 4.0 if out.dtype == torch.float32 else 2.0
    _call("gemm16", L.sfm_lin256, (_p(x16), _p(pw.w), _p(pw.bias), _p(out), M, pw.Npad, ld, out.stride(0), 1 if pw.glu else 0,
                                   _dt(), odt_id, _stream()),
          2.0 * M * pw.Npad * 256, M * 256 * 2.0 + pw.Npad * 512.0 + M * pw.N * osz,
          tag="M%d N%d K256 lin256%s o%d" % (M, pw.Npad, " glu" if pw.glu else "", int(osz)))
    return out


_HEADPOOL = {"on": _os.environ.get("SFM_HEADPOOL", "1") != "0"}


def set_headpool(flag):
    """latent heads + time pooling in one launch (sfm_headpool) in the fused path; False / SFM_HEADPOOL=0 = heads GEMM, then pool_time"""
    _HEADPOOL["on"] = bool(flag)


def headpool_tiles(Tin, Tout):
    """(frames per tile, tiles per utterance) of sfm_headpool, or None when the pair is not supported / the fusion is switched off"""
    if not _HEADPOOL["on"] or _state["gemm_variant"] != 0:
        return None
    fpt = int(_lib.load().sfm_headpool_frames_per_tile(int(Tin), int(Tout)))
    return (fpt, (Tout + fpt - 1) // fpt) if fpt > 0 else None


def headpool(xd16, pw, pooled16, part, B, Tin, Tout, gcols=16):
    """xd16 [B, Tin, 256] -> pooled16 [B, Tout, N] = time-pooled RAW head outputs, part [B, P, N / gcols, 2] = GroupNorm partial sums of
    the full-rate outputs (see include/sincformer_hip.h)."""
    _need_dev(xd16, pooled16, part)
    L = _lib.load()
    if xd16.dtype != _state["dtype"] or pw.w.dtype != _state["dtype"] or pooled16.dtype != _state["dtype"]:
        raise RuntimeError("headpool: operands must be in the stage's format")
    _call("gemm16", L.sfm_headpool, (_p(xd16), _p(pw.w), _p(pw.bias), _p(pooled16), _p(part), B, Tin, Tout, pw.Npad, xd16.stride(1),
                                     pooled16.stride(1), gcols, _dt(), _stream()),
          2.0 * B * Tin * pw.Npad * 256, B * Tin * 512.0 + pw.Npad * 512.0 + B * Tout * pw.Npad * 2.0,
          tag="M%d N%d K256 headpool->T%d" % (B * Tin, pw.Npad, Tout))


def ln_linear16(x32, ln_w, ln_b, pw, epi=EPI_NONE, out_dtype=None, eps=1e-5):
    """linear16(LayerNorm(x32[:, :256]), pw, epi) with a 16-bit result.  On the shapes sfm_lin256 takes (K = 256, plain or GLU epilogue,
    M >= 4096) the LayerNorm is the GEMM kernel's prologue (sfm_ln_lin256): one launch, the normalised 16-bit rows never reach HBM;
    otherwise sfm_layernorm + linear16.  The two routes give the same bits."""
    M = x32.shape[0]
    out = torch.empty(M, pw.N, device=x32.device, dtype=out_dtype or _state["dtype"])
    fused = (_LIN256["on"] and _state["gemm_variant"] == 0 and epi in (EPI_NONE, EPI_GLU) and (epi == EPI_GLU) == bool(pw.glu) and
             pw.K == 256 and pw.Kpad == 256 and pw.ksize == 1 and pw.Npad % (128 if pw.glu else 64) == 0 and
             pw.Npad == (2 * pw.N if pw.glu else pw.N) and pw.Npad <= 2048 and out.dtype in (torch.float16, torch.bfloat16) and
             M >= 4096 and x32.dtype == torch.float32 and x32.stride(1) == 1 and x32.stride(0) % 4 == 0 and
             x32.data_ptr() % 16 == 0 and ln_w.numel() == 256 and pw.w.dtype == _state["dtype"])
    if fused:
        _need_dev(x32, out)
        L = _lib.load()
        lw, lb = ln_w.detach().float().contiguous(), ln_b.detach().float().contiguous()
        _call("gemm16", L.sfm_ln_lin256, (_p(x32), x32.stride(0), _p(lw), _p(lb), float(eps), _p(pw.w), _p(pw.bias), _p(out), M, pw.Npad,
                                          out.stride(0), 1 if pw.glu else 0, _dt(), 1 if out.dtype == torch.float16 else 0, _stream()),
              2.0 * M * pw.Npad * 256, M * 256 * 4.0 + pw.Npad * 512.0 + M * pw.N * 2.0,
              tag="M%d N%d K256 ln+lin256%s" % (M, pw.Npad, " glu" if pw.glu else ""))
        return out
    h16 = torch.empty(M, 256, device=x32.device, dtype=_state["dtype"])
    layernorm(x32, ln_w, ln_b, out16=h16, eps=eps)
    return linear16(h16, pw, epi=epi, out=out)


def linear16(x16, pw, epi=EPI_NONE, out_dtype=None, resid=None, alpha=1.0, out=None, nsplit=0, p_drop=0.0, seed=0):
    """x16 [M, K(>=pw.K)] 16-bit contiguous rows -> [M, N]."""
    M, ld = x16.shape[0], x16.stride(0)
    if out is None:
        odt = out_dtype or (torch.float32 if epi == EPI_RESID else _state["dtype"])
        out = torch.empty(M, pw.N, device=x16.device, dtype=odt)
    if _lin256_ok(x16, pw, epi, out, resid, nsplit, p_drop):
        return lin256(x16, pw, out)
    gemm16(x16, pw, out, B=1, Lout=M, Lin=M, a_batch_stride=0, lda=ld, ldo=out.stride(0), o_batch_stride=0, epi=epi,
           resid=resid, ldr=(resid.stride(0) if resid is not None else 0), alpha=alpha, nsplit=nsplit, p_drop=p_drop,
           seed=seed)
    return out


def linear16_swish(x16, pw, p_drop=0.0, seed=0, aux=None):
    """FFN Linear with the Swish (+ hidden dropout) in the GEMM epilogue (training).
    aux is None: forward -> (d [M, N] 16-bit = drop * swish'(z), the derivative factor the backward needs, u = drop * swish(z));
    aux = saved d: backward -> g * d where g = x16 @ W^T (pw = the transposed pack, no bias)."""
    L = _lib.load()
    M, ld = x16.shape[0], x16.stride(0)
    dt = _state["dtype"]
    out = torch.empty(M, pw.N, device=x16.device, dtype=dt)
    out2 = torch.empty(M, pw.N, device=x16.device, dtype=dt) if aux is None else None
    _call("gemm16", L.sfm_gemm16_swish, (_p(x16), _p(pw.w), _p(pw.bias), _p(out), _p(aux), _p(out2), M, pw.cin, ld, pw.Kpad, pw.N,
                                         pw.Npad, pw.N, 0 if aux is None else 1, float(p_drop), int(seed) & 0xffffffff, _dt(),
                                         _stream()), 2.0 * M * pw.N * pw.K, M * (pw.cin + 2.0 * pw.N) * 2.0 + pw.Npad * pw.Kpad * 2.0,
          tag="M%d N%d K%d swish%s" % (M, pw.N, pw.K, "" if aux is None else "_bwd"))
    return (out2, out) if aux is None else out


def framed_gemm(sig, Wt, out, *, B, M, Ls, sig_batch_stride, hop, padl, K, N, o_batch_stride, ldm, ldn, mode=0,
                bias=None, out2=None, nsplit=0, gn_partial=None, gn_group=0):
    _need_dev(sig, Wt, out)
    L = _lib.load()
    Kpad, Npad = Wt.shape
    out_f32 = 1 if out.dtype == torch.float32 else 0
    _call("framed_gemm_f32", L.sfm_framed_gemm_f32, (_p(sig), _p(Wt), _p(bias), _p(out), _p(out2), _p(gn_partial), B, M, Ls,
                               sig_batch_stride, hop, padl, K, Kpad, N, Npad, nsplit, o_batch_stride, ldm, ldn, mode,
                               out_f32, gn_group, _dt(), _stream()),
          *_cost_of("framed_gemm_f32", locals()))
    return out


def pack_split16_matrix(wt_kn):
    """[K, N] fp32 -> (hi, lo) bf16 pair, n-major [round256(N), round32(K)], operand of framed_gemm_split16."""
    K, N = wt_kn.shape
    w = torch.zeros(round_up(N, 256), round_up(K, 32), device=wt_kn.device, dtype=torch.float32)
    w[:N, :K] = wt_kn.t()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous(), K, N


def framed_gemm_split16(sig, W, out, *, B, M, Ls, sig_batch_stride, hop, padl, o_batch_stride, ldm, mode=0, out2=None, nsplit=0,
                        col2_off=0):
    """W = pack_split16_matrix(...)"""
    _need_dev(sig, out)
    L = _lib.load()
    hi, lo, K, N = W
    Npad, Kpad = hi.shape
    _call("framed_gemm_split16", L.sfm_framed_gemm_split16, (_p(sig), _p(hi), _p(lo), _p(out), _p(out2), B, M, Ls, sig_batch_stride,
                                                             hop, padl, K, Kpad, N, Npad, nsplit, col2_off, o_batch_stride, ldm,
                                                             mode, _stream()),
          2.0 * B * M * K * N, 4.0 * (B * M * hop + B * M * N))
    return out


def pack_f32_matrix(wt_kn):
    """[K, N] fp32 -> zero padded [round32(K), round64(N)] operand of framed_gemm."""
    K, N = wt_kn.shape
    W = torch.zeros(round_up(K, 32), round_up(N, 64), device=wt_kn.device, dtype=torch.float32)
    W[:K, :N] = wt_kn
    return W


ATTN_QSCALE_LOG2E = 1.4426950408889634


def attention_kernel_name(B, T, H, variant=None):
    """name of the forward kernel sfm_attention_fwd_ex picks for head_dim 64 (mirrors the rule in csrc/attention.hip)"""
    v = _ATTN_VARIANT[0] if variant is None else variant
    nqt5 = (T + 511) // 512
    enough = B * H * nqt5 >= 128
    if v == 0:
        if enough and (T >= 1024 or 100 * T >= 78 * 512 * nqt5):
            v = 4
        elif enough and T <= 256 and 100 * T >= 90 * 256:
            v = 5
        else:
            v = 2 if T >= 1024 else 1
    return {1: "attn_fwd_hd64_kernel", 2: "attn_fwd_hd64x2_kernel", 3: "attn_fwd_hd64r_kernel", 4: "attn_fwd_hd64p8_kernel",
            5: "attn_fwd_hd64p4_kernel", 6: "attn_fwd_hd64q4_kernel"}[v]


def set_attention_variant(v):
    """0: chosen by shape (default), 1: 32 query rows per wave, 3: persistent ring kernel, 4 / 5: pipelined persistent kernel
    with one 8-wave / two 4-wave workgroups per CU (A/B measurements).  Host-side state only: the value is passed to
    sfm_attention_fwd_ex with every call (the library keeps no selection state)."""
    v = int(v)
    if not 0 <= v <= 6:
        raise ValueError("attention variant %d" % v)
    _ATTN_VARIANT[0] = v


_ATTN_VARIANT = [0]


def variant_overrides():
    """the kernel-selection test knobs that are NOT at their default (set_gemm_variant, set_attention_variant, SFM_GEMM_VARIANT):
    bench.py refuses to time anything while one is active"""
    out = {}
    if _state["gemm_variant"] != 0:
        out["gemm_variant"] = _state["gemm_variant"]
    if _ATTN_VARIANT[0] != 0:
        out["attention_variant"] = _ATTN_VARIANT[0]
    return out


def attention(qkv16, B, T, H, hd, out=None, prescaled=False, out_dtype=None):
    """qkv16 [B*T, 3*H*hd] (q | k | v) -> [B*T, H*hd].  prescaled: q already multiplied by
    log2(e)/sqrt(hd) (functional.pack_mhsa folds it into W_q, b_q before the 16-bit rounding)."""
    _need_dev(qkv16)
    L = _lib.load()
    D = H * hd
    ld = qkv16.stride(0)
    if qkv16.dtype != _state["dtype"]:
        raise RuntimeError("attention: Q | K | V are %s, the stage's format is %s" % (qkv16.dtype, _state["dtype"]))
    if out is None:
        out = torch.empty(B * T, D, device=qkv16.device, dtype=out_dtype or qkv16.dtype)
    _call("attention_fwd", L.sfm_attention_fwd_ex, (_p(qkv16), _p(out), B, T, H, hd, ld, out.stride(0), D, 2 * D, T * ld,
                                                    T * out.stride(0), (-1.0 if prescaled else 1.0 / math.sqrt(hd)), _dt(),
                                                    _DT_ID[out.dtype], _ATTN_VARIANT[0], _stream()),
          *_cost_of("attention_fwd", locals()))
    return out


def layernorm(x32, w, b, out16=None, out32=None, act=0, eps=1e-5):
    """rows of x32 [M, >=D] are normalised over their first D = w.numel() columns."""
    _need_dev(x32)
    L = _lib.load()
    M, D = x32.shape[0], w.numel()
    _call("layernorm", L.sfm_layernorm, (_p(x32), _p(w), _p(b), _p(out16), _p(out32), M, D, x32.stride(0),
                         out16.stride(0) if out16 is not None else 0, out32.stride(0) if out32 is not None else 0,
                         eps, act, _dt(), _stream()),
          *_cost_of("layernorm", locals()))


def gn_finalize(partial, w, b, Bn, P, G, C, rows, eps=1e-5):
    L = _lib.load()
    scale = torch.empty(Bn, C, device=w.device, dtype=torch.float32)
    shift = torch.empty(Bn, C, device=w.device, dtype=torch.float32)
    _call("gn_finalize", L.sfm_gn_finalize, (_p(partial), _p(w), _p(b), _p(scale), _p(shift), Bn, P, G, C, rows, eps, _stream()),
          *_cost_of("gn_finalize", locals()))
    return scale, shift


def gn_apply(x1, sc1, sh1, out, Bn, rows, C, act=0, x2=None, sc2=None, sh2=None):
    L = _lib.load()
    in_f32 = 1 if x1.dtype == torch.float32 else 0
    out_f32 = 1 if out.dtype == torch.float32 else 0
    _call("gn_apply", L.sfm_gn_apply, (_p(x1), _p(sc1), _p(sh1), _p(x2), _p(sc2), _p(sh2), _p(out), Bn, rows, C, in_f32, out_f32, act,
                        _dt(), _stream()),
          *_cost_of("gn_apply", locals()))
    return out


def dwconv_bn_swish(x16, wdw, bdw, bnw, bnb, bnm, bnv, B, T, C, out=None, eps=1e-5):
    L = _lib.load()
    KS = wdw.shape[-1]
    if out is None:
        out = torch.empty_like(x16)
    _call("dwconv_bn_swish", L.sfm_dwconv_bn_swish, (_p(x16), _p(wdw), _p(bdw), _p(bnw), _p(bnb), _p(bnm), _p(bnv), _p(out), B, T, C, KS,
                               eps, _dt(), _stream()),
          *_cost_of("dwconv_bn_swish", locals()))
    return out


def dwconv_folded(x16, wT, sc, sh, B, T, C, out=None, act=1):
    """depthwise conv + folded BatchNorm(eval) (+ Swish when act) with register-resident taps (KS 7 / 31);
    `out` may be 16-bit (default) or fp32."""
    L = _lib.load()
    KS = wT.shape[0]
    if out is None:
        out = torch.empty_like(x16)
    _call("dwconv_bn_swish", L.sfm_dwconv_folded, (_p(x16), _p(wT), _p(sc), _p(sh), _p(out), B, T, C, KS, int(act),
                                                   1 if out.dtype == torch.float32 else 0, _dt(), _stream()),
          2.0 * B * T * C * KS, B * T * C * 4.0)
    return out


def convert_rows(src32, dst16, M, C, Cz, ld_src, ld_dst):
    L = _lib.load()
    _call("convert_rows", L.sfm_convert_rows, (_p(src32), _p(dst16), M, C, Cz, ld_src, ld_dst, _dt(), _stream()),
          *_cost_of("convert_rows", locals()), tag="M%d C%d" % (M, C))


def transpose(src, dst, B, R, C, src_batch, src_row, dst_batch, dst_row):
    L = _lib.load()
    _call("transpose", L.sfm_transpose, (_p(src), _p(dst), B, R, C, src_batch, src_row, dst_batch, dst_row,
                         1 if src.dtype == torch.float32 else 0, 1 if dst.dtype == torch.float32 else 0, _dt(),
                         _stream()),
          *_cost_of("transpose", locals()))


def mean_time(src32, B, T, C, ld_src):
    """mean over time of fp32 [B, T, ld_src] cols [0, C) -> [B, C] (glue G2)"""
    L = _lib.load()
    out = torch.empty(B, C, device=src32.device, dtype=torch.float32)
    scratch = torch.empty(int(L.sfm_mean_time_scratch_floats(B, T, C)), device=src32.device, dtype=torch.float32)
    _call("pool_time", L.sfm_mean_time, (_p(src32), _p(out), _p(scratch), B, T, C, ld_src, _stream()), 0.0, 4.0 * B * T * C)
    return out


def pool_time_bwd(dout32, B, Tin, Tout, C):
    """adjoint of pool_time: dout fp32 [B, Tout, C] -> dsrc fp32 [B, Tin, C]"""
    L = _lib.load()
    dout32 = dout32.contiguous()
    dsrc = torch.empty(B, Tin, C, device=dout32.device, dtype=torch.float32)
    _call("pool_time_bwd", L.sfm_pool_time_bwd, (_p(dout32), _p(dsrc), B, Tin, Tout, C, C, C, _stream()), 0.0,
          4.0 * B * (Tin + Tout) * C)
    return dsrc


def pool_time(src32, dst16, dst32, B, Tin, Tout, C, ld_src, ld_dst, scale=None, shift=None):
    """adaptive average pooling over time; scale / shift [B, C]: out = scale * avg + shift (pooled GroupNorm output).
    src32: fp32, or 16-bit in the current operand format (the fused path's raw latent heads)."""
    L = _lib.load()
    src_bytes = src32.element_size()
    if src_bytes == 4:
        args = (_p(src32), _p(scale), _p(shift), _p(dst16), _p(dst32), B, Tin, Tout, C, ld_src, ld_dst, _dt(), _stream())
        fn = L.sfm_pool_time_affine
    else:
        args = (_p(src32), _DT_ID[src32.dtype], _p(scale), _p(shift), _p(dst16), _p(dst32), B, Tin, Tout, C, ld_src, ld_dst,
                _dt(), _stream())
        fn = L.sfm_pool_time_affine16
    _call("pool_time", fn, args, *_cost_of("pool_time", locals()))


def stft_lognorm_pack(re, im, dst16, M, F, zpad, ld_dst):
    L = _lib.load()
    _call("stft_lognorm_pack", L.sfm_stft_lognorm_pack, (_p(re), _p(im), _p(dst16), M, F, zpad, ld_dst, _dt(), _stream()),
          *_cost_of("stft_lognorm_pack", locals()))


def polar_mask(lm, lp, B, rows, F, phase_scale, ld_logits, mag_bias=None, nr=None, ni=None, mr=None, mi=None, er=None,
               ei=None, mmag=None, ld_enh=0):
    L = _lib.load()
    _call("polar_mask", L.sfm_polar_mask, (_p(lm), _p(lp), _p(mag_bias), _p(nr), _p(ni), _p(mr), _p(mi), _p(er), _p(ei), _p(mmag), B,
                          rows, F, float(phase_scale), ld_logits, ld_enh, _stream()),
          *_cost_of("polar_mask", locals()))


def complex_mul(sr, si, mr, mi):
    L = _lib.load()
    er, ei = torch.empty_like(sr), torch.empty_like(sr)
    _call("complex_mul", L.sfm_complex_mul, (_p(sr), _p(si), _p(mr), _p(mi), _p(er), _p(ei), sr.numel(), _stream()),
          *_cost_of("complex_mul", locals()))
    return er, ei


def istft_ola(frames, win2, out, B, T, Ln, n_fft, hop, win, ld_frames):
    L = _lib.load()
    _call("istft_ola", L.sfm_istft_ola, (_p(frames), _p(win2), _p(out), B, T, Ln, n_fft, hop, win, ld_frames, _stream()),
          *_cost_of("istft_ola", locals()))


def pack_spec(re, im, dst, M, F, ld, ld_src):
    L = _lib.load()
    _call("pack_spec", L.sfm_pack_spec, (_p(re), _p(im), _p(dst), M, F, ld, ld_src, _stream()),
          *_cost_of("pack_spec", locals()))


def sinc_filters(low_hz, band_hz, window, n_, C, K, sample_rate, min_low_hz, min_band_hz, want_filt=True):
    L = _lib.load()
    dev = low_hz.device
    Npad = round_up(C, 64)
    Wt = torch.zeros(round_up(K, 32), Npad, device=dev, dtype=torch.float32)
    filt = torch.empty(C, K, device=dev, dtype=torch.float32) if want_filt else None
    _call("sinc_filters", L.sfm_sinc_filters, (_p(low_hz), _p(band_hz), _p(window), _p(n_), _p(filt), _p(Wt), C, K, Npad,
                            float(sample_rate), float(min_low_hz), float(min_band_hz), _stream()),
          *_cost_of("sinc_filters", locals()))
    return filt, Wt


def ffn_fused(x32, lnw, lnb, w1_16, b1, w2_16, b2, alpha=0.5, eps=1e-5, out=None):
    """FeedForwardModule in one launch: x32 [M, 256] fp32 -> [M, 256] fp32."""
    _need_dev(x32)
    Lb = _lib.load()
    M, D = x32.shape
    FF = w1_16.shape[0]
    if out is None:
        out = torch.empty_like(x32)
    _call("ffn_fused", Lb.sfm_ffn_fused, (_p(x32), _p(lnw), _p(lnb), _p(w1_16), _p(b1), _p(w2_16), _p(b2), _p(out), M, D, FF,
                                          float(alpha), float(eps), _dt(), _stream()), 4.0 * M * D * FF, M * D * 8.0)
    return out


def ffn_fused_ln(x32, lnw, lnb, w1_16, b1, w2_16, b2, ln2w, ln2b, ln_f32, want_y=True, alpha=0.5, eps=1e-5):
    """ffn_fused + the LayerNorm that follows the module (ln2w / ln2b) applied in the epilogue.
    Returns (y fp32 or None, LN(y) as fp32 or 16-bit)."""
    _need_dev(x32)
    Lb = _lib.load()
    M, D = x32.shape
    FF = w1_16.shape[0]
    y = torch.empty_like(x32) if want_y else None
    ln = torch.empty(M, D, device=x32.device, dtype=torch.float32 if ln_f32 else _state["dtype"])
    _call("ffn_fused", Lb.sfm_ffn_fused_ln, (_p(x32), _p(lnw), _p(lnb), _p(w1_16), _p(b1), _p(w2_16), _p(b2), _p(y), M, D, FF,
                                             float(alpha), float(eps), _p(ln2w), _p(ln2b), _p(ln), 1 if ln_f32 else 0, _dt(),
                                             _stream()), 4.0 * M * D * FF, M * D * (8.0 if want_y else 4.0) + M * D * (4.0 if ln_f32 else 2.0))
    return y, ln


def sinc_fir16(wave, filt, out, B, L, C, K, want_stats=True, passes=0):
    """SincConv1d FIR on 16-bit operands -> out [B, L, C] channels-last (+ GroupNorm partials).  passes: 3 = split operands
    (fp32-class accuracy), 1 = operands rounded once, 0 = auto (1 for fp16 operands with a 16-bit result, else 3)."""
    Lb = _lib.load()
    dev = wave.device
    wsh = torch.empty(8 * 2 * 64 * 272, device=dev, dtype=torch.int16)
    P = Lb.sfm_sinc_fir16_tiles(L)
    part = torch.zeros(B, P, 8, 2, device=dev, dtype=torch.float32) if want_stats else None
    out_f32 = 1 if out.dtype == torch.float32 else 0
    flops, nbytes = 2.0 * B * L * C * K, B * L * 4 + B * L * C * (4 if out_f32 else 2)
    _call("sinc_fir16", Lb.sfm_sinc_fir16_ex, (_p(wave), _p(filt), _p(wsh), _p(out), _p(part), B, L, C, K, out_f32, _dt(), int(passes),
                                               _stream()), flops, nbytes)
    return part, P


def wave_moments(est, tgt):
    L = _lib.load()
    B, Ln = est.shape
    S = torch.zeros(B, 5, device=est.device, dtype=torch.float64)
    _call("loss_reduce", L.sfm_wave_moments, (_p(est), _p(tgt), _p(S), B, Ln, _p(_ws(64 * B * 5, est.device, torch.float64)), _stream()),
          0.0, 8.0 * B * Ln)
    return S


def spec_sums(pr, pi, tr, ti, out=None):
    L = _lib.load()
    S = torch.zeros(4, device=pr.device, dtype=torch.float64) if out is None else out
    _call("loss_reduce", L.sfm_spec_sums, (_p(pr), _p(pi), _p(tr), _p(ti), _p(S), pr.numel(),
                                           _p(_ws(2048 * 4, pr.device, torch.float64)), _stream()), 0.0, 16.0 * pr.numel())
    return S


def enhancer_loss_finalize(Sw, Sm, Sr, nr, B, Ln, n_mag, R=None):
    """-> [total, neg SI-SNR, L1 magnitude, MR-STFT]; R: resolutions in Sr / nr (default: all rows; 0 = none)"""
    L = _lib.load()
    out = torch.empty(4, device=Sw.device, dtype=torch.float32)
    _call("loss_reduce", L.sfm_enhancer_loss_finalize, (_p(Sw), _p(Sm), _p(Sr), _p(nr), B, Ln, n_mag,
                                                        Sr.shape[0] if R is None else int(R), _p(out), _stream()))
    return out


def sisnr_bwd(est, tgt, Sw, dwave, scale=1.0):
    L = _lib.load()
    B, Ln = est.shape
    _call("loss_bwd", L.sfm_sisnr_bwd, (_p(est), _p(tgt), _p(Sw), _p(dwave), B, Ln, float(scale), _stream()), 0.0, 12.0 * B * Ln)


def spec_loss_bwd(pr, pi, tr, ti, S, dr, di, F, ld, mode, accumulate=False, scale=1.0):
    L = _lib.load()
    _call("loss_bwd", L.sfm_spec_loss_bwd, (_p(pr), _p(pi), _p(tr), _p(ti), _p(S), _p(dr), _p(di), pr.numel(), F, ld, mode,
                                            1 if accumulate else 0, float(scale), _stream()), 0.0, 24.0 * pr.numel())


def stft_adjoint_ola(frames, dwave, B, T, Ln, n_fft, hop, win, accumulate=True, post=None):
    L = _lib.load()
    _call("loss_bwd", L.sfm_stft_adjoint_ola, (_p(frames), _p(dwave), _p(post), B, T, Ln, n_fft, hop, win,
                                               1 if accumulate else 0, _stream()), 0.0, 4.0 * (B * T * win + 2 * B * Ln))


def polar_mask_bwd(lm, lp, nr, ni, der, dei, dlog, M, F, phase_scale, ld_logits, mag_bias=None, rows_per_batch=None):
    L = _lib.load()
    _call("loss_bwd", L.sfm_polar_mask_bwd, (_p(lm), _p(lp), _p(mag_bias), _p(nr), _p(ni), _p(der), _p(dei), _p(dlog), M,
                                             rows_per_batch or M, F, float(phase_scale), ld_logits, dlog.stride(0), _stream()),
          0.0, 32.0 * M * F)


def sum_time(src32, B, T, C, ld_src):
    L = _lib.load()
    out = torch.empty(B, C, device=src32.device, dtype=torch.float32)
    scratch = torch.empty(int(L.sfm_mean_time_scratch_floats(B, T, C)), device=src32.device, dtype=torch.float32)
    _call("pool_time", L.sfm_sum_time, (_p(src32), _p(out), _p(scratch), B, T, C, ld_src, _stream()), 0.0, 4.0 * B * T * C)
    return out


def _tn_ws(L_, M, N, K, device):
    """(workspace, its size in floats) of the ordered M-split fold of the TN GEMMs, or (None, 0) in the atomics form"""
    if not _DET["on"]:
        return None, 0
    n = int(L_.sfm_tn_ws_floats(M, N, K))
    return _ws(n, device), n


def sinc_wgrad(wave, dy, C, K, exact=False):
    """dfilt [C, K] = sum_{b,l} dy[b, l, c] * wave[b, l + k - K//2]   (dy [B, L, C] 16-bit or fp32).
    A 16-bit dy with the reference's 251 taps runs on the matrix cores (waveform rounded to the compute dtype, like the
    activations of every other weight gradient on the path); exact=True or an fp32 dy keeps the fp32 vector kernel."""
    L_ = _lib.load()
    B, L = wave.shape
    if not exact and K == 251 and dy.dtype == _state["dtype"] and C % 8 == 0:
        Lc = int(L_.sfm_sinc_shift_len(L))
        xs = torch.empty(B, 8, Lc, device=wave.device, dtype=dy.dtype)
        _call("sinc_shift_pack", L_.sfm_sinc_shift_pack, (_p(wave), _p(xs), B, L, _dt(), _stream()), 0.0, B * (4.0 * L + 16.0 * Lc))
        dW = torch.zeros(C, 256, device=wave.device, dtype=torch.float32)
        ws, wsn = _tn_ws(L_, B * L, C, 256, wave.device)
        _call("sinc_wgrad16", L_.sfm_sinc_wgrad16, (_p(dy), _p(xs), _p(dW), B, L, C, _dt(), _p(ws), wsn, _stream()),
              2.0 * B * L * C * 256, B * L * (2.0 * C + 16.0))
        return dW[:, :K].contiguous()
    dfilt = torch.zeros(C, K, device=wave.device, dtype=torch.float32)
    scratch = torch.empty(int(L_.sfm_sinc_wgrad_scratch_floats(B, L, C, K)), device=wave.device, dtype=torch.float32)
    _call("sinc_wgrad", L_.sfm_sinc_wgrad, (_p(wave), _p(dy), 1 if dy.dtype == torch.float32 else 0, _p(dfilt), _p(scratch), B, L, C,
                                            K, _dt(), _stream()), 2.0 * B * L * C * K, B * L * (4.0 + C * dy.element_size()))
    return dfilt


def gn_stats(partial, rows, C, G):
    """group mean / rstd [B, G] from the per-64-row partial sums a producing GEMM wrote ([B, P, G, 2] = sum, sum of squares)"""
    s = partial.double().sum(dim=1)
    n = float(rows * (C // G))
    mean = s[..., 0] / n
    var = (s[..., 1] / n - mean * mean).clamp_min(0.0)
    return mean.float(), torch.rsqrt(var + 1e-5).float()


def gn_act_backward(dout, act, G, x1, sc1, sh1, mean1, rstd1, gamma1, x2=None, sc2=None, sh2=None, mean2=None, rstd2=None,
                    gamma2=None, dx_dtype=None):
    """backward of out = act(GN(x1) [+ GN(x2)]) on channels-last [B, L, C] tensors (PerceptionAgent nodes).
    sc / sh [B, C]: the forward's scale and shift; mean / rstd [B, G]; gamma [C].
    Returns dx1, dgamma1, dbeta1 (and dx2, dgamma2, dbeta2 when two inputs)."""
    L_ = _lib.load()
    B, L, C = x1.shape
    dx_dtype = dx_dtype or _state["dtype"]
    two = x2 is not None
    dev = x1.device
    f32 = lambda t: 1 if t.dtype == torch.float32 else 0
    c32 = lambda t: None if t is None else t.float().contiguous()
    dout = dout.contiguous()
    sc1, sh1, mean1, rstd1, gamma1 = c32(sc1), c32(sh1), c32(mean1), c32(rstd1), c32(gamma1)
    sc2, sh2, mean2, rstd2, gamma2 = c32(sc2), c32(sh2), c32(mean2), c32(rstd2), c32(gamma2)
    work = torch.zeros(3 * B * C + 3 * C, device=dev, dtype=torch.float32)       # S [B][3][C] | dparam [3][C]
    S, dparam = work[:3 * B * C], work[3 * B * C:].view(3, C)
    nbytes = float(B * L * C) * ((4 if f32(dout) else 2) + (1 + two) * (4 if f32(x1) else 2))
    ws = _ws(int(L_.sfm_gn_bwd_reduce_ws_floats(B, L, C)), dev) if _DET["on"] else None
    _call("gn_bwd_reduce", L_.sfm_gn_bwd_reduce, (_p(dout), f32(dout), _p(x1), f32(x1), _p(sc1), _p(sh1), _p(mean1), _p(rstd1), _p(x2),
                                           f32(x2) if two else 0, _p(sc2), _p(sh2), _p(mean2), _p(rstd2), _p(S), B, L, C, G,
                                           int(act), _dt(), _p(ws), _stream()), 0.0, nbytes)
    coef = torch.empty(2, 3, B, C, device=dev, dtype=torch.float32)
    _call("gn_bwd_coefs", L_.sfm_gn_bwd_coefs, (_p(S), _p(gamma1), _p(rstd1), _p(gamma2), _p(rstd2), _p(coef[0]),
                                                _p(coef[1]) if two else None, _p(dparam), B, L, C, G, _stream()), 0.0,
          4.0 * 9 * B * C)
    dx1 = torch.empty(B, L, C, device=dev, dtype=dx_dtype)
    dx2 = torch.empty(B, L, C, device=dev, dtype=dx_dtype) if two else None
    _call("gn_bwd_apply", L_.sfm_gn_bwd_apply, (_p(dout), f32(dout), _p(x1), f32(x1), _p(sc1), _p(sh1), _p(mean1), _p(rstd1), _p(coef[0]),
                                          _p(dx1), f32(dx1), _p(x2), f32(x2) if two else 0, _p(sc2), _p(sh2), _p(mean2), _p(rstd2),
                                          _p(coef[1]) if two else None, _p(dx2), f32(dx2) if two else 0, B, L, C, G, int(act),
                                          _dt(), _stream()), 0.0, nbytes + float(B * L * C) * (1 + two) * (4 if f32(dx1) else 2))
    if two:
        return dx1, dparam[1], dparam[0], dx2, dparam[2], dparam[0]
    return dx1, dparam[1], dparam[0]


# ---------------------------------------------------------------------------
# training path (ConformerBlock backward)
# ---------------------------------------------------------------------------
_WGRAD = {"stream": None, "keep": []}


class wgrad_side_stream:
    """inside this context the weight-gradient GEMMs are enqueued on a second HIP stream: they are off the critical path of
    the backward pass (nothing but the optimiser reads dW), so they fill the CUs the input-gradient chain leaves idle; on
    exit the current stream waits for them."""

    def __init__(self, enabled=True):
        self.enabled = enabled and _os.environ.get("SFM_WGRAD_STREAM", "1") != "0"

    def __enter__(self):
        if self.enabled:
            if "side" not in _WGRAD:
                _WGRAD["side"] = torch.cuda.Stream()
            _WGRAD["stream"] = _WGRAD["side"]
        return self

    def __exit__(self, *exc):
        if self.enabled:
            torch.cuda.current_stream().wait_stream(_WGRAD["side"])
            _WGRAD["stream"] = None
            # operands of the side-stream GEMMs were kept alive until the join is enqueued: whoever reuses their memory on the
            # main stream now runs after it.  (tensor.record_stream() instead defers the reuse to an event query; with the host
            # several steps ahead of the GPU the allocator then grows by hipMalloc: sporadic 1 s stalls in bench.py)
            _WGRAD["keep"].clear()
        return False


class after_wgrad:
    """host ops that post-process a weight gradient (e.g. rescaling rows) must run where the GEMM that produces it runs:
    on the side stream when wgrad_side_stream is active (the listed temporaries are kept alive for it), else in place."""

    def __init__(self, *temporaries):
        self.ws = _WGRAD["stream"]
        self.tmp = temporaries
        self.ctx = None

    def __enter__(self):
        if self.ws is not None:
            self.ctx = torch.cuda.stream(self.ws)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ws is not None:
            self.ctx.__exit__(*exc)
            _WGRAD["keep"].extend(self.tmp)
        return False


def gemm16_tn(G16, X16, dW, db=None):
    """dW[n,k] += sum_m G16[m,n] * X16[m,k]   (dW fp32, accumulated); db[n] += sum_m G16[m,n] when given."""
    ws = _WGRAD["stream"]
    if ws is not None and torch.cuda.current_stream() != ws:
        ws.wait_stream(torch.cuda.current_stream())          # operands (and the zero-filled dW) are ready
        with torch.cuda.stream(ws):
            gemm16_tn(G16, X16, dW, db)
        _WGRAD["keep"].extend((G16, X16))                      # alive until the join (see wgrad_side_stream.__exit__)
        return
    L = _lib.load()
    M, N = G16.shape
    K = X16.shape[1]
    ws, wsn = _tn_ws(L, M, N, K, G16.device)
    _call("gemm16_tn", L.sfm_gemm16_tn, (_p(G16), _p(X16), _p(dW), _p(db), M, N, K, G16.stride(0), X16.stride(0), dW.stride(0),
                                         _dt(), _p(ws), wsn, _stream()), 2.0 * M * N * K, M * (N + K) * 2.0,
          tag="M%d N%d K%d" % (M, N, K))


def conv_wgrad16(dy16, x16, B, Lout, Lin, Cin, N, ksize, stride, pad):
    """Conv1d weight / bias gradient from dY [B*Lout, N] and the channels-last input x [B, Lin, Cin] (both 16-bit):
    returns dW in torch's Conv1d layout [N, Cin, ksize] and db [N] (fp32)."""
    ws = _WGRAD["stream"]
    if ws is not None and torch.cuda.current_stream() != ws:   # off the critical path: see wgrad_side_stream
        ws.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws):
            out = conv_wgrad16(dy16, x16, B, Lout, Lin, Cin, N, ksize, stride, pad)
        _WGRAD["keep"].extend((dy16, x16))
        return out
    L = _lib.load()
    buf = torch.zeros(N * ksize * Cin + N, device=x16.device, dtype=torch.float32)       # dW | db: one fill
    dW, db = buf[:N * ksize * Cin].view(N, ksize * Cin), buf[N * ksize * Cin:]
    ws, wsn = _tn_ws(L, B * Lout, N, ksize * Cin, x16.device)
    _call("gemm16_tn", L.sfm_conv_wgrad16, (_p(dy16), _p(x16), _p(dW), _p(db), B, Lout, Lin, Cin, N, ksize, stride, pad,
                                            Lin * Cin, dy16.stride(0), dW.stride(0), _dt(), _p(ws), wsn, _stream()),
          2.0 * B * Lout * N * ksize * Cin, (B * Lout * N + B * Lin * Cin) * 2.0,
          tag="conv M%d N%d K%d s%d" % (B * Lout, N, ksize * Cin, stride))
    return dW.reshape(N, ksize, Cin).permute(0, 2, 1).contiguous(), db


def conv_dgrad16(dy16, weight, B, Lout, Lin, stride, pad, accumulate_into=None, out_dtype=torch.float32, add_even=None):
    """input gradient of Conv1d(weight [N, Cin, k], stride 1 or 2, zero padding) on channels-last tensors:
    dy16 [B, Lout, N] 16-bit -> dx [B, Lin, Cin] in `out_dtype` (fp32 or the 16-bit compute type).  Stride 1 is the
    correlation with the flipped, transposed kernel; stride 2 is that per output parity (two implicit GEMMs writing the
    even / odd rows).  Every row is written exactly once, so there is no zero-fill and no read-modify-write:
      add_even [B, ceil(Lin / stride), Cin] fp32: added to the rows of parity 0 in the GEMM's epilogue - the input gradient of a
                 parallel k = 1 convolution with the same stride (the residual blocks' skip path, agents/perception.py:121-129);
      accumulate_into (fp32 [B, Lin, Cin]): the older read-modify-write form, kept for a parity that has no tap."""
    N, Cin, k = weight.shape
    w = weight.detach().float()
    plan = []
    for r in range(stride):
        t0 = (r + pad) % stride
        taps = list(range(t0, k, stride))
        nq = (Lin - r + stride - 1) // stride
        plan.append((r, t0, taps, nq))
    covered = all(taps and nq > 0 for _, _, taps, nq in plan)
    if accumulate_into is not None:
        dx = accumulate_into
    elif covered:
        dx = torch.empty(B, Lin, Cin, device=dy16.device, dtype=out_dtype)
    else:
        dx = torch.zeros(B, Lin, Cin, device=dy16.device, dtype=out_dtype)
    for r, t0, taps, nq in plan:
        if not taps or nq <= 0:
            continue
        J = len(taps)
        c0 = (r + pad - t0) // stride
        # out[q] = sum_j' in[q - pad2 + j'] Wf[j'],  Wf[j'] = W[:, :, taps[J-1-j']]^T,  pad2 = J - 1 - c0
        wf = torch.stack([w[:, :, taps[J - 1 - j]] for j in range(J)], dim=2).permute(1, 0, 2).contiguous()   # [Cin, N, J]
        pw = pack_linear(wf)
        view = dx.reshape(B, Lin * Cin)[:, r * Cin:]
        kw = dict(B=B, Lout=nq, Lin=Lout, a_batch_stride=Lout * N, ldo=stride * Cin, o_batch_stride=Lin * Cin, stride=1,
                  pad=J - 1 - c0)
        if accumulate_into is not None:
            gemm16(dy16, pw, view, epi=EPI_RESID, resid=view, ldr=stride * Cin, r_batch_stride=Lin * Cin, alpha=1.0, **kw)
        elif add_even is not None and r == 0:
            assert add_even.dtype == torch.float32 and add_even.shape[1] == nq
            gemm16(dy16, pw, view, epi=EPI_RESID, resid=add_even, ldr=Cin, r_batch_stride=nq * Cin, alpha=1.0, **kw)
        else:
            gemm16(dy16, pw, view, **kw)
    return dx


def colsum(G, out):
    L = _lib.load()
    M, N = G.shape
    ws = _ws(int(L.sfm_colsum_ws_floats(M, N)), G.device) if _DET["on"] else None
    _call("colsum", L.sfm_colsum, (_p(G), _p(out), M, N, G.stride(0), 1 if G.dtype == torch.float32 else 0, _dt(), _p(ws), _stream()))


def layernorm_bwd(x32, gamma, dy, dres32, dgamma, dbeta, eps=1e-5, next_drop=None):
    """dy: fp32, or the 16-bit result of the GEMM that produced it (its own row stride); dres / dx fp32 [M, D].
    next_drop = (alpha, p, seed): also returns alpha * dropout(dx; p, seed) [M, D] in the 16-bit compute format, the operand the
    next backward node of the residual chain starts from (same counters as ew_train(EW_SCALE_DROP)) -> (dx, next16)."""
    L = _lib.load()
    M, D = dy.shape
    dx = torch.empty(M, D, device=dy.device, dtype=torch.float32)
    ws = _ws(int(L.sfm_layernorm_bwd_ws_floats(M, D)), dy.device) if _DET["on"] else None
    dy16 = 0 if dy.dtype == torch.float32 else 1
    if dy16 and dy.dtype != _state["dtype"]:
        raise RuntimeError("layernorm_bwd: a 16-bit dy must be in the compute format")
    if next_drop is not None:
        alpha, p, seed = next_drop
        nxt = torch.empty(M, D, device=dy.device, dtype=_state["dtype"])
        _call("layernorm_bwd", L.sfm_layernorm_bwd_next, (_p(x32), _p(gamma), _p(dy), dy16, _p(dres32), _p(dx), _p(dgamma), _p(dbeta), M, D,
                                                          x32.stride(0), dy.stride(0), D, eps, _dt(), _p(nxt), float(alpha), float(p),
                                                          int(seed) & 0xffffffff, _p(ws), _stream()))
        return dx, nxt
    _call("layernorm_bwd", L.sfm_layernorm_bwd_ex, (_p(x32), _p(gamma), _p(dy), dy16, _p(dres32), _p(dx), _p(dgamma), _p(dbeta), M, D,
                                                    x32.stride(0), dy.stride(0), D, eps, _dt(), _p(ws), _stream()))
    return dx


EW_SWISH_FWD, EW_SWISH_BWD, EW_GLU_FWD, EW_GLU_BWD, EW_SCALE_DROP, EW_GELU_FWD, EW_GELU_BWD, EW_CPEA_BWD = range(8)


def ew_train(mode, out, z=None, g=None, N=None, alpha=1.0, p=0.0, seed=0):
    L = _lib.load()
    M = out.shape[0]
    N = N if N is not None else (out.shape[1] // 2 if mode == EW_GLU_BWD else out.shape[1])
    _call("ew_train", L.sfm_ew_train, (_p(z), _p(g), _p(out), M, N, mode, 1 if (g is not None and g.dtype == torch.float32) else 0,
                                       1 if out.dtype == torch.float32 else 0, float(alpha), float(p), int(seed) & 0xffffffff,
                                       _dt(), _stream()))
    return out


def col_stats(y32, aux=None, mean=None, rstd=None):
    L = _lib.load()
    M, C = y32.shape
    S = torch.zeros(C, 2, device=y32.device, dtype=torch.float32)
    ws = _ws(int(L.sfm_col_stats_ws_floats(M, C)), y32.device) if _DET["on"] else None
    _call("col_stats", L.sfm_col_stats, (_p(y32), _p(aux), _p(mean), _p(rstd), _p(S), M, C, _p(ws), _stream()))
    return S


def add_cols(a, b, M, C, Cb):
    """out [M, C] fp32 = a[:, :C] + pad(b[:, :Cb]); a / b may be row-strided views"""
    L = _lib.load()
    out = torch.empty(M, C, device=a.device, dtype=torch.float32)
    _call("ew_train", L.sfm_add_cols, (_p(a), _p(b), _p(out), M, C, Cb, a.stride(0), b.stride(0), C, _stream()))
    return out


def lstm_hprev16(h32, B, T, H):
    L = _lib.load()
    out = torch.empty(B * T, 2 * H, device=h32.device, dtype=_state["dtype"])
    _call("convert_rows", L.sfm_lstm_hprev16, (_p(h32), _p(out), B, T, H, _dt(), _stream()))
    return out


def bn_finalize(S, gamma, beta, run_mean, run_var, M, eps, momentum, eval_mode=False):
    """-> mean, rstd, sc [1, C], sh [1, C]; the fp32 running statistics are updated in place (training)"""
    L = _lib.load()
    C = gamma.numel()
    buf = torch.empty(4, C, device=gamma.device, dtype=torch.float32)
    _call("col_stats", L.sfm_bn_finalize, (_p(S), _p(gamma), _p(beta), _p(run_mean), _p(run_var), _p(buf[0]), _p(buf[1]), _p(buf[2]),
                                           _p(buf[3]), C, M, float(eps), float(momentum), 1 if eval_mode else 0, _stream()))
    return buf[0], buf[1], buf[2:3], buf[3:4]


def bn_swish_bwd(g, y32, mean, rstd, gamma, beta, eval_mode=False):
    """backward through Swish(BatchNorm(y)): returns dy [M,C] fp32, dgamma, dbeta.  eval_mode: mean / rstd are the
    running statistics (constants), so dy carries no batch-statistics correction."""
    L = _lib.load()
    M, C = y32.shape
    S = torch.zeros(C, 2, device=y32.device, dtype=torch.float32)
    dy = torch.empty_like(y32)
    gf = 1 if g.dtype == torch.float32 else 0
    ws = _ws(int(L.sfm_col_stats_ws_floats(M, C)), y32.device) if _DET["on"] else None
    for ps in (0, 1):
        Sx = torch.zeros_like(S) if (eval_mode and ps == 1) else S
        _call("bn_swish_bwd", L.sfm_bn_swish_bwd, (_p(g), _p(y32), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(Sx), _p(dy), M, C, gf,
                                                   ps, _dt(), _p(ws), _stream()))
    return dy, S[:, 1].contiguous(), S[:, 0].contiguous()


def dwconv_wgrad(x16, dy32, B, T, C, KS):
    L = _lib.load()
    dw = torch.zeros(C, KS, device=x16.device, dtype=torch.float32)
    db = torch.zeros(C, device=x16.device, dtype=torch.float32)
    scratch = torch.empty(int(L.sfm_dwconv_wgrad_scratch_floats(B, T, C, KS)), device=x16.device, dtype=torch.float32)
    _call("dwconv_wgrad", L.sfm_dwconv_wgrad, (_p(x16), _p(dy32), _p(dw), _p(db), _p(scratch), B, T, C, KS, _dt(), _stream()))
    return dw, db


def attention_train(qkv16, B, T, H, hd, p_drop=0.0, seed=0):
    """training forward (q pre-scaled): returns O [B*T, D] 16-bit and LSE [B,H,T] fp32 (log2 domain)."""
    L = _lib.load()
    D = H * hd
    ld = qkv16.stride(0)
    out = torch.empty(B * T, D, device=qkv16.device, dtype=qkv16.dtype)
    lse = torch.empty(B, H, T, device=qkv16.device, dtype=torch.float32)
    _call("attention_fwd", L.sfm_attention_fwd_train, (_p(qkv16), _p(out), _p(lse), B, T, H, hd, ld, out.stride(0), D, 2 * D,
                                                       T * ld, T * out.stride(0), -1.0, float(p_drop), int(seed) & 0xffffffff,
                                                       _dt(), _stream()), 4.0 * B * H * T * T * hd, 4.0 * B * T * D * 2)
    return out, lse


def attention_bwd(qkv16, O16, dO16, lse, B, T, H, hd, p_drop=0.0, seed=0):
    L = _lib.load()
    D = H * hd
    dqkv = torch.empty_like(qkv16)
    if hd != 64:                                   # small-shape path (e.g. the reference's test config: 4 heads x 16)
        scratch = torch.zeros(B * T, 2 * D, device=qkv16.device, dtype=torch.float32)
        _call("attention_bwd", L.sfm_attention_bwd_generic, (_p(qkv16), _p(O16), _p(dO16), _p(lse), _p(scratch), _p(dqkv), B, T, H,
                                                             hd, qkv16.stride(0), O16.stride(0), D, 2 * D, float(p_drop),
                                                             int(seed) & 0xffffffff, _dt(), _stream()),
              10.0 * B * H * T * T * hd, 8.0 * B * T * D * 2)
        return dqkv
    delta = torch.empty(B, H, T, device=qkv16.device, dtype=torch.float32)
    _call("attention_bwd", L.sfm_attention_bwd, (_p(qkv16), _p(O16), _p(dO16), _p(lse), _p(delta), _p(dqkv), B, T, H, hd,
                                                 qkv16.stride(0), O16.stride(0), D, 2 * D, float(p_drop), int(seed) & 0xffffffff,
                                                 _dt(), _stream()), 10.0 * B * H * T * T * hd, 8.0 * B * T * D * 2)
    return dqkv


_LSTM = {"w16": _os.environ.get("SFM_LSTM_W16", "1") != "0"}


def set_lstm_w16(flag):
    """inference BiLSTM recurrence on fp16 operands (sfm_bilstm_layer_ex, default on; SFM_LSTM_W16=0 / False = fp32 W_hh and h)"""
    _LSTM["w16"] = bool(flag)


def lstm_w16():
    return _LSTM["w16"]


def bilstm_layer(xg, whh, B, T, H, w16=False):
    """one BiLSTM layer's recurrence (inference).  w16: recurrent product on fp16 operands (the fused path passes lstm_w16())"""
    L = _lib.load()
    out = torch.empty(B, T, 2 * H, device=xg.device, dtype=torch.float32)
    if w16:
        _call("bilstm_layer", L.sfm_bilstm_layer_ex, (_p(xg), _p(whh), _p(out), B, T, H, 1, _stream()),
              *_cost_of("bilstm_layer", locals()))
    else:
        _call("bilstm_layer", L.sfm_bilstm_layer, (_p(xg), _p(whh), _p(out), B, T, H, _dt(), _stream()),
              *_cost_of("bilstm_layer", locals()))
    return out


def bilstm_layer_train(xg, whh, B, T, H, w16=None):
    """forward of one BiLSTM layer that also saves the activated gates / cell states [B, T, 2, 5, H] for the BPTT.
    w16 (default: lstm_w16() when the training format is fp16): the recurrent product on fp16 operands, as under the reference's
    fp16 autocast; the saved state and the backward stay fp32"""
    L = _lib.load()
    out = torch.empty(B, T, 2 * H, device=xg.device, dtype=torch.float32)
    save = torch.empty(B, T, 2, 5, H, device=xg.device, dtype=torch.float32)
    if w16 is None:
        w16 = _LSTM["w16"] and _state["dtype"] == torch.float16
    if w16:
        _call("bilstm_layer", L.sfm_bilstm_layer_train_ex, (_p(xg), _p(whh), _p(out), _p(save), B, T, H, 1, _stream()))
    else:
        _call("bilstm_layer", L.sfm_bilstm_layer_train, (_p(xg), _p(whh), _p(out), _p(save), B, T, H, _dt(), _stream()))
    return out, save


def bilstm_layer_bwd(save, whh, dout, B, T, H):
    """dout [B, T, 2H] fp32 -> gradient w.r.t. the input projection xg [B, T, 2, 4H] fp32"""
    L = _lib.load()
    dxg = torch.empty(B, T, 2, 4 * H, device=dout.device, dtype=torch.float32)
    _call("bilstm_bwd", L.sfm_bilstm_layer_bwd, (_p(save), _p(whh), _p(dout), _p(dxg), B, T, H, _stream()))
    return dxg


def stft_lognorm_bwd(re, im, g, M, F):
    L = _lib.load()
    dre, dim_ = torch.empty_like(re), torch.empty_like(im)
    _call("stft_lognorm_bwd", L.sfm_stft_lognorm_bwd, (_p(re), _p(im), _p(g), _p(dre), _p(dim_), M, F, g.stride(0), _stream()))
    return dre, dim_


def memory_bwd(emb, params, d_out, d_gate, key_dim, value_dim, slots, temperature, want_d_emb=True):
    """-> (d_emb [B, key_dim] or None, dparams: blob laid out like `params`)"""
    L = _lib.load()
    Bn = emb.shape[0]
    d_emb = torch.empty(Bn, key_dim, device=emb.device, dtype=torch.float32) if want_d_emb else None
    dparams = torch.zeros_like(params)
    ws = None
    if _DET["on"]:
        npar = int(L.sfm_memory_param_floats(key_dim, value_dim, slots))
        if npar != params.numel():
            raise RuntimeError("memory_bwd: the parameter blob has %d floats, the kernel's layout %d" % (params.numel(), npar))
        ws = _ws(Bn * npar, emb.device)
    _call("memory_bwd", L.sfm_memory_bwd, (_p(emb), _p(params), _p(d_out), _p(d_gate), _p(d_emb), _p(dparams), Bn, key_dim, value_dim,
                                           slots, float(temperature), _p(ws), _stream()))
    return d_emb, dparams


def memory_fwd(emb, params, key_dim, value_dim, slots, temperature):
    L = _lib.load()
    Bn = emb.shape[0]
    dev = emb.device
    bias = torch.empty(Bn, value_dim, device=dev, dtype=torch.float32)
    gate = torch.empty(Bn, 1, device=dev, dtype=torch.float32)
    top = torch.empty(Bn, device=dev, dtype=torch.int32)
    sim = torch.empty(Bn, device=dev, dtype=torch.float32)
    _call("memory_fwd", L.sfm_memory_fwd, (_p(emb), _p(params), _p(bias), _p(gate), _p(top), _p(sim), Bn, key_dim, value_dim, slots,
                          float(temperature), _stream()),
          *_cost_of("memory_fwd", locals()))
    return bias, gate, top, sim


# ---------------------------------------------------------------------------
# constant operands for STFT / iSTFT as fp32 matrix products (host, float64 twiddles)
# ---------------------------------------------------------------------------
def _hann(n):
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def stft_matrix(n_fft, win, device):
    """[win, 2F] : columns [0,F) real part, [F,2F) imag part, restricted to the
    window support (torch.stft zero-pads the window to n_fft centred)."""
    F = n_fft // 2 + 1
    woff = (n_fft - win) // 2
    w = _hann(win)
    n = (np.arange(win, dtype=np.float64) + woff)[:, None]
    f = np.arange(F, dtype=np.float64)[None, :]
    ang = 2.0 * np.pi * ((n * f) % n_fft) / n_fft
    M = np.concatenate([w[:, None] * np.cos(ang), -w[:, None] * np.sin(ang)], axis=1)
    return pack_f32_matrix(torch.from_numpy(M.astype(np.float32)).to(device))


def istft_matrix(n_fft, win, device):
    """[2F(+pad), win]: irfft restricted to the window support, times the window."""
    F = n_fft // 2 + 1
    woff = (n_fft - win) // 2
    w = _hann(win)
    n = (np.arange(win, dtype=np.float64) + woff)[None, :]
    f = np.arange(F, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * ((f * n) % n_fft) / n_fft
    coef = np.full((F, 1), 2.0)
    coef[0, 0] = 1.0
    if n_fft % 2 == 0:
        coef[F - 1, 0] = 1.0
    br = coef * np.cos(ang) / n_fft * w[None, :]
    bi = -coef * np.sin(ang) / n_fft * w[None, :]
    bi[0, :] = 0.0
    if n_fft % 2 == 0:
        bi[F - 1, :] = 0.0
    M = np.concatenate([br, bi], axis=0)
    win2 = torch.from_numpy((w * w).astype(np.float32)).to(device)
    return pack_f32_matrix(torch.from_numpy(M.astype(np.float32)).to(device)), win2


# ---------------------------------------------------------------------------
# MetacognitiveArbitrationAgent / VectorQuantizer kernels (routing.hip)
# ---------------------------------------------------------------------------
def maa_update_stats(sigma, stats, num_updates, momentum=0.1):
    """train()-mode EMA of agents/maa.py:126-135, in place on the device tensors stats [2] fp32 and num_updates int64 []"""
    L = _lib.load()
    acc = torch.zeros(2, device=sigma.device, dtype=torch.float64)
    _call("maa", L.sfm_maa_update_stats, (_p(sigma), sigma.numel(), _p(acc), _p(stats), _p(num_updates), float(momentum), _stream()),
          0.0, 4.0 * sigma.numel())


def maa_forward(sigma, stats, params):
    L = _lib.load()
    n = sigma.numel()
    dev = sigma.device
    logits = torch.empty(n, 4, device=dev, dtype=torch.float32)
    probs = torch.empty(n, 4, device=dev, dtype=torch.float32)
    dec = torch.empty(n, device=dev, dtype=torch.int64)
    conf = torch.empty(n, device=dev, dtype=torch.float32)
    _call("maa", L.sfm_maa_forward, (_p(sigma), _p(stats), _p(params), _p(logits), _p(probs), _p(dec), _p(conf), n, _stream()),
          2.0 * n * (64 + 64 * 64 + 4 * 64), n * 52.0)
    return logits, probs, dec, conf


def maa_backward(sigma, stats, params, g_logits, g_probs, g_conf):
    """-> dsigma [N] fp32, dparams fp32 [4548] in the layout of `params` (weight gradients as three TN GEMMs)"""
    L = _lib.load()
    n = sigma.numel()
    dev, dt = sigma.device, _state["dtype"]
    dsig = torch.empty(n, device=dev, dtype=torch.float32)
    H = torch.empty(4, n, 64, device=dev, dtype=dt)                     # H1, H2, dZ1, dZ2
    S = torch.empty(2, n, 8, device=dev, dtype=dt)                      # GL, XN
    _call("maa", L.sfm_maa_backward, (_p(sigma), _p(stats), _p(params), _p(g_logits), _p(g_probs), _p(g_conf), _p(dsig), _p(H[0]),
                                      _p(H[1]), _p(H[2]), _p(H[3]), _p(S[0]), _p(S[1]), n, _dt(), _stream()),
          4.0 * n * (64 * 64 * 2 + 8 * 64), n * (16.0 + 4 * 128 + 32))
    buf = torch.zeros(64 * 8 + 64 + 64 * 64 + 64 + 8 * 64 + 8, device=dev, dtype=torch.float32)
    o = 0
    dW1 = buf[o:o + 512].view(64, 8); o += 512
    db1 = buf[o:o + 64]; o += 64
    dW2 = buf[o:o + 4096].view(64, 64); o += 4096
    db2 = buf[o:o + 64]; o += 64
    dW3 = buf[o:o + 512].view(8, 64); o += 512
    db3 = buf[o:o + 8]
    gemm16_tn(H[2], S[1], dW1, db1)                                      # dZ1^T x [norm | 0...]: column 0 is dW1
    gemm16_tn(H[3], H[0], dW2, db2)
    gemm16_tn(S[0], H[1], dW3, db3)
    dparams = torch.cat([dW1[:, 0], db1, dW2.reshape(-1), db2, dW3[:4].reshape(-1), db3[:4]])
    return dsig, dparams


def vq_forward(x, centroids):
    """-> quantized fp32 (shape of x), indices int64, sum of squared distances (device double [1])"""
    L = _lib.load()
    x = x.contiguous()
    q = torch.empty_like(x)
    idx = torch.empty(x.shape, device=x.device, dtype=torch.int64)
    acc = torch.zeros(1, device=x.device, dtype=torch.float64)
    _call("vq", L.sfm_vq_forward, (_p(x), _p(centroids), centroids.numel(), _p(q), _p(idx), _p(acc), x.numel(), _stream()),
          0.0, 16.0 * x.numel())
    return q, idx, acc


def vq_backward(x, idx, centroids, g_q, g_loss, beta):
    L = _lib.load()
    dx = torch.empty_like(x)
    dcent = torch.zeros(centroids.numel(), device=x.device, dtype=torch.float32)
    _call("vq", L.sfm_vq_backward, (_p(x), _p(idx), _p(centroids), centroids.numel(), _p(g_q), _p(g_loss), float(beta), _p(dx),
                                     _p(dcent), x.numel(), _stream()), 0.0, 20.0 * x.numel())
    return dx, dcent
