"""Deterministic synthetic weights and inputs (numpy only, platform independent).

There is no network for checkpoints or datasets, so every test, the golden
generator (tests/golden/make_golden.py), smoke() and bench.py draw weights and
waveforms from here.  A tensor's values depend only on (seed, key name, shape):
the generator is numpy's PCG64 seeded with [seed, crc32(key)], so the reference
modules (in the generator), the oracle and the HIP path all see bit-identical
parameters without any weight file being committed.

Scale rules (chosen so activations stay O(1) through the whole path and the
mask heads are NOT saturated, see SURVEY.md §8c "de-saturated heads"):
  * >=2-D "weight"/"in_proj_weight"/LSTM weight : N(0,1)/sqrt(fan_in)
  * 1-D "weight" (norm scales)                  : 1 + 0.1 N(0,1)
  * 1-D "bias"                                  : 0.05 N(0,1)
  * running_mean N(0,0.1); running_var U(0.5,1.5); counters 0
  * memory keys/values                          : 0.5 N(0,1)
  * SincConv1d low_hz_/band_hz_/window/n_       : left at their analytic init
    (agents/perception.py:50-77 of the reference), optionally scaled
    (sinc_scale) so the sin() path is exercised.
"""
import zlib
import math
import numpy as np

_KEEP = ("low_hz_", "band_hz_", "window", "n_")


def _rng(seed, key):
    return np.random.default_rng([int(seed), zlib.crc32(key.encode())])


def synth_tensor(key, shape, seed):
    """Value for state_dict entry `key` of the given shape (float32 / int64)."""
    shape = tuple(int(s) for s in shape)
    leaf = key.split(".")[-1]
    g = _rng(seed, key)
    if leaf in ("num_batches_tracked", "num_queries"):
        return np.zeros(shape, dtype=np.int64)
    if leaf == "usage_count":
        return np.zeros(shape, dtype=np.float32)
    if leaf == "running_mean":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf in ("keys", "values"):
        return (0.5 * g.standard_normal(shape)).astype(np.float32)
    if "bias" in leaf:
        return (0.05 * g.standard_normal(shape)).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (g.standard_normal(shape) / math.sqrt(fan_in)).astype(np.float32)
    # 1-D weight: normalisation scale
    return (1.0 + 0.1 * g.standard_normal(shape)).astype(np.float32)


def synth_state_dict(shapes, seed, keep=None, sinc_scale=None):
    """shapes: {key: shape}.  keep: {key: ndarray} analytic values to retain
    (SincConv1d parameters/buffers).  sinc_scale multiplies low_hz_/band_hz_."""
    out = {}
    for key, shape in shapes.items():
        leaf = key.split(".")[-1]
        if leaf in _KEEP and keep is not None and key in keep:
            v = np.array(keep[key], dtype=np.float32, copy=True)
            if sinc_scale is not None and leaf in ("low_hz_", "band_hz_"):
                v = (v * np.float32(sinc_scale)).astype(np.float32)
            out[key] = v
        else:
            out[key] = synth_tensor(key, shape, seed)
    return out


def synth_wave(batch, length, seed, snr_cycle=(-5.0, 0.0, 5.0, 10.0)):
    """Noisy/clean waveform pairs, float32 [B, L].

    Speech proxy N(0, 0.1^2), noise N(0, 0.3^2) mixed at SNR cycled by index
    (mixing rule of training/conformer_pipeline.py:142-150)."""
    g = np.random.default_rng([int(seed), 0xA0D10])
    clean = (0.1 * g.standard_normal((batch, length))).astype(np.float32)
    noise = (0.3 * g.standard_normal((batch, length))).astype(np.float32)
    noisy = np.empty_like(clean)
    for i in range(batch):
        snr = snr_cycle[i % len(snr_cycle)]
        cp = float(np.mean(clean[i].astype(np.float64) ** 2)) + 1e-10
        npow = float(np.mean(noise[i].astype(np.float64) ** 2)) + 1e-10
        scale = math.sqrt(cp / (npow * 10 ** (snr / 10.0)))
        noisy[i] = (clean[i] + np.float32(scale) * noise[i]).astype(np.float32)
    return noisy, clean


def synth_array(key, shape, seed, scale=1.0):
    """Generic N(0, scale^2) float32 test input keyed by name."""
    return (scale * _rng(seed, "input:" + key).standard_normal(tuple(shape))).astype(np.float32)
