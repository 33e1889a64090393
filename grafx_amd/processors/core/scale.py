"""Frequency-scale maps used when laying out filterbanks and crossover frequencies (init-time, host side).

Mirrors grafx.processors.core.scale (reference core/scale.py:8-186): Hz <-> bark (Traunmueller, Schroeder,
Wang), Hz <-> mel (HTK, Slaney), linear, log.  Forward maps take a python float, inverse maps a tensor,
as upstream.  The Traunmueller inverse reproduces the upstream control flow: the low-end correction is
applied if ANY value is below 2 bark and only otherwise is the high-end correction considered
(core/scale.py:60-65), so a grid that spans both ends gets the low correction only.
"""
import math

import numpy as np
import torch

_BARK = ("traunmuller", "schroeder", "wang")
_MEL = ("htk", "slaney")
_SLANEY_STEP = 200.0 / 3          # Hz per mel below 1 kHz
_SLANEY_KNEE_HZ = 1000.0
_SLANEY_KNEE_MEL = _SLANEY_KNEE_HZ / _SLANEY_STEP
_SLANEY_LOG = math.log(6.4) / 27.0


def hz_to_bark(f, bark_scale="traunmuller"):
    if bark_scale not in _BARK:
        raise ValueError('bark_scale should be one of "schroeder", "traunmuller" or "wang".')
    if bark_scale == "wang":
        return 6.0 * math.asinh(f / 600.0)
    if bark_scale == "schroeder":
        return 7.0 * math.asinh(f / 650.0)
    z = 26.81 * f / (1960.0 + f) - 0.53
    if z < 2:
        return z + 0.15 * (2 - z)
    if z > 20.1:
        return z + 0.22 * (z - 20.1)
    return z


def bark_to_hz(z, bark_scale="traunmuller"):
    if bark_scale not in _BARK:
        raise ValueError('bark_scale should be one of "traunmuller", "schroeder" or "wang".')
    if bark_scale == "wang":
        return 600.0 * torch.sinh(z / 6.0)
    if bark_scale == "schroeder":
        return 650.0 * torch.sinh(z / 7.0)
    low, high = z < 2, z > 20.1
    if bool(low.any()):  # in place, like upstream (the caller's grid is modified)
        z[low] = (z[low] - 0.3) / 0.85
    elif bool(high.any()):
        z[high] = (z[high] + 4.422) / 1.22
    return 1960 * ((z + 0.53) / (26.28 - z))


def hz_to_mel(f, mel_scale="htk"):
    if mel_scale not in _MEL:
        raise ValueError('mel_scale should be one of "htk" or "slaney".')
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + f / 700.0)
    if f >= _SLANEY_KNEE_HZ:
        return _SLANEY_KNEE_MEL + math.log(f / _SLANEY_KNEE_HZ) / _SLANEY_LOG
    return f / _SLANEY_STEP


def mel_to_hz(m, mel_scale="htk"):
    if mel_scale not in _MEL:
        raise ValueError('mel_scale should be one of "htk" or "slaney".')
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f = _SLANEY_STEP * m
    upper = m >= _SLANEY_KNEE_MEL
    f[upper] = _SLANEY_KNEE_HZ * torch.exp(_SLANEY_LOG * (m[upper] - _SLANEY_KNEE_MEL))
    return f


def hz_to_log(f):
    if isinstance(f, torch.Tensor):
        return torch.log(f)
    return np.log(f) if isinstance(f, np.ndarray) else math.log(f)


def log_to_hz(v):
    if isinstance(v, torch.Tensor):
        return torch.exp(v)
    return np.exp(v) if isinstance(v, np.ndarray) else math.exp(v)


def _family(scale):
    if scale in ("bark_traunmuller", "bark_schroeder", "bark_wang"):
        return "bark", scale.split("_")[1]
    if scale in ("mel_htk", "mel_slaney"):
        return "mel", scale.split("_")[1]
    if scale in ("linear", "log"):
        return scale, None
    raise ValueError(f"Unsupported scale: {scale}")


def to_scale(freqs, scale):
    kind, variant = _family(scale)
    if kind == "bark":
        return hz_to_bark(freqs, bark_scale=variant)
    if kind == "mel":
        return hz_to_mel(freqs, mel_scale=variant)
    return freqs if kind == "linear" else hz_to_log(freqs)


def from_scale(values, scale):
    kind, variant = _family(scale)
    if kind == "bark":
        return bark_to_hz(values, bark_scale=variant)
    if kind == "mel":
        return mel_to_hz(values, mel_scale=variant)
    return values if kind == "linear" else log_to_hz(values)
