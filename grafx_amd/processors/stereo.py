"""Stereo utilities (mirrors grafx.processors.stereo.StereoGain — reference stereo.py:9-48)."""
import torch.nn as nn

from .. import ops
from .core._buffer_io import BufferIO
import torch

from ..autograd import needs_grad


class StereoGain(BufferIO, nn.Module):
    def forward(self, input_signals, log_gain):
        if needs_grad(input_signals, log_gain):  # stereo.py:38-41 as torch ops (elementwise, HBM-bound either way)
            return input_signals * torch.exp(log_gain)[..., None]
        return ops.stereo_gain(input_signals, log_gain)

    accepts_mix = True   # _mix: the routing sum behind this stage, see ops.stereo_gain(mix=)

    def render_into(self, x4, out4, log_gain, _mix=None):
        if needs_grad(x4, log_gain):
            return super().render_into(x4, out4, log_gain=log_gain)
        return ops.stereo_gain(x4, log_gain, out=out4, mix=_mix)

    def parameter_size(self):
        return {"log_gain": 2}


INV_SQRT_2 = 1 / 2**0.5


class SideGainImager(nn.Module):
    """Stereo width via side-channel gain (stereo.py:51-100); elementwise."""

    def forward(self, input_signals, log_gain):
        assert input_signals.shape[1] == 2
        left, right = input_signals[:, 0, :], input_signals[:, 1, :]
        mid, side = left + right, (left - right) * torch.exp(log_gain)
        return torch.stack([(mid + side) / 2, (mid - side) / 2], 1)

    def parameter_size(self):
        return {"log_gain": 1}


class MonoToStereo(nn.Module):
    def forward(self, input_signals):
        assert input_signals.shape[1] == 1
        return input_signals.repeat(1, 2, 1)

    def parameter_size(self):
        return {}


class StereoToMidSide(nn.Module):
    def __init__(self, normalize=True):
        super().__init__()
        self.normalize = normalize

    def forward(self, input_signals):
        assert input_signals.shape[1] == 2
        if self.normalize:
            input_signals = input_signals * INV_SQRT_2
        left, right = input_signals[:, :1, :], input_signals[:, 1:, :]
        return left + right, left - right

    def parameter_size(self):
        return {}


class MidSideToStereo(nn.Module):
    def __init__(self, normalize=True):
        super().__init__()
        self.normalization_const = INV_SQRT_2 if normalize else 0.5

    def forward(self, mid, side):
        assert mid.shape[1] == 1
        return torch.cat([mid + side, mid - side], 1) * self.normalization_const

    def parameter_size(self):
        return {}
