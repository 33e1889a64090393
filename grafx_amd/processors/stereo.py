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

    def render_into(self, x4, out4, log_gain):
        if needs_grad(x4, log_gain):
            return super().render_into(x4, out4, log_gain=log_gain)
        return ops.stereo_gain(x4, log_gain, out=out4)

    def parameter_size(self):
        return {"log_gain": 2}
