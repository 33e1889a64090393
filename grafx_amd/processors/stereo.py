"""Stereo utilities (mirrors grafx.processors.stereo.StereoGain — reference stereo.py:9-48)."""
import torch.nn as nn

from .. import ops
from .core._buffer_io import BufferIO
from .core._grad import forward_only


class StereoGain(BufferIO, nn.Module):
    def forward(self, input_signals, log_gain):
        forward_only(input_signals, log_gain)
        return ops.stereo_gain(input_signals, log_gain)

    def render_into(self, x4, out4, log_gain):
        forward_only(x4, log_gain)
        return ops.stereo_gain(x4, log_gain, out=out4)

    def parameter_size(self):
        return {"log_gain": 2}
