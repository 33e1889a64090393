"""Filters (mirrors grafx.processors.filter — reference filter.py:87-168 for BiquadFilter)."""
import torch.nn as nn

from .. import ops
from .core._buffer_io import BufferIO
from .. import autograd as diff
from ..autograd import needs_grad
from .core.iir import IIRFilter


class BiquadFilter(BufferIO, nn.Module):
    def __init__(self, num_filters=1, normalized=False, **backend_kwargs):
        super().__init__()
        self.num_filters = num_filters
        self.normalized = normalized
        self.biquad = IIRFilter(order=2, **backend_kwargs)

    def forward(self, input_signals, Bs, A1_pre, A2_pre, A0=None, _out=None):
        A0 = A0 if self.normalized else None
        if needs_grad(input_signals, Bs, A1_pre, A2_pre, A0):
            return self.biquad(input_signals, *diff.biquad_coefficients(Bs, A1_pre, A2_pre, A0), out=_out)
        Bs, As = ops.biquad_coeffs(Bs, A1_pre, A2_pre, A0)
        return self.biquad(input_signals, Bs.unsqueeze(1), As.unsqueeze(1), out=_out)

    def render_into(self, x4, out4, **params):
        return self.forward(x4, _out=out4, **params)

    def parameter_size(self):
        size = {"Bs": (self.num_filters, 3), "A1_pre": self.num_filters, "A2_pre": self.num_filters}
        if self.normalized:
            size["A0"] = self.num_filters
        return size
