"""Filters (mirrors grafx.processors.filter — reference filter.py:87-168 for BiquadFilter)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .core._buffer_io import BufferIO
from .. import autograd as diff
from ..autograd import needs_grad
from .core.convolution import FIRConvolution
from .core.iir import IIRFilter
from .core.midside import lr_to_ms, ms_to_lr
from .core.utils import normalize_impulse


class FIRFilter(nn.Module):
    """Learnable FIR taps: tanh -> energy normalisation -> causal convolution (reference filter.py:20-84).
    Upstream's constructor reads ``self.processor_channel`` before assigning it (filter.py:39) and therefore
    cannot be instantiated; this class implements what the rest of that code plainly intends."""

    def __init__(self, fir_len=1023, processor_channel="mono", **backend_kwargs):
        super().__init__()
        if processor_channel not in ("mono", "stereo", "midside"):
            raise ValueError(f"Unknown channel type: {processor_channel}")
        self.fir_len = fir_len
        self.processor_channel = processor_channel
        self.num_channels = 1 if processor_channel == "mono" else 2
        backend_kwargs.pop("fir_len", None)
        self.conv = FIRConvolution(**backend_kwargs)

    def forward(self, input_signals, fir):
        fir = normalize_impulse(torch.tanh(fir))
        if self.processor_channel == "midside":
            return ms_to_lr(self.conv(lr_to_ms(input_signals), fir))
        return self.conv(input_signals, fir)

    def parameter_size(self):
        return {"fir": (self.num_channels, self.fir_len)}


class PoleZeroFilter(nn.Module):
    """Biquad cascade parameterised by complex zeros and (tanh-radius-limited) poles (reference filter.py:171-239).
    Two upstream details are kept because they change the numbers: the denominator's z^-2 coefficient uses the
    *unlimited* pole radius (filter.py:224), and one coefficient set is shared by all channels."""

    def __init__(self, num_filters=1, **backend_kwargs):
        super().__init__()
        self.num_filters = num_filters
        self.biquad = IIRFilter(order=2, **backend_kwargs)

    def forward(self, input_signals, log_gain, poles, zeros):
        p, z = torch.view_as_complex(poles.contiguous()), torch.view_as_complex(zeros.contiguous())
        p_radius, z_radius = p.abs(), z.abs()
        p = p * torch.tanh(p_radius) / (p_radius + 1e-5)
        one = torch.ones_like(p_radius)
        Bs = torch.stack([one, -2 * z.real, z_radius.square()], -1)
        As = torch.stack([one, -2 * p.real, p_radius.square()], -1)
        y = self.biquad(input_signals, Bs.unsqueeze(1), As.unsqueeze(1))
        return torch.exp(log_gain).unsqueeze(-1) * y

    def parameter_size(self):
        return {"log_gain": 1, "poles": (self.num_filters, 2), "zeros": (self.num_filters, 2)}


class BiquadFilter(BufferIO, nn.Module):
    def __init__(self, num_filters=1, normalized=False, **backend_kwargs):
        super().__init__()
        self.num_filters = num_filters
        self.normalized = normalized
        self.biquad = IIRFilter(order=2, **backend_kwargs)

    def forward(self, input_signals, Bs, A1_pre, A2_pre, A0=None, _out=None):
        A0 = A0 if self.normalized else None
        if needs_grad(input_signals, Bs, A1_pre, A2_pre, A0):
            return self.biquad(input_signals, *diff.biquad_coefficients(Bs, A1_pre, A2_pre, A0), out=_out, final=True)
        Bs, As = ops.biquad_coeffs(Bs, A1_pre, A2_pre, A0)
        return self.biquad(input_signals, Bs.unsqueeze(1), As.unsqueeze(1), out=_out)

    def render_into(self, x4, out4, **params):
        return self.forward(x4, _out=out4, **params)

    def parameter_size(self):
        size = {"Bs": (self.num_filters, 3), "A1_pre": self.num_filters, "A2_pre": self.num_filters}
        if self.normalized:
            size["A0"] = self.num_filters
        return size


# ---- single-biquad parametric filters (reference filter.py:263-560) ----------------------------------
# Coefficient formulas are a few elementwise ops on (R, 1) tensors (torch, on the GPU); the filtering is
# the HIP frequency-sampling path of IIRFilter.
class BaseParametricFilter(nn.Module):
    def __init__(self, **backend_kwargs):
        super().__init__()
        self.biquad = IIRFilter(order=2, **backend_kwargs)

    def forward(self, input_signals, w0, q_inv):
        w0, q = self.filter_parameter_activations(w0, q_inv)
        cos_w0, alpha = self.compute_common_filter_parameters(w0, q)
        Bs, As = self.get_biquad_coefficients(cos_w0, alpha)
        return self.biquad(input_signals, Bs.unsqueeze(1), As.unsqueeze(1))

    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha):
        raise NotImplementedError

    @staticmethod
    def filter_parameter_activations(w0, q_inv):
        return math.pi * torch.sigmoid(w0), torch.exp(q_inv)

    @staticmethod
    def compute_common_filter_parameters(w0, q_inv):
        return torch.cos(w0), torch.sin(w0) * q_inv * 0.5

    def parameter_size(self):
        return {"w0": 1, "q_inv": 1}


def _den(cos_w0, alpha):
    return torch.stack([1 + alpha, -2 * cos_w0, 1 - alpha], -1)


class LowPassFilter(BaseParametricFilter):
    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha):  # filter.py:416-426 (sign as in the code, not the docstring)
        c = cos_w0 - 1
        return torch.stack([c / 2, c, c / 2], -1), _den(cos_w0, alpha)


class HighPassFilter(BaseParametricFilter):
    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha):
        c = 1 + cos_w0
        return torch.stack([c / 2, -c, c / 2], -1), _den(cos_w0, alpha)


class BandPassFilter(BaseParametricFilter):
    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha):
        return torch.stack([alpha, torch.zeros_like(alpha), -alpha], -1), _den(cos_w0, alpha)


class BandRejectFilter(BaseParametricFilter):
    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha):
        one = torch.ones_like(cos_w0)
        return torch.stack([one, -2 * cos_w0, one], -1), _den(cos_w0, alpha)


class AllPassFilter(BaseParametricFilter):
    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha):
        den = _den(cos_w0, alpha)
        return den.flip(-1), den


class BaseParametricEqualizerFilter(nn.Module):
    """Stack of `num_filters` equaliser sections of one kind (filter.py:563-617)."""

    def __init__(self, num_filters=1, **backend_kwargs):
        super().__init__()
        self.num_filters = num_filters
        self.biquad = IIRFilter(order=2, **backend_kwargs)

    def forward(self, input_signals, w0, q_inv, log_gain):
        w, qi, A = self.filter_parameter_activations(w0, q_inv, log_gain)
        cw, alpha = self.compute_common_filter_parameters(w, qi)
        Bs, As = self.get_biquad_coefficients(cw, alpha, A)
        return self.biquad(input_signals, Bs.unsqueeze(1), As.unsqueeze(1))

    # the reference's static helpers (filter.py:593-604 and the per-kind coefficient maps), usable on their own
    @staticmethod
    def filter_parameter_activations(w0, q_inv, log_gain):
        """Raw parameters -> (angular frequency in (0, pi), 1/Q > 0, linear gain A > 0)."""
        return math.pi * torch.sigmoid(w0), torch.exp(q_inv), torch.exp(log_gain)

    @staticmethod
    def compute_common_filter_parameters(w0, q_inv):
        """(cos w0, alpha = sin w0 / (2 Q)) of the RBJ cookbook."""
        return torch.cos(w0), torch.sin(w0) * q_inv * 0.5

    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha, A):
        raise NotImplementedError

    def parameter_size(self):
        return {"w0": self.num_filters, "q_inv": self.num_filters, "log_gain": self.num_filters}


def _shelf_coefficients(cw, alpha, A, sg):
    """RBJ shelving sections, un-normalised a0 (filter.py:687-705 low shelf, sg = +1; 736-754 high shelf, sg = -1)."""
    ap1, am1, s = A + 1, A - 1, 2 * A.sqrt() * alpha
    Bs = torch.stack([A * (ap1 - sg * am1 * cw + s), sg * 2 * A * (am1 - sg * ap1 * cw), A * (ap1 - sg * am1 * cw - s)], -1)
    As = torch.stack([ap1 + sg * am1 * cw + s, -sg * 2 * (am1 + sg * ap1 * cw), ap1 + sg * am1 * cw - s], -1)
    return Bs, As


class PeakingFilter(BaseParametricEqualizerFilter):
    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha, A):  # filter.py:645-656
        return (torch.stack([1 + alpha * A, -2 * cos_w0, 1 - alpha * A], -1),
                torch.stack([1 + alpha / A, -2 * cos_w0, 1 - alpha / A], -1))


class LowShelf(BaseParametricEqualizerFilter):
    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha, A):
        return _shelf_coefficients(cos_w0, alpha, A, 1.0)


class HighShelf(BaseParametricEqualizerFilter):
    @staticmethod
    def get_biquad_coefficients(cos_w0, alpha, A):
        return _shelf_coefficients(cos_w0, alpha, A, -1.0)


class StateVariableFilter(nn.Module):
    """SVF-parameterised biquads (filter.py:223-300)."""

    def __init__(self, num_filters=1, **backend_kwargs):
        super().__init__()
        self.num_filters = num_filters
        self.biquad = IIRFilter(order=2, **backend_kwargs)

    def forward(self, input_signals, twoR, G, c_hp, c_bp, c_lp):
        G = torch.tan(math.pi / 2 * torch.sigmoid(G))
        twoR = F.softplus(twoR) / math.log(2) + 1e-2
        Bs, As = self.get_biquad_coefficients(twoR, G, c_hp, c_bp, c_lp)
        return self.biquad(input_signals, Bs.unsqueeze(1), As.unsqueeze(1))

    @staticmethod
    def get_biquad_coefficients(twoR, G, c_hp, c_bp, c_lp):
        G2 = G.square()
        Bs = torch.stack([c_hp + c_bp * G + c_lp * G2, -c_hp * 2 + c_lp * 2 * G2, c_hp - c_bp * G + c_lp * G2], -1)
        As = torch.stack([1 + G2 + twoR * G, 2 * G2 - 2, 1 + G2 - twoR * G], -1)
        return Bs, As

    def parameter_size(self):
        return {k: self.num_filters for k in ("twoR", "G", "c_hp", "c_bp", "c_lp")}
