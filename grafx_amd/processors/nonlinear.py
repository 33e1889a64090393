"""Memoryless distortions (mirrors grafx.processors.nonlinear — reference nonlinear.py:6-309).

Inference runs one streaming HIP pass per processor (`gfx_waveshaper_f32`, plus `gfx_row_mean_f32` when
`remove_dc` is set); when a gradient is requested the same formulas are evaluated with torch ops so that
autograd differentiates them (elementwise, no custom backward needed)."""
import torch
import torch.nn as nn

from .. import ops
from ..autograd import needs_grad
from .core._buffer_io import BufferIO


def _center(x, remove_dc):
    return x - x.mean(-1, keepdim=True) if remove_dc else x


class _Waveshaper(BufferIO, nn.Module):
    def render_into(self, x4, out4, **params):
        if needs_grad(x4, *params.values()):
            return super().render_into(x4, out4, **params)
        return self.forward(x4, _out=out4, **params)


class TanhDistortion(_Waveshaper):
    def __init__(self, pre_post_gain=True, inverse_post_gain=True, remove_dc=False, use_bias=False):
        super().__init__()
        self.pre_post_gain = pre_post_gain
        self.inverse_post_gain = inverse_post_gain
        self.remove_dc = remove_dc
        self.use_bias = use_bias

    def forward(self, input_signals, log_pre_gain=None, log_post_gain=None, bias=None, _out=None):
        pre = log_pre_gain if self.pre_post_gain else None
        inverse = self.pre_post_gain and self.inverse_post_gain
        post = log_post_gain if (self.pre_post_gain and not self.inverse_post_gain) else None
        b = bias if self.use_bias else None
        if needs_grad(input_signals, pre, post, b):
            u = _center(input_signals, self.remove_dc)
            g = torch.exp(pre).unsqueeze(-1) if pre is not None else None
            u = u * g if g is not None else u
            y = torch.tanh(u + b.unsqueeze(-1)) - torch.tanh(b.unsqueeze(-1)) if b is not None else torch.tanh(u)
            if inverse:
                return y / g
            return y * torch.exp(post).unsqueeze(-1) if post is not None else y
        return ops.waveshaper(input_signals, ops.WS_TANH, pre, post, p0=b, inverse_post_gain=inverse,
                              remove_dc=self.remove_dc, out=_out)

    def parameter_size(self):
        size = {}
        if self.pre_post_gain:
            size["log_pre_gain"] = 1
            if not self.inverse_post_gain:
                size["log_post_gain"] = 1
        if self.use_bias:
            size["bias"] = 1
        return size


class PiecewiseTanhDistortion(_Waveshaper):
    def __init__(self, pre_post_gain=True, inverse_post_gain=True, remove_dc=False):
        super().__init__()
        self.pre_post_gain = pre_post_gain
        self.inverse_post_gain = inverse_post_gain
        self.remove_dc = remove_dc

    def forward(self, input_signals, log_hardness, z_threshold, log_pre_gain=None, log_post_gain=None, _out=None):
        pre = log_pre_gain if self.pre_post_gain else None
        inverse = self.pre_post_gain and self.inverse_post_gain
        post = log_post_gain if (self.pre_post_gain and not self.inverse_post_gain) else None
        if needs_grad(input_signals, log_hardness, z_threshold, pre, post):
            u = _center(input_signals, self.remove_dc)
            g = torch.exp(pre).unsqueeze(-1) if pre is not None else None
            u = u * g if g is not None else u
            y = self.apply_distortion(u, torch.exp(log_hardness), torch.sigmoid(z_threshold))
            if inverse:
                return y / g
            return y * torch.exp(post).unsqueeze(-1) if post is not None else y
        return ops.waveshaper(input_signals, ops.WS_PIECEWISE, pre, post, p0=log_hardness, p1=z_threshold,
                              inverse_post_gain=inverse, remove_dc=self.remove_dc, out=_out)

    @staticmethod
    def apply_distortion(u, hardness, threshold):
        # the (kn, kp) / (gp, gn) ordering follows upstream (nonlinear.py:163-164)
        kn, kp = threshold[..., None, 0:1], threshold[..., None, 1:2]
        gp, gn = hardness[..., None, 0:1], hardness[..., None, 1:2]
        bp, bn = torch.tanh(kp), -torch.tanh(kn)
        above, below = u > kp, u < -kn
        hi = (1 - bp) / gp * torch.tanh(gp * (u - kp)) + bp
        lo = (1 + bn) / gn * torch.tanh(gn * (u + kn)) + bn
        return torch.where(above, hi, torch.where(below, lo, torch.tanh(u)))

    def parameter_size(self):
        size = {"log_hardness": 2, "z_threshold": 2}
        if self.pre_post_gain:
            size["log_pre_gain"] = 1
            if not self.inverse_post_gain:
                size["log_post_gain"] = 1
        return size


class _PolynomialDistortion(_Waveshaper):
    mode = None

    def __init__(self, max_order=10, pre_gain=True, remove_dc=False, use_tanh=False):
        super().__init__()
        if not 1 <= max_order <= 32:
            raise ValueError("max_order must be in [1, 32]")
        self.pre_gain = pre_gain
        self.max_order = max_order
        self.remove_dc = remove_dc
        self.use_tanh = use_tanh

    def basis(self, u):
        raise NotImplementedError

    def forward(self, input_signals, basis_weights, log_pre_gain=None, _out=None):
        pre = log_pre_gain if self.pre_gain else None
        if needs_grad(input_signals, basis_weights, pre):
            u = _center(input_signals, self.remove_dc)
            u = u * torch.exp(pre).unsqueeze(-1) if pre is not None else u
            terms = self.basis(u)                                   # (K, R, C, L)
            terms = torch.tanh(terms) if self.use_tanh else terms
            return (terms * torch.tanh(basis_weights).T[:, :, None, None]).sum(0)
        return ops.waveshaper(input_signals, self.mode, pre, None, p0=basis_weights, use_tanh=self.use_tanh,
                              remove_dc=self.remove_dc, out=_out)

    def parameter_size(self):
        size = {"basis_weights": self.max_order}
        if self.pre_gain:
            size["log_pre_gain"] = 1
        return size


class PowerDistortion(_PolynomialDistortion):
    mode = ops.WS_POWER

    def basis(self, u):
        k = torch.arange(self.max_order, device=u.device)[:, None, None, None]
        return torch.pow(u.unsqueeze(0), k)


class ChebyshevDistortion(_PolynomialDistortion):
    mode = ops.WS_CHEBYSHEV

    @staticmethod
    def apply_distortion(input_signals, basis_weights, use_tanh=False):
        """sum_k basis_weights[:, k] * T_k(x) (optionally tanh(T_k)) on (R, C, L) signals, weights (R, K) taken as
        they are (reference nonlinear.py:385-403), as torch ops."""
        K = basis_weights.shape[-1]
        terms = [torch.ones_like(input_signals), input_signals]
        for _ in range(2, K):
            terms.append(2 * input_signals * terms[-1] - terms[-2])
        terms = torch.stack(terms[:K], 0)
        terms = torch.tanh(terms) if use_tanh else terms
        return (terms * basis_weights.T[:, :, None, None]).sum(0)

    def basis(self, u):
        terms = [torch.ones_like(u), u]
        for _ in range(2, self.max_order):
            terms.append(2 * u * terms[-1] - terms[-2])
        return torch.stack(terms[: self.max_order], 0)
