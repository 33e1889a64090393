"""Differentiable surrogate of an integer delay (mirrors grafx.processors.core.delay — reference
core/delay.py:5-140): a complex "angular frequency" z (|z| squashed by tanh) is expanded to the geometric
sequence z^k, k <= N/2, whose inverse real FFT is a soft impulse; its arg-max is the hard delay, returned
with a straight-through gradient.  A (num_taps x N/2) front-end — torch ops on the GPU."""
import torch
import torch.nn as nn

from ... import autograd as diff
from ... import ops
from ...autograd import needs_grad


class NormalizedGradient(torch.autograd.Function):
    """Identity whose backward passes only the direction of the gradient (g / (|g| + 1e-7))."""

    @staticmethod
    def forward(ctx, z):
        return z

    @staticmethod
    def backward(ctx, g):
        return g / (1e-7 + g.abs())


class SurrogateDelay(nn.Module):
    def __init__(self, N, straight_through=True, radii_loss=True, normalize_gradients=True):
        super().__init__()
        self.straight_through = straight_through
        self.radii_loss = radii_loss
        self.normalize_gradients = normalize_gradients
        self.register_buffer("arange_sin", torch.arange(N // 2 + 1)[None, :])

    def forward(self, z):
        assert z.dtype == torch.cfloat
        shape = z.shape
        z = z.reshape(-1)
        loss = self.calculate_radii_loss(z)
        if self.normalize_gradients:
            z = NormalizedGradient.apply(z)
        radius = z.abs()
        z = z * torch.tanh(radius) / (radius + 1e-7)
        spec = (z[:, None] + 1e-7) ** self.arange_sin
        n = 2 * (spec.shape[-1] - 1)                                      # length 2*(N//2), as upstream
        if spec.is_cuda and spec.dtype == torch.complex64 and 1 <= n <= ops.IRDFT_MAX_N:
            # direct-sum kernels, no FFT library: gfx_irdft_f32, and with gradients its differentiable form (backward:
            # gfx_rdft_f32)
            soft = diff.irfft_small(spec.contiguous(), n) if needs_grad(z) else ops.irdft(spec, n)
        else:
            if spec.is_cuda:
                ops.fft_library_reached(f"surrogate delay's soft impulse of {n} samples")
            soft = torch.fft.irfft(spec)
        irs = self.apply_straight_through(soft) if self.straight_through else soft
        return irs.view(*shape, -1), loss

    def calculate_radii_loss(self, z):
        return (1 - torch.tanh(z.abs())).square().sum()

    def apply_straight_through(self, irs):
        return irs + (self.get_hard_irs(irs) - irs).detach()

    @torch.no_grad()
    def get_hard_irs(self, irs):
        hard = torch.zeros_like(irs)
        hard[torch.arange(len(hard), device=irs.device), irs.argmax(-1)] = 1
        return hard
