"""Envelope smoothers (mirrors grafx.processors.core.envelope — reference core/envelope.py:10-101)."""
import torch
import torch.nn as nn

from ... import autograd as diff
from ... import ops
from ...autograd import needs_grad
from .convolution import odd_length_alias, reference_aliases, resolve_flashfftconv


class TruncatedOnePoleIIRFilter(nn.Module):
    """y = relu(u * h), h[n] = (1-a) a^n for n < iir_len, a = min(sigmoid(z), 1-1e-5).

    Runs as the exact recursive form of that FIR (prefix scan) instead of an FFT convolution;
    when the reference's convolve() would alias (odd L + iir_len - 1) the full-length result is
    produced first and aliased the same way."""

    def __init__(self, iir_len=16384, flashfftconv=True, max_input_len=2**17):
        super().__init__()
        self.iir_len = iir_len
        # upstream: FIRConvolution(mode="causal", **backend_kwargs) (envelope.py:32) -> warning + native convolve()
        self.flashfftconv = resolve_flashfftconv(flashfftconv)

    def forward(self, input_signals, z_alpha):
        if needs_grad(input_signals, z_alpha):
            return diff.truncated_one_pole(input_signals, z_alpha, self.iir_len, exact=self.flashfftconv)
        L = input_signals.shape[-1]
        if not reference_aliases(L, self.iir_len, self.flashfftconv):
            return ops.onepole(input_signals, z_alpha, self.iir_len, Lout=L, relu=True)
        full = ops.onepole(input_signals, z_alpha, self.iir_len, Lout=L + self.iir_len - 1, relu=False)
        P = L + self.iir_len - 1
        if full.ndim == 2 and ops.odd_alias_supported(P) and ops._alias_pairs(P, full.shape[0]):
            return ops.odd_alias(full, 0, L, precise=True, relu=True)      # (the clamp rides on the aliasing's last pass)
        return torch.relu(odd_length_alias(full, 0, L, precise=True)).contiguous()

    def forward_energy(self, signal, z_alpha):
        """forward(mean_c signal^2, z_alpha) for a signal (R, C, L) or a strided (B, n, C, L) view, inference only: the
        compressors' envelope when this smoother's convolve() aliases (upstream's default tap counts, dynamics.py:390 +
        envelope.py:34-49) in three passes instead of six -- the energy is formed inside the scan
        (gfx_onepole_energy_f32), which also leaves the rows' maxima the aliasing scales its pairs by, and the relu rides
        on the aliasing's last pass (gfx_odd_alias_pair_precise_max_f32)."""
        L = signal.shape[-1]
        if not reference_aliases(L, self.iir_len, self.flashfftconv):
            return ops.onepole_energy(signal, z_alpha, self.iir_len, Lout=L, relu=True)
        P = L + self.iir_len - 1
        rm = {}
        full = ops.onepole_energy(signal, z_alpha, self.iir_len, Lout=P, relu=False, rowmax=rm)
        if not ops.odd_alias_supported(P) or not ops._alias_pairs(P, full.shape[0]):
            return torch.relu(odd_length_alias(full, 0, L, precise=True)).contiguous()
        return ops.odd_alias(full, 0, L, precise=True, rowmax=rm["words"], relu=True)

    def compute_impulse(self, z_alpha):
        if needs_grad(z_alpha):
            return diff.one_pole_fir(z_alpha, self.iir_len)
        return ops.onepole_fir(z_alpha, self.iir_len)


class Ballistics(nn.Module):
    """Attack/release one-pole recursion (torchcomp.compressor_core semantics as recalled)."""

    def forward(self, input_signals, z_alpha):
        if needs_grad(input_signals, z_alpha):
            return diff.BallisticsFn.apply(input_signals, z_alpha)
        return ops.ballistics(input_signals, z_alpha)
