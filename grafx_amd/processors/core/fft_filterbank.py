"""Triangular filterbank between a linear FFT grid and a perceptual band grid (init-time matrix + one matmul).

Mirrors grafx.processors.core.fft_filterbank.TriangularFilterBank (reference core/fft_filterbank.py:9-161):
`filterbank` is (num_filters, F) for synthesis (band energies -> FFT bins), `filterbank_normalized`
(F, num_filters), columns summing to one, for analysis."""
import warnings

import torch
import torch.nn as nn

from .scale import from_scale, to_scale

SCALES = ("bark_traunmuller", "bark_schroeder", "bark_wang", "mel_htk", "mel_slaney", "linear", "log")


class TriangularFilterBank(nn.Module):
    def __init__(self, num_frequency_bins, num_filters=50, scale="bark_traunmuller", f_min=40, f_max=None, sr=44100,
                 low_half_triangle=True):
        super().__init__()
        if f_max is not None and f_max > sr // 2:
            warnings.warn(f"The value for `f_max` ({f_max}) is higher than the Nyquist frequency ({sr // 2}). "
                          "The value for `f_max` will be set to the Nyquist frequency.")
            f_max = sr // 2
        fb = self.compute_matrix(num_frequency_bins, num_filters, scale, f_min, f_max, sr, low_half_triangle)
        self.num_filters = num_filters
        self.register_buffer("filterbank", fb.T)
        self.register_buffer("filterbank_normalized", fb / fb.sum(0, keepdim=True))

    def forward(self, energy, mode="synthesis"):
        if mode == "analysis":
            return torch.matmul(energy, self.filterbank_normalized)
        if mode == "synthesis":
            return torch.matmul(energy, self.filterbank)
        return energy  # upstream falls through silently on other modes

    @staticmethod
    def compute_matrix(num_frequency_bins, num_filters, scale, f_min, f_max, sr, low_half_triangle):
        assert scale in SCALES
        n_tri = num_filters - 1 if low_half_triangle else num_filters
        bins_hz = torch.linspace(0, sr // 2, num_frequency_bins)
        knots = from_scale(torch.linspace(to_scale(f_min, scale), to_scale(f_max, scale), n_tri + 2), scale)
        fb = TriangularFilterBank._create_triangular_filterbank(bins_hz, knots)
        if low_half_triangle:  # one extra band that collects whatever the triangles leave below the first knot
            fb = torch.cat([(1 - fb.sum(-1))[:, None], fb], -1)
        if bool((fb.max(dim=0).values == 0.0).any()):
            warnings.warn(f"At least one bark filterbank has all zero values. The value for `n_bins` ({n_tri}) may be "
                          f"set too high. Or, the value for `num_frequency_bins` ({num_frequency_bins}) may be set too low.")
        return fb

    @staticmethod
    def _create_triangular_filterbank(all_freqs, f_pts):
        widths = f_pts[1:] - f_pts[:-1]
        dist = f_pts[None, :] - all_freqs[:, None]            # (F, knots)
        falling = -dist[:, :-2] / widths[:-1]
        rising = dist[:, 2:] / widths[1:]
        return torch.clamp(torch.minimum(falling, rising), min=0.0)
