"""Zero-phase FIR design from a log-magnitude response (mirrors grafx.processors.core.fir —
reference core/fir.py:7-74; filterbank variant 77-127).  A tiny front-end (R x 1024 bins): exp -> irfft(n = 2*bins-1) ->
roll -> window, left to torch on the GPU; the heavy part is the zero-phase convolution that follows."""
import torch
import torch.nn as nn

from ... import ops
from ...autograd import needs_grad
from .fft_filterbank import TriangularFilterBank


def _zerophase_ir(magnitude, fir_len, window):
    """irfft(magnitude, n=fir_len) rolled to the centre and windowed (core/fir.py:20-27).  On the GPU, up to 8192 taps:
    the direct-sum kernels (any length, no FFT library) -- gfx_irdft_f32 with the roll and the window fused for
    inference, and with gradients its differentiable form (autograd.IrdftFn: backward = gfx_rdft_f32) followed by the
    roll and the window as torch ops; beyond that size, or on other dtypes, the FFT library."""
    native = (magnitude.is_cuda and fir_len <= ops.IRDFT_MAX_N and magnitude.shape[-1] == fir_len // 2 + 1
              and magnitude.dtype == torch.float32)
    if native and not needs_grad(magnitude):
        return ops.irdft(magnitude, fir_len, roll=fir_len // 2, window=window)
    if native:
        from ... import autograd as diff

        ir = diff.irfft_small(torch.complex(magnitude, torch.zeros_like(magnitude)), fir_len)
    else:
        if magnitude.is_cuda:
            ops.fft_library_reached(f"zero-phase FIR design of {fir_len} taps")
        ir = torch.fft.irfft(magnitude, n=fir_len)
    ir = torch.roll(ir, shifts=fir_len // 2, dims=-1)
    return ir if window is None else ir * window[None, :]


def get_window(window_type, window_length, **kwargs):
    makers = {"hann": torch.hann_window, "hamming": torch.hamming_window, "blackman": torch.blackman_window,
              "bartlett": torch.bartlett_window, "kaiser": torch.kaiser_window}
    if window_type in ("rectangular", "none", "boxcar", None):
        return None
    if window_type not in makers:
        raise ValueError(f"Unsupported window type: {window_type}")
    return makers[window_type](window_length, **kwargs)


def log_magnitude_to_zerophase_fir(log_magnitude, fir_len, window=None):
    lead, bins = log_magnitude.shape[:-1], log_magnitude.shape[-1]
    ir = _zerophase_ir(torch.exp(log_magnitude.reshape(-1, bins)), fir_len, window)
    return ir.view(*lead, -1)


class ZeroPhaseFIR(nn.Module):
    def __init__(self, num_magnitude_bins=1024, window="hann", **window_kwargs):
        super().__init__()
        self.num_magnitude_bins = num_magnitude_bins
        self.fir_len = 2 * num_magnitude_bins - 1
        if isinstance(window, torch.Tensor):
            self.register_buffer("window", window)
        else:
            w = get_window(window, self.fir_len, **window_kwargs)
            if w is None:
                self.window = None
            else:
                self.register_buffer("window", w)

    def forward(self, log_magnitude):
        return log_magnitude_to_zerophase_fir(log_magnitude, fir_len=self.fir_len, window=self.window)


class ZeroPhaseFilterBankFIR(nn.Module):
    """ZeroPhaseFIR whose parameter may live on a perceptual band grid: band log-magnitudes -> energies ->
    triangular synthesis filterbank -> sqrt -> the same irfft / roll / window front-end."""

    def __init__(self, num_frequency_bins=1024, use_filterbank=False, filterbank_kwargs={}, window="hann",
                 window_kwargs={}, eps=1e-7):
        super().__init__()
        self.num_frequency_bins = num_frequency_bins
        self.fir_len = 2 * num_frequency_bins - 1
        self.eps = eps
        self.use_filterbank = use_filterbank
        if use_filterbank:
            self.filterbank = TriangularFilterBank(num_frequency_bins=num_frequency_bins, **filterbank_kwargs)
        w = window if isinstance(window, torch.Tensor) else get_window(window, self.fir_len, **window_kwargs)
        self.register_buffer("window", w)

    def forward(self, log_magnitude):
        lead, bins = log_magnitude.shape[:-1], log_magnitude.shape[-1]
        magnitude = torch.exp(log_magnitude.reshape(-1, bins))
        if self.use_filterbank:
            magnitude = torch.sqrt(self.filterbank(magnitude.square()) + self.eps)
        ir = _zerophase_ir(magnitude, self.fir_len, self.window)
        return ir.view(*lead, -1)
