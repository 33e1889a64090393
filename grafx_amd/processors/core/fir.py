"""Zero-phase FIR design from a log-magnitude response (mirrors grafx.processors.core.fir —
reference core/fir.py:7-74).  A tiny front-end (R x 1024 bins): exp -> irfft(n = 2*bins-1) ->
roll -> window, left to torch on the GPU; the heavy part is the zero-phase convolution that follows."""
import torch
import torch.nn as nn


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
    ir = torch.fft.irfft(torch.exp(log_magnitude.reshape(-1, bins)), n=fir_len)
    ir = torch.roll(ir, shifts=fir_len // 2, dims=-1)
    if window is not None:
        ir = ir * window[None, :]
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
