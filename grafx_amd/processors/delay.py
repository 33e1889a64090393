"""Multitap delay (mirrors grafx.processors.delay.MultitapDelay — reference delay.py:13-185).

The impulse response is assembled from `num_segments` segments of `segment_len` samples, each holding
`num_delay_per_segment` surrogate delays (optionally coloured by a short zero-phase FIR per tap), energy
normalised, and convolved with the input — the last step, the only heavy one, is the HIP overlap-save
convolution (uniformly partitioned for the default 60 000-tap response)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .core.convolution import FIRConvolution, convolve
from .core.delay import SurrogateDelay
from .core.fir import ZeroPhaseFIR
from .core.utils import normalize_impulse


class MultitapDelay(nn.Module):
    def __init__(self, segment_len=3000, num_segments=20, num_delay_per_segment=1, processor_channel="stereo",
                 zp_filter_per_tap=True, zp_filter_bins=20, flashfftconv=True, max_input_len=2**17, pre_delay=0,
                 **surrogate_delay_kwargs):
        super().__init__()
        if processor_channel not in ("mono", "stereo", "midside"):
            raise ValueError(f"Invalid processor_channel: {processor_channel}")
        self.segment_len = segment_len
        self.num_segments = num_segments
        self.num_delay_per_segment = num_delay_per_segment
        self.zp_filter_per_tap = zp_filter_per_tap
        self.zp_filter_bins = zp_filter_bins
        self.zp_filter_len = 2 * zp_filter_bins - 1
        if zp_filter_per_tap:
            self.zp_filter = ZeroPhaseFIR(zp_filter_bins)
        self.register_buffer("window", torch.hann_window(self.zp_filter_len).view(1, 1, -1))
        self.delay = SurrogateDelay(N=segment_len, **surrogate_delay_kwargs)
        self.conv = FIRConvolution(flashfftconv=flashfftconv, max_input_len=max_input_len)
        self.pre_delay = pre_delay
        self.processor_channel = processor_channel
        self.num_channels = 1 if processor_channel == "mono" else 2

    def forward(self, input_signals, delay_z, log_fir_magnitude=None):
        ir, radii_loss = self.get_ir(delay_z, log_fir_magnitude)
        y = self.conv(input_signals, ir)   # upstream convolves directly for every channel mode (delay.py:123)
        if self.pre_delay != 0:
            y = F.pad(y, (self.pre_delay, 0))[:, :, : -self.pre_delay]
        return y, radii_loss

    def get_ir(self, delay_z, log_fir_magnitude):
        irs, radii_loss = self.delay(torch.view_as_complex(delay_z.contiguous()))     # (B, taps, T)
        if self.zp_filter_per_tap:
            B, taps, T = irs.shape
            colour = self.zp_filter(log_fir_magnitude)                                # (B, taps, 2*bins-1)
            irs = convolve(irs.reshape(B * taps, 1, T), colour.reshape(B * taps, 1, -1), mode="zerophase").view(B, taps, T)
        B, _, T = irs.shape
        c, m, p = self.num_channels, self.num_segments, self.num_delay_per_segment
        irs = irs.view(B, c, m, p, T).sum(-2).reshape(B, c, m * T)
        return normalize_impulse(irs), {"radii_reg": radii_loss}

    def parameter_size(self):
        num_delay = self.num_segments * self.num_delay_per_segment * self.num_channels
        size = {"delay_z": (num_delay, 2)}
        if self.zp_filter_per_tap:
            size["log_fir_magnitude"] = (num_delay, self.zp_filter_bins)
        return size
