"""Reverbs (mirrors grafx.processors.reverb.STFTMaskedNoiseReverb — reference reverb.py:15-228)."""
import numpy as np
import torch
import torch.nn as nn

import torch.nn.functional as F

from .. import autograd as diff
from .. import ops
from ..autograd import needs_grad
from .core._buffer_io import BufferIO
from .core.utils import normalize_impulse
from .core.convolution import convolve_taps
from .core.midside import lr_to_ms, ms_to_lr


class STFTMaskedNoiseReverb(BufferIO, nn.Module):
    def __init__(self, ir_len=60000, processor_channel="pseudo_midside", n_fft=384, hop_length=192,
                 fixed_noise=True, gain_envelope=False, flashfftconv=True, max_input_len=2**17):
        super().__init__()
        self.ir_len, self.n_fft, self.hop_length = ir_len, n_fft, hop_length
        self.num_frames = 1 + (ir_len // hop_length)
        self.num_bins = 1 + n_fft // 2
        self.register_buffer("window", torch.hann_window(n_fft))
        self.register_buffer("arange", torch.arange(self.num_frames).view(1, 1, 1, -1))
        self.fixed_noise = fixed_noise
        if not fixed_noise:
            raise NotImplementedError("fixed_noise=False (fresh noise every forward) is not part of this release")
        self.get_fixed_noise()
        self.gain_envelope = gain_envelope
        self.processor_channel = processor_channel
        if processor_channel not in ("mono", "stereo", "midside", "pseudo_midside"):
            raise ValueError(f"Invalid processor_channel: {processor_channel}")
        self._basis = {}

    def get_fixed_noise(self):
        """One-time constant, built exactly like upstream (reverb.py:101-114): RandomState(0) uniform
        noise in [-1,1) -> float32 -> centred STFT.  Init-time only; the buffer then lives on the GPU."""
        noise = np.random.RandomState(0).uniform(size=(2, self.ir_len)) * 2 - 1
        noise = torch.tensor(noise).float()
        spec = torch.stft(noise, n_fft=self.n_fft, hop_length=self.hop_length, window=self.window.cpu(),
                          return_complex=True)
        self.register_buffer("noise_stft", spec[None].contiguous())

    def _istft_basis(self, device):
        key = (device.type, device.index)
        if key not in self._basis:
            self._basis[key] = ops.istft_basis(self.window)
        return self._basis[key]

    def _ir_and_gain(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude, ms_lr):
        genv = gain_env_log_magnitude if self.gain_envelope else None
        return ops.stft_reverb_ir(self.noise_stft, init_log_magnitude, delta_log_magnitude, genv, self.window,
                                  self._istft_basis(init_log_magnitude.device), self.ir_len, self.hop_length, ms_lr)

    def compute_ir(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None):
        """Un-normalised mid/side impulse responses (R,2,ir_len) (reverb.py:161-187)."""
        if needs_grad(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude):
            return self._compute_ir_differentiable(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude)
        return self._ir_and_gain(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude, False)[0]

    def _compute_ir_differentiable(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None):
        """reverb.py:161-200 with torch ops (mask, istft) for the training path."""
        logmag = init_log_magnitude[..., None] - F.softplus(delta_log_magnitude)[..., None] * self.arange
        if self.gain_envelope:
            logmag = logmag + gain_env_log_magnitude[:, :, None, :]
        spec = self.noise_stft * torch.exp(logmag / 8)
        R = spec.shape[0]
        ir = torch.istft(spec.reshape(R * 2, self.num_bins, self.num_frames), n_fft=self.n_fft,
                         hop_length=self.hop_length, window=self.window, length=self.ir_len)
        return ir.view(R, 2, self.ir_len)

    def render_into(self, x4, out4, **params):
        if self.processor_channel == "midside":
            return super().render_into(x4, out4, **params)
        return self.forward(x4, _out=out4, **params)

    def forward(self, input_signals, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None, _out=None):
        pseudo = self.processor_channel == "pseudo_midside"
        if needs_grad(input_signals, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude):
            x = input_signals.reshape(-1, *input_signals.shape[-2:])
            ir = self._compute_ir_differentiable(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude)
            if pseudo:
                y = diff.convolve(x, normalize_impulse(ms_to_lr(ir)), "causal")
            elif self.processor_channel == "midside":
                y = ms_to_lr(diff.convolve(lr_to_ms(x), normalize_impulse(ir), "causal"))
            else:
                y = diff.convolve(x, normalize_impulse(ir), "causal")
            if _out is None:
                return y
            _out.copy_(y.view(_out.shape))
            return _out
        ir, gain = self._ir_and_gain(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude, pseudo)
        R = ir.shape[0]
        # normalize_impulse (core/utils.py:14-18) is folded into the tap -> spectrum step
        Hs = ops.fir_spectrum(ir.view(R * 2, self.ir_len), gain=gain, gain_div=2)
        if self.processor_channel == "midside":  # reverb.py:219-223
            return ms_to_lr(convolve_taps(lr_to_ms(input_signals), Hs, self.ir_len, 2, "causal"))
        return convolve_taps(input_signals, Hs, self.ir_len, 2, "causal", out=_out)

    def parameter_size(self):
        size = {"init_log_magnitude": (2, self.num_bins), "delta_log_magnitude": (2, self.num_bins)}
        if self.gain_envelope:
            size["gain_env_log_magnitude"] = (2, self.num_frames)
        return size
