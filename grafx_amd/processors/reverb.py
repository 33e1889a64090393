"""Reverbs (mirrors grafx.processors.reverb.STFTMaskedNoiseReverb — reference reverb.py:15-228)."""
import numpy as np
import torch
import torch.nn as nn

import torch.nn.functional as F

from .. import autograd as diff
from .. import ops
from ..autograd import needs_grad
from .core._buffer_io import BufferIO, Prepared, expand_shared, shared_reps
from .core.utils import normalize_impulse
from .core.convolution import convolve_taps, resolve_flashfftconv
from .core.midside import lr_to_ms, ms_to_lr


class STFTMaskedNoiseReverb(BufferIO, nn.Module):
    def __init__(self, ir_len=60000, processor_channel="pseudo_midside", n_fft=384, hop_length=192,
                 fixed_noise=True, gain_envelope=False, flashfftconv=True, max_input_len=2**17):
        super().__init__()
        self.ir_len, self.n_fft, self.hop_length = ir_len, n_fft, hop_length
        # upstream builds a FIRConvolution here (reverb.py:86-90), which without FlashFFTConv warns and goes native
        self.flashfftconv = resolve_flashfftconv(flashfftconv)
        self.num_frames = 1 + (ir_len // hop_length)
        self.num_bins = 1 + n_fft // 2
        self.register_buffer("window", torch.hann_window(n_fft))
        self.register_buffer("arange", torch.arange(self.num_frames).view(1, 1, 1, -1))
        self.fixed_noise = fixed_noise
        if fixed_noise:
            self.get_fixed_noise()
        self.gain_envelope = gain_envelope
        self.processor_channel = processor_channel
        if processor_channel not in ("mono", "stereo", "midside", "pseudo_midside"):
            raise ValueError(f"Invalid processor_channel: {processor_channel}")
        self._basis = {}
        self._envelope = {}  # overlap-added squared window per (device, frames), for _istft

    def get_fixed_noise(self):
        """One-time constant, built exactly like upstream (reverb.py:101-114): RandomState(0) uniform
        noise in [-1,1) -> float32 -> centred STFT.  Init-time only; the buffer then lives on the GPU."""
        noise = np.random.RandomState(0).uniform(size=(2, self.ir_len)) * 2 - 1
        noise = torch.tensor(noise).float()
        spec = torch.stft(noise, n_fft=self.n_fft, hop_length=self.hop_length, window=self.window.cpu(),
                          return_complex=True)
        self.register_buffer("noise_stft", spec[None].contiguous())

    def sample_noise(self, num_noises, device):
        """Fresh uniform noise per row and its STFT (reverb.py:116-128), drawn with the device generator."""
        noise = torch.rand(num_noises * 2, self.ir_len, device=device) * 2 - 1
        # (round 6: the frames on the direct-sum kernel gfx_stft_f32 -- this was the last FFT-library call of the reverb)
        spec = ops.stft(noise, self.window, self.hop_length)
        return spec.view(num_noises, 2, self.num_bins, self.num_frames)

    def _istft_basis(self, device):
        key = (device.type, device.index)
        if key not in self._basis:
            self._basis[key] = ops.istft_basis(self.window)
        return self._basis[key]

    def _ir_and_gain(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude, ms_lr):
        genv = gain_env_log_magnitude if self.gain_envelope else None
        # fixed_noise=False (reverb.py:63, 80-82, 165): fresh noise for every row, its STFT on gfx_stft_f32, mask + inverse
        # STFT + overlap-add + normalisation on the same native kernels as the fixed noise
        noise = self.noise_stft if self.fixed_noise else self.sample_noise(init_log_magnitude.shape[0],
                                                                          init_log_magnitude.device)
        return ops.stft_reverb_ir(noise, init_log_magnitude, delta_log_magnitude, genv, self.window,
                                  self._istft_basis(init_log_magnitude.device), self.ir_len, self.hop_length, ms_lr)

    def compute_ir(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None):
        """Un-normalised mid/side impulse responses (R,2,ir_len) (reverb.py:161-187)."""
        if needs_grad(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude):
            return self._compute_ir_differentiable(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude)
        return self._ir_and_gain(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude, False)[0]

    def compute_stft_mask(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None):
        """(R, 2, bins, frames) magnitude mask exp((H0 - softplus(Hd) m [+ G_m]) / 8) (reverb.py:189-200), torch ops."""
        logmag = init_log_magnitude[..., None] - F.softplus(delta_log_magnitude)[..., None] * self.arange
        if self.gain_envelope:
            logmag = logmag + gain_env_log_magnitude[:, :, None, :]
        return torch.exp(logmag / 8)

    def _compute_ir_differentiable(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None):
        """reverb.py:161-200 with torch ops (mask, istft) for the training path."""
        logmag = init_log_magnitude[..., None] - F.softplus(delta_log_magnitude)[..., None] * self.arange
        if self.gain_envelope:
            logmag = logmag + gain_env_log_magnitude[:, :, None, :]
        noise_stft = self.noise_stft if self.fixed_noise else self.sample_noise(logmag.shape[0], logmag.device)
        spec = noise_stft * torch.exp(logmag / 8)
        R = spec.shape[0]
        return self._istft(spec.reshape(R * 2, self.num_bins, self.num_frames)).view(R, 2, self.ir_len)

    def _istft(self, spec):
        """torch.istft(center=True, length=ir_len) spelled out (windowed inverse frames, overlap-add, division by the
        overlap-added squared window, trim): same arithmetic, but without istft's host-side envelope check, so that a
        training step can be captured into a HIP graph."""
        n_fft, hop, T = self.n_fft, self.hop_length, spec.shape[-1]
        total = n_fft + hop * (T - 1)
        # the frames' inverse real DFT on the direct-sum kernels (autograd.IrdftFn; n_fft <= 8192), not the FFT library
        from .. import autograd as diff

        frames = diff.irfft_small(spec.transpose(-1, -2), n_fft) * self.window
        y = self._overlap_add(frames)
        key = (spec.device.type, spec.device.index, T)
        if key not in self._envelope:
            self._envelope[key] = self._overlap_add((self.window * self.window).expand(1, T, n_fft))[0]
        a = n_fft // 2
        return y[:, a : a + self.ir_len] / self._envelope[key][a : a + self.ir_len]

    def _overlap_add(self, frames):
        """(R, T, n_fft) frames -> (R, n_fft + hop (T - 1)) with n_fft a multiple of hop: the q-th hop-sized piece of
        frame m lands in output block m + q, i.e. n_fft / hop shifted slice additions over the whole batch (F.fold
        would do the same, but its backward runs one im2col kernel per row)."""
        R, T, n_fft = frames.shape
        hop = self.hop_length
        pieces = n_fft // hop
        assert pieces * hop == n_fft
        out = frames.new_zeros(R, T + pieces - 1, hop)
        for q in range(pieces):
            out[:, q : q + T] = out[:, q : q + T] + frames[:, :, q * hop : (q + 1) * hop]
        return out.reshape(R, (T + pieces - 1) * hop)

    accepts_shared_params = True  # render_into(..., _shared_rows=n): parameters hold n rows shared by the batch
    accepts_strided_rows = True   # forward() also takes a strided (B, n, C, L) view and then returns (B, n, C, L)

    def render_into(self, x4, out4, _shared_rows=None, **params):
        if self.processor_channel == "midside":
            if _shared_rows is not None:
                params = {k: expand_shared(v, shared_reps(x4, _shared_rows)) for k, v in params.items()}
            return super().render_into(x4, out4, **params)
        return self.forward(x4, _out=out4, _shared_rows=_shared_rows, **params)

    def prepare(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None, _shared_rows=None):
        """Impulse-response synthesis + tile spectra (everything before the convolution), for the render's side stream."""
        if (not self.fixed_noise or self.processor_channel == "midside"
                or needs_grad(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude)):
            return None
        ir, gain = self._ir_and_gain(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude,
                                     self.processor_channel == "pseudo_midside")
        return Prepared(ops.fir_spectrum(ir.view(ir.shape[0] * 2, self.ir_len), gain=gain, gain_div=2))

    def forward(self, input_signals, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None, _out=None,
                _shared_rows=None, _prepared=None):
        if _prepared is not None:
            return convolve_taps(input_signals, _prepared.tensors[0], self.ir_len, 2, "causal", out=_out,
                                 exact=self.flashfftconv, h_rows=_shared_rows)
        pseudo = self.processor_channel == "pseudo_midside"
        if _shared_rows is not None and not self.fixed_noise:  # a noise per row: one parameter row per signal row
            reps = shared_reps(input_signals, _shared_rows)
            init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude = (
                expand_shared(t, reps) for t in (init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude))
            _shared_rows = None
        if needs_grad(input_signals, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude):
            # training: mask + istft as torch ops on the GPU (R x 193 x 313), the convolution below is native either way
            # the impulse responses are synthesised once per parameter row (per node when the batch shares them) and
            # the native convolution lets every batch row read them; a strided (B, n, C, L) view is read in place
            ir = self._compute_ir_differentiable(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude)
            if pseudo:
                y = diff.convolve(input_signals, normalize_impulse(ms_to_lr(ir)), "causal", exact=self.flashfftconv,
                                  final=True)
            elif self.processor_channel == "midside":
                x = input_signals.reshape(-1, *input_signals.shape[-2:])
                h = normalize_impulse(ir)
                if h.shape[0] != x.shape[0]:
                    h = expand_shared(h, x.shape[0] // h.shape[0])
                y = ms_to_lr(diff.convolve(lr_to_ms(x), h, "causal", exact=self.flashfftconv, final=True))
            else:
                y = diff.convolve(input_signals, normalize_impulse(ir), "causal", exact=self.flashfftconv, final=True)
            if _out is None:
                return y
            _out.copy_(y.view(_out.shape))
            return _out
        ir, gain = self._ir_and_gain(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude, pseudo)
        R = ir.shape[0]
        # normalize_impulse (core/utils.py:14-18) is folded into the tap -> spectrum step
        Hs = ops.fir_spectrum(ir.view(R * 2, self.ir_len), gain=gain, gain_div=2)
        if self.processor_channel == "midside":  # reverb.py:219-223
            return ms_to_lr(convolve_taps(lr_to_ms(input_signals), Hs, self.ir_len, 2, "causal", exact=self.flashfftconv))
        return convolve_taps(input_signals, Hs, self.ir_len, 2, "causal", out=_out, exact=self.flashfftconv,
                             h_rows=_shared_rows)

    def parameter_size(self):
        size = {"init_log_magnitude": (2, self.num_bins), "delta_log_magnitude": (2, self.num_bins)}
        if self.gain_envelope:
            size["gain_env_log_magnitude"] = (2, self.num_frames)
        return size


class FilteredNoiseShapingReverb(nn.Module):
    """Band-wise exponentially decaying filtered noise (mirrors reference reverb.py:231-401).

    Init: uniform noise split into `num_bands` bands by a Linkwitz-Riley crossover (host side, scipy).
    Forward: `gfx_noise_shaping_ir_f32` sums the K decaying bands into the impulse response (the reference
    materialises a (B, C, K, ir_len) envelope tensor for this), the response is energy-normalised and the
    HIP overlap-save convolution applies it (uniformly partitioned for the default 60 000 taps)."""

    def __init__(self, ir_len=60000, num_bands=12, processor_channel="midside", f_min=31.5, f_max=15000, scale="log",
                 sr=30000, zerophase=True, order=2, noise_randomness="pseudo-random", use_fade_in=False,
                 min_decay_ms=50, max_decay_ms=2000, flashfftconv=True, max_input_len=2**17):
        super().__init__()
        from .core.convolution import FIRConvolution
        from .core.noise import get_filtered_noise

        if processor_channel not in ("midside", "stereo", "mono"):
            raise ValueError(f"Unknown channel type: {processor_channel}")
        if noise_randomness not in ("pseudo-random", "fixed"):
            if noise_randomness == "random":
                raise NotImplementedError('noise_randomness="random" is an unfinished option upstream (assert False)')
            raise ValueError(f"Invalid filtered_noise argument: {noise_randomness}")
        self.num_bands = num_bands
        self.processor_channel = processor_channel
        self.num_channels = 1 if processor_channel == "mono" else 2
        self.ir_len = ir_len
        self.noise_randomness = noise_randomness
        noise_len = ir_len if noise_randomness == "fixed" else 5 * ir_len
        noise = get_filtered_noise(noise_len, num_channels=self.num_channels, num_bands=num_bands, f_min=f_min,
                                   f_max=f_max, scale=scale, sr=sr, zerophase=zerophase, order=order)
        self.register_buffer("filtered_noise", noise.unsqueeze(0))      # (1, C, K, noise_len)
        self.conv = FIRConvolution(mode="causal", flashfftconv=flashfftconv, max_input_len=max_input_len)
        ln10_20 = np.log(10) / 20
        self.min_decay = -60 / (min_decay_ms * sr / 1000) * ln10_20      # log-amplitude slope per sample
        self.max_decay = -60 / (max_decay_ms * sr / 1000) * ln10_20
        self.use_fade_in = use_fade_in
        self.register_buffer("arange", torch.arange(ir_len)[None, None, None, :])

    def get_filtered_noise(self):
        if self.noise_randomness == "fixed":
            return self.filtered_noise
        start = int(torch.randint(0, self.filtered_noise.shape[-1] - self.ir_len, (1,)))
        return self.filtered_noise[..., start : start + self.ir_len]

    def compute_ir(self, log_decay, log_gain, log_fade_in=None, z_fade_in_gain=None):
        noise = self.get_filtered_noise()
        fade = (log_fade_in, z_fade_in_gain) if self.use_fade_in else (None, None)
        if needs_grad(log_decay, log_gain, *fade):
            d = torch.sigmoid(log_decay) * (self.max_decay - self.min_decay) + self.min_decay
            env = torch.exp(self.arange * d.unsqueeze(-1))
            if self.use_fade_in:
                f = torch.sigmoid(fade[0]) * (d - self.min_decay) + self.min_decay
                env = env - torch.exp(self.arange * f.unsqueeze(-1)) * torch.sigmoid(fade[1]).unsqueeze(-1)
            return (noise * (env * log_gain.unsqueeze(-1))).sum(2)
        return ops.noise_shaping_ir(noise[0], log_decay, log_gain, fade[0], fade[1], self.ir_len, self.min_decay,
                                    self.max_decay)

    def forward(self, input_signals, log_decay, log_gain, log_fade_in=None, z_fade_in_gain=None):
        ir = normalize_impulse(self.compute_ir(log_decay, log_gain, log_fade_in, z_fade_in_gain))
        if self.processor_channel == "midside":
            return ms_to_lr(self.conv(lr_to_ms(input_signals), ir))
        return self.conv(input_signals, ir)

    def parameter_size(self):
        shape = (self.num_channels, self.num_bands)
        size = {"log_decay": shape, "log_gain": shape}
        if self.use_fade_in:
            size["log_fade_in"] = shape
            size["z_fade_in_gain"] = shape
        return size
