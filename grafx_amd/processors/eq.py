"""Equalizers (mirrors grafx.processors.eq — reference eq.py:217-336 for ParametricEqualizer)."""
import torch
import torch.nn as nn

from .. import ops
from .core._buffer_io import BufferIO, Prepared, expand_shared, shared_reps
from .core.convolution import convolve, convolve_taps
from .core.fir import ZeroPhaseFilterBankFIR, ZeroPhaseFIR
from .core.geq import GraphicEqualizerBiquad
from .. import autograd as diff
from ..autograd import needs_grad
from .core.iir import IIRFilter
from .core.midside import lr_to_ms, ms_to_lr


class ParametricEqualizer(BufferIO, nn.Module):
    def __init__(self, num_filters=10, processor_channel="mono", use_shelving_filters=True, **backend_kwargs):
        super().__init__()
        self.num_filters = num_filters
        self.use_shelving_filters = use_shelving_filters
        self.biquad = IIRFilter(order=2, **backend_kwargs)
        self.processor_channel = processor_channel
        if processor_channel not in ("mono", "stereo", "midside"):
            raise ValueError(f"Invalid processor_channel: {self.processor_channel}")

    accepts_tee = True  # render_into(..., tee=view) also leaves a copy of the input in `view`
    accepts_shared_params = True  # render_into(..., _shared_rows=n): parameters hold n rows shared by the batch
    accepts_strided_rows = True   # forward() also takes a strided (B, n, C, L) view and then returns (B, n, C, L)

    def forward(self, input_signals, w0, q_inv, log_gain, _out=None, _tee=None, _shared_rows=None):
        self._check_bands()
        if needs_grad(input_signals, w0, q_inv, log_gain):
            Bs, As = diff.PeqCoeffsFn.apply(w0, q_inv, log_gain, self.use_shelving_filters)
        else:
            Bs, As = ops.peq_coeffs(w0, q_inv, log_gain, self.use_shelving_filters)
        if self.processor_channel == "midside":
            return ms_to_lr(self.biquad(lr_to_ms(input_signals), Bs, As, shared_rows=_shared_rows, final=True))
        return self.biquad(input_signals, Bs, As, out=_out, tee=_tee, shared_rows=_shared_rows, final=True)

    def get_biquad_coefficients_with_shelving_filters(self, cos_w0, alpha, A):
        """Band 0 a low shelf, band K-1 a high shelf, peaking filters in between (eq.py:300-314), from the common
        parameters of filter.BaseParametricEqualizerFilter's helpers; (..., K, 3) numerators and denominators."""
        from .filter import HighShelf, LowShelf, PeakingFilter

        self._check_bands()
        K = self.num_filters
        parts = [LowShelf.get_biquad_coefficients(cos_w0[..., :1], alpha[..., :1], A[..., :1]),
                 PeakingFilter.get_biquad_coefficients(cos_w0[..., 1 : K - 1], alpha[..., 1 : K - 1], A[..., 1 : K - 1]),
                 HighShelf.get_biquad_coefficients(cos_w0[..., K - 1 :], alpha[..., K - 1 :], A[..., K - 1 :])]
        return torch.cat([b for b, _ in parts], -2), torch.cat([a for _, a in parts], -2)

    def _check_bands(self):
        if self.use_shelving_filters and self.num_filters < 2:
            # upstream splits the bands [1, K-2, 1] (eq.py:254, 300-302): torch.split rejects the negative size
            raise RuntimeError(f"split expects non-negative sizes, got [1, {self.num_filters - 2}, 1]: "
                               "shelving filters need num_filters >= 2")

    def prepare(self, w0, q_inv, log_gain, _shared_rows=None):
        """The parameter-only part of render_into (coefficients -> sampled response -> taps -> tile spectra), so that
        the render can run it ahead of time on a side stream; None when this configuration has no such split."""
        self._check_bands()
        if self.processor_channel == "midside" or self.biquad.backend != "fsm" or needs_grad(w0, q_inv, log_gain):
            return None
        Bs, As = ops.peq_coeffs(w0, q_inv, log_gain, self.use_shelving_filters)
        return Prepared(ops.fir_spectrum(self.biquad._taps(Bs, As)), Cf=Bs.shape[1])

    def render_into(self, x4, out4, tee=None, _shared_rows=None, _prepared=None, **params):
        if _prepared is not None:
            return convolve_taps(x4, _prepared.tensors[0], self.biquad.fsm_fir_len, _prepared.Cf, "causal", out=out4,
                                 tee=tee, exact=self.biquad.flashfftconv, h_rows=_shared_rows)
        if self.processor_channel == "midside":
            if tee is not None:
                tee.copy_(x4)
            if _shared_rows is not None:
                params = {k: expand_shared(v, shared_reps(x4, _shared_rows)) for k, v in params.items()}
            return super().render_into(x4, out4, **params)
        return self.forward(x4, _out=out4, _tee=tee, _shared_rows=_shared_rows, **params)

    def parameter_size(self):
        n_channels = 1 if self.processor_channel == "mono" else 2
        size = (n_channels, self.num_filters)
        return {k: size for k in ["w0", "q_inv", "log_gain"]}


class ZeroPhaseFIREqualizer(nn.Module):
    """Single-channel zero-phase FIR equaliser (reference eq.py:25-79): log-magnitude -> windowed
    zero-phase FIR of 2*bins-1 taps (odd, so the reference's convolve is an exact linear convolution for
    even audio lengths) -> HIP overlap-save convolution in "zerophase" mode."""

    def __init__(self, num_magnitude_bins=1024):
        super().__init__()
        self.num_magnitude_bins = num_magnitude_bins
        self.fir = ZeroPhaseFIR(num_magnitude_bins)

    def forward(self, input_signals, log_magnitude):
        fir = self.fir(log_magnitude)[:, None, :]
        return convolve(input_signals, fir, mode="zerophase")

    def parameter_size(self):
        return {"log_magnitude": self.num_magnitude_bins}


class NewZeroPhaseFIREqualizer(nn.Module):
    """Zero-phase FIR equaliser with optional perceptual filterbank parameterisation (reference eq.py:80-214).
    Taps = 2*bins-1 (odd), so for even audio lengths the reference's convolve() is an exact linear convolution
    and the whole signal path is the HIP overlap-save kernel in "zerophase" mode."""

    def __init__(self, num_frequency_bins=1024, processor_channel="mono", use_filterbank=False, filterbank_kwargs={},
                 window="hann", window_kwargs={}, eps=1e-7, flashfftconv=False):
        super().__init__()
        if processor_channel not in ("mono", "stereo", "midside"):
            raise ValueError(f"Invalid processor_channel: {processor_channel}")
        self.num_frequency_bins = num_frequency_bins
        self.processor_channel = processor_channel
        self.use_filterbank = use_filterbank
        self.fir = ZeroPhaseFilterBankFIR(num_frequency_bins=num_frequency_bins, use_filterbank=use_filterbank,
                                          filterbank_kwargs=filterbank_kwargs, window=window,
                                          window_kwargs=window_kwargs, eps=eps)

    def forward(self, input_signals, log_magnitude):
        fir = self.fir(log_magnitude)
        if self.processor_channel == "midside":
            return ms_to_lr(convolve(lr_to_ms(input_signals), fir, mode="zerophase"))
        return convolve(input_signals, fir, mode="zerophase")

    def parameter_size(self):
        n_bins = self.fir.filterbank.num_filters if self.use_filterbank else self.num_frequency_bins
        return {"log_magnitude": (1 if self.processor_channel == "mono" else 2, n_bins)}


class GraphicEqualizer(nn.Module):
    """Cascade of fixed-frequency peaking biquads (24 Bark or 31 third-octave bands; reference eq.py:339-436):
    band design on the GPU (core/geq.py), then the native frequency-sampling kernels."""

    def __init__(self, processor_channel="mono", scale="bark", sr=44100, **backend_kwargs):
        super().__init__()
        if processor_channel not in ("mono", "stereo", "midside"):
            raise ValueError(f"Invalid processor_channel: {processor_channel}")
        self.geq = GraphicEqualizerBiquad(scale=scale, sr=sr)
        self.biquad = IIRFilter(**backend_kwargs)
        self.processor_channel = processor_channel

    # band counts above this are designed in double precision on the forward path (31 third-octave bands: the lowest are
    # 9 Hz wide, and float32 coefficients 1 +- beta keep four digits of that width -- the reference's own float32 result is
    # 1e-4 .. 4e-4 from a float64 evaluation of its formulas there; 24 Bark bands meet it at 1e-5 in float32)
    PRECISE_BANDS = 24

    def forward(self, input_signals, log_gains):
        precise = (self.geq.num_bands > self.PRECISE_BANDS and self.biquad.backend == "fsm"
                   and ops.iir_fsm_native(self.biquad.fsm_fir_len) and not needs_grad(input_signals, log_gains))
        Bs, As = self.geq(log_gains, precise=precise)
        if self.processor_channel == "midside":
            return ms_to_lr(self.biquad(lr_to_ms(input_signals), Bs, As))
        return self.biquad(input_signals, Bs, As)

    def parameter_size(self):
        return {"log_gains": (1 if self.processor_channel == "mono" else 2, self.geq.num_bands)}
