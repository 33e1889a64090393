"""Dynamic range processors (mirrors grafx.processors.dynamics — reference dynamics.py:213-721)."""
import torch
import torch.nn as nn

from .. import autograd as diff
from .. import ops
from ..autograd import needs_grad
from .core._buffer_io import BufferIO, expand_shared, shared_reps
from .core.convolution import reference_aliases, resolve_flashfftconv
from .core.envelope import Ballistics, TruncatedOnePoleIIRFilter


class _Dynamics(BufferIO, nn.Module):
    _gate = False

    def __init__(self, energy_smoother="iir", gain_smoother=None, gain_smooth_in_log=False, knee="quadratic",
                 iir_len=16384, flashfftconv=True, max_input_len=2**17):
        super().__init__()
        self.iir_len = iir_len
        self.flashfftconv = resolve_flashfftconv(flashfftconv, warn=False)   # the smoother modules below warn, as upstream's
        self.energy_smoother = energy_smoother
        if energy_smoother == "iir":
            self.energy_smoother_module = TruncatedOnePoleIIRFilter(iir_len=iir_len, flashfftconv=flashfftconv)
        elif energy_smoother == "ballistics":
            self.energy_smoother_module = Ballistics()
        elif energy_smoother is not None:
            raise ValueError(f"Unknown energy_smoother: {self.energy_smoother}")
        self.gain_smoother = gain_smoother
        if gain_smoother == "iir":
            self.gain_smoother_module = TruncatedOnePoleIIRFilter(iir_len=iir_len, flashfftconv=flashfftconv)
        elif gain_smoother == "ballistics":
            self.gain_smoother_module = Ballistics()
        elif gain_smoother is not None:
            raise ValueError(f"Unknown gain_smoother: {self.gain_smoother}")
        if knee not in ("hard", "quadratic", "exponential"):
            raise ValueError(f"Unknown knee: {knee}")
        self.knee = knee
        self.gain_smooth_in_log = gain_smooth_in_log

    accepts_shared_params = True  # render_into(..., _shared_rows=n): parameters hold n rows shared by the batch
    accepts_strided_rows = True   # forward() also takes a strided (B, n, C, L) view and then returns (B, n, C, L)
    accepts_aux = True            # _aux=(dict, key): the training render's per-stage store (see forward)
    accepts_mix = True            # _mix={"mask", "out"}: the routing sum that follows, see ops.dynamics_fused(mix=)

    def render_into(self, x4, out4, _shared_rows=None, _aux=None, _mix=None, **params):
        extra = {} if _mix is None else {"_mix": _mix}
        return self.forward(x4, _out=out4, _shared_rows=_shared_rows, _aux=_aux, **extra, **params)

    def forward(self, input_signals, log_threshold, log_ratio, log_knee=None, z_alpha_pre=None, z_alpha_post=None,
                _out=None, _shared_rows=None, _aux=None, _mix=None):
        """``_aux = (store, key)`` (render_grafx's training path): the tape-free forward render leaves the smoother's scan
        in ``store[key]`` and the stage-wise backward, which re-traces this call on the same rows, hands it to the
        autograd node, so that the backward does not have to scan the input again."""
        if _shared_rows is not None and (needs_grad(input_signals, log_threshold, log_ratio, log_knee, z_alpha_pre,
                                                    z_alpha_post) or self.gain_smoother is not None
                                         or (self.energy_smoother == "ballistics" and self.gain_smoother is not None)
                                         or (self.energy_smoother == "iir" and reference_aliases(
                                             input_signals.shape[-1], self.iir_len, self.flashfftconv))):
            reps = shared_reps(input_signals, _shared_rows)  # paths without row sharing: one parameter row per signal row
            log_threshold, log_ratio, log_knee, z_alpha_pre, z_alpha_post = (
                expand_shared(t, reps) for t in (log_threshold, log_ratio, log_knee, z_alpha_pre, z_alpha_post))
            _shared_rows = None
        if needs_grad(input_signals, log_threshold, log_ratio, log_knee, z_alpha_pre, z_alpha_post):
            kept = None if _aux is None else _aux[0].pop(_aux[1], None)
            y = self._forward_differentiable(input_signals, log_threshold, log_ratio, log_knee, z_alpha_pre, z_alpha_post,
                                             u1=kept)
            if _out is None:
                return y
            _out.copy_(y.view(_out.shape))
            return _out
        L = input_signals.shape[-1]
        if self.knee == "hard":
            log_knee = None
        fusable = self.gain_smoother is None and (
            self.energy_smoother is None or (self.energy_smoother == "iir" and not reference_aliases(L, self.iir_len, self.flashfftconv))
        )
        if fusable:  # one pass: energy -> one-pole -> log -> knee -> exp -> multiply
            u1 = None
            if _aux is not None and self.energy_smoother == "iir":
                rows = input_signals.shape[0] * (input_signals.shape[1] if input_signals.ndim == 4 else 1)
                u1 = _aux[0][_aux[1]] = torch.empty((rows, L), dtype=torch.float32, device=input_signals.device)
            return ops.dynamics_fused(input_signals, log_threshold, log_ratio, log_knee, z_alpha_pre,
                                      smoother=int(self.energy_smoother == "iir"), iir_len=self.iir_len,
                                      knee=self.knee, gate=self._gate, out=_out, param_rows=_shared_rows, u1_out=u1,
                                      mix=_mix)
        # unfused configurations (a gain smoother, ballistics, or an energy smoother whose convolve() aliases): the energy
        # and the gain kernels read / write the (B, n, C, L) buffer views in place through their row maps, the smoothers
        # work on the (rows, L) envelope in between -- no flattened copy of the input, no copy of the output
        if self.energy_smoother == "ballistics" and self.gain_smoother is None:
            # energy -> attack / release recursion -> gain computer -> gain stage in one pass (gfx_dynamics_ballistics_f32)
            return ops.dynamics_ballistics(input_signals, log_threshold, log_ratio, log_knee, z_alpha_pre, self.knee, self._gate,
                                           out=_out, param_rows=_shared_rows)
        if self.energy_smoother == "ballistics":   # energy and recursion in one pass over the signal (ballistics.hip)
            energy = ops.ballistics_energy(input_signals, z_alpha_pre)
        elif self.energy_smoother == "iir" and type(self.energy_smoother_module) is TruncatedOnePoleIIRFilter:
            # energy -> truncated one-pole (-> the reference's odd-length aliasing, in double) without an energy buffer
            energy = self.energy_smoother_module.forward_energy(input_signals, z_alpha_pre)
        else:
            energy = ops.energy(input_signals)
            if self.energy_smoother is not None:
                energy = self.energy_smoother_module(energy, z_alpha=z_alpha_pre)
        if self.gain_smoother is None:   # gain computer and gain stage in one pass over the envelope
            return ops.dyn_gain_apply(input_signals, energy, log_threshold, log_ratio, log_knee, self.knee, self._gate,
                                      out=_out)
        if self.gain_smooth_in_log:  # dynamics.py:411-414
            g = ops.dyn_gain(energy, log_threshold, log_ratio, log_knee, self.knee, self._gate, log_out=True)
            return ops.apply_gain(input_signals, self.gain_smoother_module(g, z_alpha=z_alpha_post), exp_gain=True, out=_out)
        gain = ops.dyn_gain(energy, log_threshold, log_ratio, log_knee, self.knee, self._gate, log_out=False)
        return ops.apply_gain(input_signals, self.gain_smoother_module(gain, z_alpha=z_alpha_post), out=_out)

    def reads_grad_source(self, length):
        """Whether the differentiable forward of a signal of this length is the ONE native node (DynamicsFn) that can take
        its output gradient from autograd.grad_source (the stage-wise backward of render_grafx asks before it leaves a
        routing sum's adjoint un-expanded)."""
        return self.gain_smoother is None and (self.energy_smoother is None or (
            self.energy_smoother == "iir" and not reference_aliases(length, self.iir_len, self.flashfftconv)))

    def _forward_differentiable(self, x, log_threshold, log_ratio, log_knee, z_alpha_pre, z_alpha_post, u1=None):
        """dynamics.py:390-405: one native autograd node when there is no gain smoother and the energy smoother is
        the (non-aliasing) one-pole or absent; otherwise torch ops around the native (differentiable) smoothers."""
        if self.gain_smoother is None and (self.energy_smoother is None or (
                self.energy_smoother == "iir" and not reference_aliases(x.shape[-1], self.iir_len, self.flashfftconv))):
            return diff.DynamicsFn.apply(x, log_threshold, log_ratio, log_knee, z_alpha_pre,
                                         self.energy_smoother == "iir", self.iir_len, self.knee, self._gate, u1)
        x = x.reshape(-1, *x.shape[-2:])
        energy = x.square().mean(-2)
        if self.energy_smoother is not None:
            energy = self.energy_smoother_module(energy, z_alpha=z_alpha_pre)
        g = diff.log_gain(torch.log(energy + 1e-5), log_threshold - 6, log_ratio, log_knee, self.knee, self._gate)
        if self.gain_smoother is None:
            gain = torch.exp(g)
        elif self.gain_smooth_in_log:
            gain = torch.exp(self.gain_smoother_module(g, z_alpha=z_alpha_post))
        else:
            gain = self.gain_smoother_module(torch.exp(g), z_alpha=z_alpha_post)
        return gain[:, None, :] * x

    # ---- the reference classes' helper methods (dynamics.py:411-489, 643-721), as torch ops -------------------
    @classmethod
    def gain_hard_knee(cls, log_energy, log_threshold, log_ratio, _=None):
        """Log-gain of the hard knee for a log-energy envelope (threshold already shifted by the caller)."""
        return diff.log_gain(log_energy, log_threshold, log_ratio, None, "hard", cls._gate)

    @classmethod
    def gain_quad_knee(cls, log_energy, log_threshold, log_ratio, log_knee):
        return diff.log_gain(log_energy, log_threshold, log_ratio, log_knee, "quadratic", cls._gate)

    @classmethod
    def gain_exp_knee(cls, log_energy, log_threshold, log_ratio, log_knee):
        return diff.log_gain(log_energy, log_threshold, log_ratio, log_knee, "exponential", cls._gate)

    def smooth_in_log(self, gain, **gain_smooth_params):
        """Smooth the log-gain, then exponentiate (gain_smooth_in_log=True)."""
        return torch.exp(self.gain_smoother_module(gain, **gain_smooth_params))

    def smooth_in_linear(self, gain, **gain_smooth_params):
        """Exponentiate the log-gain, then smooth it."""
        return self.gain_smoother_module(torch.exp(gain), **gain_smooth_params)

    def parameter_size(self):
        size = {"log_threshold": 1, "log_ratio": 1}
        if self.knee != "hard":
            size["log_knee"] = 1
        for key, kind in (("z_alpha_pre", self.energy_smoother), ("z_alpha_post", self.gain_smoother)):
            if kind == "iir":
                size[key] = 1
            elif kind == "ballistics":
                size[key] = 2
        return size


class Compressor(_Dynamics):
    """Feed-forward compressor (reference dynamics.py:213-489)."""

    _gate = False


class NoiseGate(_Dynamics):
    """Feed-forward noise gate (reference dynamics.py:492-721)."""

    _gate = True


class ApproxCompressor(Compressor):
    """Reference dynamics.py:8-120: Compressor(energy "iir", quadratic knee) under its older parameter
    names (``z_alpha`` instead of ``z_alpha_pre``)."""

    def __init__(self, iir_len=16384, flashfftconv=True, max_input_len=2**17):
        super().__init__(energy_smoother="iir", gain_smoother=None, knee="quadratic", iir_len=iir_len,
                         flashfftconv=flashfftconv, max_input_len=max_input_len)

    def forward(self, input_signals, z_alpha, log_threshold, log_ratio, log_knee=None, _out=None, _shared_rows=None,
                _aux=None, _mix=None):
        return super().forward(input_signals, log_threshold, log_ratio, log_knee, z_alpha_pre=z_alpha, _out=_out,
                               _shared_rows=_shared_rows, _aux=_aux, _mix=_mix)

    def parameter_size(self):
        return {"z_alpha": 1, "log_threshold": 1, "log_ratio": 1, "log_knee": 1}


class ApproxNoiseGate(nn.Module):
    """Reference dynamics.py:123-210.  Its knee differs from NoiseGate's (ratio = exp(r), full-width knee,
    +1e-3 in the denominator), so the gain curve is evaluated with elementwise torch ops on the smoothed
    log-energy produced by the HIP one-pole kernel."""

    def __init__(self, freq_sample_n=16384, flashfftconv=True, max_input_len=2**17):
        super().__init__()
        self.smoother = TruncatedOnePoleIIRFilter(iir_len=freq_sample_n, flashfftconv=flashfftconv,
                                                  max_input_len=max_input_len)

    def forward(self, input_signals, z_alpha, log_threshold, log_ratio, log_knee=None):
        if needs_grad(input_signals, z_alpha, log_threshold, log_ratio, log_knee):
            energy = input_signals.square().mean(-2)
        else:
            energy = ops.energy(input_signals)
        G = torch.log(self.smoother(energy, z_alpha) + 1e-5)
        return self.compute_gain(G, log_threshold - 6, log_ratio, log_knee) * input_signals

    def compute_gain(self, log_energy, log_threshold, log_ratio, log_knee):
        """(R, 1, L) linear gain of this gate's own knee (dynamics.py:185-203)."""
        G, T = log_energy, log_threshold
        ratio, W = torch.exp(log_ratio), torch.exp(log_knee)
        below, above = G < (T - W / 2), G > (T + W / 2)
        middle = (~below) * (~above)
        out = (ratio * (G - T) + T) * below + G * above + (G + (1 - ratio) * (G - T - W / 2) ** 2 / 2 / (W + 1e-3)) * middle
        return torch.exp(out - G)[:, None, :]

    def parameter_size(self):
        return {"z_alpha": 1, "log_threshold": 1, "log_ratio": 1, "log_knee": 1}


class FactorizedCompressor(nn.Module):
    """Reference dynamics.py:724-739: a constructor only upstream (no ``forward``, no ``parameter_size``) -- kept so
    that code importing the name keeps working; calling it fails with nn.Module's NotImplementedError, as upstream."""

    def __init__(self, gain_smooth_in_log=False, with_knee=True, frame_len=1024):
        super().__init__()
        self.energy_smoother_module = Ballistics()
        self.gain_smooth_in_log = gain_smooth_in_log
        self.with_knee = with_knee
        self.frame_len = frame_len
        self.stride = frame_len // 2
        self.register_buffer("window", torch.hann_window(frame_len))


# ---- envelope followers (reference dynamics.py:1053-1117) ----------------------------------------------
class BaseEnvelopeFollower(nn.Module):
    """log(smoother(loudness) + 1e-5) with loudness = channel-mean energy or amplitude; the smoother is one of
    the HIP smoothers (truncated one-pole scan / attack-release recursion)."""

    def __init__(self, smoother, detect_with="energy"):
        super().__init__()
        if detect_with not in ("energy", "amplitude", "rms_channel"):
            raise ValueError(f"Invalid detect_with: {detect_with}")
        self.detect_with = detect_with
        self.smoother = smoother

    def forward(self, signal, *args, **kwargs):
        if self.detect_with == "energy":
            loudness = signal.square().mean(-2) if needs_grad(signal) else ops.energy(signal)
        elif self.detect_with == "amplitude":
            loudness = signal.abs().mean(-2)
        else:
            raise AttributeError('detect_with="rms_channel" reads an attribute (eps) that upstream never defines '
                                 "(dynamics.py:1071); it cannot be used there either")
        return torch.log(self.smoother(loudness, *args, **kwargs) + 1e-5)

    def parameter_size(self):
        return self.smoother.parameter_size()


class IIREnvelopeFollower(BaseEnvelopeFollower):
    def __init__(self, detect_with="energy", iir_len=16384, flashfftconv=True, max_input_len=2**17):
        super().__init__(TruncatedOnePoleIIRFilter(iir_len=iir_len, flashfftconv=flashfftconv,
                                                   max_input_len=max_input_len), detect_with=detect_with)


class BallisticsEnvelopeFollower(BaseEnvelopeFollower):
    def __init__(self, detect_with="energy"):
        super().__init__(Ballistics(), detect_with=detect_with)
