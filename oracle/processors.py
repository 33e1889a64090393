"""Oracle (test infrastructure): processor-level restatements of the reference.

Each class mirrors the reference processor's call signature
(``forward(input_signals, **params)`` + ``parameter_size()``) so it can be
dropped into ``render_grafx`` as the checker.  CPU / any float dtype.
Paths cited are relative to /root/reference/src/grafx/processors.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .lti import convolve, iir_fsm_fir, truncated_one_pole


# ----------------------------------------------------------------------------- helpers
def ms_to_lr(x):
    """core/midside.py:4-8."""
    m, s = x[..., 0:1, :], x[..., 1:2, :]
    return torch.cat([m + s, m - s], -2)


def lr_to_ms(x, mult=0.5):
    """core/midside.py:11-17."""
    l, r = x[..., 0:1, :], x[..., 1:2, :]
    y = torch.cat([l + r, l - r], -2)
    return y if mult is None else y * mult


def normalize_impulse(ir, eps=1e-12):
    """core/utils.py:14-18 — divide by sqrt(mean_c sum_t ir^2 + eps)."""
    e = ir.square().sum(2, keepdim=True).mean(1, keepdim=True)
    return ir / torch.sqrt(e + eps)


def ballistics(x, z_alpha):
    """core/envelope.py:84-101 -> ``torchcomp.compressor_core(x, zi, at, rt)`` with ``at, rt = sigmoid(z)[..., 0],
    sigmoid(z)[..., 1]`` and ``zi = 1``.

    The arithmetic lives in a third-party dependency that is ABSENT from /root/reference and from this image:
    ``torchcomp`` (pyproject.toml:19, no version pinned; the package of Yu et al., "Differentiable All-pole Filters for
    Time-varying Audio Systems", DAFx 2024 -- cited by the reference at core/envelope.py:77 and dynamics.py:226).  Its
    published algorithm, restated (torchcomp/core.py ``compressor_kernel``, the numba loop behind ``compressor_core`` on
    CPU; the paper's feed-forward compressor smooths the gain with g^[n] = alpha g^[n-1] + (1 - alpha) g[n], alpha the
    attack constant when g[n] < g^[n-1] and the release constant otherwise, and torchcomp passes ``at = 1 - alpha_at``,
    ``rt = 1 - alpha_rt``, its ``ms2coef``):

        g = zi[b]
        for t in range(T):
            f = x[b, t]
            coeff = at[b] if f < g else rt[b]
            g *= 1 - coeff
            g += coeff * f            # two products and one sum, each rounded (numba does not contract them)
            y[b, t] = g

    i.e. reading "T":  y[-1] = 1;  c = at if x[n] < y[n-1] else rt;  y[n] = (1 - c) y[n-1] + c x[n].

    PARITY UNPINNED: no wheel, no upstream golden vector and no upstream numeric test exist for this call (the reference's
    tests only smoke-test it on CUDA, tests/processors/test_dynamics.py:41-98), so nothing here can be checked against the
    third-party code itself; tests that depend on it carry the ``provisional`` marker.  What IS anchored: the call site
    (argument order, sigmoid, zi = 1: core/envelope.py:97-100) and the closed forms of tests/test_ballistics_readings.py.
    The reference's *docstring* (core/envelope.py:68-73) describes the smoother the other way round -- reading "D",
    :func:`ballistics_docstring_reading`: coefficient on y[n-1], and the *release* coefficient when u[n] < y[n-1].  The
    two are different functions of z_alpha; "T" is implemented because the code path, not the prose, is what a user's
    parameters go through.
    """
    ts = torch.sigmoid(z_alpha)
    at, rt = ts[..., 0], ts[..., 1]
    return _attack_release(x, at, rt, on_input=True)


def ballistics_coefficients(x, at, rt):
    """The recursion of :func:`ballistics` for given coefficients ``at``, ``rt`` (R,), in x's dtype: for float32 input this
    is the float32 sequential loop -- (1 - c), its product with the state, c * x and the sum each rounded once -- which
    the HIP kernels reproduce bit for bit (tests/test_gpu_ballistics.py hands the same float32 coefficients to both)."""
    return _attack_release(x, at, rt, on_input=True)


def ballistics_docstring_reading(x, z_alpha):
    """Reading "D" of core/envelope.py:68-73, NOT what the kernel implements (see :func:`ballistics`):
    alpha_A, alpha_R = sigmoid(z)[..., 0], sigmoid(z)[..., 1];  y[-1] = 1;
    y[n] = alpha_R y[n-1] + (1 - alpha_R) u[n] if u[n] < y[n-1] else alpha_A y[n-1] + (1 - alpha_A) u[n]."""
    ts = torch.sigmoid(z_alpha)
    alpha_a, alpha_r = ts[..., 0], ts[..., 1]
    return _attack_release(x, 1 - alpha_r, 1 - alpha_a, on_input=True)


def _attack_release(x, c_below, c_above, on_input=True):
    """y[n] = (1 - c) y[n-1] + c x[n], c = c_below if x[n] < y[n-1] else c_above, y[-1] = 1 (numpy loop, x's dtype)."""
    xs = x.detach().cpu().double().numpy()
    a, r = c_below.detach().cpu().double().numpy(), c_above.detach().cpu().double().numpy()
    if x.dtype == torch.float32:
        xs, a, r = xs.astype(np.float32), a.astype(np.float32), r.astype(np.float32)
    y = np.empty_like(xs)
    prev = np.ones(xs.shape[0], dtype=xs.dtype)
    one = xs.dtype.type(1)
    for n in range(xs.shape[1]):
        c = np.where(xs[:, n] < prev, a, r)
        prev = (one - c) * prev + c * xs[:, n]
        y[:, n] = prev
    return torch.from_numpy(y).to(x.dtype)


# ----------------------------------------------------------------------------- coefficient maps
def peq_biquad_coefficients(w0, q_inv, log_gain, use_shelving_filters=True):
    """eq.py:273-314 + filter.py:593-604, 645-656, 687-705, 736-754.

    (R, C_eq, K) pre-activations -> Bs, As of shape (R, C_eq, K, 3); un-normalised a0.
    """
    w = math.pi * torch.sigmoid(w0)
    qi = torch.exp(q_inv)
    A = torch.exp(log_gain)
    cw = torch.cos(w)
    alpha = torch.sin(w) * qi * 0.5

    def peaking(cw, al, A):
        b = torch.stack([1 + al * A, -2 * cw, 1 - al * A], -1)
        a = torch.stack([1 + al / A, -2 * cw, 1 - al / A], -1)
        return b, a

    def shelf(cw, al, A, sign):
        # sign=+1 low shelf (filter.py:687-705), sign=-1 high shelf (736-754)
        ap1, am1 = A + 1, A - 1
        s = 2 * A.sqrt() * al
        b0 = A * (ap1 - sign * am1 * cw + s)
        b1 = sign * 2 * A * (am1 - sign * ap1 * cw)
        b2 = A * (ap1 - sign * am1 * cw - s)
        a0 = ap1 + sign * am1 * cw + s
        a1 = -sign * 2 * (am1 + sign * ap1 * cw)
        a2 = ap1 + sign * am1 * cw - s
        return torch.stack([b0, b1, b2], -1), torch.stack([a0, a1, a2], -1)

    if not use_shelving_filters:
        return peaking(cw, alpha, A)
    K = w0.shape[-1]
    parts = []
    parts.append(shelf(cw[..., :1], alpha[..., :1], A[..., :1], +1))
    parts.append(peaking(cw[..., 1 : K - 1], alpha[..., 1 : K - 1], A[..., 1 : K - 1]))
    parts.append(shelf(cw[..., K - 1 :], alpha[..., K - 1 :], A[..., K - 1 :], -1))
    return torch.cat([p[0] for p in parts], -2), torch.cat([p[1] for p in parts], -2)


def biquad_coefficients(Bs, A1_pre, A2_pre, A0=None):
    """filter.py:144-156 — stability activations; returns (R,1,K,3) Bs, As."""
    a1 = 2 * torch.tanh(A1_pre)
    a2 = ((2 - a1.abs()) * torch.tanh(A2_pre) + a1.abs()) / 2
    As = torch.stack([torch.ones_like(a1), a1, a2], -1)
    if A0 is not None:
        As = As * A0.unsqueeze(-1)
    Bs = torch.cat([Bs[..., :1] + 1, Bs[..., 1:]], -1)
    return Bs.unsqueeze(1), As.unsqueeze(1)


# ----------------------------------------------------------------------------- processors
class _Oracle(torch.nn.Module):
    pass


class OracleStereoGain(_Oracle):
    """stereo.py:25-48."""

    def forward(self, input_signals, log_gain):
        return input_signals * torch.exp(log_gain)[..., None]

    def parameter_size(self):
        return {"log_gain": 2}


class OracleBiquadFilter(_Oracle):
    """filter.py:118-168 with the "fsm" backend (core/iir.py:147-152)."""

    def __init__(self, num_filters=1, normalized=False, fsm_fir_len=4000, **_):
        super().__init__()
        self.num_filters, self.normalized, self.fsm_fir_len = num_filters, normalized, fsm_fir_len

    def forward(self, input_signals, Bs, A1_pre, A2_pre, A0=None):
        Bs, As = biquad_coefficients(Bs, A1_pre, A2_pre, A0 if self.normalized else None)
        return convolve(input_signals, iir_fsm_fir(Bs, As, self.fsm_fir_len), "causal")

    def parameter_size(self):
        size = {"Bs": (self.num_filters, 3), "A1_pre": self.num_filters, "A2_pre": self.num_filters}
        if self.normalized:
            size["A0"] = self.num_filters
        return size


class OracleParametricEqualizer(_Oracle):
    """eq.py:243-336."""

    def __init__(self, num_filters=10, processor_channel="mono", use_shelving_filters=True,
                 fsm_fir_len=4000, **_):
        super().__init__()
        if processor_channel not in ("mono", "stereo", "midside"):
            raise ValueError(f"Invalid processor_channel: {processor_channel}")
        self.num_filters, self.processor_channel = num_filters, processor_channel
        self.use_shelving_filters, self.fsm_fir_len = use_shelving_filters, fsm_fir_len

    def forward(self, input_signals, w0, q_inv, log_gain):
        Bs, As = peq_biquad_coefficients(w0, q_inv, log_gain, self.use_shelving_filters)
        fir = iir_fsm_fir(Bs, As, self.fsm_fir_len)
        if self.processor_channel == "midside":
            return ms_to_lr(convolve(lr_to_ms(input_signals), fir, "causal"))
        return convolve(input_signals, fir, "causal")

    def parameter_size(self):
        c = 1 if self.processor_channel == "mono" else 2
        return {k: (c, self.num_filters) for k in ("w0", "q_inv", "log_gain")}


class OracleSTFTMaskedNoiseReverb(_Oracle):
    """reverb.py:57-228 (fixed noise; gain envelope optional)."""

    def __init__(self, ir_len=60000, processor_channel="pseudo_midside", n_fft=384, hop_length=192,
                 gain_envelope=False, **_):
        super().__init__()
        self.ir_len, self.n_fft, self.hop = ir_len, n_fft, hop_length
        self.num_frames, self.num_bins = 1 + ir_len // hop_length, 1 + n_fft // 2
        self.gain_envelope, self.processor_channel = gain_envelope, processor_channel
        self.register_buffer("window", torch.hann_window(n_fft))
        # reverb.py:101-114: RandomState(0) uniform noise in [-1,1), float32, centred/reflect STFT
        noise = torch.tensor(np.random.RandomState(0).uniform(size=(2, ir_len)) * 2 - 1).float()
        spec = torch.stft(noise, n_fft=n_fft, hop_length=hop_length, window=self.window, return_complex=True)
        self.register_buffer("noise_stft", spec[None])

    def compute_ir(self, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None):
        dt = init_log_magnitude.dtype
        m = torch.arange(self.num_frames, device=init_log_magnitude.device).view(1, 1, 1, -1)
        logmag = init_log_magnitude[..., None] - F.softplus(delta_log_magnitude)[..., None] * m
        if self.gain_envelope:
            logmag = logmag + gain_env_log_magnitude[:, :, None, :]
        spec = self.noise_stft.to(torch.complex128 if dt == torch.float64 else torch.complex64) * torch.exp(logmag / 8)
        r = spec.shape[0]
        ir = torch.istft(spec.reshape(r * 2, self.num_bins, self.num_frames), n_fft=self.n_fft,
                         hop_length=self.hop, window=self.window.to(dt), length=self.ir_len)
        return ir.view(r, 2, self.ir_len)

    def forward(self, input_signals, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude=None):
        ir = self.compute_ir(init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude)
        match self.processor_channel:
            case "pseudo_midside":  # reverb.py:225-228
                return convolve(input_signals, normalize_impulse(ms_to_lr(ir)), "causal")
            case "midside":  # 219-223
                return ms_to_lr(convolve(lr_to_ms(input_signals), normalize_impulse(ir), "causal"))
            case _:  # 215-217
                return convolve(input_signals, normalize_impulse(ir), "causal")

    def parameter_size(self):
        size = {"init_log_magnitude": (2, self.num_bins), "delta_log_magnitude": (2, self.num_bins)}
        if self.gain_envelope:
            size["gain_env_log_magnitude"] = (2, self.num_frames)
        return size


class OracleBallistics(_Oracle):
    def forward(self, input_signals, z_alpha):
        return ballistics(input_signals, z_alpha)


class _OracleDynamics(_Oracle):
    """Shared skeleton of dynamics.py:361-409 (Compressor) / 598-641 (NoiseGate)."""

    def __init__(self, energy_smoother="iir", gain_smoother=None, gain_smooth_in_log=False,
                 knee="quadratic", iir_len=16384, **_):
        super().__init__()
        for name, s in (("energy_smoother", energy_smoother), ("gain_smoother", gain_smoother)):
            if s not in ("iir", "ballistics", None):
                raise ValueError(f"Unknown {name}: {s}")
        if knee not in ("hard", "quadratic", "exponential"):
            raise ValueError(f"Unknown knee: {knee}")
        self.energy_smoother, self.gain_smoother = energy_smoother, gain_smoother
        self.gain_smooth_in_log, self.knee, self.iir_len = gain_smooth_in_log, knee, iir_len

    def _smooth(self, kind, u, z):
        if kind == "iir":
            return truncated_one_pole(u, z, self.iir_len)
        return ballistics(u, z)

    def forward(self, input_signals, log_threshold, log_ratio, log_knee=None, z_alpha_pre=None, z_alpha_post=None):
        energy = input_signals.square().mean(-2)
        if self.energy_smoother is not None:
            energy = self._smooth(self.energy_smoother, energy, z_alpha_pre)
        G = torch.log(energy + 1e-5)
        g = self.log_gain(G, log_threshold - 6, log_ratio, log_knee)
        if self.gain_smoother is None:
            gain = torch.exp(g)
        elif self.gain_smooth_in_log:
            gain = torch.exp(self._smooth(self.gain_smoother, g, z_alpha_post))
        else:
            gain = self._smooth(self.gain_smoother, torch.exp(g), z_alpha_post)
        return gain[:, None, :] * input_signals

    def parameter_size(self):
        size = {"log_threshold": 1, "log_ratio": 1}
        if self.knee != "hard":
            size["log_knee"] = 1
        for key, s in (("z_alpha_pre", self.energy_smoother), ("z_alpha_post", self.gain_smoother)):
            if s == "iir":
                size[key] = 1
            elif s == "ballistics":
                size[key] = 2
        return size


class OracleCompressor(_OracleDynamics):
    """dynamics.py:444-489."""

    def log_gain(self, G, T, log_ratio, log_knee):
        R = 1 + torch.exp(log_ratio)
        if self.knee == "hard":
            return torch.minimum(G, T + (G - T) / R) - G
        if self.knee == "quadratic":
            W = torch.exp(log_knee) / 2
            below, above = G < (T - W), G > (T + W)
            mid = ~below & ~above
            out = G * below + (T + (G - T) / R) * above + (G + (1 / R - 1) * (G - T + W).square() / (4 * W)) * mid
            return out - G
        k = torch.exp(log_knee)
        return (1 / R - 1) * F.softplus(k * (G - T)) / k


class OracleNoiseGate(_OracleDynamics):
    """dynamics.py:676-721."""

    def log_gain(self, G, T, log_ratio, log_knee):
        R = 1 + torch.exp(log_ratio)
        if self.knee == "hard":
            return torch.minimum(G, R * (G - T) + T) - G
        if self.knee == "quadratic":
            W = torch.exp(log_knee) / 2
            below, above = G < (T - W), G > (T + W)
            mid = ~below & ~above
            out = (R * (G - T) + T) * below + G * above + (G + (1 - R) * (G - T - W).square() / (4 * W)) * mid
            return out - G
        k = torch.exp(log_knee)
        return -torch.exp(log_ratio) * F.softplus(k * (T - G)) / k
