"""Biquad-cascade IIR filter (mirrors grafx.processors.core.iir.IIRFilter —
reference core/iir.py:25-276): the frequency-sampling ("fsm") backend and the exact recursive backends
("lfilter", "ssm") on HIP."""
import torch
import torch.nn as nn

from ... import autograd as diff
from ... import ops
from ...autograd import needs_grad
from .convolution import FIRConvolution, convolve_taps


class IIRFilter(nn.Module):
    def __init__(self, order=2, backend="fsm", flashfftconv=True, fsm_fir_len=4000,
                 fsm_max_input_len=2**17, fsm_regularization=False):
        super().__init__()
        self.backend = backend
        self.fsm_fir_len = fsm_fir_len
        self.fsm_regularization = fsm_regularization
        self.flashfftconv = False  # resolved as upstream resolves it without FlashFFTConv (FIRConvolution below warns)
        if flashfftconv:  # same precondition as upstream (iir.py:110-112)
            assert fsm_fir_len % 2 == 0
            assert fsm_max_input_len % 2 == 0
        self.order = order
        if backend == "fsm":
            # order != 2 (no processor of the package uses it): the sampled response is formed by torch ops on the GPU, as
            # upstream (iir.py:147-150 with delays = arange(order + 1)); the taps then take the native convolution
            if fsm_regularization:
                assert False  # upstream: iir.py:122-123
            self.conv = FIRConvolution(mode="causal", flashfftconv=flashfftconv, max_input_len=fsm_max_input_len)
            self.flashfftconv = self.conv.flashfftconv
            self._plans = {}
        elif backend in ("lfilter", "ssm"):
            # upstream: torchaudio.functional.lfilter per section / a state-space form on torchlpc (iir.py:154-261).
            # Here both are the exact recursive cascade as one parallel-scan HIP kernel (gfx_biquad_cascade_f32).
            # (first-order sections ride on it with a zero third coefficient; upstream's "ssm" asserts order 2, iir.py:226)
            if order not in (1, 2) or (order == 1 and backend == "ssm"):
                raise NotImplementedError("the HIP recursive kernel runs first- and second-order sections (order 1 or 2; "
                                          "backend='ssm': 2, as upstream); use backend='fsm' for higher orders")
        else:
            raise ValueError(f"Unsupported backend: {backend}")

    def _plan(self, device):
        key = (device.type, device.index)
        if key not in self._plans:
            self._plans[key] = ops.iir_fsm_plan(self.fsm_fir_len, device)
        return self._plans[key]

    FSM_NATIVE_MAX = 4096  # the Bluestein inverse DFT runs on one 8192-point LDS tile: 2N - 1 <= 8192

    def _taps(self, Bs, As):
        """(R,Cf,K,3) coefficients -> (R*Cf, N) frequency-sampled taps (iir.py:148-150): the native response +
        Bluestein kernel up to 4096 taps and the tile's own inverse transform at 8192 / 16384; other lengths: the same
        response formula as torch ops on the GPU (complex64 as upstream), inverted by the direct-sum kernel up to 8192
        taps and by the FFT library in float64 beyond."""
        N = self.fsm_fir_len
        if Bs.shape[-1] == 3 and ops.iir_fsm_native(N):  # 1..4096 (Bluestein on the LDS tile), 8192 and 16384 (tile inverse)
            return ops.iir_fsm_fir(Bs, As, N, self._plan(Bs.device))
        k = torch.arange(N // 2 + 1, device=Bs.device)
        d = torch.arange(Bs.shape[-1], device=Bs.device)
        delays = torch.exp(-1j * ((d[:, None] * k[None, :]).to(Bs.dtype) / N * 2 * torch.pi))
        resp = ((Bs.unsqueeze(-1) * delays).sum(-2) / (As.unsqueeze(-1) * delays).sum(-2)).prod(-2)
        if N <= ops.IRDFT_MAX_N and resp.dtype == torch.complex64:   # any length up to 8192: direct-sum kernel, no FFT library
            return ops.irdft(resp.contiguous(), N).reshape(-1, N)
        if resp.is_cuda:
            ops.fft_library_reached(f"IIRFilter(fsm_fir_len={N}): the taps' inverse real DFT")
        return torch.fft.irfft(resp.to(torch.complex128), dim=-1, n=N).float().reshape(-1, N)

    def fsm_fir(self, Bs, As):
        """(R,Cf,K,3) coefficients -> (R,Cf,N) FIR the FSM backend convolves with (iir.py:148-150)."""
        R, Cf = Bs.shape[0], Bs.shape[1]
        return self._taps(Bs, As).view(R, Cf, self.fsm_fir_len)

    # the two static helpers of the reference class (core/iir.py:263-276), for code that builds responses by hand
    @staticmethod
    def delay(delay_length, fir_length):
        """exp(-j 2 pi d k / N) for every delay d in ``delay_length`` and k = 0..N//2 (a trailing frequency axis)."""
        k = torch.arange(fir_length // 2 + 1, device=delay_length.device)
        k = k.reshape((1,) * delay_length.ndim + (-1,))
        return torch.exp(-1j * (delay_length.unsqueeze(-1) * k / fir_length * 2 * torch.pi))

    @staticmethod
    def iir_fsm(Bs, As, delays, eps=1e-10):
        """Sampled response of every section: sum_d B_d D_d / sum_d A_d D_d (``eps`` is accepted and unused upstream)."""
        return (Bs.unsqueeze(-1) * delays).sum(-2) / (As.unsqueeze(-1) * delays).sum(-2)

    def _process_recursive(self, input_signal, Bs, As, out=None):
        if Bs.shape[-1] == 2 and As.shape[-1] == 2:     # first-order sections: b2 = a2 = 0 (differentiable: a zero pad)
            Bs, As = torch.nn.functional.pad(Bs, (0, 1)), torch.nn.functional.pad(As, (0, 1))
        if needs_grad(input_signal, Bs, As):
            x3 = input_signal.reshape(-1, *input_signal.shape[-2:]) if input_signal.ndim == 4 else input_signal
            if self.backend == "ssm" and Bs.shape[2] > 1:
                y = self._ssm_quirk_differentiable(x3, Bs, As)
            else:
                y = diff.BiquadCascadeFn.apply(x3, Bs, As)   # native recursion both ways (autograd.BiquadCascadeFn)
            if input_signal.ndim == 4:
                y = y.view(*input_signal.shape[:2], *y.shape[1:])
            if out is None:
                return y
            out.copy_(y.view(out.shape))
            return out
        if self.backend == "ssm":
            assert Bs.shape[-1] == As.shape[-1] == 3, "The filter order must be 2."
        # "ssm" with K > 1: upstream drives every section's recursion with the original input (iir.py:226-246)
        return ops.biquad_cascade(input_signal, Bs, As, ssm_quirk=self.backend == "ssm", out=out)

    @staticmethod
    def _ssm_quirk_differentiable(x, Bs, As):
        """Upstream's "ssm" with K > 1 sections (core/iir.py:186-260) is not a cascade: every section's RECURSION is driven by
        the ORIGINAL input and only its direct term b0 by the previous section's output,
            y_k = b0_k y_{k-1} + z^-1 (b12_k / A_k) x,      b12 = b[1:] / a0 - b0 a[1:] / a0,  b0 = b[0] / a0,
        i.e. K parallel one-section filters of the input folded with the direct gains.  Spelled out like that it is K native
        single-section recursions (autograd.BiquadCascadeFn: native backward) and a few products -- the same values as the
        forward-only kernel path (gfx_biquad_cascade_f32 with ssm_quirk), with gradients (round 6: this used to raise)."""
        a0 = As[..., :1]
        Bn, a12 = Bs / a0, As[..., 1:] / a0
        b0 = Bn[..., :1]                                   # (R, Cf, K, 1)
        b12 = Bn[..., 1:] - b0 * a12                       # (R, Cf, K, 2)
        one, zero = torch.ones_like(b0[:, :, 0]), torch.zeros_like(b0[:, :, 0])
        y = x
        for k in range(Bs.shape[2]):
            Bk = torch.cat([zero, b12[:, :, k]], -1).unsqueeze(2)          # z^-1 (b12_0 + b12_1 z^-1)
            Ak = torch.cat([one, a12[:, :, k]], -1).unsqueeze(2)
            y = b0[:, :, k] * y + diff.BiquadCascadeFn.apply(x, Bk, Ak)
        return y

    def forward(self, input_signal, Bs, As, out=None, tee=None, shared_rows=None, final=False):
        """``shared_rows``: Bs/As hold that many rows, shared by the batch (signal row r uses r % shared_rows).
        ``final``: the caller returns this output as its own up to linear operations (autograd.TAPE_ONLY)."""
        if shared_rows is not None and self.backend != "fsm":
            rows = input_signal.shape[0] * input_signal.shape[1] if input_signal.ndim == 4 else input_signal.shape[0]
            Bs, As = Bs.repeat(rows // shared_rows, 1, 1, 1), As.repeat(rows // shared_rows, 1, 1, 1)  # no row sharing
            shared_rows = None
        if self.backend != "fsm":
            if tee is not None:
                tee.copy_(input_signal)
            return self._process_recursive(input_signal, Bs, As, out=out)
        if tee is not None and needs_grad(input_signal, Bs, As):
            tee.copy_(input_signal)
            tee = None
        if needs_grad(input_signal, Bs, As):  # training path: torch front-end + native conv fwd/bwd
            # a strided (B,n,C,L) view goes through as it is (the native convolution reads it in place)
            if self.fsm_fir_len <= self.FSM_NATIVE_MAX and Bs.shape[-1] == 3:  # native taps, written-out backward
                h = diff.FsmFirFn.apply(Bs, As, self.fsm_fir_len, self._plan(Bs.device))
            else:
                h = diff.fsm_fir(Bs, As, self.fsm_fir_len)
            y = diff.convolve(input_signal, h, "causal", exact=self.flashfftconv, final=final)
            if out is None:
                return y
            out.copy_(y.view(out.shape))
            return out
        R, Cf = Bs.shape[0], Bs.shape[1]
        N = self.fsm_fir_len
        h = self._taps(Bs, As)
        return convolve_taps(input_signal, ops.fir_spectrum(h), N, Cf, "causal", out=out, tee=tee, exact=self.flashfftconv,
                             h_rows=shared_rows)
