"""Oracle (test infrastructure): LTI cores of the reference, restated.

Every function cites the reference lines it follows (paths relative to
/root/reference/src/grafx/processors).  Works in whatever float dtype the
inputs carry, so tests can also run it in float64 as a tie-breaker.
"""
import math

import torch


def convolve(x, h, mode="causal"):
    """core/convolution.py:119-134.

    Zero-pad both operands to P = Lx + Lh - 1, multiply real FFTs, invert with
    the *default* irfft length 2*(P//2) (so P-1 samples when P is odd — the
    reference never passes ``n=``), then slice: causal -> [:Lx],
    zerophase -> [Lh//2 : Lh//2 + Lx].
    """
    lx, lh = x.shape[-1], h.shape[-1]
    p = lx + lh - 1
    spec = torch.fft.rfft(x, n=p) * torch.fft.rfft(h, n=p)
    y = torch.fft.irfft(spec)  # length 2*(p//2): reference quirk kept on purpose
    if mode == "causal":
        return y[..., :lx]
    if mode == "zerophase":
        return y[..., lh // 2 : lh // 2 + lx]
    return y


def linear_convolve(x, h, mode="causal"):
    """True linear convolution (what `convolve` equals whenever P is even)."""
    lx, lh = x.shape[-1], h.shape[-1]
    p = lx + lh - 1
    n = 1 << (p - 1).bit_length()
    y = torch.fft.irfft(torch.fft.rfft(x, n=n) * torch.fft.rfft(h, n=n), n=n)[..., :p]
    if mode == "causal":
        return y[..., :lx]
    if mode == "zerophase":
        return y[..., lh // 2 : lh // 2 + lx]
    return y


def fsm_delays(order, fir_len, real_dtype=torch.float32):
    """core/iir.py:269-276 — D[d,k] = exp(-j*phase), phase = d*k/N*2*pi.

    The reference forms the phase from int64 tensors: (d*k)/N is a true
    division into the default float dtype, then *2, then *pi — all in that
    float dtype — and exponentiates in the matching complex dtype.
    """
    d = torch.arange(order + 1)
    k = torch.arange(fir_len // 2 + 1)
    phase = (d[:, None] * k[None, :]).to(real_dtype) / fir_len * 2 * math.pi
    return torch.exp(-1j * phase)


def iir_fsm(Bs, As, fir_len):
    """core/iir.py:263-267 + 149 — sampled cascade response (R,Cf,N//2+1)."""
    delays = fsm_delays(Bs.shape[-1] - 1, fir_len, Bs.dtype).to(Bs.device)
    num = (Bs.unsqueeze(-1) * delays).sum(-2)
    den = (As.unsqueeze(-1) * delays).sum(-2)
    return (num / den).prod(-2)


def iir_fsm_fir(Bs, As, fir_len):
    """core/iir.py:147-150 — the length-N FIR the FSM backend convolves with."""
    return torch.fft.irfft(iir_fsm(Bs, As, fir_len), dim=-1, n=fir_len)


def one_pole_fir(z_alpha, iir_len):
    """core/envelope.py:51-60 — h[n] = (1-a) * exp(n*log a), a = min(sigmoid(z), 1-1e-5)."""
    alpha = torch.sigmoid(z_alpha).clamp(max=1 - 1e-5)
    n = torch.arange(iir_len, device=z_alpha.device)[None, :]
    return (1 - alpha) * torch.exp(n * torch.log(alpha))


def truncated_one_pole(u, z_alpha, iir_len):
    """core/envelope.py:34-49 — relu(convolve(u, h, causal)) on (R,L) tensors."""
    return torch.relu(convolve(u, one_pole_fir(z_alpha, iir_len), "causal"))
