"""Small tensor helpers with the semantics of grafx.processors.core.utils (reference core/utils.py:7-18)."""
import torch


def _log_mean_square(signal, eps):
    """log of the mean square over channels and time, per row."""
    return (signal * signal).mean(dim=(-2, -1)).add(eps).log()


def rms_difference(X, Y, eps=1e-7):
    """Sum over rows of |log-energy(X) - log-energy(Y)| (core/utils.py:7-11)."""
    return torch.sum(torch.abs(_log_mean_square(X, eps) - _log_mean_square(Y, eps)))


def normalize_impulse(ir, eps=1e-12):
    """Scale every (row) impulse response to unit energy: sum over time, mean over channels (core/utils.py:14-18)."""
    if ir.ndim != 3:
        raise AssertionError(f"impulse responses are (rows, channels, taps), got {tuple(ir.shape)}")
    energy = (ir * ir).sum(dim=-1, keepdim=True).mean(dim=-2, keepdim=True)
    return ir * torch.rsqrt(energy + eps)
