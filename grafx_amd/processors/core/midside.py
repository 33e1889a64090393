"""Left/right <-> mid/side on the channel axis (semantics of grafx.processors.core.midside, core/midside.py:4-17)."""
import torch


def _sum_and_difference(x):
    """(a, b) on the channel axis -> (a + b, a - b)."""
    first, second = x.narrow(-2, 0, 1), x.narrow(-2, 1, 1)
    return torch.cat((first + second, first - second), dim=-2)


def ms_to_lr(x):
    """(mid, side) -> (mid + side, mid - side): no scaling on the way back."""
    return _sum_and_difference(x)


def lr_to_ms(x, mult=0.5):
    """(left, right) -> mult * (left + right, left - right); ``mult=None`` leaves the sums unscaled."""
    ms = _sum_and_difference(x)
    return ms if mult is None else ms * mult
