"""L/R <-> M/S (mirrors grafx.processors.core.midside — reference core/midside.py:4-17)."""
import torch


def ms_to_lr(x):
    mid, side = torch.split(x, (1, 1), -2)
    return torch.cat([mid + side, mid - side], -2)


def lr_to_ms(x, mult=0.5):
    left, right = torch.split(x, (1, 1), -2)
    x = torch.cat([left + right, left - right], -2)
    return x if mult is None else x * mult
