"""Small tensor helpers (mirrors grafx.processors.core.utils — reference core/utils.py:7-18)."""
import torch


def rms_difference(X, Y, eps=1e-7):
    X_rms = torch.log(X.square().mean((-1, -2)) + eps)
    Y_rms = torch.log(Y.square().mean((-1, -2)) + eps)
    return (X_rms - Y_rms).abs().sum()


def normalize_impulse(ir, eps=1e-12):
    assert ir.ndim == 3
    e = ir.square().sum(2, keepdim=True).mean(1, keepdim=True)
    return ir / torch.sqrt(e + eps)
