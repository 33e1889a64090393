import torch


def sample_wise_lpc(x, A, zi=None):
    B, T = x.shape
    order = A.shape[-1]
    y = torch.zeros(B, T + order, dtype=x.dtype)
    for t in range(T):
        acc = x[:, t].clone()
        for k in range(order):
            acc = acc - A[:, t, k] * y[:, t + order - k - 1]
        y[:, t + order] = acc
    return y[:, order:]
