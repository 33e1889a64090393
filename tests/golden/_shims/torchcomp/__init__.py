"""compressor_core recursion as recalled from the public torchcomp algorithm
(Yu et al. 2024): y_t = (1-c_t) y_{t-1} + c_t x_t, c_t = at if x_t < y_{t-1} else rt."""
import torch


def compressor_core(x, zi, at, rt):
    y = torch.empty_like(x)
    prev = zi.clone()
    for t in range(x.shape[1]):
        xt = x[:, t]
        c = torch.where(xt < prev, at, rt)
        prev = (1 - c) * prev + c * xt
        y[:, t] = prev
    return y
