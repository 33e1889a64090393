import numpy as np
import scipy.signal
import torch


def lfilter(waveform, a_coeffs, b_coeffs, clamp=False, batching=True):
    x = waveform.detach().cpu().numpy()
    a = a_coeffs.detach().cpu().numpy()
    b = b_coeffs.detach().cpu().numpy()
    y = np.stack([scipy.signal.lfilter(b[i], a[i], x[..., i, :]) for i in range(a.shape[0])], -2)
    return torch.from_numpy(y).to(waveform.dtype)
