"""Band-split noise for the noise-shaping reverb (init-time, host side; mirrors grafx.processors.core.noise —
reference core/noise.py:9-73).  A Linkwitz-Riley crossover tree: at every break frequency the running signal is
split by squared Butterworth low/high-pass filters (zero-phase `sosfiltfilt`, or two causal passes); the low
branch is a band, the high branch continues to the next break."""
import numpy as np
import torch
from scipy.signal import butter, sosfilt, sosfiltfilt

from .scale import from_scale, to_scale


def apply_linkwitz_riley(input_audio, num_bands=2, f_min=40, f_max=None, scale="bark_traunmuller", sr=44100,
                         zerophase=True, order=2):
    grid = np.linspace(to_scale(f_min, scale), to_scale(f_max, scale), 2 * num_bands - 1)
    breaks = from_scale(grid[1::2], scale)

    def split(sos, sig):
        return sosfiltfilt(sos, sig) if zerophase else sosfilt(sos, sosfilt(sos, sig))

    bands, rest = [], input_audio
    for f in breaks:
        low = butter(order, f, "lowpass", fs=sr, output="sos")
        high = butter(order, f, "highpass", fs=sr, output="sos")
        bands.append(split(low, rest))
        rest = split(high, rest)
    bands.append(rest)
    return np.stack(bands, 1)


def get_filtered_noise(fir_len, num_channels=1, num_bands=12, f_min=31.5, f_max=16000, scale="log", sr=44100,
                       zerophase=True, order=2):
    noise = 2 * np.random.rand(num_channels, fir_len) - 1
    bands = apply_linkwitz_riley(noise, num_bands=num_bands, f_min=f_min, f_max=f_max, scale=scale, sr=sr,
                                 zerophase=zerophase, order=order)
    return torch.from_numpy(bands).float()
