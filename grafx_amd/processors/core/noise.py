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


OCTAVE_CENTRES = (31.5, 63, 125, 250, 500, 1000, 2000, 4000, 8000, 16000)


def octave_band_filterbank(num_taps, sample_rate):
    """(12, 1, num_taps) time-reversed windowed-sinc FIRs: a 12 Hz low-pass, ten octave band-passes centred on
    31.5 Hz ... 16 kHz (edges a half octave either side, clipped just below Nyquist) and an 18 kHz high-pass
    (reference core/noise.py:76-125; unused by the processors there, kept for users of the helper)."""
    from scipy.signal import firwin

    nyq = 0.999 * sample_rate / 2
    designs = [firwin(num_taps, 12, fs=sample_rate)]
    designs += [firwin(num_taps, [fc / 2**0.5, min(fc * 2**0.5, nyq)], fs=sample_rate, pass_zero=False) for fc in OCTAVE_CENTRES]
    designs.append(firwin(num_taps, 18000, fs=sample_rate, pass_zero=False))
    taps = torch.from_numpy(np.stack(designs).astype("float32"))
    return taps.flip(-1).unsqueeze(1)
