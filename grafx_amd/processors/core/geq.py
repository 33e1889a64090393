"""Graphic-equaliser band design (mirrors grafx.processors.core.geq.GraphicEqualizerBiquad, reference
core/geq.py:7-200): fixed centre / bandwidth tables (24 Bark bands or 31 third-octave bands), per-band
peaking biquads whose bandwidth term is rescaled so that the gain at the neighbouring band is g^0.4.
A (R, C, K) elementwise front-end on the GPU; the cascade then goes through the native FSM kernels."""
import math

import torch
import torch.nn as nn

# centre frequencies / bandwidths in Hz
_THIRD_OCT_FC = [19.69, 24.80, 31.25, 39.37, 49.61, 62.50, 78.75, 99.21, 125.0, 157.5, 198.4, 250.0, 315.0, 396.9, 500.0,
                 630.0, 793.7, 1000.0, 1260.0, 1587.0, 2000.0, 2520.0, 3175.0, 4000.0, 5040.0, 6350.0, 8000.0, 10080.0,
                 12700.0, 16000.0, 20160.0]
_THIRD_OCT_BW = [9.178, 11.56, 14.57, 18.36, 23.13, 29.14, 36.71, 46.25, 58.28, 73.43, 92.51, 116.6, 146.9, 185.0, 233.1,
                 293.7, 370.0, 466.2, 587.4, 740.1, 932.4, 1175, 1480, 1865, 2350, 2846, 3502, 4253, 5038, 5689, 5573]
_BARK_FC = [50, 150, 250, 350, 450, 570, 700, 840, 1000, 1170, 1370, 1600, 1850, 2150, 2500, 2900, 3400, 4000, 4800,
            5800, 7000, 8500, 10500, 13500]
_BARK_BW = [133.3, 160.0, 171.4, 177.8, 214.7, 235.9, 256.7, 294.4, 315.5, 370.8, 426.9, 466.2, 558.1, 651.0, 744.8,
            926.5, 1110.0, 1467.0, 1828.0, 2194.0, 2735.0, 3619.0, 5333.0, 6000.0]
_NEIGHBOUR_EXPONENT = 0.4


class GraphicEqualizerBiquad(nn.Module):
    def __init__(self, scale="bark", sr=44100):
        super().__init__()
        if scale == "bark":
            fc, bw = torch.tensor(_BARK_FC), torch.tensor(_BARK_BW)
        elif scale == "third_octave":
            fc, bw = torch.tensor(_THIRD_OCT_FC), torch.tensor(_THIRD_OCT_BW)
        else:
            raise ValueError(f"Unsupported scale: {scale}")
        c = [_NEIGHBOUR_EXPONENT] * len(fc)
        fc = fc[fc < sr / 2]
        bw = bw[: len(fc)]
        self.num_bands = len(fc)
        self.register_buffer("fc", fc)
        self.register_buffer("fB", bw)
        self.register_buffer("m2_cos_wc", -2 * torch.cos(2 * math.pi * fc / sr))
        self.register_buffer("tan_B_half", torch.tan(math.pi * bw / sr))
        self.register_buffer("c", torch.tensor(c))

    def forward(self, log_gains, precise=False):
        """``precise``: the same design carried in double precision (the float32 band tables widened, every formula in
        float64) -> float64 coefficients for ops.iir_fsm_fir's double-coefficient form."""
        if precise:
            log_gains = log_gains.double()
            tan_B_half, m2_cos_wc, c = self.tan_B_half.double(), self.m2_cos_wc.double(), self.c.double()
            g = torch.exp(log_gains)
            n2 = torch.exp(log_gains * c).square()
            scale = torch.sqrt(((1 - n2).abs() + 1e-7) / ((g.square() - n2).abs() + 1e-7))
            beta = torch.where(log_gains.abs() < 1e-3, tan_B_half.expand_as(g), tan_B_half * scale)
            gb, mid = g * beta, m2_cos_wc.expand_as(g)
            return torch.stack([1 + gb, mid, 1 - gb], -1), torch.stack([1 + beta, mid, 1 - beta], -1)
        g = torch.exp(log_gains)
        g2 = g.square()
        n2 = torch.exp(log_gains * self.c).square()            # squared gain at the neighbouring band
        scale = torch.sqrt(((1 - n2).abs() + 1e-7) / ((g2 - n2).abs() + 1e-7))
        flat = log_gains.abs() < 1e-3                          # (near-)unity bands keep the nominal bandwidth
        beta = torch.where(flat, self.tan_B_half.expand_as(g), self.tan_B_half * scale)
        gb = g * beta
        mid = self.m2_cos_wc.expand_as(g)
        return torch.stack([1 + gb, mid, 1 - gb], -1), torch.stack([1 + beta, mid, 1 - beta], -1)
