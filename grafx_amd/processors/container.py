"""Processor containers (mirrors grafx.processors.container — reference container.py:10-299).
Pure composition: they call the wrapped (HIP) processors and combine their outputs with a few
elementwise torch ops; no kernels of their own."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .core.utils import rms_difference


def _split(out):
    return out if isinstance(out, tuple) else (out, None)


class DryWet(nn.Module):
    """y = w * processor(x) + (1 - w) * x (container.py:10-82)."""

    def __init__(self, processor, external_param=True):
        super().__init__()
        self.processor = processor
        self.external_param = external_param

    def forward(self, input_signals, drywet_weight, **processor_kwargs):
        wet, extra = _split(self.processor(input_signals, **processor_kwargs))
        w = drywet_weight.view(-1, 1, 1)
        mixed = w * wet + (1 - w) * input_signals
        return mixed if extra is None else (mixed, extra)

    def parameter_size(self):
        size = self.processor.parameter_size()
        if not self.external_param:
            size["drywet_weight"] = (1,)
        return size


class SerialChain(nn.Module):
    """Processors applied one after another; parameters are a dict keyed like the processors (85-148)."""

    def __init__(self, processors):
        super().__init__()
        self.processors = nn.ModuleDict(processors)

    def forward(self, input_signals, **processors_kwargs):
        signal, intermediates = input_signals, {}
        for name, proc in self.processors.items():
            signal, extra = _split(proc(signal, **processors_kwargs[name]))
            if extra is not None:
                intermediates[name] = extra
        return signal, intermediates

    def parameter_size(self):
        return {k: v.parameter_size() for k, v in self.processors.items()}


class ParallelMix(nn.Module):
    """Weighted sum of processors fed the same input (151-222)."""

    def __init__(self, processors, activation="softmax"):
        super().__init__()
        self.processors = nn.ModuleDict(processors)
        if activation not in ("softmax", "softplus"):
            raise ValueError(f"Unsupported activation: {activation}")
        self.activation = activation
        self.mult = 1 / (math.log(2) * len(self.processors))

    def get_weight(self, weights):
        if self.activation == "softmax":
            return torch.softmax(weights, dim=-1)
        return F.softplus(weights) * self.mult

    def forward(self, input_signals, parallel_weights, **processors_kwargs):
        weights = self.get_weight(parallel_weights)
        total, intermediates = None, {}
        for i, (name, proc) in enumerate(self.processors.items()):
            out, extra = _split(proc(input_signals, **processors_kwargs[name]))
            if extra is not None:
                intermediates[name] = extra
            out = out * weights[..., i, None, None]
            total = out if total is None else total + out
        return total, intermediates

    def parameter_size(self):
        size = {k: v.parameter_size() for k, v in self.processors.items()}
        size["parallel_weights"] = len(self.processors)
        return size


class GainStagingRegularization(nn.Module):
    """Adds the input/output log-RMS difference to the intermediates (231-299)."""

    def __init__(self, processor, key="gain_reg"):
        super().__init__()
        self.processor = processor
        self.key = key

    def forward(self, input_signals, **processor_kwargs):
        out, extra = _split(self.processor(input_signals, **processor_kwargs))
        extra = {} if extra is None else extra
        assert self.key not in extra
        extra[self.key] = rms_difference(input_signals, out)
        return out, extra

    def parameter_size(self):
        return self.processor.parameter_size()
