"""HIP-backed processors with the reference's nn.Module interface
(forward(*signals, **params) + parameter_size()); see SURVEY.md §8b."""
from . import core
from .core.convolution import FIRConvolution, convolve, set_exact_convolution
from .core.envelope import Ballistics, TruncatedOnePoleIIRFilter
from .core.iir import IIRFilter
from .container import DryWet, GainStagingRegularization, ParallelMix, SerialChain
from .dynamics import ApproxCompressor, ApproxNoiseGate, Compressor, NoiseGate
from .eq import ParametricEqualizer, ZeroPhaseFIREqualizer
from .filter import (
    AllPassFilter,
    BandPassFilter,
    BandRejectFilter,
    BiquadFilter,
    HighPassFilter,
    HighShelf,
    LowPassFilter,
    LowShelf,
    PeakingFilter,
    StateVariableFilter,
)
from .reverb import STFTMaskedNoiseReverb
from .stereo import MidSideToStereo, MonoToStereo, SideGainImager, StereoGain, StereoToMidSide
