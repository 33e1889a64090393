"""HIP-backed processors with the reference's nn.Module interface
(forward(*signals, **params) + parameter_size()); see SURVEY.md §8b."""
from . import core
from .core.convolution import FIRConvolution, convolve, set_exact_convolution
from .core.envelope import Ballistics, TruncatedOnePoleIIRFilter
from .core.iir import IIRFilter
from .container import DryWet, GainStagingRegularization, ParallelMix, SerialChain
from .delay import MultitapDelay
from .dynamics import (
    ApproxCompressor,
    ApproxNoiseGate,
    BallisticsEnvelopeFollower,
    BaseEnvelopeFollower,
    Compressor,
    IIREnvelopeFollower,
    NoiseGate,
)
from .eq import GraphicEqualizer, NewZeroPhaseFIREqualizer, ParametricEqualizer, ZeroPhaseFIREqualizer
from .filter import (
    AllPassFilter,
    BandPassFilter,
    BandRejectFilter,
    BiquadFilter,
    FIRFilter,
    PoleZeroFilter,
    HighPassFilter,
    HighShelf,
    LowPassFilter,
    LowShelf,
    PeakingFilter,
    StateVariableFilter,
)
from .nonlinear import ChebyshevDistortion, PiecewiseTanhDistortion, PowerDistortion, TanhDistortion
from .reverb import FilteredNoiseShapingReverb, STFTMaskedNoiseReverb
from .stereo import MidSideToStereo, MonoToStereo, SideGainImager, StereoGain, StereoToMidSide
