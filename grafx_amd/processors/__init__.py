"""HIP-backed processors with the reference's nn.Module interface
(forward(*signals, **params) + parameter_size()); see SURVEY.md §8b."""
from . import core
from .core.convolution import FIRConvolution, convolve, set_exact_convolution
from .core.envelope import Ballistics, TruncatedOnePoleIIRFilter
from .core.iir import IIRFilter
from .dynamics import Compressor, NoiseGate
from .eq import ParametricEqualizer
from .filter import BiquadFilter
from .reverb import STFTMaskedNoiseReverb
from .stereo import StereoGain
