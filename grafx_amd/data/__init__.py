from .batch import batch_grafx
from .configs import NodeConfigs, UTILITY_TYPES
from .conversion import convert_to_tensor
from .graph import GRAFX
from .tensor import GRAFXTensor
