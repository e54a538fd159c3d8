"""Family alias module: everything of the BinaryNet family under one name (reference: QuantTorch/BinaryNet.py:1-2)."""
from .functions.binary_connect import *  # noqa: F401,F403
from .layers.binary_layers import *  # noqa: F401,F403
from .device import device  # noqa: F401  (the reference's family modules re-export it)
