"""Family alias module: everything of the DorefaNet family under one name (reference: QuantTorch/DorefaNet.py:1-2)."""
from .functions.dorefa_connect import *  # noqa: F401,F403
from .layers.dorefa_layers import *  # noqa: F401,F403
from .device import device  # noqa: F401  (the reference's family modules re-export it)
