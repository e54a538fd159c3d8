"""Family alias module: everything of the XnorNet family under one name (reference: QuantTorch/XnorNet.py:1-2)."""
from .functions.xnor_connect import *  # noqa: F401,F403
from .layers.xnor_layers import *  # noqa: F401,F403
from .device import device  # noqa: F401  (the reference's family modules re-export it)
