"""Family alias module: everything of the TernerNet family under one name (reference: QuantTorch/TernerNet.py:1-2)."""
from .functions.terner_connect import *  # noqa: F401,F403
from .layers.terner_layers import *  # noqa: F401,F403
from .device import device  # noqa: F401  (the reference's family modules re-export it)
