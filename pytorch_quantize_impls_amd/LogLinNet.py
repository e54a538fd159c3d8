"""Family alias module: everything of the LogLinNet family under one name (reference: QuantTorch/LogLinNet.py:1-2)."""
from .functions.log_lin_connect import *  # noqa: F401,F403
from .layers.log_lin_layers import *  # noqa: F401,F403
from .device import device  # noqa: F401  (the reference's family modules re-export it)
