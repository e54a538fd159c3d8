"""Public function surface — the names of QuantTorch/functions/__init__.py for the four
hot-path families (BinaryNet, TernaryNet, DoReFa-Net, XNOR-Net)."""
from .binary_connect import (BinaryConnectDeterministic, BinaryConnectStochastic, BinaryConnect,
                             BinaryDense, BinaryConv2d, AP2, ShiftBatch)
from .dorefa_connect import nnDorefaQuant, DorefaQuant, nnQuantWeight, QuantDense, QuantConv2d
from .terner_connect import (TernaryConnectDeterministic, TernaryConnectStochastic, TernaryConnect,
                             TernaryDense, TernaryConv2d)
from .xnor_connect import nnQuantXnor, QuantXnor, XNORDense, XNORConv2d
from .log_lin_connect import LogQuant, LinQuant, nnQuant, Quant
from .common import safeSign
