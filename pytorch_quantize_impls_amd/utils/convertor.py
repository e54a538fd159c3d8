"""Swap the nn.Linear / nn.Conv2d modules of a float network for a quantised layer family.

Reference: QuantTorch/utils/convertor.py:21-58.  Same contract: the network is deep-copied, every
module whose exact class is a key of the replacement table is rebuilt through the target class's
static ``convert(other, **kwargs)`` (a FRESH layer: upstream does not copy weights either,
layers/binary_layers.py:8-12), everything else is kept.

Deliberate deviations, both upstream defects:
  * ``dorefa_net_convert(net, weight_bit=3)`` upstream forwards ``weight_bit=`` to
    ``LinearDorefa.convert(other, bit_width)`` and dies with a TypeError (utils/convertor.py:47-51);
    here the argument reaches the layer as ``bit_width``.
  * ``log_lin_net_convert(net, fsr, bitwight, dtype)`` upstream forwards ``bitwight=`` to ``convert(other, ..., bit_width)``
    (utils/convertor.py:60-64): TypeError; here it arrives as ``bit_width`` (the misspelt keyword is kept as an alias).
  * ``xnor_net_convert`` upstream forwards ``quant_input`` to ``LinearXNOR.convert`` which does not take
    it (layers/xnor_layers.py:9-13); here only the conv layer receives it.
"""
from copy import deepcopy

import torch
import torch.nn as nn

from ..layers.binary_layers import LinearBin, BinConv2d
from ..layers.terner_layers import LinearTer, TerConv2d
from ..layers.dorefa_layers import LinearDorefa, DorefaConv2d
from ..layers.xnor_layers import LinearXNOR, XNORConv2d
from ..layers.log_lin_layers import LinearQuant, QuantConv2d


def _target(entry):
    if isinstance(entry, tuple):
        return entry[0], dict(entry[1])
    return entry, {}


def _convert_net(module, table):
    hit = table.get(module.__class__)   # exact class match, like upstream (subclasses are left alone)
    if hit is not None:
        cls, kwargs = _target(hit)
        return cls.convert(module, **kwargs)
    for name, child in list(module.named_children()):
        setattr(module, name, _convert_net(child, table))
    return module


def convert(module, replace_dict):
    """``replace_dict``: {source class: target class | (target class, convert kwargs)}."""
    return _convert_net(deepcopy(module), replace_dict)


def binary_net_convert(net, deterministic=True):
    kw = {"deterministic": deterministic}
    return convert(net, {nn.Linear: (LinearBin, kw), nn.Conv2d: (BinConv2d, kw)})


def ternary_net_convert(net, deterministic=True):
    """Same surgery for the ternary family (no upstream counterpart; the layers have ``convert``)."""
    kw = {"deterministic": deterministic}
    return convert(net, {nn.Linear: (LinearTer, kw), nn.Conv2d: (TerConv2d, kw)})


def dorefa_net_convert(net, weight_bit=3):
    kw = {"bit_width": weight_bit}
    return convert(net, {nn.Linear: (LinearDorefa, kw), nn.Conv2d: (DorefaConv2d, kw)})


def xnor_net_convert(net, dim=[0, 1], quant_input=False):
    return convert(net, {nn.Linear: (LinearXNOR, {"dim": dim}),
                         nn.Conv2d: (XNORConv2d, {"dim": dim, "quant_input": quant_input})})


def log_lin_net_convert(net, fsr=7, bit_width=3, dtype="lin", bitwight=None):
    """Lin / Log fixed-point family (utils/convertor.py:60-64)."""
    kw = {"fsr": fsr, "bit_width": bit_width if bitwight is None else bitwight, "dtype": dtype}
    return convert(net, {nn.Linear: (LinearQuant, kw), nn.Conv2d: (QuantConv2d, kw)})
