"""Side-channel that lets a quantiser hand its bit planes to the next layer.

The reference splits "binarise the activation" (BinaryConnect(), functions/binary_connect.py:74-83)
and "binarised layer" (LinearBin, layers/binary_layers.py:42-46) into two modules that talk through
an fp32 +-1 tensor.  On the GPU the quantiser kernel emits the bit planes in the same pass and
parks them on the fp32 tensor it returns; the next layer picks them up and never re-reads the fp32
image.  The tag is only honoured when it is provably still true:

  * it is keyed to the tensor's autograd version counter (any in-place write invalidates it);
  * views / reshapes create new tensor objects and simply lose the tag (the layer then falls back
    to checking + packing the fp32 values itself — slower, never wrong).
"""
from __future__ import annotations

from typing import Optional

import torch

from .ops import BitPlanes

_ATTR = "_qt_planes"
_ATTR_CODES = "_qt_codes"

ROWS_LAST = "rows_last"  # planes pack the LAST dimension of the (logically row-major) tensor
NHWC = "nhwc"            # planes pack the channel dimension of an [N,C,H,W] tensor, rows = N*H*W


def attach(t: torch.Tensor, planes: BitPlanes, layout: str) -> torch.Tensor:
    setattr(t, _ATTR, (planes, layout, t._version, tuple(t.shape)))
    return t


def lookup(t: torch.Tensor, layout: str) -> Optional[BitPlanes]:
    tag = getattr(t, _ATTR, None)
    if tag is None:
        return None
    planes, lay, version, shape = tag
    if lay != layout or version != t._version or shape != tuple(t.shape):
        return None
    return planes


def attach_codes(t: torch.Tensor, codes, layout: str) -> torch.Tensor:
    """Same side channel for DoReFa activation codes (ops.CodePlanes) produced by nnDorefaQuant."""
    setattr(t, _ATTR_CODES, (codes, layout, t._version, tuple(t.shape)))
    return t


def lookup_codes(t: torch.Tensor, layout: str):
    tag = getattr(t, _ATTR_CODES, None)
    if tag is None:
        return None
    codes, lay, version, shape = tag
    if lay != layout or version != t._version or shape != tuple(t.shape):
        return None
    return codes
