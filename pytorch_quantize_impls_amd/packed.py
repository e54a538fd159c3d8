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
    # tensors created under torch.inference_mode() track no version counter, so a tag on them could not be
    # invalidated: leave them un-tagged (the next layer then checks / packs the fp32 values itself)
    if not t.is_inference():
        setattr(t, _ATTR, (planes, layout, t._version, tuple(t.shape)))
    return t


def lookup(t: torch.Tensor, layout: str) -> Optional[BitPlanes]:
    tag = getattr(t, _ATTR, None)
    if tag is None or t.is_inference():
        return None
    planes, lay, version, shape = tag
    if lay != layout or version != t._version or shape != tuple(t.shape):
        return None
    return planes


def attach_codes(t: torch.Tensor, codes, layout: str) -> torch.Tensor:
    """Same side channel for DoReFa activation codes (ops.CodePlanes) produced by nnDorefaQuant."""
    if not t.is_inference():
        setattr(t, _ATTR_CODES, (codes, layout, t._version, tuple(t.shape)))
    return t


def lookup_codes(t: torch.Tensor, layout: str):
    tag = getattr(t, _ATTR_CODES, None)
    if tag is None or t.is_inference():
        return None
    codes, lay, version, shape = tag
    if lay != layout or version != t._version or shape != tuple(t.shape):
        return None
    return codes


class PackedActivation:
    """A +-1 activation that exists ONLY as bit planes (no fp32 image): what the fused inference
    epilogue (layers.fused.FusedPoolBnSign) hands to the next binarised layer.

    ``planes``: ops.BitPlanes; ``shape``: the logical shape of the +-1 tensor it stands for,
    (N, C, H, W) with NHWC planes (rows = N*H*W, K = C) or (N, K) with row planes (rows = N)."""
    is_cuda = True
    dtype = torch.float32
    requires_grad = False

    def __init__(self, planes: Optional[BitPlanes], shape, nib=None, halo=(0, 0)):
        # Between two fused binarised convs the producer may hand over the consumer's operand itself instead of bit
        # planes: ``nib`` = ops.NibPlanes of the (N, C, H, W) activation as an fp4 nibble pixel plane with a zero border
        # of ``halo`` pixels (= the consuming conv's padding); ``planes`` is then None.
        if (planes is None) == (nib is None):
            raise ValueError("a PackedActivation holds either bit planes or a nibble pixel plane")
        self.planes = planes
        self.nib = nib
        self.halo = tuple(int(v) for v in halo)
        self.shape = tuple(int(v) for v in shape)
        self.hwc = None               # (C, H, W) when the rows are a feature map flattened in (h, w, c) order (flatten_hwc)

    @property
    def device(self):
        return self.planes.device if self.planes is not None else self.nib.device

    def dim(self):
        return len(self.shape)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def flatten_hwc(self) -> "PackedActivation":
        """(N, C, H, W) NHWC planes -> (N, H*W*C) row planes in (h, w, c) order.  Needs C % 32 == 0 and
        an unpadded pixel stride so the words of one image are contiguous; the consumer's weight
        columns must be permuted to the same order (layers.fused.permute_fc_weight_hwc)."""
        if self.planes is None:
            raise ValueError("this activation was produced as a conv operand (nibble plane): no bit planes to flatten")
        N, C, H, W = self.shape
        if self.planes.ld * 32 != C:
            raise ValueError("flatten_hwc needs C to be a multiple of 128 (unpadded 16-byte pixel rows)")
        words = self.planes.sign.view(N, H * W * self.planes.ld)
        flat = PackedActivation(BitPlanes(sign=words, rows=N, K=H * W * C), (N, H * W * C))
        flat.hwc = (C, H, W)          # the feature order of the rows, for consumers that count features in NCHW order
        return flat


class CodeActivation:
    """A k-bit DoReFa activation that exists ONLY as an int8 code plane (value = inv_n * code, no fp32 image):
    what layers.fused.FusedBnDorefaQuant hands to the next DorefaConv2d / LinearDorefa in eval mode.

    ``codes``: ops.CodePlanes; ``shape``: logical shape, (N, C, H, W) with NHWC planes (rows = N*H*W, K = C) or
    (N, K).  The reference does not clamp the quantiser, so a code may not fit int8: the kernels then raise the
    shared device flag ``codes.overflow``; ``float()`` turns a raised flag into an all-NaN result on the device (no
    host sync) and ``check()`` / ``float(check=True)`` into an error — there is no fp32 image to fall back to."""
    is_cuda = True
    dtype = torch.float32
    requires_grad = False

    def __init__(self, codes, shape, halo=(0, 0)):
        self.codes = codes
        self.shape = tuple(int(v) for v in shape)
        # (hy, hx): the NHWC plane carries a zero border of that many pixels around every image,
        # rows = N*(H + 2hy)*(W + 2hx): the next conv's zero padding is then physical (un-padded kernels)
        self.halo = tuple(int(v) for v in halo)
        if any(self.halo) and len(self.shape) != 4:
            raise ValueError("only (N, C, H, W) activations carry a halo")
        if len(self.shape) == 4:
            N, C, H, W = self.shape
            want = N * (H + 2 * self.halo[0]) * (W + 2 * self.halo[1])
            if codes.rows != want:
                raise ValueError(f"code plane holds {codes.rows} pixels, shape {self.shape} with halo {self.halo} needs {want}")

    @property
    def device(self):
        return self.codes.device

    def dim(self):
        return len(self.shape)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def without_halo(self) -> "CodeActivation":
        """The same activation as a plain [N*H*W, ld] plane (one copy; identity without a halo)."""
        if not any(self.halo):
            return self
        from . import ops
        N, C, H, W = self.shape
        hy, hx = self.halo
        inner = self.codes.codes.view(N, H + 2 * hy, W + 2 * hx, -1)[:, hy:hy + H, hx:hx + W].contiguous()
        planes = ops.CodePlanes(codes=inner.view(N * H * W, -1), rows=N * H * W, K=self.codes.K, inv_n=self.codes.inv_n,
                                bit_width=self.codes.bit_width, overflow=self.codes.overflow)
        return CodeActivation(planes, self.shape)

    def flatten_hwc(self) -> "CodeActivation":
        """(N, C, H, W) -> (N, H*W*C) row codes in (h, w, c) order for a LinearDorefa whose weight columns were permuted
        to that order (layers.fused.permute_fc_weight_hwc).  A view when the pixel rows are unpadded (C % 16 == 0) and a
        row is a whole number of 128-byte GEMM stages; otherwise one gather copy into a padded plane."""
        from . import ops
        act = self.without_halo()
        N, C, H, W = act.shape
        K = H * W * C
        ld = int(act.codes.codes.shape[1])
        if ld == C and K % 128 == 0:
            rows = act.codes.codes.view(N, K)
        else:
            rows = torch.zeros((N, ops.code_ld_bytes(K)), dtype=torch.int8, device=act.device)
            rows[:, :K] = act.codes.codes.view(N, H * W, ld)[:, :, :C].reshape(N, K)
        planes = ops.CodePlanes(codes=rows, rows=N, K=K, inv_n=act.codes.inv_n, bit_width=act.codes.bit_width,
                                overflow=act.codes.overflow)
        return CodeActivation(planes, (N, K))

    def check(self):
        from . import ops
        if not ops._cfg("ASSUME_CODES_FIT") and self.codes.overflow is not None and int(self.codes.overflow.item()) != 0:
            raise RuntimeError("a DoReFa activation code exceeded int8 (|q| > 127 or NaN) inside the fused code-plane "
                               "path: run this model module by module (the fp32 route keeps unclamped activations)")
        return self

    def float(self, check: bool = False) -> torch.Tensor:
        """The fp32 image nnDorefaQuant would have returned, fl(fl(1/n) * q); (N, C, H, W) comes back channels_last.

        The int8 range flag of the chain is applied ON THE DEVICE: if any code of the chain left int8 the whole
        result is NaN (no host sync, so the host keeps enqueueing the next forward; 0.89 -> 0.77 ms on the fused
        ResNet-18).  ``check=True`` (or ``check()``) synchronises and raises instead."""
        if check:
            self.check()
        from . import ops
        flagged = not ops._cfg("ASSUME_CODES_FIT")
        if len(self.shape) == 4:
            N, _, H, W = self.shape
            return ops.codes_to_f32(self.codes, N, H, W, self.halo, 1, flagged).permute(0, 3, 1, 2)
        return ops.codes_to_f32(self.codes, self.shape[0], 1, 1, (0, 0), 1, flagged).view(self.shape)

    def avg_pool2d(self, kernel_size: int, check: bool = False) -> torch.Tensor:
        """F.avg_pool2d(self.float(), kernel_size) (kernel = stride, no padding, floor mode) in the same pass over the codes: the
        head of models/samples/ResNet_Dorefa.py (avg_pool2d(out, 4) -> Linear).  Bit-identical to pooling the fp32 image with ATen."""
        if len(self.shape) != 4:
            raise ValueError("avg_pool2d takes an (N, C, H, W) activation")
        if check:
            self.check()
        from . import ops
        N, _, H, W = self.shape
        return ops.codes_to_f32(self.codes, N, H, W, self.halo, int(kernel_size), not ops._cfg("ASSUME_CODES_FIT")).permute(0, 3, 1, 2)
