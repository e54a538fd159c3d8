"""XNOR-Net layers (reference: QuantTorch/layers/xnor_layers.py).

Upstream's eval swap crashes (free variable ``dim`` at xnor_layers.py:30,61) and XNORConv2d drops
its ``dim`` / forces quant_input=False (:47-49).  Numerics are reproduced; the crashes are fixed by
using the stored ``self.dim`` (documented in DESIGN.md)."""
import torch

from ..functions import xnor_connect, _fused
from .. import lazy
from .common import QLayer, EvalSwapMixin


class LinearXNOR(EvalSwapMixin, torch.nn.Linear, QLayer):
    """y = x . (sign(W) * mean|W|)^T + b, scale per INPUT feature (global DIM=0 upstream)."""

    @staticmethod
    def convert(other, dim=[0, 1]):
        if not isinstance(other, torch.nn.Linear):
            raise TypeError("Expected a torch.nn.Linear ! Receive:  {}".format(other.__class__))
        return LinearXNOR(other.in_features, other.out_features, other.bias is not None, dim=dim)

    def __init__(self, in_features, out_features, bias=True, dim=[0, 1]):
        super().__init__(in_features, out_features, bias=bias)
        self.lin_op = xnor_connect.XNORDense(dim=dim)
        self.dim = dim

    def _quantized_weight_for_eval(self):
        # fixed form of xnor_layers.py:30; the forward op reduces over xnor_connect.DIM whatever
        # ``dim`` says, so the eval image uses the same reduction to stay consistent with it
        return xnor_connect.xnor_weight(self.weight, xnor_connect.DIM)[0]

    def _weight_on_grid(self, w):
        # sign(W) * alpha[1, K]: per input feature every entry is 0 or +-(the column's largest magnitude)
        a = w.abs()
        return ((a == 0) | (a == a.amax(0, keepdim=True))).all()

    def forward(self, input):
        return lazy.linear_forward(self, input, "xnor")

    def _forward_impl(self, input):
        if isinstance(input, _fused.packed.PackedActivation):
            return _fused.packed_xnor_linear(self, input, hwc=input.hwc)
        if (not self.training and not torch.is_grad_enabled() and isinstance(input, torch.Tensor) and input.is_cuda
                and input.dtype == torch.float32 and input.dim() == 2 and input.numel() > 0):
            # eval mode, +-1 activation carrying its sign planes (BinaryConnect's tag): the same operands and the same GEMM as
            # for a packed activation (cached per weight version) — the module-by-module and the deferred execution agree bit for bit
            planes = _fused.packed.lookup(input, _fused.packed.ROWS_LAST)
            if planes is not None and planes.K == self.in_features and planes.rows == input.shape[0] and self._eval_on_grid():
                return _fused.packed_xnor_linear(self, _fused.packed.PackedActivation(planes, tuple(input.shape)))
        return self.lin_op.apply(input, self.weight, self.bias)


class XNORConv2d(EvalSwapMixin, torch.nn.Conv2d, QLayer):
    """conv2d(x, sign(W) * mean(|W|, dim)) (xnor_layers.py:36-69)."""

    @staticmethod
    def convert(other, dim=[0, 1], quant_input=False):
        if not isinstance(other, torch.nn.Conv2d):
            raise TypeError("Expected a torch.nn.Conv2d ! Receive:  {}".format(other.__class__))
        return XNORConv2d(other.in_channels, other.out_channels, other.kernel_size,
                          stride=other.stride, padding=other.padding, dilation=other.dilation,
                          groups=other.groups, bias=other.bias is not None, dim=dim,
                          quant_input=quant_input)

    def __init__(self, *kargs, dim=[0, 1], quant_input=False, **kwargs):
        torch.nn.Conv2d.__init__(self, *kargs, **kwargs)
        self.dim = dim
        # upstream hard-codes quant_input=False here (xnor_layers.py:49); reproduced
        self.conv_op = xnor_connect.XNORConv2d(dim, False, self.stride, self.padding,
                                               self.dilation, self.groups)
        self.binary_input = None        # like BinConv2d: None = detect +-1 activations, True / False = the caller's word

    def _quantized_weight_for_eval(self):
        return xnor_connect.xnor_weight(self.weight, self.dim)[0]

    def clamp(self):
        pass

    def _weight_on_grid(self, w):
        # sign(W) * alpha[1, 1, kh, kw]: per tap every entry is 0 or +-(the tap's largest magnitude)
        a = w.abs()
        top = a.amax((0, 1), keepdim=True)
        return ((a == 0) | (a == top)).all()

    def _taps_planes(self):
        """Cached operands of the eval-mode weight for the per-tap scaled conv: (nibble planes of sign(W), TapScales)."""
        ops = _fused.ops
        cw = ops.pixel_ld_nib_taps(self.in_channels)
        return self._eval_planes(lambda _w2: (ops.pack_conv_weight_nib(self.weight.detach(), "sign", cw=cw),
                                              ops.xnor_tap_prep(self.weight)), key="conv_taps")

    @property
    def _qt_can_defer(self) -> bool:
        """The fused / deferred inference chain runs the per-tap scaled conv: one alpha per filter tap (dim = [0, 1])."""
        d = self.dim
        return not isinstance(d, int) and sorted(int(v) for v in d) == [0, 1]

    def forward(self, input):
        return lazy.conv_forward(self, input, "xnor")

    def _forward_impl(self, input):
        if isinstance(input, _fused.packed.PackedActivation):
            return _fused.packed_xnor_conv2d(self, input)
        if (not self.training and not torch.is_grad_enabled()
                and _fused.xnor_conv_fast_applicable(input, self.weight, self.dim, self.groups, self.padding)
                and self.padding_mode == "zeros"):
            # eval mode: the weight holds sign(W) * alpha and the op quantises it again like upstream (a fixed point: the
            # scales of the image are its own magnitudes); the packed operands are cached per weight version
            known = True if _fused.packed.lookup(input, _fused.packed.NHWC) is not None else None
            out = _fused.xnor_conv2d_forward(input, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                             binary_input=known, planes=self._taps_planes())
            if out is not None:
                ops = _fused.ops
                N_, _, H, W = input.shape
                Ho, Wo = ops.conv_out_hw(H, W, self.kernel_size[0], self.kernel_size[1], self.stride, self.padding, self.dilation)
                y = out[0].view(N_, Ho, Wo, self.out_channels).permute(0, 3, 1, 2)
                if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                    y = y.contiguous()
                return y
        return self.conv_op.apply(input, self.weight, self.bias)
