"""Linear / Conv2d with Lin or Log fixed-point weights (reference: QuantTorch/layers/log_lin_layers.py).

The quantised weight levels — multiples of 2^(fsr - bit_width) up to 2^fsr (Lin, bit_width <= 8) or signed powers
of two (Log) — are exactly representable in bf16, so on a HIP device the contraction of a real-valued activation
with them runs on the bf16 matrix cores through the exact bf16-triple split (same route as the first layer of a
binary net): fp32-GEMM accuracy, and for Log weights this IS the shift-add GEMM the paper motivates.
"""
import torch

from .. import lazy
from ..functions import _fused, log_lin_connect
from .common import EvalSwapMixin, QLayer


def _exact_in_bf16(dtype, bit_width):
    return dtype == "log" or (dtype == "lin" and bit_width <= 8)


class _WeightInit:
    def reset_parameters(self):
        """uniform magnitude in [2^(fsr - bit_width), 2^fsr] with a random sign, bias 0 (log_lin_layers.py:34-38)."""
        if getattr(self, "bit_width", None) is None:     # nn.Linear/Conv2d.__init__ calls this before our fields exist
            return super().reset_parameters()
        torch.nn.init.uniform_(self.weight, 2 ** (self.fsr - self.bit_width), 2 ** self.fsr)
        self.weight.data.mul_((torch.rand_like(self.weight) < 0.5).type(self.weight.dtype) * 2 - 1)
        if self.bias is not None:
            self.bias.data.zero_()

    def clamp(self):
        self.weight.data.clamp_(-1 * 2 ** self.fsr, 2 ** self.fsr)

    def _quantized_weight_for_eval(self):
        return self.weight_op.forward(self.weight)


class LinearQuant(_WeightInit, EvalSwapMixin, torch.nn.Linear, QLayer):
    """log_lin_layers.py:6-43.  Like upstream, forward re-applies the (idempotent) weight quantiser in eval mode too."""

    @staticmethod
    def convert(other, dtype="lin", fsr=7, bit_width=3):
        if not isinstance(other, torch.nn.Linear):
            raise TypeError("Expected a torch.nn.Linear ! Receive:  {}".format(other.__class__))
        return LinearQuant(other.in_features, other.out_features, other.bias is not None, dtype=dtype, fsr=fsr,
                           bit_width=bit_width)

    def __init__(self, in_features, out_features, bias=True, dtype="lin", fsr=7, bit_width=3):
        self.bit_width, self.fsr, self.qdtype = bit_width, fsr, dtype
        torch.nn.Linear.__init__(self, in_features, out_features, bias=bias)
        self.weight_op = log_lin_connect.nnQuant(dtype=dtype, fsr=fsr, bit_width=bit_width, with_sign=True, lin_back=True)

    def forward(self, input):
        lazy.note_inference_call(self, input)
        input = lazy.resolve(input)
        wq = self.weight_op.forward(self.weight)
        if (input.is_cuda and input.dtype == torch.float32 and input.numel() > 0 and _exact_in_bf16(self.qdtype, self.bit_width)
                and not (torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad))):
            # Lin / Log levels are exact in bf16 (fp32's exponent range), not necessarily in fp16: the exact three-term route
            wt = None if self.training else self._eval_planes(
                lambda w2: _fused.ops.weight_bf16x3(w2, "raw", terms=3), key="bf16x3_raw")
            return _fused.ops.float_linear(input, wq.detach(), "raw", self.bias, weight_triples=wt, terms=3)
        _fused.note_library_path(input, "Lin/Log linear: autograd, a non-fp32 dtype or levels beyond bf16")
        return torch.nn.functional.linear(input, wq, self.bias)


class QuantConv2d(_WeightInit, EvalSwapMixin, torch.nn.Conv2d, QLayer):
    """log_lin_layers.py:46-101."""

    @staticmethod
    def convert(other, fsr=7, bit_width=3, dtype="lin"):
        if not isinstance(other, torch.nn.Conv2d):
            raise TypeError("Expected a torch.nn.Conv2d ! Receive:  {}".format(other.__class__))
        return QuantConv2d(other.in_channels, other.out_channels, other.kernel_size, stride=other.stride,
                           padding=other.padding, dilation=other.dilation, groups=other.groups,
                           bias=other.bias is not None, fsr=fsr, bit_width=bit_width, dtype=dtype)

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 fsr=7, bit_width=3, dtype="lin"):
        self.fsr, self.bit_width, self.qdtype = fsr, bit_width, dtype
        torch.nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                 dilation=dilation, groups=groups, bias=bias)
        self.weight_op = log_lin_connect.nnQuant(dtype=dtype, fsr=fsr, bit_width=bit_width, with_sign=True, lin_back=True)

    def forward(self, input):
        lazy.note_inference_call(self, input)
        input = lazy.resolve(input)
        wq = self.weight_op.forward(self.weight) if self.training else self.weight
        if (input.is_cuda and input.dtype == torch.float32 and input.numel() > 0 and input.dim() == 4
                and self.groups == 1 and self.padding_mode == "zeros" and not isinstance(self.padding, str)
                and _exact_in_bf16(self.qdtype, self.bit_width)
                and not (torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad))):
            wt = None if self.training else self._eval_planes(
                lambda _w2: _fused.ops.pack_conv_weight_bf16x3(self.weight.detach(), "raw", terms=3), key="conv_bf16x3_raw")
            N, C, H, W = input.shape
            kh, kw = int(self.weight.shape[2]), int(self.weight.shape[3])
            # levels exact in bf16, not necessarily in fp16: the exact three-term route, named explicitly
            y2 = _fused.ops.float_conv2d(input, wq.detach(), "raw", self.bias, self.stride, self.padding, self.dilation,
                                         weight_triples=wt, terms=3)
            Ho, Wo = _fused.ops.conv_out_hw(H, W, kh, kw, self.stride, self.padding, self.dilation)
            y = y2.view(N, Ho, Wo, self.weight.shape[0]).permute(0, 3, 1, 2)
            if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous()
            return y
        _fused.note_library_path(input, "Lin/Log conv: autograd, groups, a non-fp32 dtype or levels beyond bf16")
        return torch.nn.functional.conv2d(input, wq, self.bias, self.stride, self.padding, self.dilation, self.groups)
