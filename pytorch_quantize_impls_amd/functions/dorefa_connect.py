"""DoReFa-Net ops (reference: QuantTorch/functions/dorefa_connect.py)."""
import warnings

import torch

from .. import ops, packed, lazy, lazy_train
from .common import QtFunction, front, safeSign
from . import _fused

warnings.simplefilter("always", DeprecationWarning)


def _quantize(x, bit_width=3):
    """k-bit quantiser: k=1 safeSign, k=32 identity, else (1/(2^k-1)) * round((2^k-1) x) with
    round-half-even and NO clamp (dorefa_connect.py:11-25).  The scale is formed in fp32 and the
    reciprocal is multiplied (not divided) so the last ulp matches the reference."""
    if bit_width == 1:
        return safeSign(x)
    if bit_width == 32:
        return x
    if x.is_cuda and x.dtype == torch.float32:
        if 2 <= bit_width <= 8 and x.dim() >= 2 and x.numel() > 0:
            # same values as qt_dorefa_quantize_f32, plus the int8 codes the next DoReFa layer's
            # int8-MFMA path consumes (attached to the returned tensor, see packed.py)
            if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
                nhwc = x.permute(0, 2, 3, 1)
                cp, y = ops.dorefa_codes(nhwc, bit_width, ld_bytes=ops.code_ld_bytes(x.shape[1], 16))
                return packed.attach_codes(y.permute(0, 3, 1, 2), cp, packed.NHWC)
            if x.is_contiguous():
                cp, y = ops.dorefa_codes(x, bit_width)
                return packed.attach_codes(y, cp, packed.ROWS_LAST)
        return ops.dorefa_quantize(x, bit_width)
    n = torch.pow(torch.full_like(x, 2.0), bit_width) - 1
    return (1 / n) * torch.round(n * x)


def _make_quant_function(bit_width):
    class _Quant(QtFunction):
        _qt_quant_bits = bit_width          # on a deferred DorefaConv2d chain (lazy.py) the quantiser is recorded

        @classmethod
        def apply(cls, input):
            if isinstance(input, lazy.LazyActivation):
                out = lazy.quant(input, bit_width)
                if out is not None:
                    return out
                input = input.value()
            elif type(input) in lazy_train._DEFERRED:
                out = lazy_train.quant(input, bit_width)
                if out is not None:
                    return out
                input = lazy_train.resolve(input)
            return super().apply(input)

        @staticmethod
        def forward(ctx, input):
            return _quantize(input, bit_width=bit_width)

        @staticmethod
        def backward(ctx, grad_ouput):  # identity STE (dorefa_connect.py:43-44, 61-62); upstream's clone() is a copy nobody needs
            return grad_ouput

    return _Quant


def nnDorefaQuant(bit_width=3):
    """nn.Module applying the k-bit quantiser with identity gradient (dorefa_connect.py:28-45)."""
    return front(_make_quant_function(bit_width))


def DorefaQuant(x, bit_width=3):
    """Functional form of the k-bit quantiser (dorefa_connect.py:49-63)."""
    return _make_quant_function(bit_width).apply(x)


class _ignore_factor_op(QtFunction):
    """forward x*c ; backward passes the gradient through UNscaled (dorefa_connect.py:66-79)."""

    @staticmethod
    def forward(ctx, input, const):
        return input * const

    @staticmethod
    def backward(ctx, grad_ouput):
        # identity STE: the gradient itself (a clone was one 67 MB copy per quantiser and step; autograd never writes into it)
        var_grad = grad_ouput if ctx.needs_input_grad[0] else None
        return var_grad, None


class _QuantWeight(torch.nn.Module):
    def __init__(self, bit_width):
        super().__init__()
        self.bit_width = bit_width
        self.quant_op = nnDorefaQuant(bit_width)

    def forward(self, x):
        if self.bit_width == 1:
            # sign(W) * mean|W| with the scalar detached (dorefa_connect.py:99-102)
            E = ops.abs_mean(x)
            return _ignore_factor_op.apply(self.quant_op(x), E)
        if self.bit_width == 32:
            return x
        if torch.max(torch.abs(x)) == 0.0:  # all-zero guard (dorefa_connect.py:106-107)
            return torch.zeros_like(x)
        weight = torch.tanh(x)
        weight = weight / (2 * torch.max(torch.abs(weight))) + 0.5
        return 2 * self.quant_op(weight) - 1


def nnQuantWeight(bit_width=3):
    """Weight quantiser module: 2*quantize_k(tanh W / (2 max|tanh W|) + 1/2) - 1
    (dorefa_connect.py:82-113)."""
    return _QuantWeight(bit_width)


def QuantDense(bit_width=3):
    """DEPRECATED functional dense op with explicit tanh-derivative backward
    (dorefa_connect.py:116-155)."""

    class _QuantDense(QtFunction):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            max_abs = torch.max(torch.abs(torch.tanh(weight)))
            if bit_width == 1:
                weight_q = safeSign(weight) * ops.abs_mean(weight)
            elif bit_width == 32:
                weight_q = weight
            else:
                weight_q = 2 * _quantize(0.5 + torch.tanh(weight) / (2 * max_abs), bit_width=bit_width) - 1
            output = _fused.real_weight_linear(input, weight_q, bias)
            ctx.save_for_backward(input, weight, weight_q, max_abs, bias)
            return output

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, weight_q, max_abs, bias = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            if ctx.needs_input_grad[0]:
                grad_input = _fused.dense_grad_input(grad_output, weight_q, pm1=False)
            if ctx.needs_input_grad[1]:
                x_pm1 = input.dim() == 2 and _fused.packed.lookup(input, _fused.packed.ROWS_LAST) is not None
                grad_weight = _fused.dense_grad_weight(grad_output, input, x_pm1)
                if 1 < bit_width < 32:
                    grad_weight = grad_weight * (1 - torch.pow(torch.tanh(weight), 2)) / max_abs
            if bias is not None and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum(0)
            return grad_input, grad_weight, grad_bias

    return _QuantDense


def QuantConv2d(stride=1, padding=1, dilation=1, groups=1, bit_width=3):
    """DEPRECATED functional conv op; normalises by tanh(max|W|) (dorefa_connect.py:158-199)."""
    warnings.warn("Deprecated conv op ! Use layers.DorefaConv2d.", DeprecationWarning, stacklevel=2)

    class _QuantConv2d(QtFunction):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            max_weight = torch.max(torch.abs(weight))
            if bit_width == 1:
                weight_q = safeSign(weight) * ops.abs_mean(weight)
            elif bit_width == 32:
                weight_q = weight
            else:
                weight_q = 2 * _quantize(0.5 + torch.tanh(weight) / (2 * torch.tanh(max_weight)),
                                         bit_width=bit_width) - 1
            ctx.save_for_backward(input, weight, weight_q, max_weight, bias)
            return _fused.real_weight_conv2d(input, weight_q, bias, stride, padding, dilation, groups)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, weight_q, max_weight, bias = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            if ctx.needs_input_grad[0]:
                if 1 < bit_width <= _fused.LEVEL_MAX_BITS:      # odd integer levels / (2^k - 1): the levels are the exact operand, 1 / n after
                    n_w = float((1 << bit_width) - 1)
                    grad_input = _fused.conv_grad_input(input.size(), torch.round(weight_q.detach() * n_w), grad_output, stride,
                                                        padding, dilation, groups, kind="raw",
                                                        out_scale=_fused._inv_levels(bit_width))
                elif bit_width == 1:        # sign(W) * E
                    grad_input = _fused.conv_grad_input(input.size(), weight, grad_output, stride, padding, dilation, groups,
                                                        kind="binary", out_scale_dev=ops.abs_mean(weight))
                else:
                    grad_input = _fused.conv_grad_input(input.size(), weight_q, grad_output, stride, padding, dilation, groups,
                                                        kind=None)
            if ctx.needs_input_grad[1]:
                x_pm1 = input.dim() == 4 and _fused.packed.lookup(input, _fused.packed.NHWC) is not None
                grad_weight = _fused.conv_grad_weight(input, weight.shape, grad_output, stride, padding, dilation, groups, x_pm1)
                if 1 < bit_width < 32:
                    grad_weight = grad_weight * (1 - torch.pow(torch.tanh(weight), 2)) / torch.tanh(max_weight)
            if bias is not None and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum((0, 2, 3))
            if bias is not None:
                return grad_input, grad_weight, grad_bias
            return grad_input, grad_weight

    return _QuantConv2d
