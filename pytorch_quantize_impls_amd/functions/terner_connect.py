"""Ternary connect ops (reference: QuantTorch/functions/terner_connect.py)."""
import warnings

import torch

from .. import ops
from .common import QtFunction, front, safeSign, ste_mask
from . import _fused

warnings.simplefilter("always", DeprecationWarning)


class TernaryConnectDeterministic(QtFunction):
    """x_t = +1 if x >= 0.5 ; -1 if x < -0.5 ; else 0   (written in the reference as
    (s + safeSign(x - 0.5 s))/2, terner_connect.py:24-27); d x_t/d x = 1_{|x| <= 1}."""

    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        return _fused.quantize_weight_f32(input, "ternary")

    @staticmethod
    def backward(ctx, grad_output):
        input, = ctx.saved_tensors
        return ste_mask(grad_output, input)


def stochastic_ternarize(input: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """s - s*[z > |x|] for explicit uniforms z (terner_connect.py:54-56)."""
    if input.is_cuda and input.dtype == torch.float32:
        return ops.ternarize_stochastic(input, z)
    s = safeSign(input)
    return s - s * (z > torch.abs(input)).to(input.dtype)


class TernaryConnectStochastic(QtFunction):
    """x_t = sign(x) with probability |x|, else 0 (terner_connect.py:37-63)."""

    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        z = torch.rand_like(input, requires_grad=False)
        return stochastic_ternarize(input, z)

    @staticmethod
    def backward(ctx, grad_output):
        input, = ctx.saved_tensors
        return ste_mask(grad_output, input)


def TernaryConnect(stochastic=False):
    """nn.Module wrapping the ternary op (terner_connect.py:67-75)."""
    return front(TernaryConnectStochastic if stochastic else TernaryConnectDeterministic)


def _functional_ternary_weight(weight, stochastic):
    # The functional forms use torch.sign (0 -> 0), not safeSign (terner_connect.py:85-90).
    # Deterministic branch restated as-is (w = +-0.5 gives +-0.5).  The stochastic branch of the
    # reference, sign - sign(z - |w|), yields {0, +-2} (upstream bug); it is
    # reproduced literally because it is observable behaviour.
    sign = torch.sign(weight)
    if stochastic:
        z = torch.rand_like(weight, requires_grad=False)
        return sign - torch.sign(z - torch.abs(weight))
    return (sign + torch.sign(weight - 0.5 * sign)) / 2


def TernaryDense(stochastic=False):
    """Functional ternary linear op (terner_connect.py:78-108).  The image is real-valued (w = +-0.5 stays +-0.5, the stochastic
    branch gives {0, +-2}): forward and both backward contractions on the six-term real x real route for device tensors
    (a +-1 activation is the exact operand of the weight gradient)."""

    class _TernaryDense(QtFunction):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            weight_t = _functional_ternary_weight(weight, stochastic)
            ctx.save_for_backward(input, weight, weight_t, bias)
            ctx.x_is_pm1 = input.dim() == 2 and _fused.known_pm1(input, weight, _fused.packed.ROWS_LAST)
            return _fused.real_weight_linear(input, weight_t, bias)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, weight_t, bias = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            if ctx.needs_input_grad[0]:
                grad_input = _fused.dense_grad_input(grad_output, weight_t, pm1=False)
            if ctx.needs_input_grad[1]:
                grad_weight = _fused.dense_grad_weight(grad_output, input, ctx.x_is_pm1)
            if bias is not None and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum(0)
            return grad_input, grad_weight, grad_bias

    return _TernaryDense


def TernaryConv2d(stochastic=True, stride=1, padding=1, dilation=1, groups=1):
    """DEPRECATED functional ternary conv (terner_connect.py:113-153).  Backward: the image's entries are multiples of 1/2 (exact
    in fp16), so grad_input is the split gradient against the flipped image (kind "raw"); grad_weight on the +-1 routes when the
    activation is +-1."""
    warnings.warn("Deprecated conv op ! Use layers.TerConv2d.", DeprecationWarning, stacklevel=2)

    class _TernaryConv2d(QtFunction):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            weight_t = _functional_ternary_weight(weight, stochastic)
            ctx.save_for_backward(input, weight, weight_t, bias)
            ctx.x_is_pm1 = input.dim() == 4 and _fused.known_pm1(input, weight, _fused.packed.NHWC)
            return _fused.real_weight_conv2d(input, weight_t, bias, stride, padding, dilation, groups)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, weight_t, bias = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            if ctx.needs_input_grad[0]:
                grad_input = _fused.conv_grad_input(input.size(), weight_t, grad_output, stride, padding, dilation, groups,
                                                    kind="raw")
            if ctx.needs_input_grad[1]:
                grad_weight = _fused.conv_grad_weight(input, weight.shape, grad_output, stride, padding, dilation, groups,
                                                      ctx.x_is_pm1)
            if bias is not None and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum((0, 2, 3))
            if bias is not None:
                return grad_input, grad_weight, grad_bias
            return grad_input, grad_weight

    return _TernaryConv2d
