"""BinaryConnect / BinaryNet ops (reference: QuantTorch/functions/binary_connect.py).

Same public names and call conventions as the reference; tensors on a HIP device dispatch to
libqt_hip.so, CPU tensors take the equivalent torch expression.
"""
import warnings as _warnings

import torch

from .. import ops, packed, lazy, lazy_train
from .common import QtFunction, front, safeSign, ste_mask
from . import _fused

_warnings.simplefilter("always", DeprecationWarning)


def _binarize_and_tag(input: torch.Tensor) -> torch.Tensor:
    """safeSign(input); on the GPU the sign plane is produced in the same pass and attached."""
    if input.is_cuda and input.dtype == torch.float32 and input.dim() >= 2 and input.numel() > 0:
        if input.dim() == 4 and input.is_contiguous(memory_format=torch.channels_last) \
                and not input.is_contiguous():
            # NHWC storage: pack the channel dimension (what a binarised conv consumes)
            nhwc = input.permute(0, 2, 3, 1)
            planes, y = ops.sign_pack(nhwc, want_f32=True)
            y = y.permute(0, 3, 1, 2)
            return packed.attach(y, planes, packed.NHWC)
        if input.is_contiguous():
            planes, y = ops.sign_pack(input, want_f32=True)
            return packed.attach(y, planes, packed.ROWS_LAST)
    return safeSign(input)


class BinaryConnectDeterministic(QtFunction):
    """r_b = sign(r) (0 -> +1); d r_b / d r = 1_{|r| <= 1}  (binary_connect.py:14-38)."""
    _qt_records_sign = True      # on a deferred conv chain (lazy.py) the op is recorded, not executed

    @classmethod
    def apply(cls, input):
        if isinstance(input, lazy.LazyActivation):
            out = lazy.sign(input)
            if out is not None:
                return out
            input = input.value()
        elif type(input) in lazy_train._DEFERRED:
            out = lazy_train.sign(input)
            if out is not None:
                return out
            input = lazy_train.resolve(input)
        return super().apply(input)

    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        return _binarize_and_tag(input)

    @staticmethod
    def backward(ctx, grad_output):
        input, = ctx.saved_tensors
        return ste_mask(grad_output, input)


class BinaryConnectStochastic(QtFunction):
    """r_b = +1 with probability hardsigmoid(r), else -1; same STE backward
    (binary_connect.py:42-71).  The uniforms come from torch.rand_like, as in the reference."""

    @staticmethod
    def forward(ctx, input):
        ctx.save_for_backward(input)
        z = torch.rand_like(input, requires_grad=False)
        return stochastic_binarize(input, z)

    @staticmethod
    def backward(ctx, grad_output):
        input, = ctx.saved_tensors
        return ste_mask(grad_output, input)


def stochastic_binarize(input: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """-1 + 2*[z < (clamp(x,-1,1)+1)/2] for explicit uniforms z (binary_connect.py:57-61)."""
    if input.is_cuda and input.dtype == torch.float32:
        return ops.binarize_stochastic(input, z)
    p = (torch.clamp(input, -1, 1) + 1) / 2
    return -1.0 + 2.0 * (z < p).to(input.dtype)


def BinaryConnect(stochastic=False):
    """nn.Module wrapping the binarisation op, for nn.Sequential (binary_connect.py:74-83)."""
    return front(BinaryConnectStochastic if stochastic else BinaryConnectDeterministic)


class BinaryDense(QtFunction):
    """y = x . sign(W)^T + b with a plain (un-masked) backward (binary_connect.py:86-112).  Device fp32 tensors: both backward
    contractions on this backend's matrix-core routes (_fused.dense_grad_input / dense_grad_weight)."""

    @staticmethod
    def forward(ctx, input, weight, bias=None):
        ctx.save_for_backward(input, weight, bias)
        _fused.clear_last_detection()
        out = _fused.quant_linear_forward(input, weight, bias, kind="binary")
        ctx.x_is_pm1 = bool(input.is_cuda and input.dim() == 2 and (_fused.packed.lookup(input, _fused.packed.ROWS_LAST) is not None
                                                                    or _fused.last_detection_said_pm1(input)))
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input, weight, bias = ctx.saved_tensors
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            grad_input = _fused.dense_grad_input(grad_output, safeSign(weight), pm1=True)
        if ctx.needs_input_grad[1]:
            grad_weight = _fused.dense_grad_weight(grad_output, input, ctx.x_is_pm1)
        if bias is not None and ctx.needs_input_grad[2]:
            grad_bias = grad_output.sum(0)
        return grad_input, grad_weight, grad_bias


def BinaryConv2d(stride=1, padding=1, dilation=1, groups=1):
    """DEPRECATED functional conv with binarised weight (binary_connect.py:116-153); backward on the matrix-core conv routes."""
    _warnings.warn("Deprecated conv op ! Use layers.BinConv2d.", DeprecationWarning, stacklevel=2)

    class _BinaryConv2d(QtFunction):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            ctx.save_for_backward(input, weight, bias)
            _fused.clear_last_detection()
            out = _fused.quant_conv2d_forward(input, weight, bias, stride, padding, dilation,
                                              groups, kind="binary")
            ctx.x_is_pm1 = bool(input.is_cuda and input.dim() == 4 and (_fused.packed.lookup(input, _fused.packed.NHWC) is not None
                                                                        or _fused.last_detection_said_pm1(input)))
            return out

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, bias = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            if ctx.needs_input_grad[0]:
                grad_input = _fused.conv_grad_input(input.size(), weight, grad_output, stride, padding, dilation, groups,
                                                    kind="binary")
            if ctx.needs_input_grad[1]:
                grad_weight = _fused.conv_grad_weight(input, weight.shape, grad_output, stride, padding, dilation, groups,
                                                      ctx.x_is_pm1)
            if bias is not None and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum((0, 2, 3))
            if bias is not None:
                return grad_input, grad_weight, grad_bias
            return grad_input, grad_weight

    return _BinaryConv2d


def AP2(x):
    """safeSign(x) * 2^round(log2|x|) (binary_connect.py:157-169); HIP kernel for device fp32 tensors."""
    if x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and x.requires_grad):
        return ops.ap2(x)
    return safeSign(x) * torch.pow(torch.full_like(x, 2.0), torch.round(torch.log2(torch.abs(x))))


class ShiftBatch(QtFunction):
    """Shift-based batch-norm primitive (binary_connect.py:173-214).  Device fp32 tensors whose statistics / affine
    tensors broadcast over the leading dimension (how ShiftNormBatch1d / 2d call it): one HIP kernel
    (qt_shift_batch_f32); anything else: the torch expression."""

    @staticmethod
    def forward(ctx, input, running_mean, running_var, weight, bias, eps):
        per = input[0].numel() if input.dim() > 0 and input.shape[0] > 0 else -1
        if (input.is_cuda and input.dtype == torch.float32 and input.numel() > 0
                and all(t.is_cuda and t.dtype == torch.float32 and t.numel() == per and tuple(t.shape) == tuple(input.shape[1:])[-t.dim():]
                        for t in (running_mean, running_var, weight, bias))):
            out, norm_inputs, sv = ops.shift_batch(input, running_mean, running_var, weight, bias, float(eps))
            ctx.save_for_backward(input, weight, sv.view(running_var.shape), norm_inputs)
            return out
        centred = input - running_mean
        sqrtvar = torch.sqrt(running_var + eps)
        norm_inputs = centred * AP2(1 / sqrtvar)
        out = norm_inputs * AP2(weight) + bias
        ctx.save_for_backward(input, weight, sqrtvar, norm_inputs)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input, weight, sqrtvar, norm_inputs = ctx.saved_tensors
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            grad_input = grad_output * weight / sqrtvar
        if ctx.needs_input_grad[3]:
            grad_weight = grad_output * norm_inputs
        if ctx.needs_input_grad[4]:
            grad_bias = grad_output.sum(0)
        return grad_input, None, None, grad_weight, grad_bias, None
