"""XNOR-Net ops (reference: QuantTorch/functions/xnor_connect.py).

The reference module is unfinished upstream; observable numerics are reproduced, including:
  * nnQuantXnor / QuantXnor use the SIGNED mean (not mean|x|) despite their docstring (:21-28);
  * XNORDense ignores its ``dim`` argument and always reduces over the module-global DIM = 0,
    i.e. one scale per INPUT feature, shape [1, K] (:13, :112, :127);
  * XNORConv2d uses its ``dim`` in forward but the global DIM in backward (:140, :158-159).
"""
import torch

from .. import ops
from .common import front

DIM = 0


def _quantOpXnor(dim=1):
    class _QuantXNOR(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input):
            if input.is_cuda and input.dtype == torch.float32 and input.dim() == 2 and input.numel() > 0:
                out, mean = ops.xnor_act(input.detach(), dim)        # qt_xnor_act_f32: reduction + sign * mean
                ctx.save_for_backward(input, mean if dim >= 0 else mean.view(()))
                return out
            mean = torch.mean(input) if dim < 0 else torch.mean(input, dim)
            ctx.save_for_backward(input, mean)
            if dim < 0:
                return torch.sign(input) * mean
            shape = (1, -1) if dim == 0 else (-1, 1)
            return torch.sign(input) * mean.view(shape)

        @staticmethod
        def backward(ctx, grad_outputs):
            input, mean = ctx.saved_tensors
            if (input.is_cuda and input.dtype == torch.float32 and input.dim() == 2 and input.numel() > 0
                    and grad_outputs.dtype == torch.float32):
                return ops.xnor_act_backward(grad_outputs, input, mean.reshape(-1), dim)   # qt_xnor_act_backward_f32
            sgn = torch.sign(input)
            if dim < 0:
                return sgn * torch.mean(grad_outputs * sgn) + grad_outputs * mean
            shape = (1, -1) if dim == 0 else (-1, 1)
            return sgn * torch.mean(grad_outputs * sgn, dim, keepdim=True) \
                + grad_outputs * mean.view(shape).expand(input.size())

    return _QuantXNOR


def _check_dim(dim):
    if dim not in (-1, 0, 1):
        raise RuntimeError(" Please use a correct dim between -1, 0, 1")


def nnQuantXnor(dim=1):
    """Module form of the XNOR activation op on 2-D inputs (xnor_connect.py:39-52)."""
    _check_dim(dim)
    return front(_quantOpXnor(dim))


def QuantXnor(input, dim=1):
    """Functional form (xnor_connect.py:54-66)."""
    _check_dim(dim)
    return _quantOpXnor(dim).apply(input)


def xnor_weight(weight, dims):
    """sign(W) * mean(|W|, dims, keepdim) — torch.sign, so W == 0 stays 0 (xnor_connect.py:112-113)."""
    lead = None
    if isinstance(dims, int):
        lead = 1 if dims == 0 else None
    elif list(dims) == list(range(len(dims))):
        lead = len(dims)
    if weight.is_cuda and weight.dtype == torch.float32 and lead is not None and weight.numel() > 0:
        return ops.xnor_weight(weight.detach(), lead)   # qt_xnor_weight_f32 (column mean + sign*alpha)
    mean = torch.mean(torch.abs(weight), dims, keepdim=True)
    return torch.sign(weight) * mean, mean


def XNORDense(dim=[0, 1]):
    """XNOR dense op; ``dim`` is accepted and ignored like upstream (xnor_connect.py:93-132)."""

    class _XNORDense(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            if input.is_cuda and input.dtype == torch.float32 and weight.dim() == 2 and input.numel() > 0:
                # alpha on the device (qt_xnor_weight_f32), folded into the activation split; sign(W) on
                # the bf16 matrix cores: fl(x*alpha) * (+-1) is the product fl(x * (+-alpha)) exactly
                _, mean = ops.xnor_weight(weight.detach(), 1)
                ctx.save_for_backward(input, weight, mean, bias)
                return ops.float_linear(input, weight.detach(), "sign", bias, alpha=mean.view(-1))
            weight_q, mean = xnor_weight(weight, DIM)
            ctx.save_for_backward(input, weight, mean, bias)
            return torch.nn.functional.linear(input, weight_q, bias)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, mean, bias = ctx.saved_tensors
            sgn = torch.sign(weight)
            grad_input = grad_weight = grad_bias = None
            if ctx.needs_input_grad[0]:
                grad_input = grad_output.mm(sgn * mean)
            if ctx.needs_input_grad[1]:
                gw = grad_output.t().mm(input)
                grad_weight = mean * gw + sgn * torch.mean(gw * sgn, DIM, keepdim=True)
            if bias is not None and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum(0)
            return grad_input, grad_weight, grad_bias

    return _XNORDense


def XNORConv2d(dim=[0, 1], quant_input=False, stride=1, padding=1, dilation=1, groups=1):
    """XNOR conv op (xnor_connect.py:135-169)."""

    class _XNORConv2d(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            weight_b, mean_weight = xnor_weight(weight, dim)
            if quant_input:
                input = torch.sign(input) * torch.mean(torch.abs(input), 1, keepdim=True)
            ctx.save_for_backward(input, weight, mean_weight, bias)
            if (input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and input.numel() > 0
                    and groups == 1 and not isinstance(padding, str)):
                # sign(W) * alpha[kh, kw] is a REAL weight: six-term bf16 planes, implicit-GEMM conv on the
                # matrix cores (fp32-GEMM accuracy); the result keeps the input's memory format
                y2 = ops.real_conv2d(input, weight_b.detach(), bias, stride, padding, dilation)
                if y2 is not None:
                    N_, _, H, W = input.shape
                    Ho, Wo = ops.conv_out_hw(H, W, weight.shape[2], weight.shape[3], stride, padding, dilation)
                    y = y2.view(N_, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
                    if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                        y = y.contiguous()
                    return y
            return torch.nn.functional.conv2d(input, weight_b, bias=bias, stride=stride,
                                              padding=padding, dilation=dilation, groups=groups)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, mean, bias = ctx.saved_tensors
            sgn = torch.sign(weight)
            grad_input = grad_weight = grad_bias = None
            if ctx.needs_input_grad[0]:
                grad_input = torch.nn.grad.conv2d_input(input.size(), sgn * mean, grad_output,
                                                        stride=stride, padding=padding,
                                                        dilation=dilation, groups=groups)
            if ctx.needs_input_grad[1]:
                gw = torch.nn.grad.conv2d_weight(input, weight.shape, grad_output, stride=stride,
                                                 padding=padding, dilation=dilation, groups=groups)
                grad_weight = mean * gw + sgn * torch.mean(gw * sgn, DIM, keepdim=True)
            if bias is not None and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum((0, 2, 3))
            if bias is not None:
                return grad_input, grad_weight, grad_bias
            return grad_input, grad_weight

    return _XNORConv2d
