"""Lin / Log fixed-point quantisers of "Convolutional Neural Networks using Logarithmic Data Representation"
(reference: QuantTorch/functions/log_lin_connect.py).  HIP device tensors go through qt_lin_quantize_f32 /
qt_log_quantize_f32; CPU tensors evaluate the same expressions in torch.

Upstream defect kept out: ``LinQuant(lin_back=False).backward`` calls ``torch.clamp(Tensor, int, Tensor)``
(log_lin_connect.py:79), which raises under torch 2.x; here it evaluates the expression that line intends,
sign(g) * clamp(round(g/step)*step, 0, 2^fsr) (note: negative g therefore yields -0).
"""
import torch

from .. import ops
from .common import QtFunction, front


def _is_dev(t):
    return t.is_cuda and t.dtype == torch.float32


def _kernel_takes(fsr, bit_width, log):
    """Parameter window of qt_log_quantize_f32 / qt_lin_quantize_f32 (csrc/elementwise.hip); configurations outside it
    (e.g. a 32-bit Log quantiser) run the torch expression on the device, like the reference."""
    return -60 <= fsr <= 60 and 1 <= bit_width <= (16 if log else 32)


def _log_expr(x, fsr, bit_width, with_sign):
    p = torch.pow(torch.ones_like(x) * 2, torch.clamp(torch.round(torch.log2(torch.abs(x))), fsr - 2 ** bit_width, fsr))
    return torch.sign(x) * p if with_sign else p


def _lin_expr(x, fsr, bit_width, mode):
    step = torch.tensor([2.0], dtype=torch.float32, device=x.device).pow(fsr - bit_width)
    top = float(2.0 ** fsr)
    if mode == 1:
        return torch.sign(x) * torch.clamp(torch.round(torch.abs(x) / step) * step, 0, top)
    q = torch.clamp(torch.round(x / step) * step, 0, top)
    return q if mode == 0 else torch.sign(x) * q


def LogQuant(fsr=7, bit_width=3, with_sign=True, lin_back=True):
    """autograd.Function class: forward [sign(x) *] 2^clamp(round(log2|x|), fsr - 2^bit_width, fsr); backward the
    identity (lin_back) or the same quantiser applied to the gradient, signed (log_lin_connect.py:9-41)."""

    class _LogQuant(QtFunction):
        @staticmethod
        def forward(ctx, input):
            if _is_dev(input) and _kernel_takes(fsr, bit_width, True):
                return ops.log_quantize(input, fsr, bit_width, with_sign)
            return _log_expr(input, fsr, bit_width, with_sign)

        @staticmethod
        def backward(ctx, grad_output):
            if lin_back:
                return grad_output.clone()
            if _is_dev(grad_output) and _kernel_takes(fsr, bit_width, True):
                return ops.log_quantize(grad_output, fsr, bit_width, True)
            return _log_expr(grad_output, fsr, bit_width, True)

    return _LogQuant


def LinQuant(fsr=7, bit_width=3, with_sign=True, lin_back=True):
    """autograd.Function class: forward [sign(x) *] clamp(round(|x| / step) * step, 0, 2^fsr), step = 2^(fsr - bit_width)
    (bit_width 32: identity); backward the identity (lin_back) or the quantised gradient (log_lin_connect.py:43-80)."""

    class _LinQuant(QtFunction):
        @staticmethod
        def forward(ctx, input):
            if bit_width == 32:
                return input
            if _is_dev(input) and _kernel_takes(fsr, bit_width, False):
                return ops.lin_quantize(input, fsr, bit_width, 1 if with_sign else 0)
            return _lin_expr(input, fsr, bit_width, 1 if with_sign else 0)

        @staticmethod
        def backward(ctx, grad_output):
            if bit_width == 32 or lin_back:
                return grad_output.clone()
            if _is_dev(grad_output) and _kernel_takes(fsr, bit_width, False):
                return ops.lin_quantize(grad_output, fsr, bit_width, 2)
            return _lin_expr(grad_output, fsr, bit_width, 2)

    return _LinQuant


def nnQuant(dtype="lin", fsr=7, bit_width=3, with_sign=True, lin_back=True):
    """nn.Module front of LinQuant / LogQuant (log_lin_connect.py:84-100)."""
    if dtype == "lin":
        return front(LinQuant(fsr=fsr, bit_width=bit_width, with_sign=with_sign, lin_back=lin_back))
    elif dtype == "log":
        return front(LogQuant(fsr=fsr, bit_width=bit_width, with_sign=with_sign, lin_back=lin_back))
    raise RuntimeError("Only 'log' and 'lin' dtype are supported !")


def Quant(input, dtype="lin", fsr=7, bit_width=3, with_sign=True, lin_back=True):
    """Functional form (log_lin_connect.py:103-118)."""
    if dtype == "lin":
        return LinQuant(fsr=fsr, bit_width=bit_width, with_sign=with_sign, lin_back=lin_back).apply(input)
    elif dtype == "log":
        return LogQuant(fsr=fsr, bit_width=bit_width, with_sign=with_sign, lin_back=lin_back).apply(input)
    raise RuntimeError("Only 'log' and 'lin' dtype are supported !")
