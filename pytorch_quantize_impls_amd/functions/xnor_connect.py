"""XNOR-Net ops (reference: QuantTorch/functions/xnor_connect.py).

The reference module is unfinished upstream; observable numerics are reproduced, including:
  * nnQuantXnor / QuantXnor use the SIGNED mean (not mean|x|) despite their docstring (:21-28);
  * XNORDense ignores its ``dim`` argument and always reduces over the module-global DIM = 0,
    i.e. one scale per INPUT feature, shape [1, K] (:13, :112, :127);
  * XNORConv2d uses its ``dim`` in forward but the global DIM in backward (:140, :158-159).
"""
import torch

from .. import ops, packed
from . import _fused
from .common import QtFunction, front

DIM = 0


def _quantOpXnor(dim=1):
    class _QuantXNOR(QtFunction):
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
    """XNOR dense op; ``dim`` is accepted and ignored like upstream (xnor_connect.py:93-132).

    Device fp32 tensors: forward = the split GEMM of x * alpha[k] against sign(W) on the matrix cores; backward (:118-131) on the
    same routes — grad_input = (g . sign(W)) * alpha[k] (the per-column scale commutes with the contraction over n),
    grad_weight from g^T . x with the +-1 activation as the exact operand (two real operands: the dense library, counted in
    _fused.LIBRARY_PATHS) and the XNOR-Net combination mean * gw + sign(W) * mean(gw * sign(W), DIM)."""

    class _XNORDense(QtFunction):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            ctx.x_is_pm1 = False
            if input.is_cuda and input.dtype == torch.float32 and weight.dim() == 2 and input.numel() > 0:
                # alpha on the device (qt_xnor_weight_f32), folded into the activation split; sign(W) on
                # the bf16 matrix cores: fl(x*alpha) * (+-1) is the product fl(x * (+-alpha)) exactly
                _, mean = ops.xnor_weight(weight.detach(), 1)
                ctx.save_for_backward(input, weight, mean, bias)
                if input.dim() == 2:
                    planes = packed.lookup(input, packed.ROWS_LAST)
                    ctx.x_is_pm1 = planes is not None
                    if planes is not None and planes.K == weight.shape[1] and planes.rows == input.shape[0]:
                        # +-1 activation with its sign planes: x * alpha = +-alpha[k] as two-term fp16 pairs built from the
                        # bits (the operands of the eval-mode / packed path: one arithmetic for every execution of a layer)
                        wt, ap = _fused.xnor_linear_operands(weight=weight)
                        return ops.bf16_gemm(ops.bits_alpha_pairs(planes, ap), wt, bias.detach() if bias is not None else None)
                    if not ctx.x_is_pm1 and _fused._cfg("DETECT_BINARY_INPUT"):
                        ctx.x_is_pm1 = bool(_fused.detect_pm1(input, weight)[0])
                return ops.float_linear(input, weight.detach(), "sign", bias, alpha=mean.view(-1))
            weight_q, mean = xnor_weight(weight, DIM)
            ctx.save_for_backward(input, weight, mean, bias)
            return torch.nn.functional.linear(input, weight_q, bias)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, mean, bias = ctx.saved_tensors
            sgn = torch.sign(weight)
            grad_input = grad_weight = grad_bias = None
            hip = grad_output.is_cuda and grad_output.dtype == torch.float32 and grad_output.dim() == 2 and input.dim() == 2
            g2 = grad_output.contiguous() if hip else grad_output
            if ctx.needs_input_grad[0]:
                if hip:
                    grad_input = _fused.dense_grad_input(g2, sgn, pm1=True) * mean     # g . (sign(W) * alpha[k]) = (g . sign(W)) * alpha[k]
                else:
                    grad_input = _fused.lib_mm(grad_output, sgn * mean)
            if ctx.needs_input_grad[1]:
                gw = _fused.dense_grad_weight(grad_output, input, ctx.x_is_pm1)        # +-1 x: the exact operand; else six-term
                grad_weight = _fused.xnor_weight_grad(gw, weight, mean, DIM)
            if bias is not None and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum(0)
            return grad_input, grad_weight, grad_bias

    return _XNORDense


def XNORConv2d(dim=[0, 1], quant_input=False, stride=1, padding=1, dilation=1, groups=1):
    """XNOR conv op (xnor_connect.py:135-169).

    Device fp32 tensors with ``dim`` = [0, 1] (one alpha per filter tap), groups == 1:
      * +-1 activation (tag of a BinaryConnect, or detected on the device): one fp4 matrix-core pass over the sign planes with the
        taps' alphas applied on the accumulators (qt_conv2d_implicit_taps) — y = sum_taps alpha[i, j] * (integer dot over Cin);
      * real-valued activation (the first layer): six-term bf16 planes of x and of sign(W) * alpha (fp32-GEMM accuracy);
      * backward: grad_input = the split gradient against the flipped sign(W) with the flipped alphas per tap (stride 1),
        grad_weight = the +-1 weight-gradient routes (pixel-major / K-major / strided) + the XNOR-Net combination (:158-159).
    Whatever has no route here runs the reference expression on the dense library and is counted (_fused.LIBRARY_PATHS)."""

    class _XNORConv2d(QtFunction):
        @staticmethod
        def forward(ctx, input, weight, bias=None):
            ctx.x_is_pm1, ctx.taps = False, None
            fast = (not quant_input) and _fused.xnor_conv_fast_applicable(input, weight, dim, groups, padding)
            if fast:
                known = True if packed.lookup(input, packed.NHWC) is not None else None
                _fused.clear_last_detection()
                out = _fused.xnor_conv2d_forward(input, weight, bias, stride, padding, dilation, binary_input=known)
                if out is not None:
                    y2, taps = out
                    kh, kw = int(weight.shape[2]), int(weight.shape[3])
                    ctx.x_is_pm1, ctx.taps = True, taps
                    ctx.save_for_backward(input, weight, taps.alpha.view(1, 1, kh, kw), bias)
                    N_, _, H, W = input.shape
                    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, padding, dilation)
                    y = y2.view(N_, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
                    if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                        y = y.contiguous()
                    return y
            if quant_input:
                # both operands binarised (:142-143): x -> sign(x) * mean(|x|, 1) per pixel; backward sees this quantised tensor
                # (:144).  With one alpha per tap the conv is  sum_t alpha_t A[pixel(m, t)] D_t[m, co]  (D_t: integer dot of the two
                # sign planes over tap t): ONE fp4 matrix-core pass with a per-(row, tap) factor on the accumulators
                # (qt_conv2d_implicit_taps_rows); kernels beyond 48 taps: the two-term fp16 split of the quantised image x sign(W)
                on_dev = input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and input.numel() > 0
                raw = input
                need_image = (not on_dev) or any(ctx.needs_input_grad[:2])
                a_plane = None
                if on_dev:
                    input, a_plane = ops.xnor_input_quant(raw, want_image=need_image, want_scale=True)
                else:
                    input = torch.sign(input) * torch.mean(torch.abs(input), 1, keepdim=True)
                if on_dev and _fused.xnor_conv_fast_applicable(raw, weight, dim, groups, padding) and weight.is_cuda:
                    kh, kw = int(weight.shape[2]), int(weight.shape[3])
                    taps = ops.xnor_tap_prep(weight)
                    y2 = None
                    if kh * kw <= 48:
                        wp = ops.pack_conv_weight_nib(weight.detach(), "sign", cw=ops.pixel_ld_nib_taps(int(weight.shape[1])))
                        y2 = ops.conv2d_nib_taps_rows(raw, a_plane, wp, (kh, kw), taps.fwd, bias, stride, padding, dilation)
                    if y2 is None and int(weight.shape[1]) % 8 == 0:
                        if input is None:
                            input = ops.xnor_input_quant(raw)
                        y2 = ops.conv2d_real_taps(input, weight, taps.fwd, bias, stride, padding, dilation)
                    if y2 is not None:
                        ctx.taps = taps                                   # grad_input: the flipped taps, like the +-1 route
                        # always saved: a trainable bias alone still runs backward (the image slot is None then)
                        ctx.save_for_backward(input if need_image else None, weight, taps.alpha.view(1, 1, kh, kw), bias)
                        N_, _, H, W = raw.shape
                        Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, padding, dilation)
                        return y2.view(N_, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
                if input is None:
                    input = ops.xnor_input_quant(raw)
            weight_b, mean_weight = xnor_weight(weight, dim)
            ctx.save_for_backward(input, weight, mean_weight, bias)
            if (input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and input.numel() > 0
                    and groups == 1 and not isinstance(padding, str)):
                # sign(W) * alpha[kh, kw] is a REAL weight.  A strided few-channel image (the first layer): the direct kernel with
                # two-term fp16 weights (three products per element); anything else: six-term bf16 planes, implicit-GEMM conv on
                # the matrix cores (fp32-GEMM accuracy).  The result keeps the input's memory format
                y2 = None
                if ops.first_direct_applicable(int(input.shape[1]), weight.shape[2:], stride, padding, dilation):
                    fw = ops.pack_first_layer_weight(weight_b, ops._pairs(stride)[0], real=True)
                    y2 = ops.conv_first_direct(input, fw, bias, stride, padding)
                if y2 is None:
                    y2 = ops.real_conv2d(input, weight_b.detach(), bias, stride, padding, dilation)
                if y2 is not None:
                    N_, _, H, W = input.shape
                    Ho, Wo = ops.conv_out_hw(H, W, weight.shape[2], weight.shape[3], stride, padding, dilation)
                    y = y2.view(N_, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
                    if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                        y = y.contiguous()
                    return y
            _fused.note_library_path(input, "XNOR conv outside the matrix-core routes")
            return torch.nn.functional.conv2d(input, weight_b, bias=bias, stride=stride,
                                              padding=padding, dilation=dilation, groups=groups)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight, mean, bias = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            go = _fused._dense(grad_output)
            mfma = (_fused._cfg("BWD_CONV_MFMA") and ctx.taps is not None and go.is_cuda and go.dtype == torch.float32 and groups == 1
                    and not isinstance(padding, str))
            if ctx.needs_input_grad[0]:
                if mfma:
                    grad_input = ops.conv2d_grad_input_taps(input.shape, weight, go, ctx.taps.bwd, stride, padding, dilation)
                if grad_input is None:
                    grad_input = _fused.lib_conv2d_input(input.size(), torch.sign(weight) * mean, grad_output, stride, padding,
                                                         dilation, groups)
            want_bias = bias is not None and ctx.needs_input_grad[2]
            by_product = []
            if ctx.needs_input_grad[1]:
                # (quant_input: the saved input is the quantised, real-valued one — its terms go in as channel groups of the
                #  pixel-major weight-gradient kernel)
                gw = _fused.conv_grad_weight(input, weight.shape, grad_output, stride, padding, dilation, groups, ctx.x_is_pm1,
                                             by_product if want_bias else None, real_any_channels=quant_input)
                grad_weight = _fused.xnor_weight_grad(gw, weight, mean, DIM)
            if want_bias:
                grad_bias = by_product[0] if by_product else grad_output.sum((0, 2, 3))
            if bias is not None:
                return grad_input, grad_weight, grad_bias
            return grad_input, grad_weight

    return _XNORConv2d
