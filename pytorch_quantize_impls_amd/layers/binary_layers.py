"""BinaryNet layers (reference: QuantTorch/layers/binary_layers.py)."""
from math import sqrt as _sqrt

import torch

from ..functions import binary_connect, _fused
from .. import lazy
from .common import QLayer, EvalSwapMixin
from ..packed import PackedActivation as _PackedActivation


class LinearBin(EvalSwapMixin, torch.nn.Linear, QLayer):
    """nn.Linear whose weight is binarised on the fly (binary_layers.py:7-46).

    ``binary_input``: None (default) = detect +-1 activations (tag from BinaryConnect, else a
    device-side check); True = caller guarantees +-1 activations; False = never take the packed
    path.  Only consulted for device tensors.
    """

    @staticmethod
    def convert(other, deterministic=True):
        if not isinstance(other, torch.nn.Linear):
            raise TypeError("Expected a torch.nn.Linear ! Receive:  {}".format(other.__class__))
        # like upstream, a FRESH layer: weights are not copied (binary_layers.py:8-12)
        return LinearBin(other.in_features, other.out_features, other.bias is not None, deterministic)

    def __init__(self, in_features, out_features, bias=True, deterministic=True):
        torch.nn.Linear.__init__(self, in_features, out_features, bias=bias)
        self.deterministic = deterministic
        self.bin_op = binary_connect.BinaryConnectDeterministic if deterministic \
            else binary_connect.BinaryConnectStochastic
        self.binary_input = None

    def reset_parameters(self):
        self.weight.data.normal_(0, _sqrt(1. / self.in_features))
        if self.bias is not None:
            self.bias.data.zero_()

    def clamp(self):
        self.weight.data.clamp_(-1, 1)
        if self.bias is not None:  # upstream clamps the bias as well (binary_layers.py:27-28)
            self.bias.data.clamp_(-1, 1)

    def _quantized_weight_for_eval(self):
        return self.bin_op.apply(self.weight)

    def _weight_on_grid(self, w):
        return (w.abs() == 1).all()

    def forward(self, input):
        return lazy.linear_forward(self, input, "binary")

    def _forward_impl(self, input):
        if isinstance(input, _PackedActivation):
            return _fused.PACKED_FWD[isinstance(self, torch.nn.Linear)](self, input, "binary")
        if not input.is_cuda or input.dtype != torch.float32 or self.weight.dtype != torch.float32:
            # host tensors, and device models in half / bfloat16 / double: the reference expression in torch
            _fused.note_library_path(input, "non-fp32 dtype")
            w = self.bin_op.apply(self.weight) if self.training else self.weight
            return torch.nn.functional.linear(input, w, self.bias)
        if self.training:
            wq = None if self.deterministic else self.bin_op.apply(self.weight.detach())
            return _fused.QuantLinearFn.apply(input, self.weight, self.bias, "binary", wq,
                                              self.binary_input)
        # eval: weight already holds the binarised image; planes are cached
        return _eval_linear(self, input, "binary")


def _eval_linear(layer, input, kind):
    """Eval-mode forward of a device layer: F.linear(input, weight, bias) (binary_layers.py:46)
    with the packed path when the activation is +-1."""
    if torch.is_grad_enabled() and (input.requires_grad or layer.weight.requires_grad):
        # eval-mode forward with autograd on (model.eval(); model(x) without no_grad: the parameters require grad).  The
        # reference expression is F.linear on the stored image; on the quantiser's grid that is the training node with the
        # image as the pre-quantised weight — own kernels forward and backward, and its STE mask 1[|W| <= 1.001] is all ones
        # on an image in {0, +-1}, so the gradients are those of F.linear (this route used to be the dense
        # library, uncounted)
        if layer._eval_on_grid():
            return _fused.QuantLinearFn.apply(input, layer.weight, layer.bias, kind, layer.weight.detach(), layer.binary_input)
        _fused.note_library_path(input, "eval-mode forward under autograd, weight off the quantiser's grid")
        return torch.nn.functional.linear(input, layer.weight, layer.bias)
    if not layer._eval_on_grid():
        # the weight was overwritten with something that is not a quantised image (e.g. a float checkpoint loaded
        # after .eval()): upstream multiplies by whatever `weight` holds, so does this
        _fused.note_library_path(input, "eval-mode weight off the quantiser's grid")
        return torch.nn.functional.linear(input, layer.weight, layer.bias)
    K, N = input.shape[-1], layer.weight.shape[0]
    impl = _fused.ops.select_gemm_impl(_fused._cfg("GEMM_IMPL"), input.numel() // max(K, 1), N, K)
    xp, flag = _fused.activation_planes(input, layer.binary_input, impl, layer.weight)
    if xp is None:
        if _fused._cfg("FLOAT_PATH") == "bf16x3" and input.dtype == torch.float32 and input.numel() > 0:
            wt = layer._eval_planes(lambda w2: _fused.ops.weight_bf16x3(w2, kind), key="bf16x3")
            return _fused.ops.float_linear(input, layer.weight, kind, layer.bias, weight_triples=wt)
        _fused.note_library_path(input, "eval-mode linear on a real-valued activation with the float route switched off")
        return torch.nn.functional.linear(input, layer.weight, layer.bias)
    wp = layer._eval_planes(lambda w2: _fused.pack_weight(w2, kind, impl), key=impl)
    y = _fused.ops.packed_gemm(xp, wp, _fused.poison_bias(layer.bias, flag, N, input.device), impl=impl)
    return y.view(*input.shape[:-1], N)


class BinConv2d(EvalSwapMixin, torch.nn.Conv2d, QLayer):
    """nn.Conv2d with binarised weight (binary_layers.py:48-106)."""

    @staticmethod
    def convert(other, deterministic=True):
        if not isinstance(other, torch.nn.Conv2d):
            raise TypeError("Expected a torch.nn.Conv2d ! Receive:  {}".format(other.__class__))
        return BinConv2d(other.in_channels, other.out_channels, other.kernel_size,
                         stride=other.stride, padding=other.padding, dilation=other.dilation,
                         groups=other.groups, bias=other.bias is not None,
                         deterministic=deterministic)

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, deterministic=True):
        torch.nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride=stride,
                                 padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.bin_op = binary_connect.BinaryConnectDeterministic if deterministic \
            else binary_connect.BinaryConnectStochastic
        self.deterministic = deterministic
        self.binary_input = None

    def clamp(self):
        """Clamp the real weight to [-1, 1] (bias untouched, binary_layers.py:81-85)."""
        self.weight.data.clamp_(-1, 1)

    def _quantized_weight_for_eval(self):
        return self.bin_op.apply(self.weight)

    def _weight_on_grid(self, w):
        return (w.abs() == 1).all()

    def _conv_triples(self, form, terms=None):
        """Cached split image (bf16 triples / fp16 pairs, ops.FLOAT_SPLIT or ``terms``) of the eval-mode (already quantised)
        weight for real-valued inputs: 'plain' -> TriplePlanes; 's2d' -> (transformed weight shape, TriplePlanes) for the
        space-to-depth form."""
        ops = _fused.ops
        if form == "first3x3":         # MFMA row fragments of the one-pass 3 x 3 first-layer kernel (ops.pack_first3x3_weight)
            return self._eval_planes(lambda _w2: ops.pack_first3x3_weight(self.weight.detach()), key="first3x3")
        if form == "first_direct":     # fragment-ordered fp16 weight of the direct first-layer kernel (ops.pack_first_layer_weight)
            return self._eval_planes(lambda _w2: ops.pack_first_layer_weight(self.weight.detach(), self.stride[0]), key="first_direct")
        terms = ops.split_terms(terms)
        if form == "plain":
            return self._eval_planes(lambda _w2: ops.pack_conv_weight_bf16x3(self.weight.detach(), "binary", terms=terms),
                                     key=f"conv_split{terms}")

        def build(_w2):
            ws = ops.s2d_weight(self.weight.detach(), self.stride[0])
            return tuple(ws.shape), ops.pack_conv_weight_bf16x3(ws, "sign", terms=terms)
        return self._eval_planes(build, key=f"conv_split{terms}_s2d")

    def forward(self, input):
        """Eval mode without autograd on a HIP device: returns a deferred activation (``lazy.LazyActivation``: a Tensor
        that runs this conv fused with the BatchNorm / pooling / Hardtanh / BinaryConnect modules that follow it, or
        computes the plain fp32 result on any other use); every other case computes right here."""
        return lazy.conv_forward(self, input, "binary")

    def _forward_impl(self, input):
        if isinstance(input, _PackedActivation):
            return _fused.PACKED_FWD[isinstance(self, torch.nn.Linear)](self, input, "binary")
        if not input.is_cuda or input.dtype != torch.float32 or self.weight.dtype != torch.float32:
            _fused.note_library_path(input, "non-fp32 dtype")
            w = self.bin_op.apply(self.weight) if self.training else self.weight
            return torch.nn.functional.conv2d(input, w, self.bias, self.stride, self.padding,
                                              self.dilation, self.groups)
        if self.groups > 1 and self.padding_mode == "zeros" and input.dim() == 4 and not isinstance(self.padding, str):
            y = _fused.grouped_quant_conv(self, input, "binary", self.bin_op)       # every group on the groups == 1 routes
            if y is not None:
                return y
        args = (self.stride, self.padding, self.dilation, self.groups)
        if self.training:
            wq = None if self.deterministic else self.bin_op.apply(self.weight.detach())
            return _fused.QuantConv2dFn.apply(input, self.weight, self.bias, "binary", wq,
                                              self.binary_input, args)
        # eval: weight already holds the quantised image; its packed planes are cached
        if torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad):
            # eval-mode forward under autograd: the training node on the stored image (see _eval_linear)
            if self.padding_mode == "zeros" and not isinstance(self.padding, str) and self._eval_on_grid():
                return _fused.QuantConv2dFn.apply(input, self.weight, self.bias, "binary", self.weight.detach(),
                                                  self.binary_input, args)
            _fused.note_library_path(input, "eval-mode conv forward under autograd (off-grid weight / non-zero padding mode)")
            return torch.nn.functional.conv2d(input, self.weight, self.bias, *args)
        if not self._eval_on_grid():
            _fused.note_library_path(input, "eval-mode weight off the quantiser's grid")
            return torch.nn.functional.conv2d(input, self.weight, self.bias, *args)
        wp = None
        if self.groups == 1 and self.padding_mode == "zeros":
            wp = self._eval_planes(lambda _w2: _fused.ops.pack_conv_weight_nib(self.weight.detach(), "binary"),
                                   key="conv_nib")
        return _fused.quant_conv2d_forward(input, self.weight, self.bias, *args, "binary",
                                           weight_q=self.weight, weight_planes=wp,
                                           binary_input=self.binary_input, padding_mode=self.padding_mode,
                                           weight_triples_fn=self._conv_triples)


class ShiftNormBatch1d(torch.nn.Module):
    """Shift-based batch norm, 1-D (binary_layers.py:110-132); torch ops, off the hot path.
    weight/bias are created uninitialised like upstream."""
    __constants__ = ['momentum', 'eps']

    def __init__(self, in_dim, eps=1e-5, momentum=0.1):
        super().__init__()
        self.in_features = in_dim
        self.weight = torch.nn.Parameter(torch.empty(in_dim))
        self.bias = torch.nn.Parameter(torch.empty(in_dim))
        self.register_buffer('running_mean', torch.zeros(in_dim))
        self.register_buffer('running_var', torch.ones(in_dim))
        self.eps = eps
        self.momentum = momentum

    def forward(self, x):
        m = self.momentum
        self.running_mean = (1 - m) * self.running_mean + m * torch.mean(x, 0).detach()
        c = x - self.running_mean
        self.running_var = (1 - m) * self.running_var + m * torch.mean(c * binary_connect.AP2(c), 0).detach()
        return binary_connect.ShiftBatch.apply(x, self.running_mean, self.running_var, self.weight,
                                               self.bias, self.eps)


class ShiftNormBatch2d(torch.nn.Module):
    """Shift-based batch norm, 2-D (binary_layers.py:137-160)."""
    __constants__ = ['momentum', 'eps']

    def __init__(self, in_channels, eps=1e-5, momentum=0.1):
        super().__init__()
        self.in_features = in_channels
        self.weight = torch.nn.Parameter(torch.empty(in_channels))
        self.bias = torch.nn.Parameter(torch.empty(in_channels))
        self.register_buffer('running_mean', torch.zeros(in_channels))
        self.register_buffer('running_var', torch.ones(in_channels))
        self.eps = eps
        self.momentum = momentum

    @staticmethod
    def _tile(tensor, dim):
        return tensor.repeat(dim[0], dim[1], 1).transpose(2, 0)

    def forward(self, x):
        dim = x.size()[-2:]
        m = self.momentum
        self.running_mean = (1 - m) * self.running_mean + m * torch.mean(x, [0, 2, 3]).detach()
        curr_mean = self._tile(self.running_mean, dim)
        c = x - curr_mean
        self.running_var = (1 - m) * self.running_var + m * torch.mean(c * binary_connect.AP2(c), [0, 2, 3]).detach()
        return binary_connect.ShiftBatch.apply(x, curr_mean, self._tile(self.running_var, dim),
                                               self._tile(self.weight, dim),
                                               self._tile(self.bias, dim), self.eps)
