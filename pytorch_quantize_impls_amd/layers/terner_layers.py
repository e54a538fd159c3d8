"""TernaryNet layers (reference: QuantTorch/layers/terner_layers.py)."""
from math import sqrt

import torch

from ..functions import terner_connect, _fused
from .. import lazy
from .common import QLayer, EvalSwapMixin
from ..packed import PackedActivation as _PackedActivation
from .binary_layers import _eval_linear


def _ter_op(deterministic):
    return terner_connect.TernaryConnectDeterministic if deterministic \
        else terner_connect.TernaryConnectStochastic


class LinearTer(EvalSwapMixin, torch.nn.Linear, QLayer):
    """nn.Linear with a ternarised weight (terner_layers.py:10-51)."""

    @staticmethod
    def convert(other, dtype="lin", deterministic=True):
        if not isinstance(other, torch.nn.Linear):
            raise TypeError("Expected a torch.nn.Linear ! Receive:  {}".format(other.__class__))
        return LinearTer(other.in_features, other.out_features, other.bias is not None,
                         deterministic=deterministic)

    def __init__(self, in_features, out_features, bias=True, deterministic=True):
        torch.nn.Linear.__init__(self, in_features, out_features, bias=bias)
        self.deterministic = deterministic
        self.ter_op = _ter_op(deterministic)
        self.binary_input = None

    def reset_parameters(self):
        self.weight.data.normal_(0, sqrt(1. / self.in_features))
        if self.bias is not None:
            self.bias.data.zero_()

    def clamp(self):
        self.weight.data.clamp_(-1, 1)
        if self.bias is not None:
            self.bias.data.clamp_(-1, 1)

    def _quantized_weight_for_eval(self):
        return self.ter_op.apply(self.weight)

    def _weight_on_grid(self, w):
        return ((w == 0) | (w.abs() == 1)).all()

    def forward(self, input):
        return lazy.linear_forward(self, input, "ternary")

    def _forward_impl(self, input):
        if isinstance(input, _PackedActivation):
            return _fused.PACKED_FWD[isinstance(self, torch.nn.Linear)](self, input, "ternary")
        if not input.is_cuda or input.dtype != torch.float32 or self.weight.dtype != torch.float32:
            _fused.note_library_path(input, "non-fp32 dtype")
            w = self.ter_op.apply(self.weight) if self.training else self.weight
            return torch.nn.functional.linear(input, w, self.bias)
        if self.training:
            wq = None if self.deterministic else self.ter_op.apply(self.weight.detach())
            return _fused.QuantLinearFn.apply(input, self.weight, self.bias, "ternary", wq,
                                              self.binary_input)
        return _eval_linear(self, input, "ternary")


class TerConv2d(EvalSwapMixin, torch.nn.Conv2d, QLayer):
    """nn.Conv2d with a ternarised weight (terner_layers.py:54-92)."""

    @staticmethod
    def convert(other, deterministic=True):
        if not isinstance(other, torch.nn.Conv2d):
            raise TypeError("Expected a torch.nn.Conv2d ! Receive:  {}".format(other.__class__))
        return TerConv2d(other.in_channels, other.out_channels, other.kernel_size,
                         stride=other.stride, padding=other.padding, dilation=other.dilation,
                         groups=other.groups, bias=other.bias is not None,
                         deterministic=deterministic)

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, deterministic=True):
        torch.nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride=stride,
                                 padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.deterministic = deterministic
        self.ter_op = _ter_op(deterministic)
        self.binary_input = None

    def clamp(self):
        self.weight.data.clamp_(-1, 1)

    def _quantized_weight_for_eval(self):
        return self.ter_op.apply(self.weight)

    def _weight_on_grid(self, w):
        return ((w == 0) | (w.abs() == 1)).all()

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
            return self._eval_planes(lambda _w2: ops.pack_conv_weight_bf16x3(self.weight.detach(), "ternary", terms=terms),
                                     key=f"conv_split{terms}")

        def build(_w2):
            ws = ops.s2d_weight(self.weight.detach(), self.stride[0])
            return tuple(ws.shape), ops.pack_conv_weight_bf16x3(ws, "sign", terms=terms)
        return self._eval_planes(build, key=f"conv_split{terms}_s2d")

    def forward(self, input):
        """See BinConv2d.forward: eval mode without autograd on a HIP device returns a deferred activation."""
        return lazy.conv_forward(self, input, "ternary")

    def _forward_impl(self, input):
        if isinstance(input, _PackedActivation):
            return _fused.PACKED_FWD[isinstance(self, torch.nn.Linear)](self, input, "ternary")
        if not input.is_cuda or input.dtype != torch.float32 or self.weight.dtype != torch.float32:
            _fused.note_library_path(input, "non-fp32 dtype")
            w = self.ter_op.apply(self.weight) if self.training else self.weight
            return torch.nn.functional.conv2d(input, w, self.bias, self.stride, self.padding,
                                              self.dilation, self.groups)
        if self.groups > 1 and self.padding_mode == "zeros" and input.dim() == 4 and not isinstance(self.padding, str):
            y = _fused.grouped_quant_conv(self, input, "ternary", self.ter_op)       # every group on the groups == 1 routes
            if y is not None:
                return y
        args = (self.stride, self.padding, self.dilation, self.groups)
        if self.training:
            wq = None if self.deterministic else self.ter_op.apply(self.weight.detach())
            return _fused.QuantConv2dFn.apply(input, self.weight, self.bias, "ternary", wq,
                                              self.binary_input, args)
        # eval: weight already holds the quantised image; its packed planes are cached
        if torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad):
            # eval-mode forward under autograd: the training node on the stored image (see _eval_linear)
            if self.padding_mode == "zeros" and not isinstance(self.padding, str) and self._eval_on_grid():
                return _fused.QuantConv2dFn.apply(input, self.weight, self.bias, "ternary", self.weight.detach(),
                                                  self.binary_input, args)
            _fused.note_library_path(input, "eval-mode conv forward under autograd (off-grid weight / non-zero padding mode)")
            return torch.nn.functional.conv2d(input, self.weight, self.bias, *args)
        if not self._eval_on_grid():
            _fused.note_library_path(input, "eval-mode weight off the quantiser's grid")
            return torch.nn.functional.conv2d(input, self.weight, self.bias, *args)
        wp = None
        if self.groups == 1 and self.padding_mode == "zeros":
            wp = self._eval_planes(lambda _w2: _fused.ops.pack_conv_weight_nib(self.weight.detach(), "ternary"),
                                   key="conv_nib")
        return _fused.quant_conv2d_forward(input, self.weight, self.bias, *args, "ternary",
                                           weight_q=self.weight, weight_planes=wp,
                                           binary_input=self.binary_input, padding_mode=self.padding_mode,
                                           weight_triples_fn=self._conv_triples)
