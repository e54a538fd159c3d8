"""DoReFa-Net layers (reference: QuantTorch/layers/dorefa_layers.py)."""
import torch

from ..functions import dorefa_connect, _fused
from .. import lazy, lazy_train
from ..packed import CodeActivation as _CodeActivation
from .common import QLayer, EvalSwapMixin


class LinearDorefa(EvalSwapMixin, torch.nn.Linear, QLayer):
    """nn.Linear with a k-bit DoReFa weight (dorefa_layers.py:11-45)."""

    @staticmethod
    def convert(other, bit_width=3):
        if not isinstance(other, torch.nn.Linear):
            raise TypeError("Expected a torch.nn.Linear ! Receive:  {}".format(other.__class__))
        return LinearDorefa(other.in_features, other.out_features, other.bias is not None,
                            bit_width=bit_width)

    def __init__(self, in_features, out_features, bias=True, bit_width=3):
        torch.nn.Linear.__init__(self, in_features, out_features, bias=bias)
        self.bit_width = bit_width
        self.weight_op = dorefa_connect.nnQuantWeight(bit_width=bit_width)

    def extra_repr(self):
        return "bit_width = {}".format(self.bit_width)

    def _quantized_weight_for_eval(self):
        return self.weight_op.forward(self.weight)

    def _weight_on_grid(self, w):
        k = int(self.bit_width)
        if k == 1:                      # sign(W) * E: every magnitude equals the scale
            return (w.abs() == w.abs().amax()).all()
        if k > _fused.LEVEL_MAX_BITS:
            # 2^k - 1 levels beyond fp32's reach to verify (and the identity at k = 32): these widths run the real x real
            # routes, which multiply by whatever ``weight`` holds — like upstream — and need no grid guarantee
            return torch.ones((), dtype=torch.bool, device=w.device)
        n = float((1 << k) - 1)         # levels (2 q - n) / n: n * w is an integer of the parity of n, |.| <= n
        c = w * n
        r = torch.round(c)
        return ((c - r).abs() <= 1e-3).all() & (r.abs() <= n).all() & (torch.remainder(r + n, 2) == 0).all()

    def forward(self, input):
        lazy.note_inference_call(self, input)
        return lazy_train.wrap(self, self._forward_impl(lazy.resolve(input)))

    def _forward_impl(self, input):
        if isinstance(input, _CodeActivation) and (self.training or self.bit_width != 1):
            raise RuntimeError("CodeActivation inputs are an inference feature of 1-bit-weight DoReFa layers: "
                               "call .eval() first (k-bit weights: pass input.float())")
        if isinstance(input, _CodeActivation) and not self._eval_on_grid():
            raise RuntimeError("this eval-mode layer's weight no longer holds sign(W) * E (overwritten after .eval()?): "
                               "code-plane inputs need the quantised image")
        if (not isinstance(input, _CodeActivation) and input.is_cuda and not self.training and self.weight.dtype == torch.float32
                and not self._eval_on_grid()):
            _fused.note_library_path(input, "eval-mode weight off the quantiser's grid")
            return torch.nn.functional.linear(input, self.weight, self.bias)
        if input.is_cuda and self.bit_width == 1 and input.dtype == torch.float32 and self.weight.dtype == torch.float32:
            # W1Ak: int8 matrix-core path when the activation carries DoReFa codes
            if self.training:
                return _fused.DorefaW1LinearFn.apply(input, self.weight, self.bias)
            if not (torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad)):
                wc = self._eval_planes(lambda w2: _fused.ops.weight_codes(w2), key="i8")
                E = self._eval_planes(lambda w2: w2.abs().amax(), key="E")      # |w| == E everywhere after eval()
                return _fused.dorefa_w1_linear_forward(input, self.weight, self.bias, True, wc, scale=E)
        if (input.is_cuda and 2 <= self.bit_width <= _fused.LEVEL_INT8_BITS and input.dtype == torch.float32 and not self.training
                and self.weight.dtype == torch.float32 and not (torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad))):
            # WkAk inference: integer weight levels x activation codes on the int8 matrix cores
            wc = self._eval_planes(lambda w2: _fused.ops.dorefa_weight_codes(w2, self.bit_width), key="i8k")
            y = _fused.dorefa_wk_linear_forward(input, self.weight, self.bias, self.bit_width, wc)
            if y is not None:
                return y
        if (input.is_cuda and 2 <= self.bit_width <= _fused.LEVEL_MAX_BITS and input.dtype == torch.float32 and not self.training
                and self.weight.dtype == torch.float32 and not (torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad))):
            # no usable int8 codes (real-valued input, codes beyond int8) or 8-bit weights (|level| <= 255): split activation x
            # the exact level image
            terms = _fused.ops.split_terms()
            lp = self._eval_planes(lambda w2: _fused.ops.weight_bf16x3(_fused._weight_levels(w2, self.bit_width), "raw", terms=terms),
                                   key=f"levels{terms}")
            y = _fused.dorefa_levels_linear_forward(input, self.weight, self.bias, self.bit_width, lp)
            if y is not None:
                return y
        w = self.weight_op.forward(self.weight) if self.training else self.weight
        if (input.is_cuda and self.training and 2 <= self.bit_width <= _fused.LEVEL_MAX_BITS and input.dtype == torch.float32
                and self.weight.dtype == torch.float32):
            # WkAk training: level image x codes / exact split on the matrix cores, forward and both gradients
            return _fused.DorefaWkLinearFn.apply(input, w, self.bias, self.bit_width)
        if (input.is_cuda and input.dtype == torch.float32 and self.weight.dtype == torch.float32 and input.numel() > 0
                and input.dim() >= 2):
            # two real operands on the six-term planes (fp32-GEMM accuracy), forward and both gradients:
            #   bit_width = 32: the identity quantiser (functions/dorefa_connect.py:19-20, 100-101);
            #   8 < bit_width < 32: 2^k - 1 levels are past the exact level images (functions/dorefa_connect.py:21-25 accepts any
            #     k) — the quantised weight is a real tensor like any other, its straight-through backward is weight_op's;
            #   eval mode under autograd (any k): F.linear on the stored image is the reference expression
            return _fused.RealLinearFn.apply(input, w, self.bias)
        if input.is_cuda:
            _fused.note_library_path(input, "DoReFa linear in a non-fp32 dtype")
        return torch.nn.functional.linear(input, w, self.bias)


class DorefaConv2d(EvalSwapMixin, torch.nn.Conv2d, QLayer):
    """nn.Conv2d with a k-bit DoReFa weight (dorefa_layers.py:48-82)."""

    @staticmethod
    def convert(other, bit_width=3):
        if not isinstance(other, torch.nn.Conv2d):
            raise TypeError("Expected a torch.nn.Conv2d ! Receive:  {}".format(other.__class__))
        return DorefaConv2d(other.in_channels, other.out_channels, other.kernel_size,
                            stride=other.stride, padding=other.padding, dilation=other.dilation,
                            groups=other.groups, bias=other.bias is not None, bit_width=bit_width)

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, bit_width=3):
        torch.nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride=stride,
                                 padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.bit_width = bit_width
        self.weight_op = dorefa_connect.nnQuantWeight(bit_width=bit_width)

    def _quantized_weight_for_eval(self):
        return self.weight_op.forward(self.weight)

    def _weight_on_grid(self, w):
        k = int(self.bit_width)
        if k == 1:                      # sign(W) * E: every magnitude equals the scale
            return (w.abs() == w.abs().amax()).all()
        if k > _fused.LEVEL_MAX_BITS:
            # 2^k - 1 levels beyond fp32's reach to verify (and the identity at k = 32): these widths run the real x real
            # routes, which multiply by whatever ``weight`` holds — like upstream — and need no grid guarantee
            return torch.ones((), dtype=torch.bool, device=w.device)
        n = float((1 << k) - 1)         # levels (2 q - n) / n: n * w is an integer of the parity of n, |.| <= n
        c = w * n
        r = torch.round(c)
        return ((c - r).abs() <= 1e-3).all() & (r.abs() <= n).all() & (torch.remainder(r + n, 2) == 0).all()

    def forward(self, input):
        """1-bit weights, eval mode, no autograd, input carrying int8 codes on a HIP device: returns a deferred activation
        (lazy.py) that runs this conv with the BatchNorm / shortcut add / ReLU / nnDorefaQuant modules that follow it in
        the conv's code epilogue; otherwise computes here."""
        return lazy.dorefa_conv_forward(self, input)

    def _forward_impl(self, input):
        args = (self.stride, self.padding, self.dilation, self.groups)
        if isinstance(input, _CodeActivation) and (self.training or self.bit_width != 1):
            raise RuntimeError("CodeActivation inputs are an inference feature of 1-bit-weight DoReFa layers: "
                               "call .eval() first (k-bit weights: pass input.float())")
        if isinstance(input, _CodeActivation) and not self._eval_on_grid():
            raise RuntimeError("this eval-mode layer's weight no longer holds sign(W) * E (overwritten after .eval()?): "
                               "code-plane inputs need the quantised image")
        if (not isinstance(input, _CodeActivation) and input.is_cuda and not self.training and self.weight.dtype == torch.float32
                and not self._eval_on_grid()):
            _fused.note_library_path(input, "eval-mode weight off the quantiser's grid")
            return torch.nn.functional.conv2d(input, self.weight, self.bias, *args)
        if input.is_cuda and self.bit_width == 1 and input.dtype == torch.float32 and self.weight.dtype == torch.float32:
            if self.training:
                return _fused.DorefaW1Conv2dFn.apply(input, self.weight, self.bias, args)
            if not (torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad)):
                wc = None
                if self.groups == 1 and self.padding_mode == "zeros":
                    wc = self._eval_planes(lambda _w2: _fused.ops.pack_conv_weight_codes(self.weight.detach()),
                                           key="conv_i8")
                E = self._eval_planes(lambda w2: w2.abs().amax(), key="E")      # |w| == E everywhere after eval()
                return _fused.dorefa_w1_conv_forward(input, self.weight, self.bias, args, True, wc,
                                                     self.padding_mode, scale=E)
        if (input.is_cuda and 2 <= self.bit_width <= _fused.LEVEL_INT8_BITS and input.dtype == torch.float32 and not self.training
                and self.weight.dtype == torch.float32 and self.groups == 1 and self.padding_mode == "zeros"
                and not (torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad))):
            wc = self._eval_planes(
                lambda _w2: _fused.ops.pack_conv_weight_dorefa_codes(self.weight.detach(), self.bit_width), key="conv_i8k")
            y = _fused.dorefa_wk_conv_forward(input, self.weight, self.bias, args, self.bit_width, wc, self.padding_mode)
            if y is not None:
                return y
        if (input.is_cuda and 2 <= self.bit_width <= _fused.LEVEL_MAX_BITS and input.dtype == torch.float32 and not self.training
                and self.weight.dtype == torch.float32 and self.groups == 1 and self.padding_mode == "zeros"
                and not (torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad))):
            # no usable int8 codes, or 8-bit weights: split activation x the exact level image (see LinearDorefa)
            terms = _fused.ops.split_terms()
            lp = self._eval_planes(
                lambda _w2: _fused.ops.pack_conv_weight_bf16x3(_fused._weight_levels(self.weight, self.bit_width), "raw", terms=terms),
                key=f"conv_levels{terms}")
            y = _fused.dorefa_levels_conv_forward(input, self.weight, self.bias, args, self.bit_width, lp)
            if y is not None:
                return y
        w = self.weight_op.forward(self.weight) if self.training else self.weight
        if (input.is_cuda and self.training and 2 <= self.bit_width <= _fused.LEVEL_MAX_BITS and input.dtype == torch.float32
                and self.weight.dtype == torch.float32 and self.groups == 1 and self.padding_mode == "zeros"):
            return _fused.DorefaWkConv2dFn.apply(input, w, self.bias, self.bit_width, args)
        if (input.is_cuda and input.dtype == torch.float32 and self.weight.dtype == torch.float32
                and input.dim() == 4 and input.numel() > 0 and self.groups == 1 and self.padding_mode == "zeros"
                and not isinstance(self.padding, str)):
            # real x real on the six-term planes: bit_width = 32, 8 < bit_width < 32, eval mode under autograd (see LinearDorefa)
            return _fused.RealConv2dFn.apply(input, w, self.bias, args)
        if input.is_cuda:
            _fused.note_library_path(input, "DoReFa conv with groups, a non-zero padding mode or a non-fp32 dtype")
        return torch.nn.functional.conv2d(input, w, self.bias, *args)
