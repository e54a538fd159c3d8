"""Round-4 GPU parity, part c: DoReFa layers beyond the int8 code range (VERDICT r3 "missing" item 5).

  * bit_width = 8: the weight levels c = 255 w_q are odd integers up to 255 — not int8, but exact in fp16 and bf16 — so the layer
    runs on the split-activation x exact-level-image routes, forward and both gradients, training and eval mode
    (layers/dorefa_layers.py:41-45,77-82; functions/dorefa_connect.py:99-111);
  * eval-mode k-bit layers on an activation WITHOUT usable int8 codes (a real-valued input, codes beyond +-127): the same routes
    instead of the dense library;
  * bit_width = 32: the identity quantiser (functions/dorefa_connect.py:19-20,100-101) — two real operands on the six-term planes.
Every case against the fp64 evaluation of the reference expression F.linear / F.conv2d(x, weight_op(W), b), <= 1e-5 normalised
(SURVEY 8d), with _fused.LIBRARY_PATHS watched."""
import copy

import pytest
import torch

from conftest import norm_err

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import _lib  # noqa: E402
from pytorch_quantize_impls_amd.functions import _fused, nnDorefaQuant  # noqa: E402
from pytorch_quantize_impls_amd.layers import DorefaConv2d, LinearDorefa  # noqa: E402

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture()
def all_shapes_on_the_routes():
    old = _fused.BWD_MFMA_MIN_MACS
    _fused.BWD_MFMA_MIN_MACS = 0
    yield
    _fused.BWD_MFMA_MIN_MACS = old


def n(t):
    return t.detach().cpu().numpy()


def _lib_delta(before):
    return {k: v - before.get(k, 0) for k, v in _fused.LIBRARY_PATHS.items() if v != before.get(k, 0)}


def _fp64_layer(layer, x, gout=None):
    """The reference expression in fp64 on the CPU: training mode quantises the weight, eval mode uses the stored image.
    The quantised VALUES are the device layer's own fp32 image (the elementwise quantiser is pinned bit for bit elsewhere; with
    255 levels an fp64 evaluation of tanh / max|tanh| rounds a handful of the 36 864 weights of a 3 x 3 layer to the neighbouring
    level, 2 / 255 away — a difference of the reference's precision, not of this backend); the gradient still flows through the
    fp64 quantiser graph (straight-through: it does not depend on the rounding)."""
    ref = copy.deepcopy(layer).cpu().double()
    xr = x.detach().cpu().double().requires_grad_(gout is not None)
    if layer.training:
        w = ref.weight_op.forward(ref.weight)
        with torch.no_grad():
            wq_dev = layer.weight_op.forward(layer.weight).detach().cpu().double()
        w = w + (wq_dev - w).detach()
    else:
        w = ref.weight
    if isinstance(ref, torch.nn.Conv2d):
        y = torch.nn.functional.conv2d(xr, w, ref.bias, ref.stride, ref.padding, ref.dilation, ref.groups)
    else:
        y = torch.nn.functional.linear(xr, w, ref.bias)
    if gout is None:
        return y.detach()
    y.backward(gout.detach().cpu().double())
    return y.detach(), xr.grad, ref.weight.grad, (ref.bias.grad if ref.bias is not None else None)


# ---- bit_width = 8 -------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("coded", [True, False])
@pytest.mark.parametrize("Cin,Cout,ksz,s,p,H", [(64, 64, 3, 1, 1, 16), (64, 128, 3, 2, 1, 16), (64, 128, 1, 2, 0, 16)])
def test_dorefa_8bit_conv_training_vs_fp64(dev, all_shapes_on_the_routes, coded, Cin, Cout, ksz, s, p, H):
    torch.manual_seed(80 + Cin + ksz + s)
    conv = DorefaConv2d(Cin, Cout, ksz, stride=s, padding=p, bias=True, bit_width=8).to(dev).train()
    conv.weight.data.normal_(0, 0.7)
    raw = (torch.rand(6, Cin, H, H, device=dev) * 1.4).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x = nnDorefaQuant(4)(raw) if coded else raw * 1.0
    x.retain_grad()
    lib_before = dict(_fused.LIBRARY_PATHS)
    before = dict(_lib.call_counts)
    y = conv(x)
    gout = torch.randn_like(y)
    y.backward(gout)
    lib_now = _lib_delta(lib_before)
    if coded:
        assert not lib_now, lib_now
    else:
        assert set(lib_now) <= {"conv grad_weight outside the matrix-core route"}, lib_now
    assert _lib.call_counts["qt_conv2d_implicit"] - before.get("qt_conv2d_implicit", 0) >= 2     # forward + grad_x
    # the levels really leave int8: the route under test is not the int8 one
    lv = torch.round(conv.weight_op.forward(conv.weight).detach() * 255.0)
    assert float(lv.abs().max()) == 255.0 and bool((torch.remainder(lv, 2) == 1).all())
    ry, rgx, rgw, rgb = _fp64_layer(conv, x, gout)
    assert norm_err(n(y), n(ry)) <= TOL
    assert norm_err(n(x.grad), n(rgx)) <= TOL
    assert norm_err(n(conv.weight.grad), n(rgw)) <= 2 * TOL        # through tanh / max|tanh| of the weight quantiser in fp32
    assert norm_err(n(conv.bias.grad), n(rgb)) <= TOL


@pytest.mark.parametrize("coded", [True, False])
def test_dorefa_8bit_linear_training_vs_fp64(dev, all_shapes_on_the_routes, coded):
    torch.manual_seed(88)
    lin = LinearDorefa(300, 70, bias=True, bit_width=8).to(dev).train()
    lin.weight.data.normal_(0, 0.7)
    raw = (torch.rand(96, 300, device=dev) * 1.3).requires_grad_(True)
    x = nnDorefaQuant(3)(raw) if coded else raw * 1.0
    x.retain_grad()
    lib_before = dict(_fused.LIBRARY_PATHS)
    y = lin(x)
    gout = torch.randn_like(y)
    y.backward(gout)
    lib_now = _lib_delta(lib_before)
    assert not lib_now, lib_now
    ry, rgx, rgw, rgb = _fp64_layer(lin, x, gout)
    assert norm_err(n(y), n(ry)) <= TOL
    assert norm_err(n(x.grad), n(rgx)) <= TOL
    assert norm_err(n(lin.weight.grad), n(rgw)) <= 2 * TOL
    assert norm_err(n(lin.bias.grad), n(rgb)) <= TOL


@pytest.mark.parametrize("k_w", [3, 8])
@pytest.mark.parametrize("act", ["real", "codes", "beyond_int8"])
def test_dorefa_kbit_eval_layers_without_int8_codes_vs_fp64(dev, k_w, act):
    """Eval mode (weight holds the quantised image): k <= 7 on a coded activation keeps the int8 route; a real-valued activation,
    codes beyond +-127 and 8-bit weights take the level routes — none of them the dense library."""
    torch.manual_seed(k_w)
    conv = DorefaConv2d(64, 96, 3, stride=1, padding=1, bias=True, bit_width=k_w).to(dev)
    lin = LinearDorefa(320, 72, bias=True, bit_width=k_w).to(dev)
    conv.weight.data.normal_(0, 0.7)
    lin.weight.data.normal_(0, 0.7)
    conv.eval()
    lin.eval()
    scale = {"real": 1.3, "codes": 1.3, "beyond_int8": 40.0}[act]
    xc = (torch.rand(4, 64, 12, 12, device=dev) * scale).contiguous(memory_format=torch.channels_last)
    xl = torch.rand(64, 320, device=dev) * scale
    if act != "real":
        xc, xl = nnDorefaQuant(4)(xc), nnDorefaQuant(4)(xl)
    lib_before = dict(_fused.LIBRARY_PATHS)
    before = dict(_lib.call_counts)
    with torch.no_grad():
        yc, yl = conv(xc), lin(xl)
        yc2, yl2 = conv(xc), lin(xl)                     # second call: cached level planes
    assert not _lib_delta(lib_before), _lib_delta(lib_before)
    if k_w <= 7 and act == "codes":                  # the int8 route: no split of the activation
        assert all(_lib.call_counts.get(k, 0) == before.get(k, 0) for k in ("qt_f16x2_pack_f32", "qt_bf16x3_pack_f32"))
    else:
        assert any(_lib.call_counts.get(k, 0) > before.get(k, 0) for k in ("qt_f16x2_pack_f32", "qt_bf16x3_pack_f32"))
    assert torch.equal(yc, yc2) and torch.equal(yl, yl2)
    assert norm_err(n(yc), n(_fp64_layer(conv, xc))) <= TOL
    assert norm_err(n(yl), n(_fp64_layer(lin, xl))) <= TOL


# ---- bit_width = 32: the identity quantiser --------------------------------------------------------------------------------------------

@pytest.mark.parametrize("training", [True, False])
def test_dorefa_32bit_layers_vs_fp64(dev, training):
    torch.manual_seed(32)
    conv = DorefaConv2d(32, 48, 3, stride=1, padding=1, bias=True, bit_width=32).to(dev).train(training)
    lin = LinearDorefa(200, 40, bias=True, bit_width=32).to(dev).train(training)
    xc = torch.randn(4, 32, 10, 10, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xl = torch.randn(48, 200, device=dev).requires_grad_(True)
    lib_before = dict(_fused.LIBRARY_PATHS)
    yc, yl = conv(xc), lin(xl)
    gc, gl = torch.randn_like(yc), torch.randn_like(yl)
    yc.backward(gc)
    yl.backward(gl)
    lib_now = _lib_delta(lib_before)
    # the one contraction of two real operands without an own route: the conv's weight gradient (counted, not hidden)
    assert set(lib_now) <= {"conv grad_weight outside the matrix-core route"}, lib_now
    for layer, x, y, g in ((conv, xc, yc, gc), (lin, xl, yl, gl)):
        ry, rgx, rgw, rgb = _fp64_layer(layer.train(), x, g)
        layer.train(training)
        assert norm_err(n(y), n(ry)) <= TOL
        assert norm_err(n(x.grad), n(rgx)) <= TOL
        assert norm_err(n(layer.weight.grad), n(rgw)) <= TOL
        assert norm_err(n(layer.bias.grad), n(rgb)) <= TOL
