"""Host logic of the functions/layers mirror on CPU tensors, modelled on the reference's own unit
tests (tests/implementations/{BinaryNet,Terner,Dorefa}/*.py: closed-form known answers, train/eval
swap protocol, STE gradients, statistics of the stochastic ops) and on the golden vectors."""
import warnings

import numpy as np
import pytest
import torch

from conftest import same, norm_err
from pytorch_quantize_impls_amd.functions import (BinaryConnectDeterministic, BinaryConnectStochastic,
                                                  BinaryConnect, BinaryDense, TernaryConnectDeterministic,
                                                  TernaryConnectStochastic, TernaryDense, nnDorefaQuant,
                                                  DorefaQuant, nnQuantWeight, safeSign, XNORDense,
                                                  nnQuantXnor)
from pytorch_quantize_impls_amd.functions.dorefa_connect import _quantize
from pytorch_quantize_impls_amd.functions import binary_connect, terner_connect
from pytorch_quantize_impls_amd.layers import (LinearBin, BinConv2d, LinearTer, TerConv2d, LinearDorefa,
                                               DorefaConv2d, LinearXNOR, XNORConv2d, QLayer)

T = torch.tensor


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


# ---- functions vs golden ---------------------------------------------------------------------

def test_safe_sign_and_binary_connect_golden(golden):
    x = t(golden["g1_edge_x"])
    assert same(safeSign(x).numpy(), golden["g1_safe_sign"])
    assert same(BinaryConnectDeterministic.apply(x).numpy(), golden["g1_bin_det_fwd"])
    xg = x.clone().requires_grad_(True)
    BinaryConnectDeterministic.apply(xg).backward(t(golden["g1_bwd_gout"]))
    assert same(xg.grad.numpy(), golden["g1_bin_det_bwd"])


def test_ternary_connect_golden(golden):
    x = t(golden["g2_x"])
    assert same(TernaryConnectDeterministic.apply(x).numpy(), golden["g2_ter_det_fwd"])
    xg = x.clone().requires_grad_(True)
    TernaryConnectDeterministic.apply(xg).backward(t(golden["g2_bwd_gout"]))
    assert same(xg.grad.numpy(), golden["g2_ter_det_bwd"])


@pytest.mark.parametrize("k", [1, 2, 3, 4, 8, 16, 25, 31, 32])
def test_dorefa_quantize_golden(golden, k):
    x = t(golden[f"g3_quant_x_k{k}"])
    assert same(_quantize(x, k).numpy(), golden[f"g3_quant_y_k{k}"])
    assert same(nnDorefaQuant(k)(x).numpy(), golden[f"g3_quant_y_k{k}"])
    assert same(DorefaQuant(x, k).numpy(), golden[f"g3_quant_y_k{k}"])


@pytest.mark.parametrize("k", [1, 2, 3, 4, 8, 32])
def test_dorefa_weight_golden(golden, k):
    w = t(golden["g3_wq_x"]).clone().requires_grad_(True)
    q = nnQuantWeight(k)(w)
    assert same(q.detach().numpy(), golden[f"g3_wq_y_k{k}"])
    (q * t(golden["g3_wq_gout"])).sum().backward()
    assert norm_err(w.grad.numpy(), golden[f"g3_wq_grad_k{k}"]) <= 1e-6
    assert same(nnQuantWeight(3)(torch.zeros(3, 2)).numpy(), golden["g3_wq_zero_k3"])


def test_stochastic_ops_with_injected_uniforms(golden):
    x, z = t(golden["g6_x"]), t(golden["g6_z"])
    assert same(binary_connect.stochastic_binarize(x, z).numpy(), golden["g6_bin_sto"])
    assert same(terner_connect.stochastic_ternarize(x, z).numpy(), golden["g6_ter_sto"])


def test_stochastic_statistics():
    # tests/implementations/BinaryNet/function_test.py:36-44
    w = T([[1., 0., -1.]])
    r = torch.cat([BinaryConnectStochastic.apply(w) for _ in range(151)], 0).mean(0)
    assert r[0] == 1 and r[2] == -1 and -0.2 < r[1] < 0.2
    # tests/implementations/Terner/function_test.py:30-46
    x = T([[1, 0, 0.45, -1, -0.9]])
    res = torch.cat([TernaryConnectStochastic.apply(x) for _ in range(1000)], 0)
    assert set(np.unique(res.numpy()).tolist()) <= {-1.0, 0.0, 1.0}
    assert torch.all((res.mean(0) - x[0]).abs() < 5e-2)


# ---- layers vs golden --------------------------------------------------------------------------

def _mk(fam, K, N, bias):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return {"bin": lambda: LinearBin(K, N, bias=bias), "ter": lambda: LinearTer(K, N, bias=bias),
                "dorefa1": lambda: LinearDorefa(K, N, bias=bias, bit_width=1),
                "dorefa3": lambda: LinearDorefa(K, N, bias=bias, bit_width=3),
                "xnor": lambda: LinearXNOR(K, N, bias=bias)}[fam]()


def test_linear_layers_forward_backward_golden(golden):
    for name in golden["g4_lin_cases"].tolist():
        g = lambda s: golden[f"g4_lin_{name}_{s}"]
        has_b = f"g4_lin_{name}_b" in golden.files
        x, w, gout = g("x"), g("w"), g("gout")
        for fam in ("bin", "ter", "dorefa1", "dorefa3", "xnor"):
            layer = _mk(fam, x.shape[1], w.shape[0], has_b)
            layer.weight.data.copy_(t(w))
            if has_b:
                layer.bias.data.copy_(t(g("b")))
            xi = t(x).clone().requires_grad_(True)
            y = layer(xi)
            y.backward(t(gout))
            if "_pm1_" in name and not has_b and fam in ("bin", "ter"):
                assert same(y.detach().numpy(), g(f"{fam}_y")), (name, fam)
            else:
                assert norm_err(y.detach().numpy(), g(f"{fam}_y")) <= 1e-5, (name, fam)
            assert norm_err(xi.grad.numpy(), g(f"{fam}_gx")) <= 1e-5, (name, fam)
            assert norm_err(layer.weight.grad.numpy(), g(f"{fam}_gw")) <= 1e-5, (name, fam)
            if has_b:
                assert norm_err(layer.bias.grad.numpy(), g(f"{fam}_gb")) <= 1e-5


def test_conv_layers_forward_golden(golden):
    for name in golden["g4_conv_cases"].tolist():
        p = name.split("_")
        Cin, Cout, k, st, pd = int(p[0][1:]), int(p[1][1:]), int(p[2][1:]), int(p[3][1:]), int(p[4][1:])
        has_b = p[7] == "bias"
        x, w = golden[f"g4_conv_{name}_x"], golden[f"g4_conv_{name}_w"]
        ctors = {"bin": lambda: BinConv2d(Cin, Cout, k, stride=st, padding=pd, bias=has_b),
                 "ter": lambda: TerConv2d(Cin, Cout, k, stride=st, padding=pd, bias=has_b),
                 "dorefa1": lambda: DorefaConv2d(Cin, Cout, k, stride=st, padding=pd, bias=has_b, bit_width=1),
                 "xnor": lambda: XNORConv2d(Cin, Cout, k, stride=st, padding=pd, bias=has_b)}
        for fam, ctor in ctors.items():
            layer = ctor()
            layer.weight.data.copy_(t(w))
            if has_b:
                layer.bias.data.copy_(t(golden[f"g4_conv_{name}_b"]))
            y = layer(t(x)).detach().numpy()
            assert norm_err(y, golden[f"g4_conv_{name}_{fam}_y"]) <= 1e-5, (name, fam)


def test_eval_swap_protocol_golden(golden):
    w, x = golden["g5_w"], golden["g5_x"]
    for fam, ctor in {"bin": lambda: LinearBin(9, 4, bias=False), "ter": lambda: LinearTer(9, 4, bias=False),
                      "dorefa3": lambda: LinearDorefa(9, 4, bias=False, bit_width=3)}.items():
        layer = ctor()
        layer.weight.data.copy_(t(w))
        assert norm_err(layer(t(x)).detach().numpy(), golden[f"g5_{fam}_y_train"]) <= 1e-6
        layer.train(False)
        assert same(layer.weight.data.numpy(), golden[f"g5_{fam}_w_eval"])
        assert hasattr(layer.weight, "org") and same(layer.weight.org.numpy(), w)
        assert norm_err(layer(t(x)).detach().numpy(), golden[f"g5_{fam}_y_eval"]) <= 1e-6
        layer.eval()  # no-op when already in eval mode
        layer.train(True)
        assert same(layer.weight.data.numpy(), golden[f"g5_{fam}_w_back"])


# ---- known answers lifted from the reference's tests (values, not code) ----------------------------

def test_known_answer_linear_bin():
    # BinaryNet/layer_test.py:16-21: sign([.5, 0, -.5]) . [2, 1, -3] = 2 + 1 + 3 = 6
    lin = LinearBin(3, 1, bias=False)
    lin.weight.data.copy_(T([[0.5, 0, -0.5]]))
    assert lin(T([[2., 1., -3.]])).item() == 6.0
    lin2 = LinearBin(3, 1, bias=True)
    lin2.weight.data.copy_(T([[0.5, 0, -0.5]]))
    lin2.bias.data.copy_(T([3.]))
    assert lin2(T([[2., 1., -3.]])).item() == 9.0


def test_known_answer_bin_conv():
    # BinaryNet/layer_test.py:37-63 (2x2 kernel over a 2x2x2 input)
    w = T([[0.5, -0.5], [-0.5, 0.5], [1, -1], [0.5, 0.5]]).view(1, 2, 2, 2)
    x = T([[1.1, 2.1], [15, .01], [1, 0], [1., 1.0]]).view(1, 2, 2, 2)
    conv = BinConv2d(2, 1, [2, 2], stride=1, bias=False)
    conv.weight.data.copy_(w)
    expect = torch.nn.functional.conv2d(x, torch.where(w < 0, -1.0, 1.0))
    assert torch.equal(conv(x), expect)
    conv.train(False)
    assert torch.equal(conv.weight.data, torch.where(w < 0, -1.0, 1.0))
    assert torch.equal(conv(x), expect)
    conv.train(True)
    assert torch.equal(conv.weight.data, w)


def test_ste_gradients_known_answers():
    # BinaryNet/function_test.py:49-84
    inputs = T([[2., -0.5, 1.]])
    w1 = T([[0.5], [-0.5], [0.5]]).requires_grad_(True)
    loss = inputs.mm(BinaryConnectDeterministic.apply(w1))
    assert loss.item() == 2 + 0.5 + 1
    loss.backward()
    assert torch.equal(w1.grad, inputs.view(3, 1))
    w2 = T([[2.], [0.5], [0.5]]).requires_grad_(True)
    inputs.mm(BinaryConnectDeterministic.apply(w2)).backward()
    assert torch.equal(w2.grad, T([[0.], [-0.5], [1.]]))
    w3 = T([[2.], [0.5], [0.5]]).requires_grad_(True)
    inputs.mm(BinaryConnectStochastic.apply(w3)).backward()
    assert w3.grad[0] == 0.0
    # Terner/function_test.py:82-94: gradient vanishes for |x| > 1
    x = T([2, 1.0, 0.0, -1, -3]).requires_grad_(True)
    TernaryConnectDeterministic.apply(x).sum().backward()
    assert x.grad.tolist() == [0, 1, 1, 1, 0]
    # Terner/layer_test.py:247-271: LinearTer weight grad is zero where |w| > 1
    layer = LinearTer(4, 1)
    layer.weight.data.copy_(T([[1., 4., 0., 3.]]))
    (layer(torch.randn(5, 4)) ** 2).sum().backward()
    assert layer.weight.grad[0, 1] == 0 and layer.weight.grad[0, 3] == 0


def test_binary_dense_plain_backward():
    # BinaryDense backward has NO STE mask (binary_connect.py:104-112), unlike LinearBin
    x = torch.randn(4, 6, requires_grad=True)
    w = (torch.randn(3, 6) * 2).requires_grad_(True)
    b = torch.randn(3, requires_grad=True)
    g = torch.randn(4, 3)
    BinaryDense.apply(x, w, b).backward(g)
    assert torch.allclose(w.grad, g.t().mm(x.detach()))
    assert torch.allclose(x.grad, g.mm(torch.where(w.detach() < 0, -1.0, 1.0)))
    assert torch.allclose(b.grad, g.sum(0))


def test_api_surface_and_errors():
    assert issubclass(LinearBin, QLayer) and issubclass(LinearBin, torch.nn.Linear)
    assert issubclass(BinConv2d, torch.nn.Conv2d)
    with pytest.raises(TypeError, match="Expected a torch.nn.Linear"):
        LinearBin.convert(torch.nn.Conv2d(1, 1, 1))
    with pytest.raises(TypeError, match="Expected a torch.nn.Conv2d"):
        BinConv2d.convert(torch.nn.Linear(1, 1))
    conv = BinConv2d.convert(torch.nn.Conv2d(3, 8, 3, stride=2, padding=1, bias=False))
    assert (conv.in_channels, conv.out_channels, conv.stride, conv.padding, conv.bias) == (3, 8, (2, 2), (1, 1), None)
    lin = LinearTer.convert(torch.nn.Linear(5, 2))
    assert isinstance(lin, LinearTer) and lin.bias is not None
    assert isinstance(LinearDorefa.convert(torch.nn.Linear(5, 2), bit_width=2), LinearDorefa)
    with pytest.raises(RuntimeError):
        nnQuantXnor(dim=3)
    with pytest.warns(DeprecationWarning):
        binary_connect.BinaryConv2d()
    assert isinstance(BinaryConnect(), torch.nn.Module)
    lin = LinearBin(8, 4)
    assert float(lin.bias.abs().sum()) == 0.0            # reset_parameters: bias 0
    lin.weight.data.mul_(100)
    lin.bias.data.fill_(5)
    lin.clamp()
    assert float(lin.weight.abs().max()) <= 1 and float(lin.bias.max()) == 1  # bias clamped too
    seq = torch.nn.Sequential(LinearBin(8, 4), BinaryConnect(), LinearBin(4, 2))
    assert seq(torch.randn(3, 8)).shape == (3, 2)


def test_xnor_layers_numerics_and_fixed_eval():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lin = LinearXNOR(6, 3, bias=False)
    w = torch.randn(3, 6)
    lin.weight.data.copy_(w)
    x = torch.randn(4, 6)
    alpha = w.abs().mean(0, keepdim=True)           # per INPUT feature (global DIM = 0 upstream)
    assert torch.allclose(lin(x), x.mm((torch.sign(w) * alpha).t()), atol=1e-6)
    lin.train(False)                                 # upstream raises NameError here; fixed
    assert torch.allclose(lin.weight.data, torch.sign(w) * alpha)
    lin.train(True)
    assert torch.equal(lin.weight.data, w)
    conv = XNORConv2d(4, 2, 3, padding=1, bias=False)
    wc = torch.randn(2, 4, 3, 3)
    conv.weight.data.copy_(wc)
    xc = torch.randn(1, 4, 5, 5)
    a = wc.abs().mean((0, 1), keepdim=True)
    assert torch.allclose(conv(xc), torch.nn.functional.conv2d(xc, torch.sign(wc) * a, padding=1), atol=1e-5)
    conv.train(False)
    assert torch.allclose(conv.weight.data, torch.sign(wc) * a)


def test_fused_feature_classifier_validates_its_inputs():
    import torch.nn as nn
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    from pytorch_quantize_impls_amd.layers import BinConv2d, FusedFeatureClassifier, LinearBin
    feats = nn.Sequential(BinConv2d(3, 8, 3, padding=1), nn.MaxPool2d(2, 2), nn.BatchNorm2d(8), nn.Hardtanh()).eval()
    clf = nn.Sequential(BinaryConnect(stochastic=False), LinearBin(8 * 4 * 4, 10)).eval()
    m = FusedFeatureClassifier(feats, clf, (8, 4, 4))
    assert type(m.last).__name__ == "FusedConvPoolBnSign" and len(m.features) == 1
    w = clf[1].weight.data.view(10, 8, 4, 4).permute(0, 2, 3, 1).reshape(10, -1)
    assert torch.equal(m.classifier[0].weight.data, torch.where(w < 0, -1.0, 1.0))       # (h, w, c) columns, quantised image
    with pytest.raises(ValueError, match="feat_chw"):
        FusedFeatureClassifier(feats, clf, (8, 3, 3))
    with pytest.raises(ValueError, match="has to produce sign bits"):       # no BinaryConnect anywhere after the BN
        FusedFeatureClassifier(feats, nn.Sequential(LinearBin(128, 10)).eval(), (8, 4, 4))
    with pytest.raises(ValueError, match="must start with"):
        FusedFeatureClassifier(feats, nn.Sequential(BinaryConnect(stochastic=False), nn.Linear(128, 10)).eval(), (8, 4, 4))
    with pytest.raises(ValueError, match="has to produce sign bits"):
        FusedFeatureClassifier(nn.Sequential(BinConv2d(3, 8, 3)).eval(), clf, (8, 4, 4))
    # VGG style: sign before the pool, classifier without a leading BinaryConnect
    vgg = nn.Sequential(BinConv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.Hardtanh(), BinaryConnect(stochastic=False),
                        nn.MaxPool2d(2, 2)).eval()
    mv = FusedFeatureClassifier(vgg, nn.Sequential(LinearBin(128, 10)).eval(), (8, 4, 4))
    assert [type(x).__name__ for x in mv.features] == ["FusedConvPoolBnSign", "PackedMaxPool"]
    with pytest.raises(ValueError, match="eval-mode"):
        FusedFeatureClassifier(feats.train(), clf, (8, 4, 4))


def test_fused_module_host_logic():
    """Host-side logic of the inference-fusion modules that needs no GPU: operand hand-over links, ReLU placement
    parsing, shape validation of code / nibble activations, and that CPU tensors are refused loudly."""
    import pytest
    import torch
    from pytorch_quantize_impls_amd import ops, packed
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    from pytorch_quantize_impls_amd.layers import (BinConv2d, CodeMaxPool, FusedBnDorefaQuant, FusedConvPoolBnSign,
                                                   PackedMaxPool, fuse_sequential)
    seq = torch.nn.Sequential(
        BinConv2d(32, 64, 3, padding=1), torch.nn.BatchNorm2d(64), torch.nn.Hardtanh(), BinaryConnect(stochastic=False),
        BinConv2d(64, 64, 5, padding=(2, 1)), torch.nn.BatchNorm2d(64), torch.nn.Hardtanh(), BinaryConnect(stochastic=False),
        torch.nn.MaxPool2d(2, 2),
        BinConv2d(64, 32, 3, padding=0), torch.nn.MaxPool2d(2, 2), torch.nn.BatchNorm2d(32), torch.nn.Hardtanh(),
        BinaryConnect(stochastic=False)).eval()
    fused = list(fuse_sequential(seq, fuse_conv=True, packed_pool=True))
    assert [type(m) for m in fused] == [FusedConvPoolBnSign, FusedConvPoolBnSign, PackedMaxPool, FusedConvPoolBnSign]
    # each producer is told the padding of the fused conv that consumes it; the last block has no consumer
    assert fused[0].out_nib_halo == (2, 1)          # feeds the 5x5 conv padded (2, 1)
    assert fused[1].out_nib_halo is None            # feeds a PackedMaxPool, which needs bit planes
    assert fused[2].out_nib_halo == (0, 0)          # the pool feeds the un-padded 3x3 conv
    assert fused[3].out_nib_halo is None
    assert [ops.relu_mode(v) for v in (False, None, True, "post", "pre")] == [0, 0, 1, 1, 2]
    with pytest.raises(ValueError, match="relu must be"):
        ops.relu_mode("around")
    # activations validate their geometry
    codes = ops.CodePlanes(codes=torch.zeros((2 * 6 * 7, 16), dtype=torch.int8), rows=2 * 6 * 7, K=12)
    with pytest.raises(ValueError, match="needs"):
        packed.CodeActivation(codes, (2, 12, 4, 5))                 # 2*4*5 pixels expected, plane holds 2*6*7
    act = packed.CodeActivation(codes, (2, 12, 4, 5), halo=(1, 1))  # ... which is the same image with a 1-pixel halo
    assert act.without_halo().codes.rows == 2 * 4 * 5 and act.without_halo().halo == (0, 0)
    with pytest.raises(ValueError, match="either bit planes or a nibble"):
        packed.PackedActivation(None, (1, 8, 2, 2))
    with pytest.raises(ValueError, match="only un-padded"):
        CodeMaxPool(torch.nn.MaxPool2d(3, 2, padding=1))
    # no CPU fallback: the fused quantiser refuses host tensors instead of computing something else
    q = FusedBnDorefaQuant(torch.nn.BatchNorm2d(12).eval(), 4)
    with pytest.raises(TypeError, match="HIP device"):
        q(torch.zeros(1, 12, 3, 3))


def test_integer_thresholds_reproduce_the_float_predicate_exactly():
    """ops.integer_thresholds: for every integer accumulator value the one-compare form (acc < T) xor (alpha < 0) equals
    the kernel's float predicate fl(fl(acc + bias) * alpha) < -beta, brute force over the whole accumulator range,
    including alpha <= 0, always / never cases, huge offsets and NaN parameters."""
    import torch
    from pytorch_quantize_impls_amd import ops
    torch.manual_seed(3)
    C, kmax = 64, 300
    alpha = torch.randn(C) * 0.05
    beta = torch.randn(C) * 3
    bias = torch.randn(C) * 5
    alpha[0], alpha[1], beta[1] = 0.0, 0.0, -1.0                    # constant predicates
    alpha[2], beta[2] = 1e-3, 1e6                                    # never
    alpha[3], beta[3] = -1e-3, 1e6                                   # never (negative slope)
    alpha[4], beta[4] = 2.0, -1e6                                    # always
    alpha[5] = float("nan")
    beta[6] = float("nan")
    bias[7] = 1e9                                                    # acc is absorbed by the bias: a constant again
    alpha[8], beta[8], bias[8] = 0.37, -0.37 * 12.5, 0.0             # boundary between two integers
    alpha[9], beta[9], bias[9] = -1.0, 7.0, 0.0                      # exact tie at acc == 7
    for b in (bias, None):
        thr = ops.integer_thresholds(b, alpha, beta, kmax)
        bb = torch.zeros(C) if b is None else b
        k = torch.arange(-kmax, kmax + 1, dtype=torch.float32).view(-1, 1)
        want = (k + bb) * alpha < -beta
        got = (k < thr) ^ (alpha < 0)
        assert torch.equal(got, want)


def test_output_blocked_first_layer_weight_is_the_same_conv():
    """ops.d2s_first_layer_weight: the 4x4 / stride-2 / padding-1 conv with the four shifted copies of a 3x3 kernel,
    followed by depth-to-space, IS the 3x3 / stride-1 / padding-1 conv; and its space-to-depth form (ops.s2d_input /
    ops.s2d_weight with s = 2: 2x2 taps over 4C channels) is the same conv again (integer-valued inputs, so equality is exact)."""
    import torch
    from pytorch_quantize_impls_amd import ops
    torch.manual_seed(4)
    N, C, H, W, Cout = 2, 3, 8, 6, 5
    x = torch.randint(-50, 50, (N, C, H, W)).to(torch.float64)          # integer values: every summation order is exact
    w = torch.randn(Cout, C, 3, 3, dtype=torch.float64).sign()
    want = torch.nn.functional.conv2d(x, w, None, 1, 1)
    w4 = ops.d2s_first_layer_weight(w)
    assert tuple(w4.shape) == (4 * Cout, C, 4, 4)
    y4 = torch.nn.functional.conv2d(x, w4, None, 2, 1)                     # [N, (dy, dx, co), H/2, W/2]
    got = y4.view(N, 2, 2, Cout, H // 2, W // 2).permute(0, 3, 4, 1, 5, 2).reshape(N, Cout, H, W)
    assert torch.equal(got, want)
    ys = torch.nn.functional.conv2d(ops.s2d_input(x, 2, 1), ops.s2d_weight(w4, 2), None, 1, 0)
    assert torch.equal(ys, y4)
    assert ops.d2s_first_layer_applicable(3, 64, (3, 3), 1, 1, 1, 224, 224)
    assert not ops.d2s_first_layer_applicable(3, 64, (3, 3), 1, 1, 1, 223, 224)      # odd size: no 2x2 blocking
    assert not ops.d2s_first_layer_applicable(3, 48, (3, 3), 1, 1, 1, 224, 224)      # channel groups of 32 in the epilogue


def test_no_bare_dense_library_call_in_the_functional_forms():
    """every hipBLASLt / MIOpen call a backward can make goes through functions/_fused.py, where it is counted
    (LIBRARY_PATHS); the Function classes themselves contain no `.mm(` / `torch.nn.grad.` / F.linear-on-a-device-tensor shortcut."""
    import glob
    import os
    import re
    import pytorch_quantize_impls_amd.functions as F_
    root = os.path.dirname(F_.__file__)
    bad = []
    for path in sorted(glob.glob(os.path.join(root, "*.py"))):
        if os.path.basename(path) == "_fused.py":
            continue
        text = open(path, encoding="utf-8").read()
        code = "\n".join(ln.split("#", 1)[0] for ln in text.split("\n"))        # comments stripped (docstrings cite upstream lines)
        code = re.sub(r'"""(.|\n)*?"""', "", code)
        for pat in (r"\.mm\(", r"torch\.nn\.grad\."):
            for m in re.finditer(pat, code):
                bad.append((os.path.basename(path), m.group(0)))
    assert not bad, bad


def test_every_dense_library_call_of_the_layers_is_counted():
    """a layer that falls back to F.linear / F.conv2d on a device tensor says so (note_library_path within the
    three code lines above the call), so _fused.LIBRARY_PATHS sees every hipBLASLt / MIOpen forward the layers can make."""
    import glob
    import os
    import re
    import pytorch_quantize_impls_amd.layers as L_
    root = os.path.dirname(L_.__file__)
    bad, seen = [], 0
    for path in sorted(glob.glob(os.path.join(root, "*.py"))):
        text = open(path, encoding="utf-8").read()
        text = re.sub(r'"""(.|\n)*?"""', lambda m: "\n" * m.group(0).count("\n"), text)       # docstrings cite upstream lines
        lines = [ln.split("#", 1)[0].rstrip() for ln in text.split("\n")]
        code = [(i, ln) for i, ln in enumerate(lines) if ln.strip()]
        for j, (i, ln) in enumerate(code):
            if re.search(r"(functional|\bF)\.(linear|conv2d)\(", ln):
                seen += 1
                before = " ".join(c for _, c in code[max(0, j - 3):j])
                if "note_library_path" not in before:
                    bad.append((os.path.basename(path), i + 1))
    assert seen >= 10 and not bad, bad
