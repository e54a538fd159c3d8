"""Lin / Log fixed-point family: oracle vs the reference's golden vectors (G9), CPU layers vs golden."""
import os

import numpy as np
import pytest
import torch

from pytorch_quantize_impls_amd.functions import LinQuant, LogQuant, Quant, nnQuant
from pytorch_quantize_impls_amd.layers import LinearQuant, QuantConv2d

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g9():
    return np.load(os.path.join(HERE, "golden", "golden_loglin_v1.npz"))


def same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32) | (np.isnan(a) * 0x7fffffff).astype(np.uint32),
                                                  b.view(np.uint32) | (np.isnan(b) * 0x7fffffff).astype(np.uint32))


def test_oracle_lin_log_vs_reference_vectors(oracle, g9):
    for fsr, bits in g9["g9_cfgs"].tolist():
        for sign in (1, 0):
            for nm in ("edge", "rand"):
                x = g9[f"g9_{nm}"]
                tag = f"f{fsr}_b{bits}_s{sign}_{nm}"
                assert same(oracle.lin_quantize(x, fsr, bits, sign), g9[f"g9_lin_{tag}"]), ("lin", tag)
                assert same(oracle.log_quantize(x, fsr, bits, bool(sign)), g9[f"g9_log_{tag}"]), ("log", tag)
        for nm in ("edge", "rand"):
            assert same(oracle.log_quantize(g9[f"g9_{nm}"], fsr, bits, True), g9[f"g9_logbwd_f{fsr}_b{bits}_{nm}"])
    assert same(oracle.lin_quantize(g9["g9_edge"], 7, 32, 1), g9["g9_lin_f7_b32_s1_edge"])


def test_cpu_functions_vs_reference_vectors(g9):
    for fsr, bits in g9["g9_cfgs"].tolist():
        for sign in (True, False):
            x = torch.from_numpy(g9["g9_rand"])
            tag = f"f{fsr}_b{bits}_s{int(sign)}_rand"
            assert same(LinQuant(fsr, bits, sign).apply(x).numpy(), g9[f"g9_lin_{tag}"])
            assert same(LogQuant(fsr, bits, sign).apply(x).numpy(), g9[f"g9_log_{tag}"])
            assert same(Quant(x, "log", fsr, bits, sign).numpy(), g9[f"g9_log_{tag}"])
            assert same(nnQuant("lin", fsr, bits, sign)(x).numpy(), g9[f"g9_lin_{tag}"])
    with pytest.raises(RuntimeError, match="Only 'log' and 'lin'"):
        nnQuant("exp")
    # quantised-gradient backward: log = golden; lin = the oracle's restatement of the intended expression
    g = torch.from_numpy(g9["g9_rand"])
    xin = torch.ones_like(g, requires_grad=True)
    LogQuant(7, 3, True, lin_back=False).apply(xin).backward(g)
    assert same(xin.grad.numpy(), g9["g9_logbwd_f7_b3_rand"])


def test_lin_quantised_gradient_backward_vs_oracle(oracle, g9):
    g = torch.from_numpy(g9["g9_rand"])
    xin = torch.ones_like(g, requires_grad=True)
    LinQuant(2, 3, True, lin_back=False).apply(xin).backward(g)      # upstream crashes here under torch 2.x
    want = oracle.lin_quantize(g9["g9_rand"], 2, 3, 2)
    assert same(xin.grad.numpy(), want)
    assert not want[g9["g9_rand"] < 0].any()                       # negative g clamps to zero (signed zero)


def test_cpu_layers_vs_reference_vectors(g9):
    for name in g9["g9_layer_cases"].tolist():
        p = name.split("_")
        dtype = p[1]
        w, b, x = (torch.from_numpy(g9[f"g9_{name}_{k}"]) for k in ("w", "b", "x"))
        fsr, bits = int(p[-2][1:]), int(p[-1][1:])
        if name.startswith("lin_"):
            layer = LinearQuant(w.shape[1], w.shape[0], True, dtype=dtype, fsr=fsr, bit_width=bits)
        else:
            layer = QuantConv2d(w.shape[1], w.shape[0], w.shape[2], stride=int(p[5][1:]), padding=int(p[6][1:]),
                                bias=True, fsr=fsr, bit_width=bits, dtype=dtype)
        layer.weight.data.copy_(w); layer.bias.data.copy_(b)
        xi = x.clone().requires_grad_(True)
        y = layer(xi)
        y.backward(torch.from_numpy(g9[f"g9_{name}_gout"]))
        tol = lambda a, ref: np.abs(a - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)
        assert tol(y.detach().numpy(), g9[f"g9_{name}_y"]), name
        assert tol(xi.grad.numpy(), g9[f"g9_{name}_gx"]) and tol(layer.weight.grad.numpy(), g9[f"g9_{name}_gw"]), name
        layer.train(False)
        assert tol(layer(x).detach().numpy(), g9[f"g9_{name}_y_eval"]), name
        if name.startswith("lin_"):
            assert np.array_equal(layer.weight.data.numpy(), g9[f"g9_{name}_w_eval"])
        layer.train(True)
        assert torch.equal(layer.weight.data, w)                     # eval swap restored the real weight


def test_layer_plumbing():
    lin = LinearQuant.convert(torch.nn.Linear(6, 4, bias=False), dtype="log", fsr=2, bit_width=3)
    assert isinstance(lin, LinearQuant) and lin.bias is None and lin.qdtype == "log"
    mag = lin.weight.detach().abs()
    assert float(mag.min()) >= 2 ** (2 - 3) and float(mag.max()) <= 2 ** 2 and (lin.weight < 0).any() and (lin.weight > 0).any()
    with pytest.raises(TypeError, match="Expected a torch.nn.Conv2d"):
        QuantConv2d.convert(torch.nn.Linear(2, 2))
    conv = QuantConv2d.convert(torch.nn.Conv2d(3, 5, 3, stride=2, padding=1), fsr=1, bit_width=2, dtype="lin")
    assert conv.stride == (2, 2) and conv.bit_width == 2
    conv.weight.data.fill_(100.0); conv.clamp()
    assert float(conv.weight.detach().max()) == 2.0


def test_ap2_oracle_and_cpu_function_vs_reference_vector(oracle, g9):
    from pytorch_quantize_impls_amd.functions import AP2
    x = g9["g9_ap2_in"]
    assert same(oracle.ap2(x), g9["g9_ap2_out"])
    assert same(AP2(torch.from_numpy(x)).numpy(), g9["g9_ap2_out"])
