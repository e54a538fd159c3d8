"""Pin the round-2 oracle restatements (XNOR activation op, functional ternary / DoReFa weight formulas) against
vectors produced by the reference itself (tests/golden/make_golden_r2.py), and the package's CPU branch of the same
functions against the same vectors.  Float tails: max|a-b| / max|b| <= 1e-5."""
import warnings

import numpy as np
import pytest
import torch

from conftest import same, norm_err

TOL = 1e-5


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


@pytest.mark.parametrize("dim", [-1, 0, 1])
def test_g10_xnor_act_oracle_and_cpu_branch(golden_r2, oracle, dim):
    from pytorch_quantize_impls_amd.functions.xnor_connect import QuantXnor
    for tag in golden_r2["g10_cases"]:
        x, g = golden_r2[f"g10_{tag}_x"], golden_r2[f"g10_{tag}_g"]
        y_ref, gx_ref = golden_r2[f"g10_{tag}_d{dim}_y"], golden_r2[f"g10_{tag}_d{dim}_gx"]
        y, _ = oracle.xnor_act(x, dim)
        assert norm_err(y, y_ref) <= TOL, (tag, dim)
        assert np.array_equal(y == 0, y_ref == 0)          # torch.sign(0) = 0 survives
        assert norm_err(oracle.xnor_act_backward(g, x, dim), gx_ref) <= TOL, (tag, dim)
        xi = _t(x).requires_grad_(True)
        yc = QuantXnor(xi, dim=dim)
        yc.backward(_t(g))
        assert norm_err(yc.detach().numpy(), y_ref) <= TOL and norm_err(xi.grad.numpy(), gx_ref) <= TOL


def test_g11_functional_weight_formulas(golden_r2, oracle):
    """The weight image each functional form contracts with, recovered from the reference's grad_input = g . W_q."""
    for tag in golden_r2["g11_lin_cases"]:
        w = golden_r2[f"g11_lin_{tag}_w"]
        g = golden_r2[f"g11_lin_{tag}_g"].astype(np.float64)
        for name, wq in (("ter", oracle.functional_ternary_weight(w)), ("q1", oracle.functional_quant_weight(w, 1)),
                         ("q3", oracle.functional_quant_weight(w, 3)), ("q32", oracle.functional_quant_weight(w, 32))):
            assert norm_err(g @ wq.astype(np.float64), golden_r2[f"g11_lin_{tag}_{name}_gx"]) <= TOL, (tag, name)
    wt = oracle.functional_ternary_weight(np.array([0.5, -0.5, 0.0, -0.0, 0.49999997, -0.50000006, 0.7, -2.0], np.float32))
    assert wt.tolist() == [0.5, -0.5, 0.0, 0.0, 0.0, -1.0, 1.0, -1.0]


@pytest.mark.parametrize("name", ["ter", "q1", "q3", "q32"])
def test_g11_functional_linear_cpu_branch(golden_r2, oracle, name):
    from pytorch_quantize_impls_amd.functions.dorefa_connect import QuantDense
    from pytorch_quantize_impls_amd.functions.terner_connect import TernaryDense
    op = {"ter": lambda: TernaryDense(stochastic=False), "q1": lambda: QuantDense(1), "q3": lambda: QuantDense(3),
          "q32": lambda: QuantDense(32)}[name]()
    for tag in golden_r2["g11_lin_cases"]:
        x, w, g = (golden_r2[f"g11_lin_{tag}_{k}"] for k in ("x", "w", "g"))
        has_b = f"g11_lin_{tag}_b" in golden_r2.files
        xi, wi = _t(x).requires_grad_(True), _t(w).requires_grad_(True)
        bi = _t(golden_r2[f"g11_lin_{tag}_b"]).requires_grad_(True) if has_b else None
        y = op.apply(xi, wi, bi) if has_b else op.apply(xi, wi)
        y.backward(_t(g))
        assert norm_err(y.detach().numpy(), golden_r2[f"g11_lin_{tag}_{name}_y"]) <= TOL
        assert norm_err(xi.grad.numpy(), golden_r2[f"g11_lin_{tag}_{name}_gx"]) <= TOL
        assert norm_err(wi.grad.numpy(), golden_r2[f"g11_lin_{tag}_{name}_gw"]) <= TOL
        # oracle: the forward is the fp64 contraction with the restated weight image
        wq = {"ter": oracle.functional_ternary_weight(w), "q1": oracle.functional_quant_weight(w, 1),
              "q3": oracle.functional_quant_weight(w, 3), "q32": oracle.functional_quant_weight(w, 32)}[name]
        yo = oracle.linear(x, wq, golden_r2[f"g11_lin_{tag}_b"] if has_b else None)
        assert norm_err(yo, golden_r2[f"g11_lin_{tag}_{name}_y"]) <= TOL


@pytest.mark.parametrize("name", ["ter", "q1", "q3", "bin"])
def test_g11_functional_conv_cpu_branch(golden_r2, oracle, name):
    from pytorch_quantize_impls_amd.functions.binary_connect import BinaryConv2d
    from pytorch_quantize_impls_amd.functions.dorefa_connect import QuantConv2d
    from pytorch_quantize_impls_amd.functions.terner_connect import TernaryConv2d
    for tag in golden_r2["g11_conv_cases"]:
        parts = dict((p[0], int(p[1:])) for p in str(tag).split("_")[:6])
        st, pd = parts["s"], parts["p"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            op = {"ter": lambda: TernaryConv2d(stochastic=False, stride=st, padding=pd), "q1": lambda: QuantConv2d(st, pd, bit_width=1),
                  "q3": lambda: QuantConv2d(st, pd, bit_width=3), "bin": lambda: BinaryConv2d(st, pd)}[name]()
        x, w, g = (golden_r2[f"g11_conv_{tag}_{k}"] for k in ("x", "w", "g"))
        has_b = f"g11_conv_{tag}_b" in golden_r2.files
        xi, wi = _t(x).requires_grad_(True), _t(w).requires_grad_(True)
        bi = _t(golden_r2[f"g11_conv_{tag}_b"]).requires_grad_(True) if has_b else None
        y = op.apply(xi, wi, bi) if has_b else op.apply(xi, wi)
        y.backward(_t(g))
        assert norm_err(y.detach().numpy(), golden_r2[f"g11_conv_{tag}_{name}_y"]) <= TOL
        assert norm_err(xi.grad.numpy(), golden_r2[f"g11_conv_{tag}_{name}_gx"]) <= TOL
        assert norm_err(wi.grad.numpy(), golden_r2[f"g11_conv_{tag}_{name}_gw"]) <= TOL
        wq = {"ter": lambda: oracle.functional_ternary_weight(w), "q1": lambda: oracle.functional_quant_weight(w, 1, conv=True),
              "q3": lambda: oracle.functional_quant_weight(w, 3, conv=True), "bin": lambda: oracle.safe_sign(w)}[name]()
        yo = oracle.conv2d(x, wq, golden_r2[f"g11_conv_{tag}_b"] if has_b else None, stride=st, padding=pd)
        assert norm_err(yo, golden_r2[f"g11_conv_{tag}_{name}_y"]) <= TOL


def test_g13_shift_batch_oracle_and_cpu_branch(golden_r2, oracle):
    from pytorch_quantize_impls_amd.functions.binary_connect import ShiftBatch
    from pytorch_quantize_impls_amd.layers import ShiftNormBatch1d, ShiftNormBatch2d
    for tag in golden_r2["g13_cases"]:
        x, mean, var, w, b = (golden_r2[f"g13_{tag}_{k}"] for k in ("x", "mean", "var", "w", "b"))
        want = golden_r2[f"g13_{tag}_y"]
        got, _ = oracle.shift_batch(x, mean, var, w, b, 1e-4)
        assert norm_err(got, want) <= TOL, tag
        y = ShiftBatch.apply(_t(x), _t(mean), _t(var), _t(w), _t(b), 1e-4)
        assert same(y.numpy(), want), tag
    for name, cls in (("bn1d", ShiftNormBatch1d), ("bn2d", ShiftNormBatch2d)):
        x = golden_r2[f"g13_{name}_x"]
        m = cls(x.shape[1])
        m.weight.data.copy_(_t(golden_r2[f"g13_{name}_w"])); m.bias.data.copy_(_t(golden_r2[f"g13_{name}_b"]))
        y = m(_t(x))
        assert same(y.detach().numpy(), golden_r2[f"g13_{name}_y"]), name
        assert same(m.running_var.numpy(), golden_r2[f"g13_{name}_running_var"])
