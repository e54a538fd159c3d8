"""The oracle's restatement of XNORConv2d(quant_input=True) (oracle.xnor_input_quant / xnor_conv2d_forward / _backward,
functions/xnor_connect.py:135-169) against the round-5 reference vectors (tests/golden/make_golden_r5.py: G21 fp64 samples of the
reference FUNCTION forward + backward, G22 digests on power-of-two operands).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from pytorch_quantize_impls_amd import synth

G21 = ["conv2", "conv3", "odd_96_72_9_s2", "small_8_16_7_nobias", "c3_first_layer"]
G22 = ["xnorconv_qi_conv2", "xnorconv_qi_conv3", "xnorconv_qi_odd_96_72_9_s2"]


@pytest.fixture(scope="module")
def g5():
    return np.load(os.path.join(GOLDEN_DIR, "golden_r5_v1.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def h5():
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r5.json")) as fh:
        return json.load(fh)["cases"]


def g21_operands(g5, name):
    """The operands make_golden_r5.py fed the reference (regenerated from the counter PRNG)."""
    B, Cin, Cout, H, k, s, p, seed, has_bias = (int(v) for v in g5[f"g21_{name}_geom"])
    x = synth.normal(seed, (B, Cin, H, H))
    x.reshape(-1)[::97] = 0.0
    w = synth.normal(seed + 1, (Cout, Cin, k, k), 0.05)
    w[0, 0, 0, 0] = 0.0
    b = synth.normal(seed + 2, (Cout,)) if has_bias else None
    Ho = (H + 2 * p - k) // s + 1
    go = synth.normal(seed + 3, (B, Cout, Ho, Ho))
    return x, w, b, go, s, p


def g22_operands(c):
    sg = synth.pm1(c["x_seed"], (c["B"], c["Cin"], c["H"], c["H"]))
    a = np.floor(synth.uniform(c["a_seed"], (c["B"], 1, c["H"], c["H"]), -3.0, 3.0)).astype(np.float32)
    x = (sg * np.exp2(a)).astype(np.float32)
    ws = synth.pm1(c["w_seed"], (c["Cout"], c["Cin"], c["k"], c["k"]))
    e = np.floor(synth.uniform(c["e_seed"], (c["k"], c["k"]), -6.0, 2.0)).astype(np.float32)
    return x, (ws * np.exp2(e)[None, None]).astype(np.float32)


def sampled(g5, name, key, full):
    want = g5[f"g21_{name}_{key}"]
    mx, st = g5[f"g21_{name}_{key}_max"]
    got = np.asarray(full, dtype=np.float64).reshape(-1)[::int(st)]
    assert got.shape == want.shape
    return float(np.abs(got - want).max() / mx)


@pytest.mark.parametrize("name", G21)
def test_oracle_xnor_conv_quant_input_fp64_vectors(oracle, g5, name):
    x, w, b, go, s, p = g21_operands(g5, name)
    y = oracle.xnor_conv2d_forward(x, w, b, s, p, quant_input=True)
    assert sampled(g5, name, "y", y) <= 1e-5             # fp32 alpha / scales vs the fp64 evaluation: SURVEY 8(d)'s bar
    gx, gw, gb = oracle.xnor_conv2d_backward(go, x, w, s, p, quant_input=True)
    for key, t in (("gx", gx), ("gw", gw)) + ((("gb", gb),) if b is not None else ()):
        assert sampled(g5, name, key, t) <= 1e-12, key


@pytest.mark.parametrize("name", G22)
def test_oracle_xnor_conv_quant_input_digest(oracle, h5, name):
    c = h5[name]
    x, w = g22_operands(c)
    y = oracle.xnor_conv2d_forward(x, w, None, c["stride"], c["pad"], quant_input=True)
    assert hashlib.sha256(np.ascontiguousarray(y, dtype=np.float32).tobytes()).hexdigest() == c["sha256_f32"]


def test_input_quantiser_keeps_zeros_and_scales_per_pixel(oracle):
    x = np.array([[[[0.5, -2.0]], [[0.0, 4.0]], [[-1.5, 0.0]]]], dtype=np.float32)          # [1, 3, 1, 2]
    q = oracle.xnor_input_quant(x)
    a0, a1 = np.float32(2.0 / 3.0), np.float32(2.0)
    assert np.array_equal(q, np.array([[[[a0, -a1]], [[0.0, a1]], [[-a0, 0.0]]]], dtype=np.float32))


# ---- G23: DoReFa layers at 8 < bit_width < 32 (tests/golden/make_golden_r5b.py) -----------------------------------------------------

G23 = ["conv12_s1_codes", "conv16_s2_real", "conv16_s1_codes", "lin12_real", "lin16_codes", "lin9_codes"]


@pytest.mark.parametrize("name", G23)
def test_oracle_dorefa_9_to_16_bit_forward_vs_reference_fp64(oracle, name):
    """The oracle's nnQuantWeight + conv2d / linear (functions/dorefa_connect.py:99-111, layers/dorefa_layers.py:41-45, 77-82) against the
    reference's own layers run in double precision: the weights sit away from the rounding boundaries, so the fp32 quantiser picks
    the reference's levels (checked: n * w_q is an odd-parity integer grid point within 2^-24 n)."""
    g = np.load(os.path.join(GOLDEN_DIR, "golden_r5b_v1.npz"), allow_pickle=False)
    x, w, b, want = (g[f"g23_{name}_{k}"] for k in ("x", "w", "b", "y"))
    geom = [int(v) for v in g[f"g23_{name}_geom"]]
    bits = geom[-2]
    wq = oracle.dorefa_weight(w, bits)
    n = float((1 << bits) - 1)
    lv = np.round(wq.astype(np.float64) * n)
    assert np.abs(wq.astype(np.float64) * n - lv).max() < 0.02 and np.all(np.mod(lv + n, 2) == 0) and np.abs(lv).max() == n
    if name.startswith("conv"):
        y = oracle.conv2d(x, wq, b, geom[5], geom[6])
    else:
        y = oracle.linear(x, wq, b)
    assert np.abs(y.astype(np.float64) - want).max() <= 1e-5 * np.abs(want).max()
