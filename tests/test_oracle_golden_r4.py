"""The oracle's XNOR-Net restatement (oracle.xnor_conv2d_* / xnor_dense_*, functions/xnor_connect.py:93-169) against the round-4
reference vectors (tests/golden/make_golden_r4.py: digests of the reference LAYER on power-of-two-per-tap weights, fp64 samples of
the reference FUNCTIONS forward + backward at the AlexNet shapes).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from pytorch_quantize_impls_amd import synth


@pytest.fixture(scope="module")
def g4():
    return np.load(os.path.join(GOLDEN_DIR, "golden_r4_v1.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def h4():
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r4.json")) as fh:
        return json.load(fh)["cases"]


def pow2_tap_weight(case):
    """The generator's weight: +-2^e[i, j] per tap (its exponents travel with the digest)."""
    s = synth.pm1(case["w_seed"], (case["Cout"], case["Cin"], case["k"], case["k"]))
    return (s * np.exp2(np.asarray(case["tap_exponents"], dtype=np.float32))[None, None]).astype(np.float32)


@pytest.mark.parametrize("name", ["xnorconv_conv2", "xnorconv_conv3", "xnorconv_conv5", "xnorconv_odd_96_72_9_s2"])
def test_oracle_xnor_conv_digest(oracle, h4, name):
    c = h4[name]
    x = synth.pm1(c["x_seed"], (c["B"], c["Cin"], c["H"], c["H"]))
    y = oracle.xnor_conv2d_forward(x, pow2_tap_weight(c), None, c["stride"], c["pad"])
    assert hashlib.sha256(np.ascontiguousarray(y, dtype=np.float32).tobytes()).hexdigest() == c["sha256_f32"]


def _sampled(g4, name, key, full):
    want = g4[f"g17_{name}_{key}"]
    mx, st = g4[f"g17_{name}_{key}_max"]
    got = np.asarray(full, dtype=np.float64).reshape(-1)[::int(st)]
    assert got.shape == want.shape
    return float(np.abs(got - want).max() / mx)


@pytest.mark.parametrize("name", ["conv2", "conv3", "conv5", "odd_96_72_9_s2"])
def test_oracle_xnor_conv_fp64_vectors(oracle, g4, name):
    B, Cin, Cout, H, k, s, p, seed = (int(v) for v in g4[f"g17_{name}_geom"])
    x = synth.pm1(seed, (B, Cin, H, H))
    w = synth.normal(seed + 1, (Cout, Cin, k, k), 0.05)
    w[0, 0, 0, 0] = 0.0
    b = synth.normal(seed + 2, (Cout,))
    y = oracle.xnor_conv2d_forward(x, w, b, s, p)
    # the oracle forms alpha and sign(W) * alpha in fp32 like the reference's fp32 run; the vector is the fp64 evaluation: the
    # parity bar of SURVEY 8(d) (1e-5 normalised) is what relates the two
    assert _sampled(g4, name, "y", y) <= 1e-5
    go = synth.normal(seed + 3, tuple(y.shape))
    gx, gw, gb = oracle.xnor_conv2d_backward(go, x, w, s, p)
    for key, t in (("gx", gx), ("gw", gw), ("gb", gb)):
        assert _sampled(g4, name, key, t) <= 1e-12, key


@pytest.mark.parametrize("name", ["fc1", "fc3", "fc_odd"])
def test_oracle_xnor_dense_fp64_vectors(oracle, g4, name):
    B, K, N, seed = (int(v) for v in g4[f"g17_{name}_geom"])
    x = synth.pm1(seed, (B, K))
    w = synth.normal(seed + 1, (N, K), 0.05)
    w[0, 0] = 0.0
    b = synth.normal(seed + 2, (N,))
    y = oracle.xnor_dense_forward(x, w, b)
    assert _sampled(g4, name, "y", y) <= 1e-5
    go = synth.normal(seed + 3, tuple(y.shape))
    gx, gw, gb = oracle.xnor_dense_backward(go, x, w)
    for key, t in (("gx", gx), ("gw", gw), ("gb", gb)):
        assert _sampled(g4, name, key, t) <= 1e-12, key
