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


# ---- round 4, second half: G19 (LinearXNOR digests at the classifier shapes) and G20 (8- / 32-bit DoReFa layers in fp64) ------------------

def _g19_operands(seed, B, K, N):
    """The operands tests/golden/make_golden_r4c.py fed the reference: x +-1, |W[n, k]| = 2^e[k], bias multiples of 1/4."""
    x = synth.pm1(seed, (B, K))
    e = np.floor(synth.uniform(seed + 1, (K,), -6.0, -2.0)).astype(np.int32)
    sgn = synth.pm1(seed + 2, (N, K))
    w = (sgn * np.exp2(e.astype(np.float32))[None, :]).astype(np.float32)
    b = (np.round(synth.normal(seed + 3, (N,)) * 4) / 4).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize("name", ["xnor_fc3_b256", "xnor_fc_ragged"])        # (the 4096-wide cases: GPU tests only — half a minute of CPU each)
def test_oracle_linear_xnor_digest(oracle, name):
    """oracle.xnor_dense_forward reproduces the reference LinearXNOR's SHA-256 (exact sums: alpha a power of two per input feature)."""
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r4c.json")) as fh:
        c = json.load(fh)["cases"][name]
    x, w, b = _g19_operands(c["seed"], c["B"], c["K"], c["N"])
    y = oracle.xnor_dense_forward(x, w, b)
    assert hashlib.sha256(np.ascontiguousarray(y, dtype=np.float32).tobytes()).hexdigest() == c["sha256_f32"]


@pytest.mark.parametrize("name", ["conv8_s1_codes", "conv8_s2_real", "conv8_1x1_codes", "lin8_codes", "lin8_real", "conv32_s1", "lin32"])
def test_oracle_dorefa_8_and_32_bit_forward_vs_reference_fp64(oracle, name):
    """The oracle's restatement of nnQuantWeight + conv2d / linear (functions/dorefa_connect.py:99-111, layers/dorefa_layers.py:41-45,
    77-82) against the reference's own layers run in double precision (G20): the 8-bit weights were built away from the rounding
    boundaries, so the fp32 quantiser picks the reference's levels."""
    g = np.load(os.path.join(GOLDEN_DIR, "golden_r4d_v1.npz"), allow_pickle=False)
    x, w, b, want = (g[f"g20_{name}_{k}"] for k in ("x", "w", "b", "y"))
    geom = [int(v) for v in g[f"g20_{name}_geom"]]
    bits = geom[-2]
    wq = oracle.dorefa_weight(w, bits)
    if bits == 8:
        lv = np.round(wq.astype(np.float64) * 255.0)
        assert np.abs(wq * 255.0 - lv).max() < 1e-3 and np.all(np.mod(lv, 2) == 1) and np.abs(lv).max() == 255
    if name.startswith("conv"):
        y = oracle.conv2d(x, wq, b, geom[5], geom[6])
    else:
        y = oracle.linear(x, wq, b)
    assert np.abs(y.astype(np.float64) - want).max() <= 1e-5 * np.abs(want).max()
