"""The oracle's C restatement against the round-3 reference digests / vectors (tests/golden/make_golden_r3.py): the strided
C4 shapes and the remaining C5 channel classes, so that the GPU tests of those shapes have a comparator that is not the same
device's dense library.  CPU only; the small shapes, the big ones are covered on the GPU."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, norm_err
from pytorch_quantize_impls_amd import synth


@pytest.fixture(scope="module")
def hashes_r3():
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r3.json")) as fh:
        return json.load(fh)["cases"]


@pytest.fixture(scope="module")
def golden_r3():
    return np.load(os.path.join(GOLDEN_DIR, "golden_r3_v1.npz"), allow_pickle=False)


def _inputs(h):
    shape = (h["B"], h["Cin"], h["H"], h["H"])
    if h["x"] == "pm1":
        x = synth.pm1(h["x_seed"], shape)
    else:
        x = np.floor(synth.uniform(h["x_seed"], shape, 0.0, 16.0)).clip(0, 15).astype(np.float32)
    w = synth.uniform(h["w_seed"], (h["Cout"], h["Cin"], h["k"], h["k"]), h["w_lo"], h["w_hi"])
    return x, w


@pytest.mark.parametrize("case", ["binconv_c4_64_128_32_s2", "binconv_c4_64_128_32_1x1s2", "binconv_c4_256_512_8_s2",
                                  "w1a4_core_c4_64_128_32_s2", "w1a4_core_c4_64_128_32_1x1s2", "terconv_c5_512_512_14"])
def test_oracle_reproduces_the_reference_digests(oracle, hashes_r3, case):
    h = hashes_r3[case]
    x, w = _inputs(h)
    fwd = oracle.ter_conv2d_forward if case.startswith("terconv") else oracle.bin_conv2d_forward
    y = fwd(x, w, None, h["stride"], h["pad"])
    yi = y.astype(np.int32)
    assert np.array_equal(yi.astype(np.float32), y)
    assert hashlib.sha256(np.ascontiguousarray(yi).tobytes()).hexdigest() == h["sha256_int32"]
    assert float(y.astype(np.float64).sum()) == h["sum"]


def test_strided_backward_vectors_are_consistent_with_the_oracle_forward(oracle, golden_r3):
    """G15 (fp64 autograd of the reference's strided BinConv2d): forward equals the oracle's; grad_weight is the STE-masked
    correlation of input and gradient, grad_bias the gradient's sum (functions/binary_connect.py:31-38,141-143)."""
    for name in golden_r3["g15_cases"]:
        Cin, Cout, H, k, s, p = (int(v) for v in golden_r3[f"g15_{name}_geom"])
        x, w, b, go = (golden_r3[f"g15_{name}_{t}"] for t in ("x", "w", "b", "go"))
        y = oracle.bin_conv2d_forward(x, w, b, s, p)
        assert norm_err(y, golden_r3[f"g15_{name}_y"]) <= 1e-6
        assert norm_err(go.astype(np.float64).sum((0, 2, 3)), golden_r3[f"g15_{name}_gb"]) <= 1e-6
        gw = golden_r3[f"g15_{name}_gw"]
        assert (gw[np.abs(w) > 1.001] == 0).all()
        # one tap checked by hand: dW[co, ci, i, j] = sum g[n, co, y, x] * x[n, ci, s y + i - p, s x + j - p]
        i = j = k // 2
        Ho = go.shape[2]
        xs = np.zeros((x.shape[0], Cin, Ho, Ho), np.float64)
        for yy in range(Ho):
            for xx in range(Ho):
                hh, ww = s * yy + i - p, s * xx + j - p
                if 0 <= hh < H and 0 <= ww < H:
                    xs[:, :, yy, xx] = x[:, :, hh, ww]
        ref = np.einsum("nohw,nihw->oi", go.astype(np.float64), xs)
        ref[np.abs(w[:, :, i, j]) > 1.001] = 0
        assert norm_err(gw[:, :, i, j], ref) <= 1e-6
