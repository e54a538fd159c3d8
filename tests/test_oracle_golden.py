"""Pin the CPU oracle (oracle/) against vectors produced by the reference itself.

Integer / byte / index results are compared bit-exactly; float tails with the normalised metric
max|a-b|/max|b| <= 1e-5 stated by BASELINE.json's north_star."""
import hashlib

import numpy as np
import pytest

from conftest import same, norm_err
from pytorch_quantize_impls_amd import synth

TOL = 1e-5


def test_g1_safe_sign_and_ste(golden, oracle):
    x = golden["g1_edge_x"]
    assert same(oracle.safe_sign(x), golden["g1_safe_sign"])
    assert same(oracle.safe_sign(x), golden["g1_bin_det_fwd"])
    assert same(oracle.ste_mask(golden["g1_bwd_gout"], x), golden["g1_bin_det_bwd"])
    assert same(oracle.ste_mask(np.ones(6, np.float32), golden["g1_mask_x"]), golden["g1_mask"])
    assert golden["g1_mask"].tolist() == [0, 1, 1, 1, 0, 1]


def test_g2_ternary(golden, oracle):
    x = golden["g2_x"]
    assert same(oracle.ternarize(x), golden["g2_ter_det_fwd"])
    assert same(oracle.ste_mask(golden["g2_bwd_gout"], x), golden["g2_ter_det_bwd"])
    # known answers of the reference's own test (tests/implementations/Terner/function_test.py:10-27)
    assert oracle.ternarize(np.array([0.75, 0.5, 0.25, 0.0, -1, -0.2], np.float32)).tolist() == [1, 1, 0, 0, -1, 0]
    assert oracle.ternarize(np.array([1, 0, 0.51, 0.1, 0, -1, -0.2, .7], np.float32)).tolist() == [1, 0, 1, 0, 0, -1, 0, 1]


@pytest.mark.parametrize("k", [1, 2, 3, 4, 8, 16, 25, 31, 32])
def test_g3_dorefa_quantize(golden, oracle, k):
    assert same(oracle.dorefa_quantize(golden[f"g3_quant_x_k{k}"], k), golden[f"g3_quant_y_k{k}"])


@pytest.mark.parametrize("k", [1, 2, 3, 4, 8, 32])
def test_g3_dorefa_weight(golden, oracle, k):
    got = oracle.dorefa_weight(golden["g3_wq_x"], k)
    ref = golden[f"g3_wq_y_k{k}"]
    if k in (1, 32):
        assert norm_err(got, ref) <= 1e-6
    else:
        n = 2 ** k - 1
        codes_got = np.rint((got + 1) / 2 * n)
        codes_ref = np.rint((ref + 1) / 2 * n)
        assert np.max(np.abs(codes_got - codes_ref)) <= 1  # tanh ulp may move a tie by one code
        assert np.mean(codes_got != codes_ref) <= 0.05
    assert same(oracle.dorefa_weight(np.zeros((3, 2), np.float32), 3), golden["g3_wq_zero_k3"])


def _lin_case(golden, name):
    g = lambda s: golden[f"g4_lin_{name}_{s}"]
    b = g("b") if f"g4_lin_{name}_b" in golden.files else None
    return g("x"), g("w"), b, g("gout")


def test_g4_linear_layers(golden, oracle):
    for name in golden["g4_lin_cases"].tolist():
        x, w, b, gout = _lin_case(golden, name)
        exact = ("_pm1_" in name) and b is None
        for fam, wq in (("bin", oracle.safe_sign(w)), ("ter", oracle.ternarize(w)),
                        ("dorefa1", oracle.dorefa_weight(w, 1)), ("xnor", oracle.xnor_dense_weight(w))):
            y = oracle.linear(x, wq, b)
            ref = golden[f"g4_lin_{name}_{fam}_y"]
            if exact and fam in ("bin", "ter"):
                assert same(y, ref), (name, fam)
            else:
                assert norm_err(y, ref) <= TOL, (name, fam)
        # backward of LinearBin / LinearTer: grad_x = g.Wq ; grad_W = (g^T.x)*1[|W|<=1.001]
        for fam, wq in (("bin", oracle.safe_sign(w)), ("ter", oracle.ternarize(w))):
            gx = oracle.linear(gout, wq.T.copy())
            assert norm_err(gx, golden[f"g4_lin_{name}_{fam}_gx"]) <= TOL
            gw = oracle.ste_mask(oracle.linear(gout.T.copy(), x.T.copy()), w)
            assert norm_err(gw, golden[f"g4_lin_{name}_{fam}_gw"]) <= TOL
            if b is not None:
                assert norm_err(gout.sum(0), golden[f"g4_lin_{name}_{fam}_gb"]) <= TOL


def test_g4_packed_gemm_equals_reference(golden, oracle):
    """The bit-plane formulation reproduces the reference's fp32 result bit-exactly."""
    for name in golden["g4_lin_cases"].tolist():
        if "_pm1_" not in name:
            continue
        x, w, b, _ = _lin_case(golden, name)
        K = x.shape[1]
        y = oracle.xnor_gemm(oracle.sign_pack(x), oracle.sign_pack(w), K, b)
        m, s = oracle.ternary_pack(w)
        yt = oracle.tern_gemm(oracle.sign_pack(x), m, s, K, b)
        if b is None:
            assert same(y, golden[f"g4_lin_{name}_bin_y"])
            assert same(yt, golden[f"g4_lin_{name}_ter_y"])
        else:
            assert norm_err(y, golden[f"g4_lin_{name}_bin_y"]) <= TOL
            assert norm_err(yt, golden[f"g4_lin_{name}_ter_y"]) <= TOL


def _conv_cfg(name):
    p = name.split("_")
    return dict(Cin=int(p[0][1:]), Cout=int(p[1][1:]), k=int(p[2][1:]), stride=int(p[3][1:]),
                pad=int(p[4][1:]), H=int(p[5][1:]), pm1=p[6] == "pm1", bias=p[7] == "bias")


def test_g4_conv_layers(golden, oracle):
    for name in golden["g4_conv_cases"].tolist():
        c = _conv_cfg(name)
        x, w = golden[f"g4_conv_{name}_x"], golden[f"g4_conv_{name}_w"]
        b = golden[f"g4_conv_{name}_b"] if c["bias"] else None
        for fam, wq in (("bin", oracle.safe_sign(w)), ("ter", oracle.ternarize(w)),
                        ("dorefa1", oracle.dorefa_weight(w, 1)), ("xnor", oracle.xnor_conv_weight(w))):
            y = oracle.conv2d(x, wq, b, c["stride"], c["pad"])
            ref = golden[f"g4_conv_{name}_{fam}_y"]
            if c["pm1"] and not c["bias"] and fam in ("bin", "ter"):
                assert same(y, ref), (name, fam)
            else:
                assert norm_err(y, ref) <= TOL, (name, fam)


def test_g5_eval_swap(golden, oracle):
    w, x = golden["g5_w"], golden["g5_x"]
    for fam, q in (("bin", oracle.safe_sign), ("ter", oracle.ternarize)):
        assert same(q(w), golden[f"g5_{fam}_w_eval"])
        assert same(w, golden[f"g5_{fam}_w_back"])
        assert norm_err(oracle.linear(x, q(w)), golden[f"g5_{fam}_y_train"]) <= TOL
        assert norm_err(oracle.linear(x, q(w)), golden[f"g5_{fam}_y_eval"]) <= TOL


def test_g6_stochastic_with_injected_uniforms(golden, oracle):
    x, z = golden["g6_x"], golden["g6_z"]
    assert same(oracle.binarize_stochastic(x, z), golden["g6_bin_sto"])
    assert same(oracle.ternarize_stochastic(x, z), golden["g6_ter_sto"])


def test_g7_c1_mlp(golden, oracle):
    """Config 1: 784-512-10 MLP, batch 128, CPU plumbing (BASELINE.json configs[0])."""
    x = synth.normal(0x5EED + 1, (128, 784))
    w1 = synth.normal(0x5EED + 11, (512, 784), std=(1 / 784) ** 0.5)
    w2 = synth.normal(0x5EED + 12, (10, 512), std=(1 / 512) ** 0.5)
    h = np.maximum(oracle.linear_bin_forward(x, w1), 0)
    assert norm_err(h, golden["g7_c1_hidden_prebn"]) <= TOL
    hs = golden["g7_c1_hidden_sign"]  # sign(BN(h)): taken from the reference (BN is off-path)
    logits = oracle.linear_bin_forward(hs, w2)
    logp = logits - np.log(np.sum(np.exp(logits - logits.max(1, keepdims=True)), 1, keepdims=True)) \
        - logits.max(1, keepdims=True)
    assert norm_err(logp, golden["g7_c1_logits"]) <= TOL


@pytest.mark.parametrize("case", ["linbin_c2_slice", "linbin_c3_fc1_slice", "linbin_odd",
                                  "linter_c2_slice", "linter_odd"])
def test_g7_digests_packed_oracle(golden_hashes, oracle, case):
    """SHA-256 of the int32 result of config-sized layers, inputs regenerated from the PRNG."""
    h = golden_hashes[case]
    x = synth.pm1(h["x_seed"], (h["B"], h["K"]))
    w = synth.uniform(h["w_seed"], (h["N"], h["K"]), h["w_lo"], h["w_hi"])
    if case.startswith("linbin"):
        y = oracle.xnor_gemm(oracle.sign_pack(x), oracle.sign_pack(w), h["K"])
    else:
        m, s = oracle.ternary_pack(w)
        y = oracle.tern_gemm(oracle.sign_pack(x), m, s, h["K"])
    assert hashlib.sha256(y.astype(np.int32).tobytes()).hexdigest() == h["sha256_int32"]


def test_nibble_formulation_equals_reference(golden, oracle):
    """The fp4-nibble restatement (what the MFMA kernel computes) reproduces the reference too."""
    for name in golden["g4_lin_cases"].tolist():
        if "_pm1_" not in name or name.endswith("_bias"):
            continue
        x, w, b, _ = _lin_case(golden, name)
        K = x.shape[1]
        xn = oracle.pack_nib(x)
        assert np.array_equal(xn, oracle.bits_to_nib(oracle.sign_pack(x), None, K))
        assert same(oracle.nib_gemm(xn, oracle.pack_nib(w), K), golden[f"g4_lin_{name}_bin_y"])
        wt = oracle.pack_nib(w, ternary=True)
        m, s = oracle.ternary_pack(w)
        assert np.array_equal(wt, oracle.bits_to_nib(s, m, K))
        assert same(oracle.nib_gemm(xn, wt, K), golden[f"g4_lin_{name}_ter_y"])


def test_g8_dorefa_w1a4(golden, oracle):
    for name in golden["g8_cases"].tolist():
        x, w = golden[f"g8_{name}_x"], golden[f"g8_{name}_w"]
        b = golden[f"g8_{name}_b"] if f"g8_{name}_b" in golden.files else None
        xq = oracle.dorefa_quantize(np.maximum(x, 0), 4)
        assert same(xq, golden[f"g8_{name}_xq"])
        wq = oracle.dorefa_weight(w, 1)
        if name.startswith("lin"):
            assert norm_err(oracle.linear(xq, wq, b), golden[f"g8_{name}_y"]) <= TOL
            assert norm_err(oracle.dorefa_w1a_linear(x, w, b), golden[f"g8_{name}_y"]) <= TOL   # int8 factoring
            assert norm_err(oracle.linear(xq, wq, b), golden[f"g8_{name}_y_eval"]) <= TOL
            gout = golden[f"g8_{name}_gout"]
            assert norm_err(oracle.linear(gout.T.copy(), xq.T.copy()), golden[f"g8_{name}_gw"]) <= TOL   # unscaled by E
            gx = oracle.linear(gout, wq.T.copy()) * (x > 0)
            assert norm_err(gx, golden[f"g8_{name}_gx"]) <= TOL
        else:
            p = name.split("_")
            st, pd = int(p[4][1:]), int(p[5][1:])
            assert norm_err(oracle.conv2d(xq, wq, b, st, pd), golden[f"g8_{name}_y"]) <= TOL
