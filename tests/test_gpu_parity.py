"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the reference-generated
golden vectors.  Integer / bit / index results: bit-exact.  Float tails: max|a-b|/max|b| <= 1e-5.
Every test asserts that the libqt_hip.so entry points actually ran (no silent torch path)."""
import copy
import hashlib
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import same, norm_err

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import _lib, lazy, ops, packed, synth  # noqa: E402
from pytorch_quantize_impls_amd.functions import (BinaryConnectDeterministic, BinaryConnectStochastic,  # noqa: E402
                                                  TernaryConnectDeterministic, BinaryConnect, BinaryDense,
                                                  nnDorefaQuant, safeSign)
from pytorch_quantize_impls_amd.functions import binary_connect, terner_connect  # noqa: E402
from pytorch_quantize_impls_amd.layers import (LinearBin, LinearTer, BinConv2d, TerConv2d,  # noqa: E402
                                               LinearDorefa, DorefaConv2d, LinearXNOR, XNORConv2d)

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    name, cus = _lib.device_info()
    assert name.startswith("gfx950"), name
    return torch.device("cuda:0")


def g(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def n(t):
    return t.detach().cpu().numpy()


def planes_np(p):
    return n(p).view(np.uint32)


def _unpack_bits(words: torch.Tensor, K: int) -> torch.Tensor:
    """[rows, ld] int32 bit-plane words -> [rows, K] 0 / 1 (device tensor)."""
    w = words.to(torch.int64) & 0xFFFFFFFF
    bits = (w.unsqueeze(-1) >> torch.arange(32, device=words.device)) & 1
    return bits.reshape(words.shape[0], -1)[:, :K]


class used:
    """Context manager asserting that the named C-ABI entry points ran inside the block."""

    def __init__(self, *names):
        self.names = names

    def __enter__(self):
        self.before = dict(_lib.call_counts)

    def __exit__(self, *exc):
        if exc[0] is None:
            for k in self.names:
                assert _lib.call_counts[k] > self.before.get(k, 0), f"{k} did not run"


# ---- elementwise ------------------------------------------------------------------------------------

def test_elementwise_edges_golden(dev, golden, oracle):
    with used("qt_binarize_f32", "qt_ste_mask_f32", "qt_ternarize_f32"):
        x = g(golden["g1_edge_x"], dev)
        assert same(n(ops.binarize(x)), golden["g1_safe_sign"])
        assert same(n(ops.ste_mask(g(golden["g1_bwd_gout"], dev), x)), golden["g1_bin_det_bwd"])
        xt = g(golden["g2_x"], dev)
        assert same(n(ops.ternarize(xt)), golden["g2_ter_det_fwd"])
        assert same(n(ops.ste_mask(g(golden["g2_bwd_gout"], dev), xt)), golden["g2_ter_det_bwd"])


@pytest.mark.parametrize("size", [1, 3, 4, 5, 63, 64, 1000, 4099, 1 << 20])
@pytest.mark.parametrize("offset", [0, 1])
def test_elementwise_random_vs_oracle(dev, oracle, size, offset):
    x = synth.uniform(size + 7, (size + offset,), -1.6, 1.6)
    x[::7] = 0.0
    x[::11] = -0.0
    if size > 10:
        x[5], x[6], x[8] = np.nan, 0.5, -0.5
    xd = g(x, dev)[offset:]            # offset=1 -> 4-byte aligned only: exercises the scalar head
    xs = x[offset:]
    assert same(n(ops.binarize(xd)), oracle.safe_sign(xs))
    assert same(n(ops.ternarize(xd)), oracle.ternarize(xs))
    go = synth.normal(size, (size + offset,))[offset:]
    assert same(n(ops.ste_mask(g(go, dev), xd)), oracle.ste_mask(go, xs))
    z = synth.uniform(size + 1, xs.shape, 0.0, 1.0)
    assert same(n(ops.binarize_stochastic(xd, g(z, dev))), oracle.binarize_stochastic(xs, z))
    assert same(n(ops.ternarize_stochastic(xd, g(z, dev))), oracle.ternarize_stochastic(xs, z))


@pytest.mark.parametrize("k", [1, 2, 3, 4, 8, 16, 25, 31, 32])
def test_dorefa_quantize(dev, golden, oracle, k):
    with used("qt_dorefa_quantize_f32"):
        assert same(n(ops.dorefa_quantize(g(golden[f"g3_quant_x_k{k}"], dev), k)), golden[f"g3_quant_y_k{k}"])
    x = synth.uniform(900 + k, (5000,), -2.0, 3.0)
    assert same(n(ops.dorefa_quantize(g(x, dev), k)), oracle.dorefa_quantize(x, k))
    assert same(n(nnDorefaQuant(k)(g(x, dev))), oracle.dorefa_quantize(x, k))


def test_stochastic_golden(dev, golden):
    x, z = g(golden["g6_x"], dev), g(golden["g6_z"], dev)
    assert same(n(binary_connect.stochastic_binarize(x, z)), golden["g6_bin_sto"])
    assert same(n(terner_connect.stochastic_ternarize(x, z)), golden["g6_ter_sto"])


# ---- packing ------------------------------------------------------------------------------------------

PACK_SHAPES = [(1, 1), (1, 31), (2, 32), (3, 33), (5, 64), (4, 100), (7, 127), (9, 128), (3, 129),
               (128, 784), (33, 4096), (2, 9216), (300, 4), (17, 36), (64, 1000)]


@pytest.mark.parametrize("rows,K", PACK_SHAPES)
def test_sign_pack_vs_oracle(dev, oracle, rows, K):
    x = synth.uniform(rows * 131 + K, (rows, K), -1.0, 1.0)
    x[0, 0] = -0.0
    if K > 3:
        x[-1, 3] = np.nan
        x[0, K - 1] = -1e-45
    with used("qt_sign_pack_f32"):
        p, y = ops.sign_pack(g(x, dev), want_f32=True)
    want = oracle.sign_pack(x)
    assert p.ld == want.shape[1] and p.ld % 4 == 0
    assert np.array_equal(planes_np(p.sign), want)          # includes the all-zero pad words
    assert same(n(y), oracle.safe_sign(x))
    p2, _ = ops.sign_pack(g(x, dev))
    assert np.array_equal(planes_np(p2.sign), want)


@pytest.mark.parametrize("rows,K", PACK_SHAPES)
def test_ternary_pack_vs_oracle(dev, oracle, rows, K):
    x = synth.uniform(rows * 17 + K, (rows, K), -1.2, 1.2)
    x[0, 0] = 0.5
    if K > 2:
        x[0, 1], x[0, 2] = -0.5, np.nan
    with used("qt_ternary_pack_f32"):
        p = ops.ternary_pack(g(x, dev))
    m, s = oracle.ternary_pack(x)
    assert np.array_equal(planes_np(p.mask), m) and np.array_equal(planes_np(p.sign), s)


def test_sign_pack_strided_rows(dev, oracle):
    """Row stride larger than K (a column slice of a wider matrix)."""
    big = synth.uniform(77, (40, 256), -1, 1)
    view = g(big, dev)[:, 64:64 + 96]
    p, _ = ops.sign_pack(view)
    assert np.array_equal(planes_np(p.sign), oracle.sign_pack(big[:, 64:160]))
    view2 = g(big, dev)[:, 1:1 + 99]        # 4-byte aligned rows, K % 4 != 0 -> ballot path
    p2, _ = ops.sign_pack(view2)
    assert np.array_equal(planes_np(p2.sign), oracle.sign_pack(big[:, 1:100]))


def test_check_pm1(dev):
    a = synth.pm1(5, (300, 77))
    assert int(ops.check_pm1(g(a, dev)).item()) == 0
    for bad in (0.0, 0.5, np.nan, 1.0000001, -2.0):
        b = a.copy()
        b[123, 45] = bad
        assert int(ops.check_pm1(g(b, dev)).item()) != 0


# ---- packed GEMMs ----------------------------------------------------------------------------------------

GEMM_SHAPES = [(1, 1, 1), (5, 7, 31), (5, 7, 32), (5, 7, 33), (128, 128, 512), (129, 127, 100),
               (130, 260, 784), (64, 10, 4096), (257, 65, 9216), (3, 300, 1000), (200, 130, 96)]


@pytest.fixture(params=[0, 1, 2], ids=["auto", "tiled", "skinny"])
def popc_kernel(request):
    """Run a popcount-GEMM test under the automatic choice and with each kernel chosen (an argument of the
    qt_*_gemm_variant entry points; the library keeps no state)."""
    ops.POPC_VARIANT = request.param
    yield request.param
    ops.POPC_VARIANT = 0


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES + [(64, 4096, 4096), (1, 1000, 9216), (256, 10, 4096), (70, 33, 10000)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_xnor_gemm_vs_oracle(dev, oracle, popc_kernel, M, N, K, with_bias):
    x = synth.pm1(M * 7 + K, (M, K))
    w = synth.uniform(N * 5 + K, (N, K), -1, 1)
    b = synth.normal(N, (N,)) if with_bias else None
    with used("qt_sign_pack_f32", "qt_xnor_gemm_variant" if popc_kernel else "qt_xnor_gemm"):
        xp, wp = ops.sign_pack(g(x, dev))[0], ops.sign_pack(g(w, dev))[0]
        y = n(ops.xnor_gemm(xp, wp, None if b is None else g(b, dev)))
    want_int = oracle.linear(x, oracle.safe_sign(w))          # the reference computation, fp32
    if b is None:
        assert same(y, want_int)
        assert same(y, oracle.xnor_gemm(oracle.sign_pack(x), oracle.sign_pack(w), K))
    else:
        # integer part exact, bias added once in fp32
        assert same(y, want_int + b[None, :])
        assert norm_err(y, oracle.linear(x, oracle.safe_sign(w), b)) <= TOL


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES + [(64, 1000, 4096), (256, 10, 4096)])
def test_tern_gemm_vs_oracle(dev, oracle, popc_kernel, M, N, K):
    x = synth.pm1(M * 3 + K, (M, K))
    w = synth.uniform(N * 9 + K, (N, K), -1.5, 1.5)
    with used("qt_ternary_pack_f32", "qt_tern_gemm_variant" if popc_kernel else "qt_tern_gemm"):
        y = n(ops.tern_gemm(ops.sign_pack(g(x, dev))[0], ops.ternary_pack(g(w, dev))))
    assert same(y, oracle.linear(x, oracle.ternarize(w)))


# ---- matrix-core formulation (nibble planes + MX-fp4 MFMA) ------------------------------------------------

@pytest.mark.parametrize("rows,K", PACK_SHAPES)
def test_nib_pack_vs_oracle(dev, oracle, rows, K):
    x = synth.uniform(rows * 31 + K, (rows, K), -1.2, 1.2)
    x[0, 0] = -0.0
    if K > 2:
        x[0, 1], x[0, 2] = 0.5, np.nan
    with used("qt_sign_pack_nib_f32", "qt_ternary_pack_nib_f32", "qt_bits_to_nib"):
        pn = ops.sign_pack_nib(g(x, dev))
        tn = ops.ternary_pack_nib(g(x, dev))
        assert pn.ld % 32 == 0
        assert np.array_equal(planes_np(pn.words), oracle.pack_nib(x))
        assert np.array_equal(planes_np(tn.words), oracle.pack_nib(x, ternary=True))
        # 1-bit planes -> nibbles gives the same image
        assert np.array_equal(planes_np(ops.bits_to_nib(ops.sign_pack(g(x, dev))[0]).words), oracle.pack_nib(x))
        assert np.array_equal(planes_np(ops.bits_to_nib(ops.ternary_pack(g(x, dev))).words),
                              oracle.pack_nib(x, ternary=True))


NIB_GEMM_SHAPES = GEMM_SHAPES + [(256, 256, 256), (512, 256, 4096), (255, 257, 300), (300, 520, 1000)]


@pytest.mark.parametrize("M,N,K", NIB_GEMM_SHAPES)
@pytest.mark.parametrize("variant", [None, 5, 6, 7, 8, 9, 10, 15, 16, 20, 21, 22, 23, 24])
def test_nib_gemm_vs_oracle(dev, oracle, M, N, K, variant):
    x = synth.pm1(M * 7 + K, (M, K))
    w = synth.uniform(N * 5 + K, (N, K), -1.5, 1.5)
    b = synth.normal(N, (N,))
    with used("qt_nib_gemm" if variant is None else "qt_nib_gemm_variant"):
        xn = ops.sign_pack_nib(g(x, dev))
        y = n(ops.nib_gemm(xn, ops.sign_pack_nib(g(w, dev)), variant=variant))
        yt = n(ops.nib_gemm(xn, ops.ternary_pack_nib(g(w, dev)), g(b, dev), variant=variant))
    assert same(y, oracle.linear(x, oracle.safe_sign(w)))                    # the reference computation
    assert same(yt, oracle.linear(x, oracle.ternarize(w)) + b[None, :])       # integer part exact + bias once


def test_nib_gemm_equals_popcount_gemm_large(dev):
    """Both formulations are bit-identical on a large ragged problem (MFMA tile edges, K tail)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    x = torch.randn((1000, 5000), device=dev, generator=gen)
    w = torch.randn((777, 5000), device=dev, generator=gen)
    ref = ops.xnor_gemm(ops.sign_pack(x)[0], ops.sign_pack(w)[0])
    for variant in (None, 5, 6, 7, 8, 9, 10, 15, 16, 20, 21, 22, 23, 24):
        assert torch.equal(ops.nib_gemm(ops.sign_pack_nib(x), ops.sign_pack_nib(w), variant=variant), ref)
    reft = ops.tern_gemm(ops.sign_pack(x)[0], ops.ternary_pack(w))
    assert torch.equal(ops.nib_gemm(ops.bits_to_nib(ops.sign_pack(x)[0]), ops.bits_to_nib(ops.ternary_pack(w))), reft)


def test_layers_route_large_shapes_to_mfma(dev, oracle):
    assert ops.select_gemm_impl("auto", 4096, 4096, 4096) == "mfma"
    assert ops.select_gemm_impl("auto", 256, 4096, 4096) == "mfma" and ops.select_gemm_impl("auto", 4096, 10, 4096) == "valu"
    assert ops.select_gemm_impl("auto", 1024, 4096, 1024) == "valu" and ops.select_gemm_impl("auto", 128, 512, 784) == "valu"
    from pytorch_quantize_impls_amd.functions import _fused
    _fused.GEMM_IMPL = "mfma"      # small shapes below: force the formulation under test
    try:
        _route_mfma_body(dev, oracle)
    finally:
        _fused.GEMM_IMPL = "auto"


def _route_mfma_body(dev, oracle):
    x = synth.normal(41, (256, 512))
    w = synth.uniform(42, (320, 512), -1, 1)
    want = oracle.linear_bin_forward(oracle.safe_sign(x), w)
    for cls, wantf in ((LinearBin, want), (LinearTer, oracle.linear_ter_forward(oracle.safe_sign(x), w))):
        layer = cls(512, 320, bias=False).to(dev)
        layer.weight.data.copy_(g(w, dev))
        with used("qt_nib_gemm", "qt_bits_to_nib"):
            y = layer(BinaryConnectDeterministic.apply(g(x, dev)))
        assert same(n(y), wantf)
        layer.eval()
        with torch.no_grad(), used("qt_nib_gemm"):
            y2 = layer(BinaryConnectDeterministic.apply(g(x, dev)))
        assert same(n(y2), wantf)


def test_gemm_empty_and_errors(dev):
    xp = ops.sign_pack(torch.ones((4, 64), device=dev))[0]
    wp = ops.sign_pack(torch.ones((0, 64), device=dev))[0]
    assert ops.xnor_gemm(xp, wp).shape == (4, 0)
    wp2 = ops.sign_pack(torch.ones((3, 96), device=dev))[0]
    with pytest.raises(ValueError):
        ops.xnor_gemm(xp, wp2)
    with pytest.raises(ValueError):
        ops.tern_gemm(xp, ops.sign_pack(torch.ones((3, 64), device=dev))[0])


# ---- reference-generated golden vectors through the layers ---------------------------------------------------

def _mk(fam, K, N, bias, dev):
    layer = {"bin": LinearBin, "ter": LinearTer}[fam](K, N, bias=bias)
    return layer.to(dev)


def test_linear_layers_golden_forward_backward(dev, golden):
    for name in golden["g4_lin_cases"].tolist():
        gg = lambda s: golden[f"g4_lin_{name}_{s}"]
        has_b = f"g4_lin_{name}_b" in golden.files
        x, w, gout = gg("x"), gg("w"), gg("gout")
        for fam in ("bin", "ter"):
            layer = _mk(fam, x.shape[1], w.shape[0], has_b, dev)
            layer.weight.data.copy_(g(w, dev))
            if has_b:
                layer.bias.data.copy_(g(gg("b"), dev))
            xi = g(x, dev).requires_grad_(True)
            before = _lib.call_counts["qt_xnor_gemm"] + _lib.call_counts["qt_tern_gemm"]
            y = layer(xi)
            y.backward(g(gout, dev))
            ran_packed = _lib.call_counts["qt_xnor_gemm"] + _lib.call_counts["qt_tern_gemm"] > before
            assert ran_packed == ("_pm1_" in name), name   # +-1 input is detected on the device
            if "_pm1_" in name and not has_b:
                assert same(n(y), gg(f"{fam}_y")), (name, fam)
            else:
                assert norm_err(n(y), gg(f"{fam}_y")) <= TOL, (name, fam)
            assert norm_err(n(xi.grad), gg(f"{fam}_gx")) <= TOL
            assert norm_err(n(layer.weight.grad), gg(f"{fam}_gw")) <= TOL
            if has_b:
                assert norm_err(n(layer.bias.grad), gg(f"{fam}_gb")) <= TOL


def test_conv_layers_golden_forward(dev, golden):
    for name in golden["g4_conv_cases"].tolist():
        p = name.split("_")
        Cin, Cout, k, st, pd = int(p[0][1:]), int(p[1][1:]), int(p[2][1:]), int(p[3][1:]), int(p[4][1:])
        has_b = p[7] == "bias"
        x, w = golden[f"g4_conv_{name}_x"], golden[f"g4_conv_{name}_w"]
        for fam, cls in (("bin", BinConv2d), ("ter", TerConv2d)):
            layer = cls(Cin, Cout, k, stride=st, padding=pd, bias=has_b).to(dev)
            layer.weight.data.copy_(g(w, dev))
            if has_b:
                layer.bias.data.copy_(g(golden[f"g4_conv_{name}_b"], dev))
            y = n(layer(g(x, dev)))
            ref = golden[f"g4_conv_{name}_{fam}_y"]
            if p[6] == "pm1" and not has_b:
                assert same(y, ref), (name, fam)
            else:
                assert norm_err(y, ref) <= TOL, (name, fam)


@pytest.mark.parametrize("case", ["binconv_alex_conv2", "binconv_alex_conv3", "binconv_alex_conv5"])
@pytest.mark.parametrize("fmt", ["nchw", "channels_last", "tagged"])
def test_conv_reference_digests(dev, golden_hashes, case, fmt):
    """AlexNet-Bin conv layers at batch 2: SHA-256 of the int32 result the
    REFERENCE produced, through the packed conv path (pixel planes -> packed im2col -> MFMA GEMM)."""
    h = golden_hashes[case]
    x = synth.pm1(h["x_seed"], (h["B"], h["Cin"], h["H"], h["H"]))
    w = synth.uniform(h["w_seed"], (h["Cout"], h["Cin"], h["k"], h["k"]), -1.0, 1.0)
    conv = BinConv2d(h["Cin"], h["Cout"], h["k"], stride=h["stride"], padding=h["pad"]).to(dev)
    conv.weight.data.copy_(g(w, dev))
    conv.bias.data.zero_()
    xd = g(x, dev)
    if fmt != "nchw":
        xd = xd.contiguous(memory_format=torch.channels_last)
    if fmt == "tagged":
        xd = BinaryConnectDeterministic.apply(xd)
        assert packed.lookup(xd, packed.NHWC) is not None
    for implicit, entries in ((True, ("qt_conv2d_implicit",)), (False, ("qt_im2col_words", "qt_nib_gemm"))):
        ops.CONV_IMPLICIT = implicit
        try:
            for training in (True, False):
                conv.train(training)
                with torch.no_grad(), used(*entries):
                    y = lazy.resolve(conv(xd))           # eval mode defers the conv: take its own fp32 result here
                assert y.shape == (h["B"], h["Cout"], h["H"], h["H"])
                assert y.is_contiguous(memory_format=torch.channels_last) == (fmt != "nchw")
                yi = n(y.contiguous()).astype(np.int32)
                assert hashlib.sha256(yi.tobytes()).hexdigest() == h["sha256_int32"], (fmt, training, implicit)
        finally:
            ops.CONV_IMPLICIT = True


def test_conv_backward_matches_dense(dev):
    """Training-mode BinConv2d / TerConv2d: packed forward, autograd backward = dense conv + STE mask."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    x = torch.randn((3, 32, 9, 9), device=dev, generator=gen).sign()
    for cls in (BinConv2d, TerConv2d):
        layer = cls(32, 6, 3, stride=2, padding=1).to(dev)
        layer.weight.data.mul_(8)          # some |w| > 1.001 so the STE mask matters
        xi = x.clone().requires_grad_(True)
        y = layer(xi)
        go = torch.randn(y.shape, device=dev, generator=gen)
        y.backward(go)
        wq = (ops.binarize if cls is BinConv2d else ops.ternarize)(layer.weight.detach())
        xr = x.clone().requires_grad_(True)
        wr = wq.clone().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, layer.bias, stride=2, padding=1)
        yr.backward(go)
        assert torch.equal(y, yr + 0)      # integer-valued + same bias add
        assert norm_err(n(xi.grad), n(xr.grad)) <= TOL
        mask = (layer.weight.detach().abs() <= 1.001).float()
        assert norm_err(n(layer.weight.grad), n(wr.grad * mask)) <= TOL


def test_eval_swap_on_device(dev, golden):
    w, x = golden["g5_w"], golden["g5_x"]
    for fam, cls in (("bin", LinearBin), ("ter", LinearTer)):
        layer = cls(9, 4, bias=False).to(dev)
        layer.weight.data.copy_(g(w, dev))
        assert norm_err(n(layer(g(x, dev))), golden[f"g5_{fam}_y_train"]) <= TOL
        layer.train(False)
        assert same(n(layer.weight.data), golden[f"g5_{fam}_w_eval"])
        assert norm_err(n(layer(g(x, dev))), golden[f"g5_{fam}_y_eval"]) <= TOL
        # packed eval path with cached planes: +-1 input
        xb = synth.pm1(3, (6, 9))
        want = xb @ golden[f"g5_{fam}_w_eval"].T
        with torch.no_grad():
            y1 = n(layer(g(xb, dev)))
            y2 = n(layer(g(xb, dev)))
        assert same(y1, want.astype(np.float32)) and same(y2, y1)
        layer.train(True)
        assert same(n(layer.weight.data), golden[f"g5_{fam}_w_back"])


def test_binaryconnect_tag_feeds_linear_without_recheck(dev, oracle):
    x = synth.normal(21, (70, 333))
    w = synth.uniform(22, (50, 333), -1, 1)
    layer = LinearBin(333, 50, bias=False).to(dev)
    layer.weight.data.copy_(g(w, dev))
    c0 = _lib.call_counts["qt_check_pm1_f32"]
    xs = BinaryConnect()(g(x, dev))
    assert packed.lookup(xs, packed.ROWS_LAST) is not None
    y = layer(xs)
    assert _lib.call_counts["qt_check_pm1_f32"] == c0           # tag used, no device check/sync
    assert same(n(y), oracle.linear_bin_forward(oracle.safe_sign(x), w))
    xs.mul_(1.0)                                                   # in-place write invalidates the tag
    assert packed.lookup(xs, packed.ROWS_LAST) is None
    assert same(n(layer(xs)), oracle.linear_bin_forward(oracle.safe_sign(x), w))
    # non-binary input takes the general path and still matches
    assert norm_err(n(layer(g(x, dev))), oracle.linear_bin_forward(x, w)) <= TOL


def test_binary_dense_function(dev, oracle):
    x = synth.pm1(31, (33, 100))
    w = synth.uniform(32, (9, 100), -2, 2)
    b = synth.normal(33, (9,))
    xi, wi, bi = g(x, dev).requires_grad_(True), g(w, dev).requires_grad_(True), g(b, dev).requires_grad_(True)
    y = BinaryDense.apply(xi, wi, bi)
    assert same(n(y), oracle.linear(x, oracle.safe_sign(w)) + b[None, :])
    go = synth.normal(34, (33, 9))
    y.backward(g(go, dev))
    assert norm_err(n(wi.grad), go.T @ x) <= TOL          # plain backward: no STE mask
    assert norm_err(n(xi.grad), go @ oracle.safe_sign(w)) <= TOL


# ---- config-sized cases ------------------------------------------------------------------------------------

@pytest.mark.parametrize("case", ["linbin_odd", "linbin_c2_slice", "linbin_c3_fc1_slice", "linbin_c2_full",
                                  "linter_odd", "linter_c2_slice", "linter_c3_fc1_slice", "linter_c2_full"])
def test_reference_digests(dev, golden_hashes, case):
    """SHA-256 over the int32 result the REFERENCE produced (tests/golden/golden_hashes.json)."""
    h = golden_hashes[case]
    x = synth.pm1(h["x_seed"], (h["B"], h["K"]))
    w = synth.uniform(h["w_seed"], (h["N"], h["K"]), h["w_lo"], h["w_hi"])
    cls = LinearBin if case.startswith("linbin") else LinearTer
    layer = cls(h["K"], h["N"], bias=False).to(dev)
    layer.weight.data.copy_(g(w, dev))
    from pytorch_quantize_impls_amd.functions import _fused
    for impl, entry in (("valu", "qt_xnor_gemm" if cls is LinearBin else "qt_tern_gemm"), ("mfma", "qt_nib_gemm")):
        _fused.GEMM_IMPL = impl
        try:
            with torch.no_grad(), used(entry):
                y = layer(BinaryConnectDeterministic.apply(g(x, dev)))
        finally:
            _fused.GEMM_IMPL = "auto"
        yi = n(y).astype(np.int32)
        assert hashlib.sha256(yi.tobytes()).hexdigest() == h["sha256_int32"], impl


def test_c2_full_size_properties(dev):
    """Size-independent properties at BASELINE.json's full size (4096 x 4096 x 4096)."""
    B = K = N = 4096
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    x = torch.randn((B, K), device=dev, generator=gen)
    w = torch.randn((N, K), device=dev, generator=gen)
    xp, xs = ops.sign_pack(x, want_f32=True)
    wp, ws = ops.sign_pack(w, want_f32=True)
    y = ops.xnor_gemm(xp, wp)
    # (1) parity and range: y == K (mod 2), |y| <= K
    yi = y.to(torch.int64)
    assert torch.equal(yi.to(torch.float32), y)
    assert bool(((yi - K) % 2 == 0).all()) and int(yi.abs().max()) <= K
    # (2) checksum of checksums against an independent fp64 evaluation: sum_n y[m,n] = x[m].sum_n w[n]
    colsum = ws.to(torch.float64).sum(0)
    assert torch.equal(y.to(torch.float64).sum(1), xs.to(torch.float64) @ colsum)
    rowsum = xs.to(torch.float64).sum(0)
    assert torch.equal(y.to(torch.float64).sum(0), ws.to(torch.float64) @ rowsum)
    # (3) antisymmetry: negating the activations negates the output
    y_neg = ops.xnor_gemm(ops.sign_pack(-xs)[0], wp)
    assert torch.equal(y_neg, -y)
    # (4) a random 64x64 sub-block against the dense fp32 product of the +-1 images
    idx = torch.randint(0, B, (64,), device=dev, generator=gen)
    jdx = torch.randint(0, N, (64,), device=dev, generator=gen)
    sub = xs[idx].to(torch.float64) @ ws[jdx].to(torch.float64).t()
    assert torch.equal(y[idx][:, jdx].to(torch.float64), sub)
    # (5) ternary at full size: checksum of checksums
    tp = ops.ternary_pack(w)
    yt = ops.tern_gemm(xp, tp)
    wt = ops.ternarize(w).to(torch.float64)
    assert torch.equal(yt.to(torch.float64).sum(1), xs.to(torch.float64) @ wt.sum(0))
    assert torch.equal(yt[idx][:, jdx].to(torch.float64), xs[idx].to(torch.float64) @ wt[jdx].t())


@pytest.mark.usefixtures("exact_split")
def test_alexnet_bin_layerwise(dev):
    """Every binarised layer of AlexNet-Bin on the GPU against the same layer on
    the CPU, fed with the CPU model's own intermediate activations (so a sign flip near a BN threshold
    in one layer cannot mask or fake a mismatch in the next)."""
    import bench_models
    torch.manual_seed(5)
    cpu = bench_models.AlexNetBin()
    bench_models.randomize_bn(cpu)
    cpu.eval()
    gpu = bench_models.AlexNetBin()
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.to(dev).eval()
    captured = {}
    hooks = []
    names = {m: k for k, m in cpu.named_modules()}
    for m in cpu.modules():
        if isinstance(m, (BinConv2d, LinearBin)):
            hooks.append(m.register_forward_hook(lambda mod, inp, out: captured.__setitem__(names[mod], (inp[0], out))))
    with torch.no_grad():
        cpu(torch.randn(2, 3, 224, 224))
    for h in hooks:
        h.remove()
    gmods = dict(gpu.named_modules())
    assert len(captured) == 8
    for name, (xin, yout) in captured.items():
        binary = bool(((xin == 1) | (xin == -1)).all())
        packed_entries = ("qt_nib_gemm", "qt_xnor_gemm", "qt_conv2d_implicit")
        float_entries = ("qt_bf16x3_pack_f32", "qt_conv_first_direct_f32")   # real-valued input: split + implicit GEMM, or (strided
        float_before = sum(_lib.call_counts[k] for k in float_entries)        # first layers since round 4) the direct kernel
        before = sum(_lib.call_counts[k] for k in packed_entries)
        with torch.no_grad():
            y = lazy.resolve(gmods[name](xin.to(dev)))
        ran_float = sum(_lib.call_counts[k] for k in float_entries) > float_before
        ran_packed = sum(_lib.call_counts[k] for k in packed_entries) > before and not ran_float
        assert ran_packed == binary, name                   # only features.0 sees real pixels ...
        assert ran_float == (not binary), name              # ... and it takes the bf16x3 path, not a library
        assert norm_err(n(y), yout.numpy()) <= TOL, name
        if binary:   # integer part exact: subtract the bias and compare as integers
            b = gmods[name].bias.detach().cpu().numpy()
            shape = (1, -1, 1, 1) if y.dim() == 4 else (1, -1)
            assert np.array_equal(np.rint(n(y) - b.reshape(shape)), np.rint(yout.numpy() - b.reshape(shape))), name


# ---- DoReFa W1A4: int8 code planes + int8 MFMA --------------------------------------------------------------

def test_dorefa_codes_and_i8_gemm_vs_oracle(dev, oracle):
    for (M, N, K) in [(5, 7, 31), (130, 70, 200), (300, 260, 1000), (257, 129, 4096)]:
        x = np.maximum(synth.normal(M + K, (M, K)) * 1.5, 0)
        w = synth.uniform(N + K, (N, K), -1.5, 1.5)
        b = synth.normal(N, (N,))
        with used("qt_dorefa_codes_i8", "qt_weight_codes_i8", "qt_i8_gemm"):
            cp, y = ops.dorefa_codes(g(x, dev), 4)
            wc = ops.weight_codes(g(w, dev))
            out = n(ops.i8_gemm(cp, wc, 0.25, g(b, dev), scale_dev=torch.tensor(2.0, device=dev)))
        assert same(n(y), oracle.dorefa_quantize(x, 4))
        q = np.rint(np.float32(15) * x)
        assert np.array_equal(n(cp.codes)[:, :K].astype(np.float32), q) and not n(cp.codes)[:, K:].any()
        assert np.array_equal(n(wc.codes)[:, :K].astype(np.float32), oracle.safe_sign(w))
        assert cp.usable()
        want = (q.astype(np.float64) @ oracle.safe_sign(w).astype(np.float64).T).astype(np.float32) * np.float32(0.5) + b
        assert same(out, want)                      # integer part exact, one scale multiply, bias once
    # the reference does not clamp: a code beyond int8 must be flagged, not wrapped
    big = np.zeros((3, 40), np.float32)
    big[1, 7] = 9.0                                   # 15 * 9 = 135 > 127
    assert not ops.dorefa_codes(g(big, dev), 4)[0].usable()
    assert ops.dorefa_codes(g(big * 0.9, dev), 4)[0].usable()   # 121.5 -> 122 fits


def test_dorefa_w1a4_layers_golden(dev, golden):
    from pytorch_quantize_impls_amd.functions import nnDorefaQuant as Q
    for name in golden["g8_cases"].tolist():
        x, w = golden[f"g8_{name}_x"], golden[f"g8_{name}_w"]
        has_b = f"g8_{name}_b" in golden.files
        if name.startswith("lin"):
            layer = LinearDorefa(w.shape[1], w.shape[0], bias=has_b, bit_width=1).to(dev)
        else:
            p = name.split("_")
            layer = DorefaConv2d(w.shape[1], w.shape[0], w.shape[2], stride=int(p[4][1:]), padding=int(p[5][1:]),
                                 bias=has_b, bit_width=1).to(dev)
        layer.weight.data.copy_(g(w, dev))
        if has_b:
            layer.bias.data.copy_(g(golden[f"g8_{name}_b"], dev))
        xi = g(x, dev)
        if not name.startswith("lin"):
            xi = xi.contiguous(memory_format=torch.channels_last)
        xi.requires_grad_(True)
        with used("qt_dorefa_codes_i8", "qt_i8_gemm" if name.startswith("lin") else "qt_conv2d_implicit"):
            xq = Q(4)(torch.relu(xi))
            y = layer(xq)
        assert same(n(xq), golden[f"g8_{name}_xq"])
        assert norm_err(n(y), golden[f"g8_{name}_y"]) <= TOL, name
        if name.startswith("lin"):
            y.backward(g(golden[f"g8_{name}_gout"], dev))
            assert norm_err(n(xi.grad), golden[f"g8_{name}_gx"]) <= TOL
            assert norm_err(n(layer.weight.grad), golden[f"g8_{name}_gw"]) <= TOL
            layer.train(False)
            with torch.no_grad(), used("qt_i8_gemm"):
                ye = layer(Q(4)(torch.relu(g(x, dev))))
            assert norm_err(n(ye), golden[f"g8_{name}_y_eval"]) <= TOL


# ---- XNOR-Net: HIP weight quantiser (sign * column mean), dense fp32 contraction -----------------------------

def test_xnor_weight_kernel_vs_oracle(dev, oracle):
    for shape, lead in (((7, 33), 1), ((300, 1000), 1), ((6, 5, 3, 3), 2), ((64, 96, 5, 5), 2), ((3, 1), 1)):
        w = synth.uniform(sum(shape), shape, -2, 2)
        w.reshape(-1)[::7] = 0.0
        with used("qt_xnor_weight_f32"):
            wq, alpha = ops.xnor_weight(g(w, dev), lead)
        want = oracle.xnor_dense_weight(w) if lead == 1 else oracle.xnor_conv_weight(w)
        assert alpha.shape == ((1,) * lead + tuple(shape[lead:]))
        assert norm_err(n(wq), want) <= TOL   # fp32 column sums: order-dependent in the last ulps
        assert np.array_equal(n(wq) == 0, w == 0)           # torch.sign keeps zeros


def test_xnor_layers_golden(dev, golden):
    for name in golden["g4_lin_cases"].tolist():
        gg = lambda s: golden[f"g4_lin_{name}_{s}"]
        has_b = f"g4_lin_{name}_b" in golden.files
        x, w, gout = gg("x"), gg("w"), gg("gout")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            layer = LinearXNOR(x.shape[1], w.shape[0], bias=has_b).to(dev)
        layer.weight.data.copy_(g(w, dev))
        if has_b:
            layer.bias.data.copy_(g(gg("b"), dev))
        xi = g(x, dev).requires_grad_(True)
        with used("qt_xnor_weight_f32"):
            y = layer(xi)
        y.backward(g(gout, dev))
        assert norm_err(n(y), gg("xnor_y")) <= TOL, name
        assert norm_err(n(xi.grad), gg("xnor_gx")) <= TOL
        assert norm_err(n(layer.weight.grad), gg("xnor_gw")) <= TOL
    for name in golden["g4_conv_cases"].tolist():
        p = name.split("_")
        Cin, Cout, k, st, pd = int(p[0][1:]), int(p[1][1:]), int(p[2][1:]), int(p[3][1:]), int(p[4][1:])
        has_b = p[7] == "bias"
        layer = XNORConv2d(Cin, Cout, k, stride=st, padding=pd, bias=has_b).to(dev)
        layer.weight.data.copy_(g(golden[f"g4_conv_{name}_w"], dev))
        if has_b:
            layer.bias.data.copy_(g(golden[f"g4_conv_{name}_b"], dev))
        before = dict(_lib.call_counts)
        y = layer(g(golden[f"g4_conv_{name}_x"], dev))
        # real-valued input: alpha + sign(W) * alpha image (qt_xnor_weight_f32), six-term conv; +-1 input: alpha + Horner
        # tables (qt_xnor_tap_prep_f32), per-tap scaled fp4 conv (round 4)
        assert sum(_lib.call_counts[k] - before.get(k, 0) for k in ("qt_xnor_weight_f32", "qt_xnor_tap_prep_f32")) > 0
        assert norm_err(n(y), golden[f"g4_conv_{name}_xnor_y"]) <= TOL, name


# ---- fused inference epilogue -------------------------------------------------------------------

def test_fused_pool_bn_sign_pack_vs_unfused(dev, oracle):
    from pytorch_quantize_impls_amd.layers import FusedPoolBnSign
    gen = torch.Generator(device=dev)
    gen.manual_seed(11)
    for (N, C, H, pk, ps) in [(3, 64, 13, 3, 2), (2, 192, 55, 3, 2), (4, 36, 9, 1, 1), (5, 128, 1, 1, 1)]:
        x = (torch.randn((N, C, H, H), device=dev, generator=gen) * 20).round()      # integer-valued like conv outputs
        x = x.contiguous(memory_format=torch.channels_last)
        bn = torch.nn.BatchNorm2d(C).to(dev).eval()
        bn.running_mean.copy_(torch.randn(C, device=dev, generator=gen) * 3 + 0.37)
        bn.running_var.copy_(torch.rand(C, device=dev, generator=gen) * 50 + 1)
        bn.weight.data.copy_(torch.randn(C, device=dev, generator=gen))               # negative gammas too
        bn.bias.data.copy_(torch.randn(C, device=dev, generator=gen))
        pool = torch.nn.MaxPool2d(pk, ps) if pk > 1 else None
        with torch.no_grad(), used("qt_pool_affine_sign_pack_nhwc"):
            act = FusedPoolBnSign(bn, pool)(x)
            xin = pool(x) if pool is not None else x
            alpha, beta = (1.0 / torch.sqrt(bn.running_var + bn.eps)) * bn.weight, None
            beta = bn.bias - bn.running_mean * alpha
            t = xin * alpha.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)                 # the folded expression
            want = ops.sign_pack(ops.binarize(t).permute(0, 2, 3, 1).contiguous())[0]
        assert act.shape == tuple(xin.shape)
        assert torch.equal(act.planes.sign, want.sign)
        # and against torch's own eval BatchNorm + Hardtanh + sign: identical except (rarely) at a threshold
        with torch.no_grad():
            ref = torch.nn.functional.hardtanh(bn(xin))
            ref_bits = ops.sign_pack(ops.binarize(ref).permute(0, 2, 3, 1).contiguous())[0].sign
        diff = (act.planes.sign ^ ref_bits)
        flips = sum(bin(int(v) & 0xFFFFFFFF).count("1") for v in diff.flatten().tolist() if v)
        assert flips <= max(2, int(1e-5 * xin.numel())), flips


def test_fused_alexnet_matches_unfused(dev):
    import bench_models
    torch.manual_seed(9)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    # the device fold (thresholds bisected on this device's own F.batch_norm): the fused form IS the module graph, bit for bit.
    # (The "reference" fold, whose BatchNorm arithmetic is ATen-CPU's, is pinned block by block with tie accounting in
    # tests/test_gpu_r2.py::test_c3_fused_alexnet_layerwise.)
    fused = bench_models.FusedAlexNetBin(model, fold="device")
    x = torch.randn((4, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad(), used("qt_pool_affine_sign_pack_nhwc", "qt_conv2d_implicit", "qt_xnor_gemm"), lazy.eager():
        yf = fused(x)
        yu = model(x)                  # module by module (the deferred execution of the same graph: test_gpu_lazy.py)
    assert yf.shape == yu.shape == (4, 10)
    assert torch.equal(yf, yu)


# ---- real-valued activations: exact bf16 triple split + bf16 MFMA GEMM -----------------------------------------

@pytest.mark.usefixtures("exact_split")
def test_bf16x3_split_is_exact(dev):
    x = np.concatenate([synth.normal(1, (5000,)) * 3, synth.uniform(2, (5000,), -1e-3, 1e-3),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.14159274, 1e-20, -1e20, 65504.0, 1.0000001], np.float32)])
    x = x.reshape(1, -1).astype(np.float32)
    with used("qt_bf16x3_pack_f32"):
        tp = ops.split_bf16x3(g(x, dev))
    raw = n(tp.data).view(np.uint16)[0, :3 * x.shape[1]].astype(np.uint32) << 16
    terms = raw.view(np.float32).reshape(-1, 3).astype(np.float64)
    assert np.array_equal(terms.sum(1).astype(np.float32), x[0])                 # hi + mid + lo == x exactly
    assert not n(tp.data)[0, 3 * x.shape[1]:].any()                                  # zero pad
    w = synth.uniform(3, (4, 37), -1.5, 1.5)
    w[0, 0] = 0.0
    for kind, q in (("binary", np.where(w < 0, -1.0, 1.0)), ("sign", np.sign(w)),
                    ("ternary", np.where(w >= 0.5, 1.0, np.where(w < -0.5, -1.0, 0.0)))):
        wt = ops.weight_bf16x3(g(w, dev), kind)
        vals = (n(wt.data).view(np.uint16)[:, :3 * 37].astype(np.uint32) << 16).view(np.float32).reshape(4, 37, 3)
        assert np.array_equal(vals, np.repeat(q[:, :, None], 3, 2).astype(np.float32)), kind


@pytest.mark.parametrize("M,N,K", [(5, 7, 31), (130, 70, 363), (300, 260, 1000), (257, 129, 4096), (1024, 1024, 784)])
@pytest.mark.usefixtures("exact_split")
def test_float_linear_bf16x3_vs_fp64(dev, M, N, K):
    x = synth.normal(M + K, (M, K)) * 2.0
    w = synth.uniform(N + K, (N, K), -1.5, 1.5)
    b = synth.normal(N, (N,))
    alpha = np.abs(synth.normal(K, (K,))) + 0.1
    for kind, q in (("binary", np.where(w < 0, -1.0, 1.0)), ("ternary", np.where(w >= 0.5, 1.0, np.where(w < -0.5, -1.0, 0.0)))):
        with used("qt_bf16x3_pack_f32", "qt_bf16_gemm"):
            y = n(ops.float_linear(g(x, dev), g(w, dev), kind, g(b, dev)))
        ref = x.astype(np.float64) @ q.T + b
        assert norm_err(y, ref) <= TOL, kind                        # fp32-GEMM class accuracy (1.6e-6 at K=4096)
    ya = n(ops.float_linear(g(x, dev), g(w, dev), "sign", None, alpha=g(alpha, dev)))
    refa = (x * alpha[None, :]).astype(np.float32).astype(np.float64) @ np.sign(w).T
    assert norm_err(ya, refa) <= TOL


@pytest.mark.usefixtures("exact_split")
def test_float_conv_bf16x3_vs_fp64(dev):
    gen = torch.Generator(device=dev)
    gen.manual_seed(21)
    for (Cin, Cout, k, st, pd, H) in [(3, 192, 11, 4, 2, 224), (3, 64, 3, 1, 1, 32), (20, 9, 5, 2, 2, 17)]:
        x = torch.randn((2, Cin, H, H), device=dev, generator=gen)
        w = torch.randn((Cout, Cin, k, k), device=dev, generator=gen)
        b = torch.randn((Cout,), device=dev, generator=gen)
        with used("qt_bf16x3_pack_f32", "qt_conv2d_implicit"):
            y2 = ops.float_conv2d(x, w, "binary", b, st, pd)
        ops.CONV_IMPLICIT = False
        try:
            with used("qt_im2col_words", "qt_bf16_gemm"):
                y2e = ops.float_conv2d(x, w, "binary", b, st, pd)
        finally:
            ops.CONV_IMPLICIT = True
        assert torch.equal(y2, y2e)     # same operands, same accumulation order: bitwise equal
        Ho = (H + 2 * pd - k) // st + 1
        y = y2.view(2, Ho, Ho, Cout).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x.double(), ops.binarize(w).double(), b.double(), st, pd)
        assert norm_err(n(y), n(ref)) <= TOL


@pytest.mark.usefixtures("exact_split")
def test_strided_first_layer_conv_s2d_vs_fp64(dev, s2d_first_layer):
    """BinConv2d on real pixels with stride > 1 goes through the space-to-depth form; same numbers."""
    from pytorch_quantize_impls_amd.functions import _fused
    gen = torch.Generator(device=dev)
    gen.manual_seed(33)
    for (Cin, Cout, k, st, pd, H) in [(3, 192, 11, 4, 2, 224), (3, 16, 7, 2, 3, 33), (1, 8, 5, 3, 1, 20)]:
        x = torch.randn((2, Cin, H, H), device=dev, generator=gen).contiguous(memory_format=torch.channels_last)
        conv = BinConv2d(Cin, Cout, k, stride=st, padding=pd).to(dev)
        ref = torch.nn.functional.conv2d(x.double(), ops.binarize(conv.weight.detach()).double(),
                                         conv.bias.detach().double(), st, pd)
        outs = []
        for use in (True, False):
            _fused.USE_S2D = use
            try:
                with torch.no_grad(), used("qt_conv2d_implicit", "qt_bf16x3_pack_f32"):
                    outs.append(conv(x))
            finally:
                _fused.USE_S2D = True
        assert outs[0].shape == ref.shape
        assert norm_err(n(outs[0]), n(ref)) <= TOL and norm_err(n(outs[1]), n(ref)) <= TOL


# ---- utils: packed checkpoint on a device model -----------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("family", ["binary", "ternary", "dorefa1"])
def test_packed_state_device_planes_equal_cpu_planes(dev, family):
    import torch.nn as nn
    from pytorch_quantize_impls_amd import utils as U
    conv = {"binary": U.binary_net_convert, "ternary": U.ternary_net_convert,
            "dorefa1": lambda n: U.dorefa_net_convert(n, weight_bit=1)}[family]
    torch.manual_seed(11)
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.Hardtanh(), nn.Flatten(),
                        nn.Linear(8 * 5 * 5, 37))
    m = conv(net)
    for p in m.parameters():
        p.data.uniform_(-1.2, 1.2)
    m.eval()
    st_cpu = U.packed_state_dict(m)
    before = dict(_lib.call_counts)
    md = copy.deepcopy(m).to(dev)
    st_dev = U.packed_state_dict(md)
    name = "qt_ternary_pack_f32" if family == "ternary" else "qt_sign_pack_f32"
    assert _lib.call_counts.get(name, 0) - before.get(name, 0) == 2      # packed by the HIP kernels
    for k, e in st_cpu["layers"].items():
        assert torch.equal(e["sign"], st_dev["layers"][k]["sign"])
        if family == "ternary":
            assert torch.equal(e["mask"], st_dev["layers"][k]["mask"])
    fresh = conv(net).to(dev)
    U.load_packed_state_dict(fresh, st_cpu)
    x = torch.randn(4, 3, 5, 5)
    y_dev = fresh(x.to(dev)).cpu()
    y_cpu = m(x)
    assert (y_dev - y_cpu).abs().max() <= 1e-5 * y_cpu.abs().max()


# ---- conv with the threshold-bit epilogue + pooling on bits (no fp32 activation between binarised layers) ----

@pytest.mark.gpu
def test_pool_bits_vs_numpy(dev):
    rng = np.random.default_rng(5)
    for (N, C, H, W, k, s) in [(2, 40, 7, 9, 3, 2), (1, 192, 13, 13, 3, 2), (3, 33, 6, 6, 2, 2), (2, 64, 5, 5, 3, 1)]:
        ld = ops.packed_ld(C)
        bits = rng.integers(0, 2, size=(N, H, W, C), dtype=np.uint8)
        alpha = rng.standard_normal(C).astype(np.float32)
        alpha[::7] = 0.0
        plane, _ = ops.sign_pack(g(np.where(bits.reshape(-1, C) == 1, -1.0, 1.0).astype(np.float32), dev))
        na = ops.neg_alpha_words(g(alpha, dev))
        with used("qt_pool_bits"):
            out, (Ho, Wo) = ops.pool_bits(plane, N, H, W, k, s, na)
        assert (Ho, Wo) == ((H - k) // s + 1, (W - k) // s + 1) and out.ld == ld
        win = np.stack([bits[:, i:i + s * Ho:s, j:j + s * Wo:s, :] for i in range(k) for j in range(k)])
        want = np.where(alpha < 0, win.max(0), win.min(0)).astype(np.uint8)        # OR where alpha < 0, else AND
        got, _ = ops.sign_pack(g(np.where(want.reshape(-1, C) == 1, -1.0, 1.0).astype(np.float32), dev))
        assert torch.equal(out.sign, got.sign)


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,k,st,pd,pool", [(64, 96, 3, 1, 1, None), (32, 70, 5, 1, 2, (3, 2)), (96, 256, 3, 1, 1, (3, 2)),
                                                   (64, 192, 3, 2, 1, (2, 2)), (40, 33, 1, 1, 0, None)])
@pytest.mark.parametrize("kind", ["binary", "ternary"])
def test_fused_conv_pool_bn_sign_vs_oracle(dev, oracle, Cin, Cout, k, st, pd, pool, kind):
    from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d, FusedConvPoolBnSign, FusedPoolBnSign
    from pytorch_quantize_impls_amd import packed as pk
    N, H = 3, 13
    x = synth.pm1(Cin + Cout, (N, Cin, H, H))
    w = synth.uniform(Cout + k, (Cout, Cin, k, k), -1.2, 1.2)
    b = synth.normal(k + 3, (Cout,)) * 2
    cls = BinConv2d if kind == "binary" else TerConv2d
    conv = cls(Cin, Cout, k, stride=st, padding=pd).to(dev)
    conv.weight.data.copy_(g(w, dev)); conv.bias.data.copy_(g(b, dev))
    conv.eval()
    bn = torch.nn.BatchNorm2d(Cout).to(dev).eval()
    bn.running_mean.copy_(g(synth.normal(1, (Cout,)) * 5, dev)); bn.running_var.copy_(g(synth.uniform(2, (Cout,), 1, 60), dev))
    bn.weight.data.copy_(g(synth.normal(3, (Cout,)), dev)); bn.bias.data.copy_(g(synth.normal(4, (Cout,)), dev))   # negative gammas too
    mp = torch.nn.MaxPool2d(*pool) if pool else None
    fused = FusedConvPoolBnSign(conv, bn, mp)
    xin = g(x, dev).contiguous(memory_format=torch.channels_last)
    act_in = pk.PackedActivation(ops.sign_pack(xin.permute(0, 2, 3, 1).contiguous())[0], tuple(xin.shape))
    with torch.no_grad(), used("qt_conv2d_implicit_bits"):
        act = fused(act_in)                       # packed in, packed out
        act_t = fused(xin)                        # +-1 fp32 tensor in (detected), packed out
        two_step = FusedPoolBnSign(bn, mp)(conv(xin)) if Cout % 4 == 0 else None   # fp32 conv output -> pool/BN/sign kernel
    assert torch.equal(act.planes.sign, act_t.planes.sign) and act.shape == act_t.shape
    if two_step is not None:                                           # (that kernel needs C % 4 == 0; the bit route does not)
        assert act.shape == two_step.shape and torch.equal(act.planes.sign, two_step.planes.sign)
    from pytorch_quantize_impls_amd.layers import fold_batchnorm
    alpha, beta = (n(t) for t in fold_batchnorm(bn))
    wq = oracle.safe_sign(w) if kind == "binary" else oracle.ternarize(w)
    want, (Ho, Wo) = oracle.bin_conv_pool_bn_sign_planes(x, wq, b, st, pd, 1, alpha, beta, *(pool or (1, 1)))
    assert act.shape == (N, Cout, Ho, Wo)
    assert np.array_equal(n(act.planes.sign).view(np.uint32), want)    # and to the CPU restatement of the chain


@pytest.mark.gpu
def test_fused_first_layer_conv_bits_match_fp32_route(dev):
    """Real-valued input (bf16-triple conv, space-to-depth form): the threshold-bit epilogue sees the same
    accumulators as the fp32 epilogue, so the planes must equal conv -> FusedPoolBnSign exactly."""
    from pytorch_quantize_impls_amd.layers import BinConv2d, FusedConvPoolBnSign, FusedPoolBnSign
    torch.manual_seed(3)
    for (Cin, Cout, k, st, pd, HW, pool) in [(3, 192, 11, 4, 2, 67, (3, 2)), (3, 64, 3, 1, 1, 20, None), (3, 40, 5, 2, 2, 31, (2, 2))]:
        conv = BinConv2d(Cin, Cout, k, stride=st, padding=pd).to(dev).eval()
        bn = torch.nn.BatchNorm2d(Cout).to(dev).eval()
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 30); bn.weight.data.normal_(); bn.bias.data.normal_()
        mp = torch.nn.MaxPool2d(*pool) if pool else None
        x = torch.randn((2, Cin, HW, HW), device=dev).contiguous(memory_format=torch.channels_last)
        before = dict(_lib.call_counts)
        with torch.no_grad():
            act = FusedConvPoolBnSign(conv, bn, mp)(x)
            ref = FusedPoolBnSign(bn, mp)(conv(x))
        # strided first layers: the direct kernel (round 4), both epilogues; stride 1: the implicit GEMM on bf16 triples
        ran = {k_: _lib.call_counts[k_] - before.get(k_, 0) for k_ in ("qt_conv2d_implicit_bits", "qt_conv_first_direct_bits_f32",
                                                                      "qt_conv2d_implicit", "qt_conv_first_direct_f32",
                                                                      "qt_conv3x3_first_f32")}
        # (3 -> 64, 3 x 3, stride 1: the one-pass first-layer kernel of round 6, threshold-bit and fp32 epilogue of ONE accumulation)
        assert (ran["qt_conv_first_direct_bits_f32"] == 1 and ran["qt_conv_first_direct_f32"] == 1) if st > 1 else \
            ran["qt_conv3x3_first_f32"] == 2, ran
        assert act.shape == ref.shape and torch.equal(act.planes.sign, ref.planes.sign)


@pytest.mark.gpu
def test_fused_alexnet_conv_bits_equals_fp32_fusion(dev):
    import bench_models
    torch.manual_seed(9)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn((4, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad(), used("qt_conv2d_implicit_bits", "qt_pool_bits"):
        y_bits = bench_models.FusedAlexNetBin(model, fuse_conv=True)(x)
    before = dict(_lib.call_counts)
    with torch.no_grad():
        y_f32 = bench_models.FusedAlexNetBin(model, fuse_conv=False)(x)
    assert _lib.call_counts["qt_conv2d_implicit_bits"] == before["qt_conv2d_implicit_bits"]
    assert torch.equal(y_bits, y_f32)


@pytest.mark.gpu
def test_conv_ping_pong_kernels_equal_double_buffered(dev, s2d_first_layer):
    """qt_conv2d_implicit_variant(2): the ping-pong main loop (ring of 4 64-byte stages, wrapped W pieces for the
    192- and 64-wide tiles) must reproduce the double-buffered conv kernels (variant 1) and the automatic choice bit for
    bit, all four tile widths."""
    from pytorch_quantize_impls_amd.layers import BinConv2d
    try:
        for (Cin, Cout, k, st, pd, H) in [(64, 256, 3, 1, 1, 13), (96, 192, 5, 1, 2, 14), (32, 128, 3, 2, 1, 17), (64, 40, 1, 1, 0, 9),
                                          (3, 192, 11, 4, 2, 67)]:
            conv = BinConv2d(Cin, Cout, k, stride=st, padding=pd).to(dev).eval()
            x = g(synth.pm1(Cin + H, (3, Cin, H, H)) if Cin > 3 else synth.normal(7, (3, Cin, H, H)), dev)
            x = x.contiguous(memory_format=torch.channels_last)
            outs = []
            for which in (0, 1, 2, 4):
                ops.CONV_VARIANT = which
                with torch.no_grad(), used("qt_conv2d_implicit_variant" if which else "qt_conv2d_implicit"):
                    outs.append(conv(x).clone())
            assert all(torch.equal(outs[0], o) for o in outs[1:]), (Cin, Cout, k)
    finally:
        ops.CONV_VARIANT = 0


@pytest.mark.gpu
@pytest.mark.usefixtures("exact_split")
def test_s2d_triple_pack_row_staged_kernel_equals_generic(dev):
    """Channels-last images take the LDS-staged kernel, NCHW storage the generic gather: identical planes
    (incl. zero padding rows/columns, ragged sizes, the non-vectorisable W*C % 4 != 0 case)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    for (N, C, H, W, s, pad) in [(2, 3, 224, 224, 4, 2), (3, 3, 33, 35, 2, 3), (2, 1, 20, 23, 3, 1), (1, 4, 17, 9, 2, 0), (2, 3, 31, 30, 4, (1, 2))]:
        x = torch.randn((N, C, H, W), device=dev, generator=gen)
        xl = x.contiguous(memory_format=torch.channels_last)
        with used("qt_bf16x3_s2d_pack_f32"):
            a, hw_a = ops.s2d_triple_pack(xl, s, pad)
            b, hw_b = ops.s2d_triple_pack(x.contiguous(), s, pad)
        assert hw_a == hw_b and a.data.shape == b.data.shape
        assert torch.equal(a.data, b.data), (N, C, H, W, s, pad)


@pytest.mark.gpu
def test_padded_pixel_planes_route_equals_bounds_checked_route(dev, oracle):
    """Packed-activation convs expand their bits into a physically zero-padded plane and run the un-padded
    (VALID) conv kernels; the bounds-checked kernels on the original plane must give the same bits / floats."""
    from pytorch_quantize_impls_amd.functions import _fused
    from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d
    from pytorch_quantize_impls_amd import packed as pk
    for (cls, Cin, Cout, k, st, pd, dil, H) in [(BinConv2d, 64, 192, 5, 1, 2, 1, 13), (TerConv2d, 96, 256, 3, 2, 1, 1, 15),
                                               (BinConv2d, 32, 70, 3, 1, (2, 1), 2, 12), (BinConv2d, 192, 384, 3, 1, 1, 1, 13)]:
        conv = cls(Cin, Cout, k, stride=st, padding=pd, dilation=dil).to(dev).eval()
        x = g(synth.pm1(Cin + Cout, (3, Cin, H, H + 1)), dev).contiguous(memory_format=torch.channels_last)
        act = pk.PackedActivation(ops.sign_pack(x.permute(0, 2, 3, 1).contiguous())[0], tuple(x.shape))
        outs = []
        for flag in (True, False):
            _fused.PAD_PLANES = flag
            try:
                with torch.no_grad(), used("qt_bits_to_nib_pad" if flag else "qt_bits_to_nib", "qt_conv2d_implicit"):
                    outs.append(lazy.resolve(conv(act)))
            finally:
                _fused.PAD_PLANES = True
        assert torch.equal(outs[0], outs[1]), (Cin, Cout, k)
        wq = oracle.safe_sign(n(conv.weight)) if cls is BinConv2d else n(conv.weight)
        ref = oracle.conv2d(n(x), wq, n(conv.bias), st, pd, dil)
        assert norm_err(n(outs[0]), ref) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kw_bits,ka_bits", [(2, 2), (3, 4), (4, 4), (7, 3)])
def test_dorefa_wk_ak_inference_on_int8_matrix_cores(dev, oracle, kw_bits, ka_bits):
    """Eval-mode LinearDorefa / DorefaConv2d with k-bit weights (2..7) on code-carrying activations: integer
    weight levels x activation codes on the int8 MFMA kernels, against the fp64 evaluation of the reference
    formula (functions/dorefa_connect.py:82-113 weights, :11-25 activations) and the layer's own dense route."""
    from pytorch_quantize_impls_amd.functions import nnDorefaQuant as Q
    # linear
    x = np.abs(synth.normal(kw_bits + 10, (37, 200))) * 0.6
    w = synth.normal(ka_bits + 20, (29, 200)) * 0.8
    b = synth.normal(3, (29,))
    layer = LinearDorefa(200, 29, bit_width=kw_bits).to(dev)
    layer.weight.data.copy_(g(w, dev)); layer.bias.data.copy_(g(b, dev))
    layer.eval()
    wq = oracle.dorefa_weight(w, kw_bits).astype(np.float64)
    with torch.no_grad(), used("qt_dorefa_codes_i8", "qt_i8_gemm"):
        xq = Q(ka_bits)(g(x, dev))
        y = layer(xq)
        codes = ops.dorefa_weight_codes(layer.weight.detach(), kw_bits)
    n_w = (1 << kw_bits) - 1
    lv = n(codes.codes)[:, :200].astype(np.int64)
    assert np.array_equal(lv, np.rint(wq * n_w).astype(np.int64)) and np.all(np.abs(lv) % 2 == 1) and np.abs(lv).max() <= n_w
    ref = oracle.dorefa_quantize(x, ka_bits).astype(np.float64) @ wq.T + b
    assert norm_err(n(y), ref) <= TOL
    with torch.no_grad():
        dense = torch.nn.functional.linear(xq.clone(), layer.weight, layer.bias)      # clone: no tag -> library route
    assert norm_err(n(y), n(dense)) <= TOL
    # conv
    xc = np.abs(synth.normal(kw_bits + 30, (2, 24, 9, 9))) * 0.5
    wc = synth.normal(ka_bits + 40, (40, 24, 3, 3)) * 0.7
    conv = DorefaConv2d(24, 40, 3, stride=1, padding=1, bit_width=kw_bits).to(dev)
    conv.weight.data.copy_(g(wc, dev))
    conv.eval()
    with torch.no_grad(), used("qt_conv2d_implicit"):
        xq = Q(ka_bits)(g(xc, dev).contiguous(memory_format=torch.channels_last))
        yc = conv(xq)
    refc = oracle.conv2d(oracle.dorefa_quantize(xc, ka_bits), oracle.dorefa_weight(wc, kw_bits), n(conv.bias), 1, 1)
    assert norm_err(n(yc), refc) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("cls", ["bin", "ter"])
@pytest.mark.usefixtures("exact_split")
def test_linear_backward_on_bf16_matrix_cores_vs_fp64(dev, cls):
    """Large training-mode LinearBin / LinearTer: grad_x = g . Q(W) and grad_W = (g^T . x) * STE mask run through the
    exact-split bf16 GEMM (x is a tagged +-1 activation); compare with the fp64 evaluation."""
    from pytorch_quantize_impls_amd.layers import LinearTer
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    M, K, N = 640, 768, 512
    torch.manual_seed(21)
    layer = (LinearBin if cls == "bin" else LinearTer)(K, N).to(dev)
    layer.weight.data.uniform_(-1.3, 1.3)
    pre = torch.randn((M, K), device=dev, requires_grad=True)
    gout = torch.randn((M, N), device=dev)
    before = dict(_lib.call_counts)
    y = layer(BinaryConnect()(pre))
    y.backward(gout)
    assert _lib.call_counts["qt_bf16_gemm"] - before.get("qt_bf16_gemm", 0) == 2       # both backward GEMMs
    w = layer.weight.detach().double()
    wq = (torch.where(w < 0, -1.0, 1.0) if cls == "bin" else torch.where(w >= 0.5, 1.0, torch.where(w < -0.5, -1.0, 0.0))).double()
    xq = torch.where(pre.detach() < 0, -1.0, 1.0).double()
    gx_ref = (gout.double() @ wq) * (pre.detach().abs() <= 1.001).double()
    gw_ref = (gout.double().t() @ xq) * (w.abs() <= 1.001).double()
    assert norm_err(n(pre.grad), n(gx_ref)) <= TOL
    assert norm_err(n(layer.weight.grad), n(gw_ref)) <= TOL
    assert norm_err(n(layer.bias.grad), n(gout.double().sum(0))) <= TOL


# ---- seeded fuzz over shapes: conv / fused conv blocks / packed GEMM against the oracle (bit-exact cases) ----------

@pytest.mark.gpu
def test_fuzz_packed_conv_vs_oracle(dev, oracle):
    from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d, FusedConvPoolBnSign, fold_batchnorm
    from pytorch_quantize_impls_amd import packed as pk
    rng = np.random.default_rng(20260928)
    for it in range(40):
        Cin = int(rng.choice([1, 3, 8, 24, 32, 40, 64, 96, 130]))
        Cout = int(rng.choice([1, 7, 32, 48, 64, 100, 192, 200, 260]))
        kh, kw = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        st = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        pd = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        dl = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        N = int(rng.integers(1, 4))
        H = int(rng.integers(dl[0] * (kh - 1) + 1, 15)) + 2
        W = int(rng.integers(dl[1] * (kw - 1) + 1, 15)) + 2
        ter = bool(rng.integers(0, 2))
        cls = TerConv2d if ter else BinConv2d
        conv = cls(Cin, Cout, (kh, kw), stride=st, padding=pd, dilation=dl, bias=False).to(dev)
        w = synth.uniform(1000 + it, (Cout, Cin, kh, kw), -1.4, 1.4)
        conv.weight.data.copy_(g(w, dev))
        conv.eval()
        x = synth.pm1(2000 + it, (N, Cin, H, W))
        wq = oracle.ternarize(w) if ter else oracle.safe_sign(w)
        want = oracle.conv2d(x, wq, None, st, pd, dl)
        xt = g(x, dev).contiguous(memory_format=torch.channels_last)
        act = pk.PackedActivation(ops.sign_pack(xt.permute(0, 2, 3, 1).contiguous())[0], tuple(xt.shape))
        cfg = (it, Cin, Cout, kh, kw, st, pd, dl, N, H, W, ter)
        with torch.no_grad():
            y_t = conv(xt)                 # +-1 tensor in: bounds-checked conv kernels
            y_p = conv(act)                # packed in: physically padded plane + un-padded kernels
        assert np.array_equal(n(y_t), want), cfg
        assert np.array_equal(n(y_p), want), cfg
        # fused block on the same conv (pooling when the output is large enough)
        Ho, Wo = want.shape[2:]
        pool = torch.nn.MaxPool2d(2, 2) if min(Ho, Wo) >= 2 and it % 2 == 0 else None
        bn = torch.nn.BatchNorm2d(Cout).to(dev).eval()
        bn.running_mean.copy_(g(synth.normal(3000 + it, (Cout,)) * 3, dev))
        bn.running_var.copy_(g(synth.uniform(4000 + it, (Cout,), 0.5, 40), dev))
        bn.weight.data.copy_(g(synth.normal(5000 + it, (Cout,)), dev))
        bn.bias.data.copy_(g(synth.normal(6000 + it, (Cout,)), dev))
        with torch.no_grad():
            out = FusedConvPoolBnSign(conv, bn, pool)(act)
        alpha, beta = (n(t) for t in fold_batchnorm(bn))
        want_bits, _ = oracle.bin_conv_pool_bn_sign_planes(x, wq, None, st, pd, dl, alpha, beta, *((2, 2) if pool else (1, 1)))
        assert np.array_equal(n(out.planes.sign).view(np.uint32), want_bits), cfg


@pytest.mark.gpu
def test_fuzz_packed_gemm_vs_oracle(dev, oracle):
    rng = np.random.default_rng(77)
    for it in range(30):
        M, N, K = (int(v) for v in rng.integers(1, 700, size=3))
        if it % 5 == 0:
            M, N = 384 * int(rng.integers(1, 3)) + int(rng.integers(0, 9)), 192 * int(rng.integers(1, 3))
        x = synth.pm1(100 + it, (M, K))
        w = synth.uniform(200 + it, (N, K), -1.5, 1.5)
        ref_b = oracle.linear(x, oracle.safe_sign(w))
        ref_t = oracle.linear(x, oracle.ternarize(w))
        xb = ops.sign_pack(g(x, dev))[0]
        wb, wt = ops.sign_pack(g(w, dev))[0], ops.ternary_pack(g(w, dev))
        assert np.array_equal(n(ops.xnor_gemm(xb, wb)), ref_b), (M, N, K)
        assert np.array_equal(n(ops.tern_gemm(xb, wt)), ref_t), (M, N, K)
        xn = ops.bits_to_nib(xb)
        assert np.array_equal(n(ops.nib_gemm(xn, ops.bits_to_nib(wb))), ref_b), (M, N, K)
        assert np.array_equal(n(ops.nib_gemm(xn, ops.bits_to_nib(wt))), ref_t), (M, N, K)


@pytest.mark.gpu
def test_fuzz_real_input_conv_vs_fp64(dev, s2d_first_layer):
    """First-layer style convs on real-valued inputs (exact bf16-triple route; space-to-depth form when strided)."""
    rng = np.random.default_rng(4242)
    from pytorch_quantize_impls_amd.layers import TerConv2d
    for it in range(24):
        Cin = int(rng.choice([1, 3, 4, 6, 16]))
        Cout = int(rng.choice([5, 32, 64, 96, 192]))
        k = int(rng.integers(1, 8))
        st = int(rng.integers(1, 5))
        pd = int(rng.integers(0, 4))
        H, W = int(rng.integers(k, 40)) + 3, int(rng.integers(k, 40)) + 3
        cls = TerConv2d if it % 3 == 0 else BinConv2d
        conv = cls(Cin, Cout, k, stride=st, padding=pd).to(dev)
        conv.weight.data.uniform_(-1.3, 1.3)
        conv.bias.data.normal_()
        conv.binary_input = False
        x = torch.randn((2, Cin, H, W), device=dev) * 2.0
        if it % 2:
            x = x.contiguous(memory_format=torch.channels_last)
        w = conv.weight.detach().double()
        wq = (torch.where(w < 0, -1.0, 1.0) if cls is BinConv2d
              else torch.where(w >= 0.5, 1.0, torch.where(w < -0.5, -1.0, 0.0))).double()
        ref = torch.nn.functional.conv2d(x.double(), wq, conv.bias.detach().double(), st, pd)
        for mode in ("train", "eval"):
            conv.train(mode == "train")
            with torch.no_grad(), used("qt_conv2d_implicit"):
                y = lazy.resolve(conv(x))
            assert y.shape == ref.shape and norm_err(n(y), n(ref)) <= TOL, (it, Cin, Cout, k, st, pd, H, W, mode)
        conv.train()


@pytest.mark.gpu
def test_pack_pair_equals_single_operand_packs(dev):
    for (M, N, K) in [(4096, 4096, 4096), (300, 77, 1000), (5, 9, 12), (64, 64, 31), (1, 1, 4)]:
        x = g(synth.normal(M + K, (M, K)), dev)
        w = g(synth.uniform(N + K, (N, K), -1.5, 1.5), dev)
        for kind in ("binary", "ternary"):
            with used("qt_pack_pair_nib_f32"):
                xp, wp = ops.pack_linear_operands(x, w, kind, "mfma")
            assert torch.equal(xp.words, ops.sign_pack_nib(x).words)
            assert torch.equal(wp.words, (ops.sign_pack_nib(w) if kind == "binary" else ops.ternary_pack_nib(w)).words)
    # the layer takes this route for an un-tagged +-1 activation in training mode (large shape -> matrix cores)
    layer = LinearBin(2048, 1024).to(dev)
    xs = torch.randn((2048, 2048), device=dev).sign()
    xs[xs == 0] = 1
    with used("qt_pack_pair_nib_f32", "qt_nib_gemm"):
        y = layer(xs)
    ref = torch.nn.functional.linear(xs.double(), torch.where(layer.weight.detach() < 0, -1.0, 1.0).double(),
                                     layer.bias.detach().double())
    assert norm_err(n(y), n(ref)) <= TOL


@pytest.mark.gpu
def test_graphed_fused_alexnet_replay_equals_eager(dev):
    """utils.graphed: the fused inference forward captured as a hipGraph (C-ABI launches included) replays
    bit-identically for new inputs."""
    import bench_models
    from pytorch_quantize_impls_amd.utils import graphed
    torch.manual_seed(2)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    fused = bench_models.FusedAlexNetBin(model)
    xs = [torch.randn((4, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    gm = graphed(fused, xs[0])
    before = dict(_lib.call_counts)
    for x in xs[::-1]:
        with torch.no_grad():
            want = fused(x)
        n_calls = sum(_lib.call_counts.values())
        got = gm(x).clone()
        assert sum(_lib.call_counts.values()) == n_calls          # replay: no Python-side launches
        assert torch.equal(got, want)
    with pytest.raises(ValueError, match="captured for input"):
        gm(xs[0][:2])


@pytest.mark.gpu
def test_conv_wrappers_reject_mismatched_pixel_planes(dev):
    """A pixel plane that does not hold N*H*W pixels would make the gather read out of bounds: refuse it."""
    x = g(synth.pm1(1, (2, 32, 6, 6)), dev).contiguous(memory_format=torch.channels_last)
    px = ops.pack_pixels_nib(x)
    wp = ops.pack_conv_weight_nib(g(synth.uniform(2, (8, 32, 3, 3), -1, 1), dev), "binary")
    with pytest.raises(ValueError, match="pixel plane holds"):
        ops.conv2d_nib(px, (2, 32, 8, 8), wp, (3, 3), None, 1, 1, 1)
    codes, _ = ops.dorefa_codes(torch.rand((2 * 6 * 6, 32), device=dev), 4, want_f32=False, ld_bytes=32)
    wc = ops.pack_conv_weight_codes(g(synth.uniform(3, (8, 32, 3, 3), -1, 1), dev))
    with pytest.raises(ValueError, match="pixel plane holds"):
        ops.conv2d_codes(codes, (2, 32, 7, 6), wc, (3, 3), 1.0, None, 1, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("C,Cout,k,st,pd", [(32, 48, 3, 1, 1), (64, 200, 5, 1, 2), (96, 64, 3, 2, 1), (32, 16, 3, 1, (2, 0))])
def test_physical_padding_equals_bounds_checked_conv(dev, monkeypatch, C, Cout, k, st, pd):
    """ops.PAD_PIXEL_PLANES (zero border made physical, un-padded kernels) is bit-identical to the per-tap
    bounds-checked kernels, for nibble planes and for DoReFa int8 code planes."""
    N, H, W = 3, 9, 11
    x = g(synth.pm1(5, (N, C, H, W)), dev).contiguous(memory_format=torch.channels_last)
    px = ops.pack_pixels_nib(x)
    wp = ops.pack_conv_weight_nib(g(synth.uniform(6, (Cout, C, k, k), -1, 1), dev), "ternary")
    b = g(synth.uniform(7, (Cout,), -1, 1), dev)
    codes, _ = ops.dorefa_codes(torch.rand((N * H * W, C), device=dev), 4, want_f32=False, ld_bytes=C)
    wc = ops.pack_conv_weight_codes(g(synth.uniform(8, (Cout, C, k, k), -1, 1), dev), ternary=True)
    outs = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "PAD_PIXEL_PLANES", flag)
        outs[flag] = (ops.conv2d_nib(px, (N, C, H, W), wp, (k, k), b, st, pd, 1),
                      ops.conv2d_codes(codes, (N, C, H, W), wc, (k, k), 1.0 / 15, b, st, pd, 1))
    assert torch.equal(outs[True][0], outs[False][0])
    assert torch.equal(outs[True][1], outs[False][1])
    ph, pw = (pd, pd) if isinstance(pd, int) else pd
    q = ops.pad_pixel_plane(codes.codes, N, H, W, pd).view(N, H + 2 * ph, W + 2 * pw, C)
    assert torch.equal(q[:, ph:ph + H, pw:pw + W], codes.codes.view(N, H, W, C))
    assert int(q.abs().sum()) == int(codes.codes.abs().sum())


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k,relu,res", [((3, 64, 5, 7), 4, True, None), ((2, 36, 4, 4), 4, True, "codes"),
                                              ((2, 128, 3, 5), 3, True, "f32"), ((2, 20, 6, 3), 4, False, "f32bn"),
                                              ((5, 48), 2, True, "codes"), ((4, 7, 3, 3), 8, True, "f32")])
def test_fused_bn_dorefa_quant_vs_oracle(dev, oracle, shape, k, relu, res):
    """layers.FusedBnDorefaQuant (BatchNorm + residual + ReLU + k-bit quantiser in one kernel, int8 code plane out)
    against the oracle's fp32 restatement of the chain: codes and fp32 image bit for bit."""
    from pytorch_quantize_impls_amd.layers import FusedBnDorefaQuant, fold_batchnorm
    from pytorch_quantize_impls_amd import packed
    C = shape[1]
    x = synth.uniform(11, shape, -3, 3)
    bn = (torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d)(C)
    bn.running_mean.copy_(torch.from_numpy(synth.uniform(12, (C,), -1, 1)))
    bn.running_var.copy_(torch.from_numpy(synth.uniform(13, (C,), 0.5, 4)))
    bn.weight.data.copy_(torch.from_numpy(synth.uniform(14, (C,), -1.5, 1.5)))
    bn.bias.data.copy_(torch.from_numpy(synth.uniform(15, (C,), -0.5, 0.5)))
    bn = bn.to(dev).eval()
    bn_r = copy.deepcopy(bn)
    bn_r.weight.data.mul_(0.5)
    xg = g(x, dev)
    if len(shape) == 4:
        xg = xg.contiguous(memory_format=torch.channels_last)
    alpha, beta = (v.cpu().numpy() for v in fold_batchnorm(bn))
    residual = res_np = res_aff = res_bn = None
    if res == "codes":
        r_in = synth.uniform(16, shape, 0, 2)
        r2 = g(r_in, dev)
        r2 = r2.permute(0, 2, 3, 1).reshape(-1, C) if len(shape) == 4 else r2
        rc, ry = ops.dorefa_codes(r2.contiguous(), k, want_f32=True)
        residual = packed.CodeActivation(rc, shape)
        res_np = ry.cpu().numpy().reshape((shape[0],) + tuple(shape[2:]) + (C,))
        res_np = np.moveaxis(res_np, -1, 1) if len(shape) == 4 else res_np
        np.testing.assert_array_equal(residual.float().cpu().numpy(), res_np)      # code plane round trip
    elif res is not None:
        res_np = synth.uniform(17, shape, -2, 2)
        residual = g(res_np, dev)
        if res == "f32bn":
            res_bn = bn_r
            res_aff = tuple(v.cpu().numpy() for v in fold_batchnorm(bn_r))
    mod = FusedBnDorefaQuant(bn, k, relu=relu)
    act = mod(xg, residual=residual, residual_bn=res_bn)
    want_q, want_y = oracle.affine_relu_dorefa_codes(x, alpha, beta, k, relu, res_np, res_aff)
    assert act.shape == tuple(shape) and act.codes.codes.shape[1] % 16 == 0
    got_q = act.codes.codes[:, :C].cpu().numpy().astype(np.float32)
    got_q = got_q.reshape((shape[0],) + tuple(shape[2:]) + (C,))
    got_q = np.moveaxis(got_q, -1, 1) if len(shape) == 4 else got_q
    fits = np.abs(want_q) <= 127
    np.testing.assert_array_equal(got_q[fits], want_q[fits])
    assert int(act.codes.overflow.item()) == int(not fits.all())
    assert int(act.codes.codes[:, C:].abs().sum()) == 0                                # pad bytes stay zero
    if fits.all():
        np.testing.assert_array_equal(act.float().cpu().numpy(), want_y)
    else:
        with pytest.raises(RuntimeError, match="exceeded int8"):
            act.check()
        with pytest.raises(RuntimeError, match="exceeded int8"):
            act.float(check=True)
        assert bool(torch.isnan(act.float()).all())          # device-side predication: no sync, NaN result


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,ksz,st,pd,k,res", [(32, 64, 3, 1, 1, 4, "codes"), (64, 40, 3, 2, 1, 4, "f32bn"),
                                                       (16, 200, 1, 2, 0, 3, None), (48, 130, 3, 1, 1, 2, "f32"),
                                                       (128, 256, 3, 1, 1, 4, "codes"), (64, 64, 3, 1, 1, 8, None)])
def test_conv_code_epilogue_equals_conv_then_fused_quantiser(dev, oracle, Cin, Cout, ksz, st, pd, k, res):
    """layers.FusedDorefaConvBnQuant (BatchNorm / residual / ReLU / quantiser in the conv kernel's epilogue) is
    bit-identical to DorefaConv2d -> FusedBnDorefaQuant (itself pinned to the oracle), and to the oracle's chain on
    the oracle's own conv when that conv is exact in fp32."""
    from pytorch_quantize_impls_amd.layers import DorefaConv2d, FusedBnDorefaQuant, FusedDorefaConvBnQuant, fold_batchnorm
    N, H, W = 3, 9, 8
    torch.manual_seed(7)
    conv = DorefaConv2d(Cin, Cout, ksz, stride=st, padding=pd, bias=True, bit_width=1).to(dev).eval()
    bn = torch.nn.BatchNorm2d(Cout).to(dev)
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 4); bn.weight.data.normal_(); bn.bias.data.normal_()
    bn.eval()
    bn_r = copy.deepcopy(bn)
    bn_r.weight.data.mul_(-0.7)
    x_codes, _ = ops.dorefa_codes(torch.rand((N * H * W, Cin), device=dev) * 1.5, 4, want_f32=False,
                                  ld_bytes=ops.code_ld_bytes(Cin, 16))
    act = packed.CodeActivation(x_codes, (N, Cin, H, W))
    Ho, Wo = ops.conv_out_hw(H, W, ksz, ksz, st, pd, 1)
    residual = res_bn = None
    if res == "codes":
        r_codes, _ = ops.dorefa_codes(torch.rand((N * Ho * Wo, Cout), device=dev) * 2, k, want_f32=False,
                                      ld_bytes=ops.code_ld_bytes(Cout, 16))
        residual = packed.CodeActivation(r_codes, (N, Cout, Ho, Wo))
    elif res is not None:
        residual = torch.randn((N, Cout, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)
        res_bn = bn_r if res == "f32bn" else None
    with torch.no_grad():
        fused = FusedDorefaConvBnQuant(conv, bn, k)(act, residual=residual, residual_bn=res_bn)
        two_step = FusedBnDorefaQuant(bn, k)(conv(act), residual=residual, residual_bn=res_bn)
    assert fused.shape == two_step.shape == (N, Cout, Ho, Wo)
    assert torch.equal(fused.codes.codes, two_step.codes.codes)
    assert int(fused.codes.overflow.item()) == int(two_step.codes.overflow.item())
    # the shared flag: both calls OR into the input activation's flag
    assert fused.codes.overflow.data_ptr() == x_codes.overflow.data_ptr()


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,ksz,st,pd,halo", [(32, 64, 3, 1, 1, (1, 1)), (64, 40, 3, 2, 1, (0, 0)), (96, 200, 5, 1, 2, (2, 1)),
                                                      (32, 7, 1, 1, 0, (1, 0)), (128, 384, 3, 1, 1, (1, 1))])
def test_nibble_plane_epilogue_equals_bits_then_expand(dev, Cin, Cout, ksz, st, pd, halo):
    """qt_conv2d_implicit_nib (threshold bits written as the next conv's fp4 nibble pixel plane, with a zero halo) ==
    qt_conv2d_implicit_bits followed by qt_bits_to_nib_pad; qt_pool_bits_nib == qt_pool_bits + qt_bits_to_nib_pad."""
    N, H, W = 3, 10, 9
    x = g(synth.pm1(31, (N, Cin, H, W)), dev).contiguous(memory_format=torch.channels_last)
    px = ops.pack_pixels_nib(x)
    wp = ops.pack_conv_weight_nib(g(synth.uniform(32, (Cout, Cin, ksz, ksz), -1, 1), dev), "ternary")
    b = g(synth.uniform(33, (Cout,), -2, 2), dev)
    alpha, beta = g(synth.uniform(34, (Cout,), -1, 1), dev), g(synth.uniform(35, (Cout,), -3, 3), dev)
    Ho, Wo = ops.conv_out_hw(H, W, ksz, ksz, st, pd, 1)
    bits = ops.conv2d_nib(px, (N, Cin, H, W), wp, (ksz, ksz), b, st, pd, 1, epi=(alpha, beta))
    want = ops.bits_to_nib_pad(bits, N, Ho, Wo, halo, ld=ops.pixel_ld_nib(Cout))
    junk = torch.full(tuple(want.words.shape), 0x55555555, dtype=torch.int32, device=dev)   # poison the block torch.empty reuses
    del junk
    got = ops.conv2d_nib(px, (N, Cin, H, W), wp, (ksz, ksz), b, st, pd, 1, epi=ops.NibEpilogue(alpha, beta, halo))
    assert isinstance(got, ops.NibPlanes) and got.rows == want.rows and got.K == Cout
    assert torch.equal(got.words, want.words)
    if min(Ho, Wo) >= 2:
        na = ops.neg_alpha_words(alpha)
        pooled, (Hq, Wq) = ops.pool_bits(bits, N, Ho, Wo, 2, 2, na)
        want_p = ops.bits_to_nib_pad(pooled, N, Hq, Wq, halo, ld=ops.pixel_ld_nib(Cout))
        junk = torch.full(tuple(want_p.words.shape), 0x55555555, dtype=torch.int32, device=dev)
        del junk
        got_p, hw = ops.pool_bits_nib(bits, N, Ho, Wo, 2, 2, na, halo)
        assert hw == (Hq, Wq) and torch.equal(got_p.words, want_p.words)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,Cout,H,W", [("binary", 64, 16, 12), ("ternary", 96, 10, 14), ("binary", 32, 6, 6)])
def test_first_layer_output_blocked_form_equals_direct_form(dev, monkeypatch, kind, Cout, H, W):
    """The 2x2 output-blocked form of a real-input 3x3 / stride-1 / padding-1 first layer (4x4 stride-2 conv with the
    four shifted kernels, depth-to-space in the nibble epilogue) computes the same products as the direct form: with
    inputs whose partial sums are exact in fp32 (multiples of 1/8) the operand handed to the next conv is identical
    bit for bit; with Gaussian inputs only exact threshold ties may differ."""
    from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d, FusedConvPoolBnSign, fused as fused_mod
    monkeypatch.setattr(ops, "FIRST_3X3", False)      # the forms compared here are the routes the one-pass kernel (round 6) replaced
    N, Cin = 3, 3
    conv = (BinConv2d if kind == "binary" else TerConv2d)(Cin, Cout, 3, padding=1).to(dev)
    conv.weight.data.copy_(g(synth.uniform(41, (Cout, Cin, 3, 3), -1.2, 1.2), dev))
    conv.bias.data.copy_(g(synth.uniform(42, (Cout,), -1, 1), dev))
    conv.binary_input = False
    conv.eval()
    bn = torch.nn.BatchNorm2d(Cout).to(dev).eval()
    bn.running_mean.copy_(g(synth.normal(43, (Cout,)), dev)); bn.running_var.copy_(g(synth.uniform(44, (Cout,), 0.5, 4), dev))
    bn.weight.data.copy_(g(synth.normal(45, (Cout,)), dev)); bn.bias.data.copy_(g(synth.normal(46, (Cout,)), dev))
    blk = FusedConvPoolBnSign(conv, bn)
    blk.out_nib_halo = (1, 1)
    x_exact = g(np.round(synth.normal(47, (N, Cin, H, W)) * 16) / 8, dev).contiguous(memory_format=torch.channels_last)
    x_gauss = g(synth.normal(48, (N, Cin, H, W)), dev).contiguous(memory_format=torch.channels_last)
    # third form: the direct 3x3 kernel on the padded pair / triple plane (what the fused stacks use by default)
    monkeypatch.setattr(fused_mod, "DIRECT_FIRST_LAYER", True)
    before = dict(_lib.call_counts)
    with torch.no_grad():
        direct = (blk(x_exact), blk(x_gauss))
    ran = sum(_lib.call_counts.get(k, 0) - before.get(k, 0) for k in ("qt_conv3x3_direct_nib", "qt_conv3x3_direct_pairs"))
    assert ran == 2          # (fp16 pair pixels by default since round 4: qt_conv3x3_direct_pairs; bf16 triples: _nib with elem = 2)
    monkeypatch.setattr(fused_mod, "DIRECT_FIRST_LAYER", False)
    outs = {}
    for flag in (True, False):
        monkeypatch.setattr(fused_mod, "D2S_FIRST_LAYER", flag)
        before = dict(_lib.call_counts)
        with torch.no_grad():
            outs[flag] = (blk(x_exact), blk(x_gauss))
        used = {k_: v - before.get(k_, 0) for k_, v in _lib.call_counts.items() if v - before.get(k_, 0)}
        assert used.get("qt_conv2d_implicit_nib") == 2, used
    a, b = outs[True], outs[False]
    assert a[0].shape == b[0].shape == (N, Cout, H, W) and a[0].halo == (1, 1)
    assert torch.equal(a[0].nib.words, b[0].nib.words)
    diff = (a[1].nib.words != b[1].nib.words).float().mean().item()
    assert diff < 1e-3, diff
    assert direct[0].halo == (1, 1) and torch.equal(direct[0].nib.words, b[0].nib.words)
    assert (direct[1].nib.words != b[1].nib.words).float().mean().item() < 1e-3
    # pooled first block (bit planes into the pooling kernel) through the direct kernel as well
    blk_p = FusedConvPoolBnSign(conv, bn, torch.nn.MaxPool2d(2, 2))
    with torch.no_grad():
        monkeypatch.setattr(fused_mod, "DIRECT_FIRST_LAYER", True)
        p1 = blk_p(x_exact)
        monkeypatch.setattr(fused_mod, "DIRECT_FIRST_LAYER", False)
        p0 = blk_p(x_exact)
    assert torch.equal(p1.planes.sign, p0.planes.sign)


@pytest.mark.gpu
@pytest.mark.parametrize("C,Cout,N,H,W,kind", [(64, 64, 3, 14, 17, "binary"), (64, 128, 2, 9, 30, "ternary"),
                                               (128, 128, 2, 12, 12, "binary"), (128, 40, 1, 5, 7, "ternary"),
                                               (64, 100, 5, 33, 6, "binary"), (40, 64, 2, 8, 8, "ternary"),
                                               (64, 64, 1, 1, 1, "binary"), (128, 96, 2, 1, 5, "ternary"),
                                               (64, 7, 3, 2, 1, "binary"), (64, 64, 40, 17, 19, "ternary"),
                                               # planes at least 256 wide / high (the lean epilogue's one-compare carries),
                                               # all four tile shapes of the fp4 kernel
                                               (64, 64, 1, 3, 300, "binary"), (64, 128, 1, 300, 5, "ternary"),
                                               (64, 64, 1, 257, 254, "binary"), (128, 64, 2, 2, 254, "ternary"),
                                               (128, 128, 1, 4, 260, "binary"), (64, 128, 3, 254, 3, "binary")])
def test_direct_conv3x3_equals_implicit_gemm(dev, monkeypatch, C, Cout, N, H, W, kind):
    """qt_conv3x3_direct_nib (input patch loaded once per tile, taps read from LDS) against the implicit-GEMM kernels:
    threshold bits and the next conv's nibble halo plane, bit for bit (incl. the halo the kernel writes itself)."""
    x = g(synth.pm1(51, (N, C, H, W)), dev).contiguous(memory_format=torch.channels_last)
    bits = ops.sign_pack(x.permute(0, 2, 3, 1).contiguous())[0]
    px = ops.bits_to_nib_pad(bits, N, H, W, (1, 1), ld=ops.pixel_ld_nib(C))
    wp = ops.pack_conv_weight_nib(g(synth.uniform(52, (Cout, C, 3, 3), -1, 1), dev), kind)
    b = g(synth.uniform(53, (Cout,), -3, 3), dev)
    alpha, beta = g(synth.uniform(54, (Cout,), -1, 1), dev), g(synth.uniform(55, (Cout,), -20, 20), dev)
    assert ops.direct_conv3x3_applicable(C, Cout, (3, 3), 1, 1, 1, (1, 1), ops.NibEpilogue(alpha, beta, (1, 1)))
    assert ops.direct_conv3x3_applicable(C, Cout, (3, 3), 1, 1, 1, (1, 1), (alpha, beta)) == (ops.pixel_ld_nib(C) == 8 or Cout == 128)
    want_bits = ops.conv2d_nib(px, (N, C, H + 2, W + 2), wp, (3, 3), b, 1, 0, 1, epi=(alpha, beta))
    want_nib = ops.conv2d_nib(px, (N, C, H + 2, W + 2), wp, (3, 3), b, 1, 0, 1, epi=ops.NibEpilogue(alpha, beta, (1, 1)))
    for shape in (want_bits.sign.shape, want_nib.words.shape):          # poison what torch.empty will hand out
        junk = torch.full(tuple(shape), 0x55555555, dtype=torch.int32, device=dev)
        del junk
    before = dict(_lib.call_counts)
    got_bits = ops.conv3x3_direct_nib(px, N, C, H, W, wp, b, (alpha, beta))
    got_nib = ops.conv3x3_direct_nib(px, N, C, H, W, wp, b, ops.NibEpilogue(alpha, beta, (1, 1)))
    assert _lib.call_counts["qt_conv3x3_direct_nib"] - before.get("qt_conv3x3_direct_nib", 0) == 2
    assert torch.equal(got_bits.sign, want_bits.sign)
    assert torch.equal(got_nib.words, want_nib.words)


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,ksz,st,pd,kind", [(32, 64, 3, 1, 1, "binary"), (96, 200, 5, 1, 2, "ternary"),
                                                      (64, 40, 3, 2, 0, "binary"), (256, 384, 3, 1, 1, "ternary")])
def test_integer_threshold_epilogue_equals_float_epilogue(dev, Cin, Cout, ksz, st, pd, kind):
    """The one-compare epilogue (per-channel integer thresholds from ops.integer_thresholds) produces the bits of the
    float epilogue for every sign / magnitude of the folded BatchNorm, as bit planes and as nibble planes."""
    N, H, W = 3, 11, 9
    x = g(synth.pm1(61, (N, Cin, H, W)), dev).contiguous(memory_format=torch.channels_last)
    px = ops.pack_pixels_nib(x)
    wp = ops.pack_conv_weight_nib(g(synth.uniform(62, (Cout, Cin, ksz, ksz), -1, 1), dev), kind)
    b = g(synth.uniform(63, (Cout,), -4, 4), dev)
    alpha = g(synth.uniform(64, (Cout,), -0.2, 0.2), dev)
    beta = g(synth.uniform(65, (Cout,), -8, 8), dev)
    alpha[0] = 0.0
    alpha[1], beta[1] = 0.0, -1.0
    beta[2] = 1e6
    beta[3] = -1e6
    thr = ops.integer_thresholds(b, alpha, beta, Cin * ksz * ksz)
    args = (px, (N, Cin, H, W), wp, (ksz, ksz), b, st, pd, 1)
    want = ops.conv2d_nib(*args, epi=(alpha, beta))
    got = ops.conv2d_nib(*args, epi=(alpha, beta, thr))
    assert torch.equal(got.sign, want.sign)
    assert 0.02 < float((want.sign != 0).float().mean())            # not a degenerate all-zero comparison
    want_n = ops.conv2d_nib(*args, epi=ops.NibEpilogue(alpha, beta, (1, 1)))
    got_n = ops.conv2d_nib(*args, epi=ops.NibEpilogue(alpha, beta, (1, 1), thr=thr))
    assert torch.equal(got_n.words, want_n.words)


@pytest.mark.gpu
@pytest.mark.parametrize("Cout,N,H,W,k,relu,res", [(64, 3, 12, 9, 4, True, True), (64, 2, 32, 32, 4, True, False),
                                                   (40, 4, 7, 5, 3, "pre", True), (64, 1, 1, 1, 4, False, True)])
def test_direct_code_conv_equals_implicit_code_epilogue(dev, monkeypatch, Cout, N, H, W, k, relu, res):
    """qt_conv3x3_direct_codes (direct 3x3 kernel, int8 code planes, DoReFa code epilogue) == qt_conv2d_implicit_codes on
    the same halo planes, byte for byte including the halo it writes and the shared overflow flag."""
    from pytorch_quantize_impls_amd.layers import DorefaConv2d, FusedDorefaConvBnQuant
    C = 64
    torch.manual_seed(71)
    conv = DorefaConv2d(C, Cout, 3, padding=1, bias=True, bit_width=1).to(dev).eval()
    bn = torch.nn.BatchNorm2d(Cout).to(dev)
    bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 3); bn.weight.data.uniform_(-0.6, 0.6); bn.bias.data.uniform_(-0.2, 0.3)
    bn.eval()

    def halo_act(x2, Cn, kk):
        cp, _ = ops.dorefa_codes(x2, kk, want_f32=False, ld_bytes=ops.code_ld_bytes(Cn, 16))
        q = ops.pad_pixel_plane(cp.codes, N, H, W, (1, 1))
        cp = ops.CodePlanes(codes=q, rows=int(q.shape[0]), K=Cn, inv_n=cp.inv_n, bit_width=kk, overflow=cp.overflow)
        return packed.CodeActivation(cp, (N, Cn, H, W), halo=(1, 1))

    act = halo_act(torch.rand((N * H * W, C), device=dev) * 1.2, C, 4)
    residual = halo_act(torch.rand((N * H * W, Cout), device=dev), Cout, k) if res else None
    blk = FusedDorefaConvBnQuant(conv, bn, k, relu=relu, out_halo=1)
    outs = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "DIRECT_CONV3X3_CODES", flag)
        junk = torch.full((N * (H + 2) * (W + 2), ops.code_ld_bytes(Cout, 16)), 85, dtype=torch.int8, device=dev)
        del junk
        before = dict(_lib.call_counts)
        with torch.no_grad():
            outs[flag] = blk(act, residual=residual)
        used = {k_: v - before.get(k_, 0) for k_, v in _lib.call_counts.items() if v - before.get(k_, 0)}
        assert ("qt_conv3x3_direct_codes" in used) == flag and ("qt_conv2d_implicit_codes" in used) == (not flag), used
    a, b = outs[True], outs[False]
    assert a.halo == b.halo == (1, 1) and a.shape == b.shape
    assert torch.equal(a.codes.codes, b.codes.codes)
    assert int(a.codes.overflow.item()) == int(b.codes.overflow.item())


@pytest.mark.gpu
@pytest.mark.parametrize("halo", [(1, 1), (2, 0), (0, 3)])
def test_zero_halo_touches_only_the_border(dev, halo):
    N, H, W, C = 3, 5, 7, 48
    hy, hx = halo
    plane = torch.randint(1, 100, (N * (H + 2 * hy) * (W + 2 * hx), C), dtype=torch.int8, device=dev)
    before = plane.clone().view(N, H + 2 * hy, W + 2 * hx, C)
    ops.zero_halo(plane, N, H, W, halo)
    after = plane.view(N, H + 2 * hy, W + 2 * hx, C)
    assert torch.equal(after[:, hy:hy + H, hx:hx + W], before[:, hy:hy + H, hx:hx + W])
    assert int(after.sum()) == int(before[:, hy:hy + H, hx:hx + W].sum())


@pytest.mark.gpu
def test_fused_dorefa_pre_relu_and_code_pool_vs_oracle(dev, oracle):
    """The two other module orders of the reference's DoReFa examples: ReLU in front of the BatchNorm
    (models/FullNet/DorefaMNIST.py:46-48) and MaxPool2d after the quantiser (models/samples/AlexNet_Dorefa.py:38-41),
    the latter as a max over the int8 codes."""
    from pytorch_quantize_impls_amd.layers import (CodeMaxPool, DorefaConv2d, FusedBnDorefaQuant, FusedDorefaConvBnQuant,
                                                   fold_batchnorm)
    N, C, H, W, k = 3, 40, 9, 11, 3
    x = synth.uniform(21, (N, C, H, W), -2, 2)
    bn = torch.nn.BatchNorm2d(C).to(dev)
    bn.running_mean.copy_(g(synth.uniform(22, (C,), -0.5, 0.5), dev))
    bn.running_var.copy_(g(synth.uniform(23, (C,), 0.5, 4), dev))
    bn.weight.data.copy_(g(synth.uniform(24, (C,), -1.5, 1.5), dev))
    bn.bias.data.copy_(g(synth.uniform(25, (C,), -0.5, 0.5), dev))
    bn.eval()
    alpha, beta = (n(v) for v in fold_batchnorm(bn))
    xg = g(x, dev).contiguous(memory_format=torch.channels_last)
    act = FusedBnDorefaQuant(bn, k, relu="pre")(xg)
    want_q, want_y = oracle.affine_relu_dorefa_codes(x, alpha, beta, k, "pre")
    assert (np.abs(want_q) <= 127).all()
    np.testing.assert_array_equal(n(act.float()), want_y)
    # pre-ReLU in the conv epilogue == conv -> one-pass form
    conv = DorefaConv2d(C, 24, 3, padding=1, bias=False, bit_width=1).to(dev).eval()
    with torch.no_grad():
        e1 = FusedDorefaConvBnQuant(conv, torch.nn.BatchNorm2d(24).to(dev).eval(), k, relu="pre")(act)
        e2 = FusedBnDorefaQuant(torch.nn.BatchNorm2d(24).to(dev).eval(), k, relu="pre")(conv(act))
    assert torch.equal(e1.codes.codes, e2.codes.codes)
    # MaxPool on the codes == MaxPool of the fp32 image (oracle.maxpool2d), with and without an output halo
    for (pk, ps, halo) in ((2, 2, (0, 0)), (3, 2, (1, 1)), (2, 1, (2, 1))):
        pooled = CodeMaxPool(torch.nn.MaxPool2d(pk, ps), out_halo=halo)(act)
        want = oracle.maxpool2d(want_y, pk, ps)
        assert pooled.halo == halo and pooled.shape == want.shape
        np.testing.assert_array_equal(n(pooled.float()), want)
        full = pooled.codes.codes.view(N, want.shape[2] + 2 * halo[0], want.shape[3] + 2 * halo[1], -1)
        assert int(full.abs().sum()) == int(pooled.without_halo().codes.codes.abs().sum())      # zero border
    with pytest.raises(ValueError, match="relu must be"):
        FusedBnDorefaQuant(bn, k, relu="both")


@pytest.mark.gpu
@pytest.mark.parametrize("C1", [64, 40])
def test_fused_dorefa_cnn_chain_matches_module_graph(dev, C1):
    """A small CNN in the module order of models/samples/AlexNet_Dorefa.py (conv, BatchNorm, ReLU, quant, MaxPool ...
    flatten, LinearDorefa, BatchNorm1d, ReLU, quant, LinearDorefa) run (a) module by module and (b) as code planes end
    to end: code-epilogue conv, pool on codes, flatten_hwc into the int8 GEMM, one-pass quantiser after the FC."""
    from pytorch_quantize_impls_amd.functions import nnDorefaQuant
    from pytorch_quantize_impls_amd.layers import (CodeMaxPool, DorefaConv2d, FusedBnDorefaQuant, FusedDorefaConvBnQuant,
                                                   LinearDorefa, permute_fc_weight_hwc)
    torch.manual_seed(5)
    k, N = 4, 8
    conv0 = DorefaConv2d(3, 32, 3, padding=1, bias=False, bit_width=1)
    conv1 = DorefaConv2d(32, C1, 3, padding=1, bias=True, bit_width=1)
    lin2, lin3 = LinearDorefa(C1 * 4 * 4, 128, bit_width=1), LinearDorefa(128, 10, bit_width=1)
    bn0, bn1, bn2 = torch.nn.BatchNorm2d(32), torch.nn.BatchNorm2d(C1), torch.nn.BatchNorm1d(128)
    for i, b in enumerate((bn0, bn1, bn2)):
        b.running_mean.copy_(torch.from_numpy(synth.normal(30 + i, (b.num_features,))) * 0.1)
        b.running_var.copy_(torch.from_numpy(synth.uniform(40 + i, (b.num_features,), 0.5, 2)))
        b.weight.data.copy_(torch.from_numpy(synth.uniform(50 + i, (b.num_features,), 0.3, 0.8)))
        b.bias.data.copy_(torch.from_numpy(synth.uniform(60 + i, (b.num_features,), 0.0, 0.4)))
    mods = torch.nn.ModuleList([conv0, conv1, lin2, lin3, bn0, bn1, bn2]).to(dev).eval()
    quant, pool = nnDorefaQuant(k), torch.nn.MaxPool2d(2)
    x = g(synth.normal(70, (N, 3, 16, 16)), dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        h = pool(quant(torch.relu(bn0(conv0(x)))))
        h = pool(quant(torch.relu(bn1(conv1(h)))))
        h = quant(torch.relu(bn2(lin2(h.reshape(N, -1)))))
        want = lin3(h)
        # fused: the FC weight takes the (h, w, c) column order of flatten_hwc
        lin2f = LinearDorefa(C1 * 4 * 4, 128, bit_width=1).to(dev)
        lin2f.weight.data.copy_(permute_fc_weight_hwc(lin2.weight.org if hasattr(lin2.weight, "org") else lin2.weight, C1, 4, 4))
        lin2f.bias.data.copy_(lin2.bias)
        lin2f.eval()
        before = dict(_lib.call_counts)
        # device fold: BatchNorm evaluated in this device's own arithmetic (layers.fused.device_bn_fold) -> the module graph's codes
        a = CodeMaxPool(pool, out_halo=1)(FusedBnDorefaQuant(bn0, k, fold="device")(conv0(x)))
        a = CodeMaxPool(pool)(FusedDorefaConvBnQuant(conv1, bn1, k, fold="device")(a))
        flat = a.flatten_hwc()
        assert torch.equal(flat.float(), a.float().permute(0, 2, 3, 1).reshape(N, -1))
        a2 = FusedBnDorefaQuant(bn2, k)(lin2f(flat))
        got = lin3(a2)
        used = {k_: v - before.get(k_, 0) for k_, v in _lib.call_counts.items() if v - before.get(k_, 0)}
    assert used.get("qt_conv2d_implicit_codes") == 1 and used.get("qt_pool_codes_i8") == 2 and used.get("qt_i8_gemm") == 2, used
    assert isinstance(got, torch.Tensor) and got.shape == want.shape
    # the 4-D chain: identical codes (device fold).  The BatchNorm1d block behind the FC is folded the reference way (the device
    # arithmetic is probed for 4-D activations only): its codes may differ from the module graph's at rounding boundaries only —
    # counted, each within 1e-5 of a half-integer of n v — and with equal codes the logits agree to the float tail
    nlev = float((1 << k) - 1)
    with torch.no_grad():
        h4 = pool(quant(torch.relu(bn1(conv1(pool(quant(torch.relu(bn0(conv0(x))))))))))
        assert torch.equal(a.float(), h4)
        v2 = torch.relu(bn2(lin2(h4.reshape(N, -1))))
        c_mod, c_fus = torch.round(v2 * nlev), torch.round(a2.float() * nlev)
        flipped = c_mod != c_fus
        assert int(flipped.sum()) <= 2
        if flipped.any():
            u = v2[flipped] * nlev
            assert float(((u - torch.floor(u)) - 0.5).abs().max()) <= 1e-5 * max(1.0, float((v2 * nlev).abs().mean()))
        else:
            assert norm_err(n(got), n(want)) <= TOL
        assert norm_err(n(lin3(a2.float())), n(got)) <= TOL          # the same codes through the module's own forward


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,ksz,st,pd,in_halo,out_halo,res_halo", [
    (64, 64, 3, 1, 1, (1, 1), (1, 1), (1, 1)), (32, 100, 3, 2, 1, (1, 1), (1, 1), None),
    (16, 48, 1, 2, 0, (1, 1), (0, 0), (2, 1)), (64, 130, 3, 1, (1, 0), (2, 1), (1, 2), (0, 0)),
    (128, 64, 5, 1, 2, (2, 2), (3, 0), None), (32, 32, 3, 1, 2, (1, 1), (1, 1), (1, 1))])
def test_halo_planes_equal_plain_planes(dev, Cin, Cout, ksz, st, pd, in_halo, out_halo, res_halo):
    """Code planes with a physical zero border (CodeActivation.halo): a conv reading one (fp32 output and code
    epilogue), the epilogue writing one and a residual held in one give exactly the plain-plane results; a padding
    larger than the halo falls back to the stripped plane."""
    from pytorch_quantize_impls_amd.layers import DorefaConv2d, FusedDorefaConvBnQuant
    N, H, W = 3, 10, 7
    torch.manual_seed(11)
    conv = DorefaConv2d(Cin, Cout, ksz, stride=st, padding=pd, bias=True, bit_width=1).to(dev).eval()
    bn = torch.nn.BatchNorm2d(Cout).to(dev)
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 4); bn.weight.data.normal_(); bn.bias.data.normal_()
    bn.eval()
    x_codes, _ = ops.dorefa_codes(torch.rand((N * H * W, Cin), device=dev), 4, want_f32=False,
                                  ld_bytes=ops.code_ld_bytes(Cin, 16))
    plain = packed.CodeActivation(x_codes, (N, Cin, H, W))

    def with_halo(act, halo):
        Nn, C, Hh, Ww = act.shape
        q = ops.pad_pixel_plane(act.codes.codes, Nn, Hh, Ww, halo)
        cp = ops.CodePlanes(codes=q, rows=q.shape[0], K=act.codes.K, inv_n=act.codes.inv_n,
                            bit_width=act.codes.bit_width, overflow=act.codes.overflow)
        return packed.CodeActivation(cp, act.shape, halo=halo)

    haloed = with_halo(plain, in_halo)
    assert torch.equal(haloed.without_halo().codes.codes, plain.codes.codes)
    assert torch.equal(haloed.float(), plain.float())
    Ho, Wo = ops.conv_out_hw(H, W, ksz, ksz, st, pd, 1)
    res_plain = res_h = None
    if res_halo is not None:
        r_codes, _ = ops.dorefa_codes(torch.rand((N * Ho * Wo, Cout), device=dev) * 2, 4, want_f32=False,
                                      ld_bytes=ops.code_ld_bytes(Cout, 16))
        res_plain = packed.CodeActivation(r_codes, (N, Cout, Ho, Wo))
        res_h = with_halo(res_plain, res_halo) if any(res_halo) else res_plain
    with torch.no_grad():
        before = dict(_lib.call_counts)
        y_h = lazy.resolve(conv(haloed))
        used = {k_: v - before.get(k_, 0) for k_, v in _lib.call_counts.items() if v - before.get(k_, 0)}
        assert torch.equal(y_h, conv(plain))
        fits = all(p_ <= h_ for p_, h_ in zip((pd, pd) if isinstance(pd, int) else pd, in_halo))
        assert ("qt_conv2d_implicit_halo" in used) == fits, used
        want = FusedDorefaConvBnQuant(conv, bn, 4)(plain, residual=res_plain)
        # the output plane comes from torch.empty: poison the block the caching allocator will hand out, so a border
        # byte the kernel failed to write shows up
        junk = torch.full((N * (Ho + 2 * out_halo[0]) * (Wo + 2 * out_halo[1]), ops.code_ld_bytes(Cout, 16)), 85,
                          dtype=torch.int8, device=dev)
        del junk
        got = FusedDorefaConvBnQuant(conv, bn, 4, out_halo=out_halo)(haloed, residual=res_h)
    assert got.halo == tuple(out_halo) and got.shape == want.shape
    assert torch.equal(got.without_halo().codes.codes, want.codes.codes)
    if any(out_halo):       # the border really is zero
        full = got.codes.codes.view(N, Ho + 2 * out_halo[0], Wo + 2 * out_halo[1], -1)
        assert int(full.abs().sum()) == int(want.codes.codes.abs().sum())


@pytest.mark.gpu
def test_fuzz_conv_code_epilogue_vs_oracle(dev, oracle):
    """Random DorefaConv2d geometries through the code epilogue: against the two-step HIP path (bit-identical) and
    against the oracle's conv + chain on the same codes.  The oracle's fp32 conv sums in another order than the
    int32 accumulate, so an output whose n*t lands within float rounding of a .5 tie may differ by one code: such
    positions are identified from the oracle's own pre-rounding value and excluded (they are < 1e-3 of the outputs)."""
    from pytorch_quantize_impls_amd.layers import DorefaConv2d, FusedBnDorefaQuant, FusedDorefaConvBnQuant, fold_batchnorm
    rng = np.random.default_rng(20260929)
    checked = skipped = 0
    for it in range(30):
        Cin = int(rng.choice([4, 16, 24, 32, 64, 100]))
        Cout = int(rng.choice([1, 7, 30, 32, 64, 100, 130, 192, 260]))
        kh, kw = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        st = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        pd = (int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        dl = (int(rng.integers(1, 3)), 1)
        N = int(rng.integers(1, 4))
        H = int(rng.integers(dl[0] * (kh - 1) + 1, 12)) + 2
        W = int(rng.integers(kw, 12)) + 2
        k = int(rng.choice([2, 3, 4, 5]))
        relu = bool(it % 5)
        res_kind = ("codes", "f32", "f32bn", None)[it % 4]
        cfg = (it, Cin, Cout, kh, kw, st, pd, dl, N, H, W, k, relu, res_kind)
        conv = DorefaConv2d(Cin, Cout, (kh, kw), stride=st, padding=pd, dilation=dl, bias=bool(it % 3), bit_width=1).to(dev)
        conv.weight.data.copy_(g(synth.uniform(100 + it, (Cout, Cin, kh, kw), -1, 1), dev))
        conv.eval()
        bn = torch.nn.BatchNorm2d(Cout).to(dev).eval()
        bn.running_mean.copy_(g(synth.normal(200 + it, (Cout,)) * 0.2, dev))
        bn.running_var.copy_(g(synth.uniform(300 + it, (Cout,), 0.5, 4), dev))
        bn.weight.data.copy_(g(synth.normal(400 + it, (Cout,)) * 0.3, dev))
        bn.bias.data.copy_(g(synth.normal(500 + it, (Cout,)) * 0.3, dev))
        bn_r = copy.deepcopy(bn)
        bn_r.weight.data.mul_(0.5)
        xin = synth.uniform(600 + it, (N, H, W, Cin), 0, 1.2)
        x_codes, x_img = ops.dorefa_codes(g(xin.reshape(-1, Cin), dev), 4, want_f32=True, ld_bytes=ops.code_ld_bytes(Cin, 16))
        act = packed.CodeActivation(x_codes, (N, Cin, H, W))
        Ho, Wo = ops.conv_out_hw(H, W, kh, kw, st, pd, dl)
        residual = res_bn = res_np = res_aff = None
        if res_kind == "codes":
            r_codes, r_img = ops.dorefa_codes(g(synth.uniform(700 + it, (N * Ho * Wo, Cout), 0, 1), dev), k, want_f32=True,
                                              ld_bytes=ops.code_ld_bytes(Cout, 16))
            residual = packed.CodeActivation(r_codes, (N, Cout, Ho, Wo))
            res_np = np.moveaxis(n(r_img).reshape(N, Ho, Wo, Cout), -1, 1)
        elif res_kind is not None:
            res_np = synth.uniform(800 + it, (N, Cout, Ho, Wo), -1, 1)
            residual = g(res_np, dev).contiguous(memory_format=torch.channels_last)
            if res_kind == "f32bn":
                res_bn, res_aff = bn_r, tuple(n(v) for v in fold_batchnorm(bn_r))
        with torch.no_grad():
            fused = FusedDorefaConvBnQuant(conv, bn, k, relu=relu)(act, residual=residual, residual_bn=res_bn)
            two = FusedBnDorefaQuant(bn, k, relu=relu)(conv(act), residual=residual, residual_bn=res_bn)
        assert torch.equal(fused.codes.codes, two.codes.codes), cfg
        # oracle: conv of the fp32 activation image with the eval weight (sign(W) * E), then the chain
        x_nchw = np.moveaxis(n(x_img).reshape(N, H, W, Cin), -1, 1)
        yc = oracle.conv2d(x_nchw, n(conv.weight), None if conv.bias is None else n(conv.bias), st, pd, dl)
        alpha, beta = (n(v) for v in fold_batchnorm(bn))
        want_q, _ = oracle.affine_relu_dorefa_codes(yc, alpha, beta, k, relu, res_np, res_aff)
        got_q = np.moveaxis(n(fused.codes.codes[:, :Cout]).astype(np.float32).reshape(N, Ho, Wo, Cout), -1, 1)
        fits = np.abs(want_q) <= 126                      # beyond int8 the kernel writes 0 and raises the flag
        if not (np.abs(want_q) <= 128).all():
            assert int(fused.codes.overflow.item()) == 1, cfg
        diff = (got_q != want_q) & fits
        assert np.abs((got_q - want_q)[fits]).max(initial=0) <= 1, cfg
        assert diff.mean() <= 2e-3, (cfg, diff.mean())
        checked += diff.size
        skipped += int(diff.sum())
    assert skipped <= 1e-3 * checked, (skipped, checked)


@pytest.mark.gpu
def test_fused_dorefa_resnet_matches_module_graph(dev):
    """C4 in its fused inference form (code planes between the DorefaConv2d layers) against the module-by-module
    eval graph: BatchNorm folding re-associates fp32 rounding, so a code may flip at a rounding boundary — the
    logits agree to the float tail and the codes of the first block agree except at such ties."""
    import bench_models
    torch.manual_seed(0)
    m = bench_models.DorefaResNet18()
    bench_models.randomize_bn(m, seed=3)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_var.mul_(4.0)
    m = m.to(dev).to(memory_format=torch.channels_last).eval()
    f = bench_models.FusedDorefaResNet18(m, fuse_conv=False)
    fc = bench_models.FusedDorefaResNet18(m, fuse_conv=True)
    x = torch.randn((16, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        before = dict(_lib.call_counts)
        got = f(x)
        used = {k_: v - before.get(k_, 0) for k_, v in _lib.call_counts.items() if v - before.get(k_, 0)}
        before = dict(_lib.call_counts)
        got_c = fc(x)
        used_c = {k_: v - before.get(k_, 0) for k_, v in _lib.call_counts.items() if v - before.get(k_, 0)}
        want = m(x)
        a0 = f.q0(f.stem(x))
        ref0 = m.quant(torch.relu(m.bn(m.stem(x))))
    assert used.get("qt_affine_dorefa_codes_i8") == 17 and used.get("qt_conv2d_implicit", 0) >= 19, used
    assert "qt_dorefa_codes_i8" not in used
    # conv-epilogue form: 16 code-epilogue convs, 3 fp32 shortcut convs, only the stem quantiser as its own pass
    assert used_c.get("qt_conv2d_implicit_codes") == 16 and used_c.get("qt_conv2d_implicit_halo") == 3, used_c
    assert used_c.get("qt_affine_dorefa_codes_halo_i8") == 1 and "qt_pad_pixel_plane" not in used_c     # ... straight into the halo plane
    assert used_c.get("qt_codes_to_f32") == 1                                                            # the head: image + avg_pool2d, one pass
    assert torch.equal(got_c, got)
    with torch.no_grad():
        assert torch.equal(bench_models.FusedDorefaResNet18(m, fuse_conv=True, halo=0)(x), got)
    # "reference"-fold codes of the first block vs the module graph's: they may differ at rounding boundaries only — counted, each
    # exactly one level apart with n v within 1e-5 of a half-integer
    with torch.no_grad():
        v0 = torch.relu(m.bn(m.stem(x)))
    diff0 = a0.float() != ref0
    assert float(diff0.float().mean()) < 1e-4, float(diff0.float().mean())
    if diff0.any():
        u = v0[diff0] * 15.0
        assert float(((a0.float() - ref0)[diff0].abs() * 15.0 - 1.0).abs().max()) <= 1e-3
        assert float(((u - torch.floor(u)) - 0.5).abs().max()) <= 1e-5 * max(1.0, float((v0 * 15.0).abs().mean()))
    # with the device fold the fused forms ARE the module graph: identical logits
    with torch.no_grad():
        got_d = bench_models.FusedDorefaResNet18(m, fuse_conv=True, fold="device")(x)
        got_d0 = bench_models.FusedDorefaResNet18(m, fuse_conv=False, fold="device")(x)
    assert torch.equal(got_d, want) and torch.equal(got_d0, want)


# ---- Lin / Log fixed-point family ---------------------------------------------------------------

@pytest.fixture(scope="module")
def g9():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_loglin_v1.npz"))


def _same_nan(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and \
        np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)])


@pytest.mark.gpu
def test_lin_log_kernels_vs_reference_vectors(dev, oracle, g9):
    from pytorch_quantize_impls_amd.functions import LinQuant, LogQuant
    for fsr, bits in g9["g9_cfgs"].tolist():
        for sign in (1, 0):
            for nm in ("edge", "rand"):
                x = g(g9[f"g9_{nm}"], dev)
                tag = f"f{fsr}_b{bits}_s{sign}_{nm}"
                with used("qt_lin_quantize_f32", "qt_log_quantize_f32"):
                    yl = LinQuant(fsr, bits, bool(sign)).apply(x)
                    yg = LogQuant(fsr, bits, bool(sign)).apply(x)
                assert _same_nan(n(yl), g9[f"g9_lin_{tag}"]), ("lin", tag)
                assert _same_nan(n(yg), g9[f"g9_log_{tag}"]), ("log", tag)
        gr = g(g9["g9_rand"], dev)
        xin = torch.ones_like(gr, requires_grad=True)
        LogQuant(fsr, bits, True, lin_back=False).apply(xin).backward(gr)
        assert _same_nan(n(xin.grad), g9[f"g9_logbwd_f{fsr}_b{bits}_rand"])
        xin = torch.ones_like(gr, requires_grad=True)
        LinQuant(fsr, bits, True, lin_back=False).apply(xin).backward(gr)
        assert _same_nan(n(xin.grad), oracle.lin_quantize(g9["g9_rand"], fsr, bits, 2))
    # odd length / misaligned views exercise the scalar head and tail of the elementwise kernel
    x = g(synth.normal(3, (1003,)) * 9, dev)
    assert _same_nan(n(ops.log_quantize(x[1:], 3, 2)), oracle.log_quantize(n(x)[1:], 3, 2))
    assert _same_nan(n(ops.lin_quantize(x[1:], 3, 2, 1)), oracle.lin_quantize(n(x)[1:], 3, 2, 1))


@pytest.mark.gpu
def test_lin_log_layers_vs_reference_vectors(dev, g9):
    """LinearQuant / QuantConv2d: device forward (exact-bf16 weight levels x bf16-triple activations on the matrix
    cores in eval / no-grad mode, dense library under autograd) against the reference's outputs and gradients."""
    from pytorch_quantize_impls_amd.layers import LinearQuant, QuantConv2d
    for name in g9["g9_layer_cases"].tolist():
        p = name.split("_")
        dtype, fsr, bits = p[1], int(p[-2][1:]), int(p[-1][1:])
        w, b, x = (g(g9[f"g9_{name}_{k}"], dev) for k in ("w", "b", "x"))
        if name.startswith("lin_"):
            layer = LinearQuant(w.shape[1], w.shape[0], True, dtype=dtype, fsr=fsr, bit_width=bits).to(dev)
        else:
            layer = QuantConv2d(w.shape[1], w.shape[0], w.shape[2], stride=int(p[5][1:]), padding=int(p[6][1:]),
                                bias=True, fsr=fsr, bit_width=bits, dtype=dtype).to(dev)
        layer.weight.data.copy_(w); layer.bias.data.copy_(b)
        xi = x.clone().requires_grad_(True)
        y = layer(xi)
        y.backward(g(g9[f"g9_{name}_gout"], dev))
        assert norm_err(n(y), g9[f"g9_{name}_y"]) <= TOL, name
        assert norm_err(n(xi.grad), g9[f"g9_{name}_gx"]) <= TOL and norm_err(n(layer.weight.grad), g9[f"g9_{name}_gw"]) <= TOL
        with torch.no_grad(), used("qt_bf16x3_pack_f32", "qt_bf16_gemm" if name.startswith("lin_") else "qt_conv2d_implicit"):
            y_ng = layer(x)                                        # training-mode weights, matrix-core route
        assert norm_err(n(y_ng), g9[f"g9_{name}_y"]) <= TOL, name
        layer.train(False)
        with torch.no_grad(), used("qt_bf16_gemm" if name.startswith("lin_") else "qt_conv2d_implicit"):
            ye = layer(x)
        assert norm_err(n(ye), g9[f"g9_{name}_y_eval"]) <= TOL, name


@pytest.mark.gpu
def test_ap2_kernel_vs_reference_vector(dev, g9):
    from pytorch_quantize_impls_amd.functions import AP2
    with used("qt_ap2_f32"):
        y = AP2(g(g9["g9_ap2_in"], dev))
    assert _same_nan(n(y), g9["g9_ap2_out"])


@pytest.mark.gpu
def test_real_x_real_six_term_route_vs_fp64(dev):
    """REAL activations x REAL weights through six-term bf16 planes (XNORConv2d's contraction): fp32-GEMM accuracy."""
    torch.manual_seed(12)
    for (M, N, K) in [(5, 7, 31), (300, 130, 777), (1024, 512, 2048)]:
        x = torch.randn((M, K), device=dev) * 3
        w = torch.randn((N, K), device=dev) * 0.2
        b = torch.randn((N,), device=dev)
        with used("qt_bf16x6_pack_f32", "qt_bf16_gemm"):
            y = ops.real_linear(x, w, b)
        ref = x.double() @ w.double().t() + b.double()
        assert norm_err(n(y), n(ref)) <= TOL, (M, N, K)
    for (Cin, Cout, k, st, pd, H) in [(3, 16, 3, 1, 1, 12), (32, 48, 5, 2, 2, 15), (64, 192, 3, 1, 1, 13)]:
        x = torch.randn((2, Cin, H, H), device=dev)
        w = torch.randn((Cout, Cin, k, k), device=dev) * 0.3
        b = torch.randn((Cout,), device=dev)
        with used("qt_bf16x6_pack_f32", "qt_conv2d_implicit"):
            y2 = ops.real_conv2d(x, w, b, st, pd, 1)
        Ho = (H + 2 * pd - k) // st + 1
        ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), st, pd)
        assert norm_err(n(y2.view(2, Ho, Ho, Cout).permute(0, 3, 1, 2)), n(ref)) <= TOL


@pytest.mark.gpu
def test_xnor_conv2d_layer_runs_on_the_matrix_cores(dev, golden):
    name = golden["g4_conv_cases"].tolist()[0]
    p = name.split("_")
    Cin, Cout, k, st, pd = int(p[0][1:]), int(p[1][1:]), int(p[2][1:]), int(p[3][1:]), int(p[4][1:])
    has_b = p[7] == "bias"
    layer = XNORConv2d(Cin, Cout, k, stride=st, padding=pd, bias=has_b).to(dev)
    layer.weight.data.copy_(g(golden[f"g4_conv_{name}_w"], dev))
    if has_b:
        layer.bias.data.copy_(g(golden[f"g4_conv_{name}_b"], dev))
    with used("qt_xnor_weight_f32", "qt_bf16x6_pack_f32", "qt_conv2d_implicit"):
        y = layer(g(golden[f"g4_conv_{name}_x"], dev))
    assert norm_err(n(y), golden[f"g4_conv_{name}_xnor_y"]) <= TOL


@pytest.mark.gpu
def test_fused_relu_bn_sign_mlp_pattern(dev):
    """Linear -> ReLU -> BatchNorm1d -> BinaryConnect (benchmark/BinaryNet/MLPBin.py:42-44) fused: fuse_sequential
    recognises the ReLU in front of the BatchNorm; planes equal the un-fused modules' sign bits."""
    from pytorch_quantize_impls_amd.layers import fuse_sequential, FusedPoolBnSign
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    torch.manual_seed(6)
    seq = torch.nn.Sequential(LinearBin(200, 96), torch.nn.ReLU(), torch.nn.BatchNorm1d(96), BinaryConnect(),
                              LinearBin(96, 10)).to(dev).eval()
    bn = seq[2]
    bn.running_mean.normal_(2, 3); bn.running_var.uniform_(0.5, 20); bn.weight.data.normal_(); bn.bias.data.normal_()
    fused = fuse_sequential(seq)
    assert isinstance(fused[1], FusedPoolBnSign) and fused[1].pre_relu and len(fused) == 3
    x = torch.randn((64, 200), device=dev).sign()
    with torch.no_grad(), used("qt_pool_affine_sign_pack_nib_nhwc"):         # (batch 64: the pass also writes the next GEMM's nibble rows)
        yf = fused(x)
        h = seq[0](x)
        act = fused[1](h)
        alpha, beta = (1.0 / torch.sqrt(bn.running_var + bn.eps)) * bn.weight, None
        beta = bn.bias - bn.running_mean * alpha
        want = ops.sign_pack(ops.binarize(torch.relu(h) * alpha + beta))[0]
    assert torch.equal(act.planes.sign, want.sign)
    # against the un-fused modules (the device's own BatchNorm arithmetic): the device fold gives the same bits, hence the same logits
    with torch.no_grad():
        yu = seq(x)
        yd = fuse_sequential(seq, fold="device")(x)
    assert torch.equal(yd, yu)
    # the reference fold differs from it only at ties: flipped signs counted, each within 1e-5 of the BatchNorm threshold
    with torch.no_grad():
        v = bn(torch.relu(h))
        dev_bits = ops.sign_pack(ops.binarize(v))[0].sign
    flipped = _unpack_bits(act.planes.sign ^ dev_bits, 96).bool()
    assert int(flipped.sum()) <= 2 and (not flipped.any() or float(v[flipped].abs().max()) <= 1e-5 * float(v.abs().mean()))


@pytest.mark.gpu
def test_assume_codes_fit_skips_the_overflow_sync_and_allows_graph_capture(dev):
    """ops.ASSUME_CODES_FIT: a DoReFa W1A4 block whose activations are clipped to [0, 1] never overflows int8, so the
    per-tensor host check can be waived; the forward then captures as a hipGraph and replays identically."""
    from pytorch_quantize_impls_amd.functions import nnDorefaQuant
    from pytorch_quantize_impls_amd.utils import graphed
    torch.manual_seed(8)
    net = torch.nn.Sequential(torch.nn.Hardtanh(0.0, 1.0), nnDorefaQuant(4), DorefaConv2d(16, 32, 3, padding=1, bit_width=1),
                              torch.nn.Hardtanh(0.0, 1.0), nnDorefaQuant(4), DorefaConv2d(32, 8, 3, padding=1, bit_width=1)).to(dev).eval()
    x = torch.rand((4, 16, 10, 10), device=dev).contiguous(memory_format=torch.channels_last) * 1.5
    with torch.no_grad():
        want = net(x)
    ops.ASSUME_CODES_FIT = True
    try:
        with torch.no_grad():
            assert torch.equal(net(x), want)
        gm = graphed(net, x)
        x2 = torch.rand_like(x)
        with torch.no_grad():
            want2 = net(x2)
        assert torch.equal(gm(x2), want2)
    finally:
        ops.ASSUME_CODES_FIT = False
