"""Round-4 GPU parity: the XNOR-Net family on its own kernels (VERDICT r3 "next" item 1).

  * qt_conv2d_implicit_taps — conv2d(x, sign(W) * alpha[1, 1, kh, kw]) for +-1 activations as ONE fp4 matrix-core pass with the taps'
    alphas applied on the accumulators (functions/xnor_connect.py:135-146, layers/xnor_layers.py:36-69):
      - bit-exact against the digests of the REFERENCE layer on power-of-two-per-tap weights at the AlexNet conv2 / conv3 / conv5
        shapes (tests/golden/make_golden_r4.py G16): tap order, tap boundaries, padding, factor table;
      - <= 1e-5 (normalised) against the fp64 evaluation of the reference FUNCTIONS, forward and backward (G17);
      - <= 1e-5 against the oracle at ragged shapes, every tile configuration, taps whose alpha is 0;
  * the backward of XNORConv2d / XNORDense on the matrix-core routes, no dense-library call (_fused.LIBRARY_PATHS);
  * qt_xnor_tap_prep_f32 (alpha + Horner tables) against numpy."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, norm_err

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import _lib, ops, synth  # noqa: E402
from pytorch_quantize_impls_amd.functions import BinaryConnectDeterministic, _fused, xnor_connect  # noqa: E402
from pytorch_quantize_impls_amd.layers import XNORConv2d, LinearXNOR  # noqa: E402

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g4():
    return np.load(os.path.join(GOLDEN_DIR, "golden_r4_v1.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def h4():
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r4.json")) as fh:
        return json.load(fh)["cases"]


@pytest.fixture()
def all_shapes_on_the_routes():
    old = _fused.BWD_MFMA_MIN_MACS
    _fused.BWD_MFMA_MIN_MACS = 0
    yield
    _fused.BWD_MFMA_MIN_MACS = old


def n(t):
    return t.detach().cpu().numpy()


def t32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


# ---- alpha + Horner tables ---------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("Cout,Cin,k", [(576, 192, 5), (1152, 576, 3), (7, 3, 11), (5, 4, 1), (64, 64, 7), (3, 2, 32)])
def test_tap_prep_vs_numpy(dev, Cout, Cin, k):
    w = synth.normal(Cout + k, (Cout, Cin, k, k), 0.1)
    if k >= 3:
        w[:, :, 1, 1] = 0.0                      # a tap whose alpha is 0: inherits its predecessor's scale
        w[:, :, 0, 0] = 0.0                      # ... a leading one: the first non-zero scale
    ts = ops.xnor_tap_prep(t32(w, dev))
    T = k * k
    alpha = np.abs(w.astype(np.float64)).mean((0, 1)).reshape(-1)
    assert np.abs(n(ts.alpha).astype(np.float64) - alpha).max() <= 1e-6 * alpha.max()       # three-level fp32 sum
    a32 = n(ts.alpha)
    for tab, order in ((n(ts.fwd), a32), (n(ts.bwd), a32[::-1])):
        nz = order[order != 0]
        eff, prev = [], (nz[0] if nz.size else np.float32(1))
        for v in order:
            prev = v if v != 0 else prev
            eff.append(prev)
        eff = np.asarray(eff, dtype=np.float32)
        assert tab.shape == (T + 1,) and tab[0] == 1.0 and tab[T] == eff[-1]
        assert np.array_equal(tab[1:T], eff[:-1] / eff[1:])
    # eval mode: the scales are given (the weight already holds sign(W) * alpha)
    ts2 = ops.xnor_tap_prep(alpha=ts.alpha)
    assert torch.equal(ts2.tables, ts.tables)


# ---- the reference layer's digests (bit-exact) ---------------------------------------------------------------------------------------

def _pow2_weight(c):
    s = synth.pm1(c["w_seed"], (c["Cout"], c["Cin"], c["k"], c["k"]))
    return (s * np.exp2(np.asarray(c["tap_exponents"], dtype=np.float32))[None, None]).astype(np.float32)


@pytest.mark.parametrize("name", ["xnorconv_conv2", "xnorconv_conv3", "xnorconv_conv5", "xnorconv_odd_96_72_9_s2"])
@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_xnor_conv_layer_reference_digest(dev, h4, name, channels_last, mode):
    c = h4[name]
    x = t32(synth.pm1(c["x_seed"], (c["B"], c["Cin"], c["H"], c["H"])), dev)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    conv = XNORConv2d(c["Cin"], c["Cout"], c["k"], stride=c["stride"], padding=c["pad"]).to(dev)
    conv.weight.data.copy_(t32(_pow2_weight(c), dev))
    conv.bias.data.zero_()
    if mode == "eval":
        conv.eval()
    before = dict(_lib.call_counts)
    with torch.no_grad():
        y = conv(BinaryConnectDeterministic.apply(x))              # (eval mode: a deferred activation until it is used)
        a = np.ascontiguousarray(n(y.contiguous()), dtype=np.float32)
    assert _lib.call_counts["qt_conv2d_implicit_taps"] == before.get("qt_conv2d_implicit_taps", 0) + 1
    assert a.shape[1] == c["Cout"]
    assert hashlib.sha256(a.tobytes()).hexdigest() == c["sha256_f32"], (float(a.astype(np.float64).sum()), c["sum"])


# ---- the reference functions in fp64: forward + backward ----------------------------------------------------------------------------

def _sampled_err(g4, name, key, full):
    want = g4[f"g17_{name}_{key}"]
    mx, st = g4[f"g17_{name}_{key}_max"]
    got = n(full.contiguous()).astype(np.float64).reshape(-1)[::int(st)]
    assert got.shape == want.shape
    return float(np.abs(got - want).max() / mx)


@pytest.mark.parametrize("name", ["conv2", "conv3", "conv5", "odd_96_72_9_s2"])
@pytest.mark.parametrize("channels_last", [False, True])
def test_xnor_conv_function_vs_reference_fp64(dev, g4, name, channels_last, all_shapes_on_the_routes):
    B, Cin, Cout, H, k, s, p, seed = (int(v) for v in g4[f"g17_{name}_geom"])
    x = t32(synth.pm1(seed, (B, Cin, H, H)), dev)
    w = synth.normal(seed + 1, (Cout, Cin, k, k), 0.05)
    w[0, 0, 0, 0] = 0.0
    w = t32(w, dev).requires_grad_(True)
    b = t32(synth.normal(seed + 2, (Cout,)), dev).requires_grad_(True)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    xs = BinaryConnectDeterministic.apply(x.requires_grad_(True))
    xs.retain_grad()
    op = xnor_connect.XNORConv2d([0, 1], False, s, p, 1, 1)
    _fused.LIBRARY_PATHS.clear()
    before = dict(_lib.call_counts)
    y = op.apply(xs, w, b)
    assert _lib.call_counts["qt_conv2d_implicit_taps"] == before.get("qt_conv2d_implicit_taps", 0) + 1
    assert _sampled_err(g4, name, "y", y) <= TOL
    go = t32(synth.normal(seed + 3, tuple(y.shape)), dev)
    if channels_last:
        go = go.contiguous(memory_format=torch.channels_last)
    y.backward(go)
    assert _sampled_err(g4, name, "gw", w.grad) <= TOL
    assert _sampled_err(g4, name, "gb", b.grad) <= TOL
    assert _sampled_err(g4, name, "gx", xs.grad) <= TOL
    if s == 1:          # strided grad_input has no per-tap route (only the real-valued first layer is strided in the reference's models)
        assert not _fused.LIBRARY_PATHS, dict(_fused.LIBRARY_PATHS)
        assert _lib.call_counts["qt_conv2d_implicit_taps"] == before.get("qt_conv2d_implicit_taps", 0) + 2


@pytest.mark.parametrize("name", ["fc1", "fc3", "fc_odd"])
def test_xnor_dense_function_vs_reference_fp64(dev, g4, name, all_shapes_on_the_routes):
    B, K, N, seed = (int(v) for v in g4[f"g17_{name}_geom"])
    x = t32(synth.pm1(seed, (B, K)), dev)
    w = synth.normal(seed + 1, (N, K), 0.05)
    w[0, 0] = 0.0
    w = t32(w, dev).requires_grad_(True)
    b = t32(synth.normal(seed + 2, (N,)), dev).requires_grad_(True)
    xs = BinaryConnectDeterministic.apply(x.requires_grad_(True))
    xs.retain_grad()
    _fused.LIBRARY_PATHS.clear()
    y = xnor_connect.XNORDense([0, 1]).apply(xs, w, b)
    assert _sampled_err(g4, name, "y", y) <= TOL
    y.backward(t32(synth.normal(seed + 3, tuple(y.shape)), dev))
    for key, t in (("gx", xs.grad), ("gw", w.grad), ("gb", b.grad)):
        assert _sampled_err(g4, name, key, t) <= TOL, key
    assert not _fused.LIBRARY_PATHS, dict(_fused.LIBRARY_PATHS)


# ---- ragged shapes / every tile configuration / zero taps, against the oracle ------------------------------------------------------

TAPS_SHAPES = [
    # N, Cin, Cout, H, W, k, s, p          tile the dispatch takes (csrc/conv_taps.hip)
    (2, 64, 64, 9, 11, 3, 1, 1),            # 64-wide, padded: Conv64
    (2, 64, 64, 12, 10, 3, 1, 0),           # ... un-padded: ConvV64
    (3, 40, 128, 14, 14, 3, 1, 1),          # channels padded 40 -> 64; 128-wide
    (2, 130, 200, 8, 9, 3, 1, 0),           # 130 -> 192 channels (3 k-steps per tap), 192-wide + ragged column tile, un-padded
    (2, 192, 192, 13, 13, 5, 1, 2),         # 25 taps
    (1, 64, 256, 30, 30, 1, 1, 0),          # one tap: no factor at all; 256 -> 128-wide tiles
    (2, 128, 96, 10, 10, 7, 2, 3),          # 49 taps, stride 2
    (1, 256, 512, 6, 6, 3, 1, 1),           # small M, long K: skinny tiles
    (1, 512, 512, 4, 4, 3, 1, 0),           # ... un-padded: ConvVSkinny
    (8, 256, 256, 8, 8, 3, 1, 0),           # M = 288 .. un-padded small-M rules
    (70, 256, 384, 8, 8, 3, 1, 0),          # M = 2520
    (2, 64, 72, 21, 5, 3, (2, 1), (1, 0)),  # anisotropic stride / padding
    (2, 64, 64, 11, 11, 3, 1, 1, 2),        # dilation 2
]


@pytest.mark.parametrize("shape", TAPS_SHAPES)
@pytest.mark.parametrize("with_bias", [False, True])
def test_xnor_conv_vs_oracle(dev, oracle, shape, with_bias):
    N, Cin, Cout, H, W, k, s, p = shape[:8]
    d = shape[8] if len(shape) > 8 else 1
    seed = N * 1000 + Cin * 7 + Cout + k
    x = synth.pm1(seed, (N, Cin, H, W))
    w = synth.normal(seed + 1, (Cout, Cin, k, k), 0.3)
    w[w == 0] = 0.1
    if k >= 3:
        w[:, :, 0, 1] = 0.0                    # a tap with alpha == 0 (all its weights zero)
        w[1, 0, 2, 2] = 0.0                    # a lone zero weight: torch.sign keeps it at 0
    b = synth.normal(seed + 2, (Cout,)) if with_bias else None
    conv = XNORConv2d(Cin, Cout, k, stride=s, padding=p, dilation=d, bias=with_bias).to(dev)
    conv.weight.data.copy_(t32(w, dev))
    if with_bias:
        conv.bias.data.copy_(t32(b, dev))
    before = dict(_lib.call_counts)
    with torch.no_grad():
        y = conv(BinaryConnectDeterministic.apply(t32(x, dev)))
    assert _lib.call_counts["qt_conv2d_implicit_taps"] == before.get("qt_conv2d_implicit_taps", 0) + 1
    want = oracle.xnor_conv2d_forward(x, w, b, s, p, d)
    assert norm_err(n(y), want) <= TOL


def test_xnor_conv_untagged_pm1_input_is_detected(dev, oracle):
    """+-1 activation WITHOUT a quantiser's tag (e.g. behind a MaxPool): detected on the device, same route."""
    x = synth.pm1(5, (2, 64, 8, 8))
    w = synth.normal(6, (64, 64, 3, 3), 0.2)
    conv = XNORConv2d(64, 64, 3, padding=1, bias=False).to(dev)
    conv.weight.data.copy_(t32(w, dev))
    before = dict(_lib.call_counts)
    with torch.no_grad():
        y = conv(t32(x, dev))
        yr = conv(t32(x, dev) * 0.75)                   # a real-valued input: the six-term route
    assert _lib.call_counts["qt_conv2d_implicit_taps"] == before.get("qt_conv2d_implicit_taps", 0) + 1
    assert norm_err(n(y), oracle.xnor_conv2d_forward(x, w, None, 1, 1)) <= TOL
    assert norm_err(n(yr), oracle.xnor_conv2d_forward(x * 0.75, w, None, 1, 1)) <= TOL


def test_xnor_layers_training_step_without_the_dense_library(dev, all_shapes_on_the_routes):
    """conv -> flatten -> linear, one SGD-less step: every gradient against the fp64 evaluation of the same graph built from the
    reference expressions, and no hipBLASLt / MIOpen call on the way."""
    torch.manual_seed(3)
    conv = XNORConv2d(64, 64, 3, padding=1).to(dev)
    fc = LinearXNOR(64 * 6 * 6, 24).to(dev)
    x = torch.randn((4, 64, 6, 6), device=dev)
    _fused.LIBRARY_PATHS.clear()
    h = BinaryConnectDeterministic.apply(conv(BinaryConnectDeterministic.apply(x)))
    y = fc(h.reshape(4, -1))
    y.square().sum().backward()
    assert not _fused.LIBRARY_PATHS, dict(_fused.LIBRARY_PATHS)

    def ref():
        cw, cb = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
        fw, fb = fc.weight.detach().double().requires_grad_(True), fc.bias.detach().double().requires_grad_(True)
        xs = torch.where(x < 0, -1.0, 1.0).double()
        a = cw.abs().mean((0, 1), keepdim=True)
        h0 = torch.nn.functional.conv2d(xs, _XnorW.apply(cw, a, 0), cb, padding=1)
        hs = _Ste.apply(h0)
        a2 = fw.abs().mean(0, keepdim=True)
        yy = torch.nn.functional.linear(hs.reshape(4, -1), _XnorW.apply(fw, a2, 0), fb)
        yy.square().sum().backward()
        return yy, cw.grad, cb.grad, fw.grad, fb.grad

    yy, gcw, gcb, gfw, gfb = ref()
    assert norm_err(n(y), n(yy)) <= TOL
    for got, want in ((conv.weight.grad, gcw), (conv.bias.grad, gcb), (fc.weight.grad, gfw), (fc.bias.grad, gfb)):
        assert norm_err(n(got), n(want)) <= TOL


class _XnorW(torch.autograd.Function):
    """sign(W) * alpha with the reference's hand-written weight gradient (xnor_connect.py:126-127, 158-159)."""

    @staticmethod
    def forward(ctx, w, alpha, dim0):
        ctx.save_for_backward(w, alpha)
        ctx.dim0 = dim0
        return torch.sign(w) * alpha

    @staticmethod
    def backward(ctx, g):
        w, alpha = ctx.saved_tensors
        sgn = torch.sign(w)
        return alpha * g + sgn * torch.mean(g * sgn, ctx.dim0, keepdim=True), None, None


class _Ste(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.where(x < 0, -1.0, 1.0).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * (x.abs() <= 1.001).to(g.dtype)


# ---- the XNOR-Net flavour of config 3 as the un-modified module graph ---------------------------------------------------------------

import bench_models  # noqa: E402
from pytorch_quantize_impls_amd import lazy  # noqa: E402


def _alexnet_xnor(dev, seed=0):
    torch.manual_seed(seed)
    m = bench_models.alexnet_xnor(num_classes=10)
    for mod in m.modules():                            # fresh layers: weights of a trained-looking scale
        if isinstance(mod, (XNORConv2d, LinearXNOR)):
            mod.weight.data.normal_(0, 0.05)
            mod.bias.data.normal_(0, 0.1)
    bench_models.randomize_bn(m, seed)
    return m.to(dev).to(memory_format=torch.channels_last).eval()


def test_xnor_alexnet_module_graph_runs_the_tap_kernels_and_equals_the_eager_graph(dev):
    """xnor_net_convert(AlexNet) in eval mode under no_grad: every XNORConv2d is deferred and runs fused with the pooling /
    BatchNorm / Hardtanh / BinaryConnect modules behind it (per-tap scaled conv with the threshold epilogue, LinearXNOR on the
    packed bits); the logits equal the module-by-module evaluation BIT FOR BIT (device thresholds on the same fp32 conv values)."""
    m = _alexnet_xnor(dev)
    x = torch.randn(64, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    _fused.LIBRARY_PATHS.clear()
    with torch.no_grad():
        before = dict(_lib.call_counts)
        lazy.STATS.clear()
        y = m(x)
        stats = dict(lazy.STATS)
        used = {k: _lib.call_counts[k] - before.get(k, 0) for k in _lib.call_counts}
        with lazy.eager():
            e = m(x)
    assert stats.get("deferred") == 5 and stats.get("fused") == 5 and not stats.get("materialised"), stats
    assert used.get("qt_conv2d_implicit_taps_bits", 0) + used.get("qt_conv2d_implicit_taps_nib", 0) == 4, used
    # the three LinearXNOR layers on sign bits: two digit-plane GEMMs, the 10-way head in one streaming launch
    assert used.get("qt_bits_alpha_digits_i8", 0) == 2 and used.get("qt_i8_gemm_splitk", 0) == 2 and used.get("qt_xnor_head_i8", 0) == 1, used
    assert not _fused.LIBRARY_PATHS, dict(_fused.LIBRARY_PATHS)
    assert torch.isfinite(y).all()
    assert torch.equal(y, e)
    # ... and the explicit fused form (its first classifier layer reads the (h, w, c)-flattened bits in the weight's NCHW order)
    from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
    with torch.no_grad():
        f = FusedFeatureClassifier(m.features, m.classifieur, (256, 6, 6), fold="device")(x)
    assert torch.equal(f, y)


def test_xnor_alexnet_eager_graph_vs_the_oracle_chain(dev, oracle):
    """The module-by-module graph's conv outputs against the oracle (fp32 restatement of the reference), layer by layer on the
    +-1 activations the device produced: <= 1e-5 normalised at every XNORConv2d / LinearXNOR of the network (batch 2)."""
    m = _alexnet_xnor(dev, 1)
    x = torch.randn(2, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    seen = []

    def hook(mod, inp, out):
        seen.append((mod, inp[0].detach(), out.detach()))
    hs = [mod.register_forward_hook(hook) for mod in m.modules() if isinstance(mod, (XNORConv2d, LinearXNOR))]
    with torch.no_grad(), lazy.eager():
        m(x)
    for h in hs:
        h.remove()
    assert len(seen) == 8
    for mod, xin, out in seen:
        w = n(mod.weight)            # eval mode: the quantised image, which the op quantises again like upstream (a fixed point
        b = n(mod.bias)              # unless a weight is exactly zero)
        if isinstance(mod, XNORConv2d):
            want = oracle.xnor_conv2d_forward(n(xin.contiguous()), w, b, mod.stride, mod.padding, mod.dilation)
        else:
            want = oracle.xnor_dense_forward(n(xin), w, b)
        assert norm_err(n(out.contiguous()), want) <= TOL, type(mod).__name__


@pytest.mark.parametrize("B,K,N", [(4, 9216, 4096), (5, 100, 33), (256, 4096, 10), (3, 64, 7)])
def test_linear_xnor_on_packed_bits_vs_oracle(dev, oracle, B, K, N):
    from pytorch_quantize_impls_amd import packed
    x = synth.pm1(B + K, (B, K))
    w = synth.normal(B + K + 1, (N, K), 0.05)
    w[0, 0] = 0.0
    b = synth.normal(B + K + 2, (N,))
    lin = LinearXNOR(K, N).to(dev)
    lin.weight.data.copy_(t32(w, dev))
    lin.bias.data.copy_(t32(b, dev))
    lin.eval()
    planes, _ = ops.sign_pack(t32(x, dev))
    with torch.no_grad():
        y = lin(packed.PackedActivation(planes, (B, K)))
        y2 = lin(BinaryConnectDeterministic.apply(t32(x, dev)))
    # eval mode: the weight holds sign(W) * alpha and the op quantises it AGAIN, like upstream (layers/xnor_layers.py:24-33 +
    # xnor_connect.py:112): a column with an exact zero gets alpha * (N - 1) / N the second time
    want = oracle.xnor_dense_forward(x, oracle.xnor_dense_weight(w), b)
    assert norm_err(n(y), want) <= TOL and norm_err(n(y2), want) <= TOL


# ---- direct first-layer conv (csrc/conv_first_direct.hip) ----------------------------------------------------------------------------

FIRST_SHAPES = [
    # N, C, H, W, Cout, k, s, p
    (2, 3, 224, 224, 192, 11, 4, 2),        # AlexNet conv1 (models/Alexnet/Alexnet_Bin.py:13)
    (3, 3, 67, 45, 72, 11, 4, 2),           # ragged map, channel tail (72 = 2 tiles + 8)
    (2, 3, 30, 34, 200, 7, 2, 3),           # stride 2: channels padded 3 -> 4 in the patch; two channel blocks (grid.y = 2)
    (2, 1, 28, 28, 40, 5, 2, 0),            # one channel, no padding
    (1, 4, 33, 31, 64, 8, 4, 1),            # 4 channels, even kernel
    (2, 3, 19, 19, 24, 5, 3, 2),            # stride 3 (Cp = 4)
    (5, 2, 16, 40, 33, 4, 4, 0),            # kernel == stride
]


@pytest.mark.parametrize("shape", FIRST_SHAPES)
@pytest.mark.parametrize("kind", ["binary", "ternary", "xnor"])
@pytest.mark.parametrize("channels_last", [False, True])
def test_first_layer_direct_conv_vs_oracle(dev, oracle, shape, kind, channels_last):
    from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d
    N, C, H, W, Cout, k, s, p = shape
    seed = N * 100 + H + Cout
    x = synth.normal(seed, (N, C, H, W)) * 2.0
    x[0, 0, :3, :3] *= 37.0                          # a tile whose range differs from its neighbours' (per-tile scale)
    w = synth.uniform(seed + 1, (Cout, C, k, k), -1.3, 1.3)
    b = synth.normal(seed + 2, (Cout,))
    cls = {"binary": BinConv2d, "ternary": TerConv2d, "xnor": XNORConv2d}[kind]
    conv = cls(C, Cout, k, stride=s, padding=p).to(dev)
    conv.weight.data.copy_(t32(w, dev))
    conv.bias.data.copy_(t32(b, dev))
    conv.binary_input = False
    xd = t32(x, dev)
    if channels_last:
        xd = xd.contiguous(memory_format=torch.channels_last)
    want = {"binary": lambda: oracle.bin_conv2d_forward(x, w, b, s, p), "ternary": lambda: oracle.ter_conv2d_forward(x, w, b, s, p),
            "xnor": lambda: oracle.xnor_conv2d_forward(x, w, b, s, p)}[kind]()
    for mode in ("train", "eval"):
        conv.train(mode == "train")
        before = dict(_lib.call_counts)
        with torch.no_grad(), lazy.eager():
            y = conv(xd)
        assert _lib.call_counts["qt_conv_first_direct_f32"] == before.get("qt_conv_first_direct_f32", 0) + 1, mode
        assert tuple(y.shape) == want.shape
        assert norm_err(n(y), want) <= TOL, (mode, norm_err(n(y), want))
    if kind != "xnor":
        wq = {"binary": oracle.safe_sign, "ternary": oracle.ternarize}[kind](w)
        fw = ops.pack_first_layer_weight(t32(wq, dev), s)
        y2 = ops.conv_first_direct(xd, fw, t32(b, dev), s, p)
        Ho, Wo = want.shape[2], want.shape[3]
        assert norm_err(n(y2.view(N, Ho, Wo, Cout).permute(0, 3, 1, 2)), want) <= TOL


def test_first_layer_direct_conv_threshold_bits_equal_the_unfused_chain(dev):
    """conv1 -> MaxPool -> BatchNorm -> Hardtanh -> BinaryConnect deferred (direct kernel with the threshold epilogue + pooling on
    bits) against the same modules run one by one: identical sign planes."""
    from pytorch_quantize_impls_amd.layers import BinConv2d
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    torch.manual_seed(5)
    for cls in (BinConv2d, XNORConv2d):
        conv = cls(3, 192, 11, stride=4, padding=2).to(dev)
        conv.weight.data.normal_(0, 0.3)
        conv.binary_input = False
        bn = torch.nn.BatchNorm2d(192).to(dev)
        bn.running_mean.normal_(0, 3.0)
        bn.running_var.uniform_(5, 50)
        bn.weight.data.normal_(0, 1.0)
        seq = torch.nn.Sequential(conv, torch.nn.MaxPool2d(3, 2), bn, torch.nn.Hardtanh(), BinaryConnect()).eval()
        x = torch.randn(6, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
        before = dict(_lib.call_counts)
        with torch.no_grad():
            d = seq(x)
            assert isinstance(d, lazy.LazyActivation)
            bits = d._qt.force().planes.sign.clone()
            with lazy.eager():
                e = seq(x)
        assert _lib.call_counts["qt_conv_first_direct_bits_f32"] == before.get("qt_conv_first_direct_bits_f32", 0) + 1
        want = ops.sign_pack(e.permute(0, 2, 3, 1).contiguous())[0].sign
        assert torch.equal(bits, want), cls.__name__


# ---- batch-256 digests of the reference layers (tests/golden/make_golden_r4b.py G18) -----------------------------

@pytest.mark.parametrize("name", ["c3_binconv3_b256", "c5_terconv_b256"])
@pytest.mark.parametrize("mode", ["eval_deferred", "eval_eager", "train"])
def test_conv_layers_reproduce_the_reference_digest_at_batch_256(dev, name, mode):
    """BinConv2d at AlexNet's conv3 and TerConv2d at VGG-16's last block, the configs' own batch of 256, +-1 inputs: the SHA-256 of
    the fp32 result is the reference layer's (exact integer sums) — through the deferred inference graph (materialised), the
    module-by-module eval path and the training-mode forward, channels-last storage."""
    from pytorch_quantize_impls_amd import lazy
    from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r4b.json")) as fh:
        c = json.load(fh)["cases"][name]
    B, Cin, Cout, H, k = c["B"], c["Cin"], c["Cout"], c["H"], c["k"]
    conv = {"BinConv2d": BinConv2d, "TerConv2d": TerConv2d}[c["layer"]](Cin, Cout, k, stride=c["stride"], padding=c["pad"]).to(dev)
    conv.weight.data.copy_(t32(synth.uniform(c["w_seed"], (Cout, Cin, k, k), -1.0, 1.0), dev))
    conv.bias.data.copy_(t32(np.round(synth.normal(c["b_seed"], (Cout,)) * 4), dev))
    x = t32(synth.pm1(c["x_seed"], (B, Cin, H, H)), dev).contiguous(memory_format=torch.channels_last)
    conv.train(mode == "train")
    with torch.no_grad():
        if mode == "eval_deferred":
            y = conv(x)
            assert isinstance(y, lazy.LazyActivation)
            y = y.value()
        else:
            with lazy.eager():
                y = conv(x)
    a = np.ascontiguousarray(y.detach().float().contiguous().cpu().numpy(), dtype=np.float32)
    assert a.shape == (B, Cout, H, H)
    assert hashlib.sha256(a.tobytes()).hexdigest() == c["sha256_f32_nchw"], (float(a.astype(np.float64).sum()), c["sum"])


@pytest.mark.parametrize("real", [False, True])
def test_first_layer_threshold_epilogue_is_the_float_formula_bit_for_bit(dev, real):
    """The threshold epilogue compares the raw accumulator with ONE per-channel fp32 threshold (found in the kernel by bisection with
    the epilogue's own arithmetic) instead of evaluating  fl(fl(y * alpha) + beta) < 0  per value, y = the fp32 route's output.  The
    bits must be that formula's on y for every kind of channel: alpha > 0, alpha < 0, alpha == 0 (either sign of beta), beta == 0
    with zero bias (ties at +-0), huge / tiny alpha, infinite beta, NaN beta — and for pixels that are +-inf or NaN (both sides
    poisoned the same way), on tiles whose power-of-two scales differ by 2^60."""
    torch.manual_seed(31 + real)
    N, C, H, W, Cout, k, s_, p_ = 3, 3, 67, 75, 96, 11, 4, 2
    x = torch.randn(N, C, H, W, device=dev)
    x[0] *= 2.0 ** 40
    x[1] *= 2.0 ** -30
    x[2, :, :20, :20] = torch.round(x[2, :, :20, :20] * 4) / 4          # exact sums: accumulators that hit thresholds exactly
    x[2, 1, 40, 40] = float("inf")
    x[2, 2, 60, 10] = float("nan")
    x[2, 0, 5, 70] = float("-inf")
    x = x.contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, C, k, k, device=dev).sign()
    if real:
        w = w * torch.rand(1, 1, k, k, device=dev) * 0.1
    fw = ops.pack_first_layer_weight(w, s_, real=real)
    bias = torch.randn(Cout, device=dev)
    bias[::5] = 0.0
    alpha = torch.randn(Cout, device=dev)
    beta = torch.randn(Cout, device=dev) * 3.0
    alpha[1::8] = 0.0
    beta[1::16] = -beta[1::16].abs()
    beta[2::10] = 0.0
    alpha[3] = 1e30
    alpha[4] = -1e-30
    alpha[6] = 1e-42                                    # subnormal
    beta[7] = float("inf")
    beta[8] = float("-inf")
    beta[9] = float("nan")
    alpha[11] = float("nan")
    y = ops.conv_first_direct(x, fw, bias, s_, p_)
    bits = ops.conv_first_direct(x, fw, bias, s_, p_, epi=(alpha, beta)).sign
    want_neg = (y * alpha + beta) < 0                   # torch evaluates mul then add in fp32, like the fp32 route + the fold
    got_neg = torch.zeros_like(want_neg)
    for c in range(Cout):
        got_neg[:, c] = ((bits[:, c // 32] >> (c % 32)) & 1).bool()
    bad = (got_neg != want_neg).nonzero()
    assert bad.numel() == 0, (bad[:10].tolist(), [float(y[i, c]) for i, c in bad[:10].tolist()])
    assert bool(want_neg.any()) and bool((~want_neg).any())
    assert (bits[:, Cout // 32:] == 0).all()


# ---- un-modified TRAINING graphs reach the fused training nodes (lazy_train.py) --------------------------------

def _train_step(net, model, x, t, loss_fn):
    model.zero_grad(set_to_none=True)
    loss = loss_fn(net(x), t)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone()


def _assert_same_step(m_a, m_b, loss_a, loss_b, library=()):
    """``library``: parameters whose gradient comes from the dense library (the fp32 stem conv: MIOpen's weight gradient adds
    with atomics, two runs of the SAME graph differ in the last bit — tools/probes/lazy_train_dbg.py)."""
    assert torch.equal(loss_a, loss_b), (float(loss_a), float(loss_b))
    for (k, p), (_, q_) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert (p.grad is None) == (q_.grad is None), k
        if p.grad is not None and k in library:
            assert norm_err(n(p.grad), n(q_.grad)) <= TOL, k
        elif p.grad is not None:
            assert torch.equal(p.grad, q_.grad), k
    for (k, p), (_, q_) in zip(m_a.named_buffers(), m_b.named_buffers()):
        assert torch.equal(p, q_), k


def test_alexnet_module_graph_trains_on_the_fused_chain_by_itself(dev):
    """bench_models.AlexNetBin, un-modified, in training mode: every [MaxPool2d, BatchNorm, Hardtanh, BinaryConnect] run (and the
    reshape in front of the classifier's BinaryConnect) is recorded by lazy_train.py and runs as the SAME autograd node as the
    explicit TrainFusedAlexNetBin form — loss, every gradient and every BatchNorm buffer bit for bit — with no torch / MIOpen
    pooling, BatchNorm or Hardtanh kernel in the step (7 fused nodes, nothing replayed); ``lazy_train.eager()`` restores the
    module-by-module step."""
    import copy
    import bench_models
    from pytorch_quantize_impls_amd import lazy_train
    torch.manual_seed(21)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model, 2)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    twin = copy.deepcopy(model)
    x = torch.randn(16, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    t = torch.randint(0, 10, (16,), device=dev)
    lazy_train.STATS.clear()
    before = dict(_lib.call_counts)
    la = _train_step(model, model, x, t, torch.nn.functional.nll_loss)
    assert _lib.call_counts["qt_pool_bn_sign_train_f32"] - before.get("qt_pool_bn_sign_train_f32", 0) == 7
    assert lazy_train.STATS["fused:sign"] == 7 and not any(k.startswith("replayed") for k in lazy_train.STATS), dict(lazy_train.STATS)
    lb = _train_step(bench_models.TrainFusedAlexNetBin(twin), twin, x, t, torch.nn.functional.nll_loss)
    _assert_same_step(model, twin, la, lb)
    # the profiler's view: no MIOpen BatchNorm / torch pooling kernel in the un-modified step
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        _train_step(model, model, x, t, torch.nn.functional.nll_loss)
    names = [e.key for e in prof.key_averages()]
    assert not [k for k in names if "atch" in k and "orm" in k and "qt_" not in k], names
    assert not [k for k in names if "max_pool" in k.lower() and "qt_" not in k], names
    # switched off: the module-by-module step (MIOpen's BatchNorm: same loss to rounding, not bit for bit)
    before = dict(_lib.call_counts)
    with lazy_train.eager():
        lc = _train_step(twin, twin, x, t, torch.nn.functional.nll_loss)
    assert _lib.call_counts["qt_pool_bn_sign_train_f32"] == before.get("qt_pool_bn_sign_train_f32", 0)
    assert abs(float(lc) - float(la)) <= 5e-2 * abs(float(la))


class _ModuleStem(torch.nn.Module):
    def __init__(self, bn, quant):
        super().__init__()
        self.bn, self.quant = bn, quant

    def forward(self, x):
        return self.quant(torch.relu(self.bn(x)))


@pytest.mark.parametrize("w_bits", [1, 3])
def test_dorefa_resnet_module_graph_trains_on_the_fused_chain_by_itself(dev, w_bits):
    """bench_models.DorefaResNet18, un-modified, in training mode: BatchNorm [+ shortcut] -> ReLU -> nnDorefaQuant behind each of the
    16 DorefaConv2d runs as one _TrainBnActQuantFn node and the 3 shortcut BatchNorms as its quantiser-less form — the step of
    TrainFusedDorefaResNet18 bit for bit (with the fp32 stem's BatchNorm left to MIOpen in both, the one chain that does not
    start at a quantised layer)."""
    import copy
    import bench_models
    from pytorch_quantize_impls_amd import lazy_train
    torch.manual_seed(22)
    m = bench_models.DorefaResNet18(w_bits=w_bits, a_bits=4)
    bench_models.randomize_bn(m, seed=3)
    m = m.to(dev).to(memory_format=torch.channels_last).train()
    twin = copy.deepcopy(m)
    x = torch.randn(32, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    t = torch.randint(0, 10, (32,), device=dev)
    lazy_train.STATS.clear()
    before = dict(_lib.call_counts)
    la = _train_step(m, m, x, t, torch.nn.functional.cross_entropy)
    assert _lib.call_counts["qt_bn_train_stats_f32"] - before.get("qt_bn_train_stats_f32", 0) == 19      # 16 convs + 3 shortcuts
    assert _lib.call_counts["qt_bn_act_train_backward_f32"] - before.get("qt_bn_act_train_backward_f32", 0) == 19
    assert lazy_train.STATS["fused:quant"] == 16 and lazy_train.STATS["replayed:bn"] == 3, dict(lazy_train.STATS)
    explicit = bench_models.TrainFusedDorefaResNet18(twin)
    explicit.q0 = _ModuleStem(twin.bn, twin.quant)
    lb = _train_step(explicit, twin, x, t, torch.nn.functional.cross_entropy)
    _assert_same_step(m, twin, la, lb, library=("stem.weight",))


def test_training_chain_outside_the_grammar_is_the_module_chain(dev):
    """Recorded calls whose consumer is not BinaryConnect / nnDorefaQuant: BatchNorm by the quantiser-less node (fp64 parity like
    the fused forms), the rest by torch on the real tensor; a BatchNorm on C % 4 != 0 channels, ceil-mode pooling, an eval-mode
    BatchNorm and dropout are not recorded at all; the last layer's output is the real tensor for C++ callers."""
    from pytorch_quantize_impls_amd import lazy_train
    from pytorch_quantize_impls_amd.layers import BinConv2d, LinearBin
    torch.manual_seed(23)
    conv = BinConv2d(16, 24, 3, padding=1).to(dev).train()
    bn = torch.nn.BatchNorm2d(24).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    x = torch.where(torch.randn(6, 16, 9, 9, device=dev) < 0, -1.0, 1.0)
    out = conv(x)
    assert type(out) is lazy_train.TrainOut
    (g,) = torch.autograd.grad(out.sum(), conv.weight)                   # C++ entry: the ordinary tensor and graph
    assert torch.isfinite(g).all()
    h = bn(conv(x))
    assert type(h) is lazy_train.TrainChain
    y = torch.tanh(h)
    rbn = torch.nn.BatchNorm2d(24).double().train()
    rbn.load_state_dict({k: v.double().cpu() if v.is_floating_point() else v.cpu() for k, v in bn.state_dict().items()
                         if k != "num_batches_tracked"}, strict=False)
    rbn.running_mean.zero_()
    rbn.running_var.fill_(1.0)
    with lazy_train.eager():
        ref = torch.tanh(rbn(conv(x).double().cpu()))
    assert norm_err(n(y), n(ref)) <= TOL
    assert norm_err(n(bn.running_mean), n(rbn.running_mean)) <= 1e-6 and int(bn.num_batches_tracked) == 1
    # not recorded
    bn6 = torch.nn.BatchNorm2d(6).to(dev).train()
    assert type(bn6(BinConv2d(16, 6, 3).to(dev).train()(x))) is torch.Tensor
    assert type(torch.nn.functional.max_pool2d(conv(x), 2, 2, ceil_mode=True)) is torch.Tensor
    assert type(bn.eval()(conv(x))) is torch.Tensor
    assert type(torch.nn.functional.dropout(conv(x), 0.5, True)) is torch.Tensor
    lin = LinearBin(32, 10).to(dev).train()
    o = lin(torch.randn(4, 32, device=dev))
    loss = torch.nn.functional.cross_entropy(o, torch.randint(0, 10, (4,), device=dev))
    loss.backward()
    assert lin.weight.grad is not None and type(loss) is torch.Tensor
