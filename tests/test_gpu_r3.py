"""Round-3 GPU parity: the training-path routes VERDICT r2 listed as missing.

  * strided convs (the 3x3 / stride-2 and 1x1 / stride-2 convs of models/Resnet/Resnet_bin.py:20-33): grad_input through the
    zero-dilated gradient, grad_weight through the space-to-depth image / the sub-sampled GEMM, against fp64 of
    torch.nn.grad.conv2d_input / conv2d_weight (functions/binary_connect.py:141-143);
  * DoReFa layers with k-bit weights in TRAINING mode (layers/dorefa_layers.py:41-45,77-82, functions/dorefa_connect.py:99-111):
    forward and every gradient against the fp64 evaluation of the reference expression, no dense-library call;
  * the DoReFa ResNet-18 training step (W1A4 and W3A4) without a dense-library path.

Float tails: max|a-b| / max|b| <= 1e-5."""
import copy

import numpy as np
import pytest
import torch

from conftest import norm_err

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import _lib, ops  # noqa: E402
from pytorch_quantize_impls_amd.functions import BinaryConnectDeterministic, nnDorefaQuant, _fused  # noqa: E402
from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d, DorefaConv2d, LinearDorefa  # noqa: E402

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


def n(t):
    return t.detach().cpu().numpy()


@pytest.fixture()
def all_shapes_on_the_routes():
    """The routes are dispatched from 2^27 MACs on; the tests exercise them at every shape."""
    old = _fused.BWD_MFMA_MIN_MACS
    _fused.BWD_MFMA_MIN_MACS = 0
    yield
    _fused.BWD_MFMA_MIN_MACS = old


# ---- strided convs: grad_input ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("N,Cin,Cout,H,W,k,s,p,cl", [(4, 64, 128, 32, 32, 3, 2, 1, True), (4, 64, 128, 32, 32, 1, 2, 0, True),
                                                      (3, 24, 40, 15, 17, 3, 2, 1, False), (2, 16, 8, 14, 9, 1, 2, 0, True),
                                                      (2, 8, 12, 19, 20, 3, 3, 1, True), (3, 32, 32, 16, 16, 5, 2, 2, False),
                                                      (2, 256, 512, 8, 8, 3, 2, 1, True)])
@pytest.mark.parametrize("kind", ["sign", "raw"])
def test_strided_grad_input_vs_fp64(dev, N, Cin, Cout, H, W, k, s, p, cl, kind):
    g = torch.Generator(device=dev).manual_seed(N * 100 + Cin + k)
    if kind == "sign":
        wq = torch.randint(-1, 2, (Cout, Cin, k, k), generator=g, device=dev).float()
    else:       # odd integer levels of a 3-bit DoReFa weight
        wq = (2 * torch.randint(0, 8, (Cout, Cin, k, k), generator=g, device=dev) - 7).float()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    go = torch.randn((N, Cout, Ho, Wo), device=dev, generator=g)
    if cl:
        go = go.contiguous(memory_format=torch.channels_last)
    before = dict(_lib.call_counts)
    got = ops.conv2d_grad_input_q((N, Cin, H, W), wq, go, s, p, 1, kind=kind)
    assert got is not None and _lib.call_counts["qt_conv2d_implicit"] > before.get("qt_conv2d_implicit", 0)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), wq.double(), go.double(), stride=s, padding=p)
    assert norm_err(n(got), n(ref)) <= TOL


# ---- strided convs: grad_weight -----------------------------------------------------------------------------------------

@pytest.mark.parametrize("N,Cin,Cout,H,k,s,p,cl", [(8, 64, 128, 32, 3, 2, 1, True), (8, 64, 128, 32, 1, 2, 0, True),
                                                    (5, 16, 40, 16, 3, 2, 1, False), (3, 128, 256, 16, 3, 2, 1, True),
                                                    (4, 256, 512, 8, 1, 2, 0, True), (3, 12, 36, 9, 1, 3, 0, False),
                                                    (2, 8, 32, 12, 4, 3, 1, True)])
@pytest.mark.parametrize("levels", [1.0, 15.0])
def test_strided_grad_weight_vs_fp64(dev, N, Cin, Cout, H, k, s, p, cl, levels):
    g = torch.Generator(device=dev).manual_seed(N * 100 + Cin + k)
    if levels == 1.0:
        x = torch.randint(-1, 2, (N, Cin, H, H), generator=g, device=dev).float()
    else:       # a 4-bit DoReFa image q / 15 exactly as the quantiser forms it
        x = nnDorefaQuant(4)(torch.rand((N, Cin, H, H), generator=g, device=dev) * 1.5).detach()
    Ho = (H + 2 * p - k) // s + 1
    go = torch.randn((N, Cout, Ho, Ho), device=dev, generator=g)
    if cl:
        x, go = x.contiguous(memory_format=torch.channels_last), go.contiguous(memory_format=torch.channels_last)
    assert ops.wgrad_strided_applicable(x.shape, go.shape, (k, k), s, p, 1)
    got = ops.conv2d_grad_weight_strided(x, go, (k, k), s, p, x_levels=levels)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), go.double(), stride=s, padding=p)
    assert got is not None and norm_err(n(got), n(ref)) <= TOL


def test_strided_route_applicability_rules():
    x, gsh = (8, 64, 32, 32), (8, 128, 16, 16)
    assert ops.wgrad_strided_applicable(x, gsh, (3, 3), 2, 1, 1) and ops.wgrad_strided_applicable(x, gsh, (1, 1), 2, 0, 1)
    assert not ops.wgrad_strided_applicable(x, gsh, (3, 3), 1, 1, 1)            # stride 1: the pixel-major route itself
    assert not ops.wgrad_strided_applicable(x, gsh, (3, 3), 2, 1, 2)            # dilated
    assert not ops.wgrad_strided_applicable(x, gsh, (5, 5), 2, 1, 1)            # taps beyond offsets {-1, 0} of the s2d image
    assert not ops.wgrad_strided_applicable((8, 64, 31, 31), gsh, (3, 3), 2, 1, 1)   # odd extent
    assert not ops.wgrad_strided_applicable((8, 4, 32, 32), gsh, (3, 3), 2, 1, 1)    # too few channels for a tile


@pytest.mark.parametrize("cls,k,s,p", [(BinConv2d, 3, 2, 1), (TerConv2d, 1, 2, 0), (BinConv2d, 1, 2, 0), (TerConv2d, 3, 2, 1)])
def test_strided_binarised_conv_backward_has_no_library_path(dev, all_shapes_on_the_routes, cls, k, s, p):
    """Training-mode BinConv2d / TerConv2d with stride 2 on a BinaryConnect-tagged activation (ResNet_bin stage transition):
    forward, grad_input, grad_weight (with the quantiser's STE mask) and grad_bias on this backend's kernels."""
    torch.manual_seed(k + s)
    conv = cls(64, 128, k, stride=s, padding=p).to(dev)
    conv.weight.data.uniform_(-1.3, 1.3)
    xr = torch.randn(6, 64, 32, 32, device=dev, requires_grad=True)
    xs = BinaryConnectDeterministic.apply(xr.contiguous(memory_format=torch.channels_last))
    xs.retain_grad()
    lib_before = dict(_fused.LIBRARY_PATHS)
    y = conv(xs)
    gout = torch.randn_like(y)
    y.backward(gout)
    assert dict(_fused.LIBRARY_PATHS) == lib_before, _fused.LIBRARY_PATHS
    wq = (torch.where(conv.weight < 0, -1.0, 1.0) if cls is BinConv2d else ops.ternarize(conv.weight.detach())).double()
    gi = torch.nn.grad.conv2d_input(xs.shape, wq, gout.double(), stride=s, padding=p)
    gw = torch.nn.grad.conv2d_weight(xs.detach().double(), conv.weight.shape, gout.double(), stride=s, padding=p)
    gw = torch.where(conv.weight.detach().abs() > 1.001, torch.zeros_like(gw), gw)
    assert norm_err(n(xs.grad), n(gi)) <= TOL
    assert norm_err(n(conv.weight.grad), n(gw)) <= TOL
    assert norm_err(n(conv.bias.grad), n(gout.double().sum((0, 2, 3)))) <= TOL


# ---- DoReFa, k-bit weights, training mode --------------------------------------------------------------------------------

def _fp64_layer_grads(layer, x, gout):
    """Forward + gradients of the reference expression F.conv2d / F.linear(x, weight_op(W), b) in fp64 on the CPU."""
    ref = copy.deepcopy(layer).cpu().double().train()
    xr = x.detach().cpu().double().requires_grad_(True)
    w = ref.weight_op.forward(ref.weight)
    if isinstance(ref, torch.nn.Conv2d):
        y = torch.nn.functional.conv2d(xr, w, ref.bias, ref.stride, ref.padding, ref.dilation, ref.groups)
    else:
        y = torch.nn.functional.linear(xr, w, ref.bias)
    y.backward(gout.detach().cpu().double())
    return y.detach(), xr.grad, ref.weight.grad, (ref.bias.grad if ref.bias is not None else None)


@pytest.mark.parametrize("k_w", [2, 3, 4, 7])
@pytest.mark.parametrize("coded", [True, False])
@pytest.mark.parametrize("Cin,Cout,ksz,s,p,H", [(64, 64, 3, 1, 1, 16), (64, 128, 3, 2, 1, 16), (64, 128, 1, 2, 0, 16)])
def test_dorefa_kbit_conv_training_vs_fp64(dev, all_shapes_on_the_routes, k_w, coded, Cin, Cout, ksz, s, p, H):
    torch.manual_seed(k_w * 10 + Cin + ksz)
    conv = DorefaConv2d(Cin, Cout, ksz, stride=s, padding=p, bias=True, bit_width=k_w).to(dev).train()
    conv.weight.data.normal_(0, 0.7)
    raw = (torch.rand(6, Cin, H, H, device=dev) * 1.4).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x = nnDorefaQuant(4)(raw) if coded else raw * 1.0
    x.retain_grad()
    lib_before = dict(_fused.LIBRARY_PATHS)
    before = dict(_lib.call_counts)
    y = conv(x)
    gout = torch.randn_like(y)
    y.backward(gout)
    lib_now = {k: v - lib_before.get(k, 0) for k, v in _fused.LIBRARY_PATHS.items() if v != lib_before.get(k, 0)}
    if coded:
        assert not lib_now, lib_now                                  # forward, grad_x and grad_W on this backend's kernels
        assert _lib.call_counts["qt_conv2d_implicit"] - before.get("qt_conv2d_implicit", 0) >= 2
    else:                                                            # a real activation: only grad_W has two real operands
        assert set(lib_now) <= {"conv grad_weight outside the matrix-core route"}, lib_now
    ry, rgx, rgw, rgb = _fp64_layer_grads(conv, x, gout)
    assert norm_err(n(y), n(ry)) <= TOL
    assert norm_err(n(x.grad), n(rgx)) <= TOL
    assert norm_err(n(conv.weight.grad), n(rgw)) <= 2 * TOL        # through tanh / max|tanh| of the weight quantiser in fp32
    assert norm_err(n(conv.bias.grad), n(rgb)) <= TOL


@pytest.mark.parametrize("k_w", [2, 3, 5])
@pytest.mark.parametrize("coded", [True, False])
def test_dorefa_kbit_linear_training_vs_fp64(dev, all_shapes_on_the_routes, k_w, coded):
    torch.manual_seed(k_w)
    lin = LinearDorefa(300, 70, bias=True, bit_width=k_w).to(dev).train()
    lin.weight.data.normal_(0, 0.7)
    raw = (torch.rand(96, 300, device=dev) * 1.3).requires_grad_(True)
    x = nnDorefaQuant(3)(raw) if coded else raw * 1.0
    x.retain_grad()
    lib_before = dict(_fused.LIBRARY_PATHS)
    y = lin(x)
    gout = torch.randn_like(y)
    y.backward(gout)
    lib_now = {k: v - lib_before.get(k, 0) for k, v in _fused.LIBRARY_PATHS.items() if v != lib_before.get(k, 0)}
    if coded:
        assert not lib_now, lib_now
    ry, rgx, rgw, rgb = _fp64_layer_grads(lin, x, gout)
    assert norm_err(n(y), n(ry)) <= TOL
    assert norm_err(n(x.grad), n(rgx)) <= TOL
    assert norm_err(n(lin.weight.grad), n(rgw)) <= 2 * TOL
    assert norm_err(n(lin.bias.grad), n(rgb)) <= TOL


@pytest.mark.parametrize("k_w", [1, 3])
@pytest.mark.parametrize("ksz,s,p", [(3, 1, 1), (3, 2, 1), (1, 2, 0)])
def test_dorefa_training_with_codes_beyond_int8_vs_fp64(dev, all_shapes_on_the_routes, k_w, ksz, s, p):
    """The reference's quantiser does not clamp (functions/dorefa_connect.py:11-25): relu(bn(x)) + shortcut reaches values
    whose 4-bit codes exceed 127 (even 256).  Forward then takes the exact-split route, the weight gradient contracts the
    two base-256 digits of the codes — still no dense-library call, still the fp64 result."""
    torch.manual_seed(k_w + ksz + s)
    conv = DorefaConv2d(64, 96, ksz, stride=s, padding=p, bias=True, bit_width=k_w).to(dev).train()
    conv.weight.data.normal_(0, 0.7)
    raw = (torch.rand(5, 64, 16, 16, device=dev) * 40.0).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x = nnDorefaQuant(4)(raw)                     # codes up to 600
    x.retain_grad()
    assert float(x.max()) * 15 > 300
    lib_before = dict(_fused.LIBRARY_PATHS)
    y = conv(x)
    gout = torch.randn_like(y)
    y.backward(gout)
    lib_now = {k: v - lib_before.get(k, 0) for k, v in _fused.LIBRARY_PATHS.items() if v != lib_before.get(k, 0)}
    assert not lib_now, lib_now
    ry, rgx, rgw, rgb = _fp64_layer_grads(conv, x, gout)
    assert norm_err(n(y), n(ry)) <= TOL
    assert norm_err(n(x.grad), n(rgx)) <= TOL
    assert norm_err(n(conv.weight.grad), n(rgw)) <= 2 * TOL
    assert norm_err(n(conv.bias.grad), n(rgb)) <= TOL


@pytest.mark.parametrize("w_bits", [1, 3])
def test_dorefa_resnet18_training_step_runs_on_this_backend(dev, w_bits):
    """C4's network in TRAINING mode, batch 64: forward + backward of every DorefaConv2d (stride-1 and stride-2 3x3, 1x1
    stride-2 shortcuts) without a dense-library contraction.  The first-layer stem and the classifier head are plain
    nn.Conv2d / nn.Linear in the reference too (models/samples/ResNet_Dorefa.py) and are not this path's."""
    import bench_models
    torch.manual_seed(4)
    m = bench_models.DorefaResNet18(w_bits=w_bits, a_bits=4)
    bench_models.randomize_bn(m, seed=3)
    m = m.to(dev).to(memory_format=torch.channels_last).train()
    x = torch.randn(64, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    t = torch.randint(0, 10, (64,), device=dev)
    _fused.LIBRARY_PATHS.clear()
    loss = torch.nn.functional.cross_entropy(m(x), t)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    assert not _fused.LIBRARY_PATHS, dict(_fused.LIBRARY_PATHS)
    for name, p_ in m.named_parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all(), name


# ---- two-term fp16 split of real-valued operands (ops.FLOAT_SPLIT = "f16x2", csrc/split_f16.hip) ---------------------------

def _decode_pairs(tp, K):
    raw = n(tp.data).view(np.float16)[:, :2 * K].astype(np.float64).reshape(tp.rows, K, 2)
    s = n(tp.scale).astype(np.float64)
    return raw, s


def test_f16x2_split_meets_its_stated_bound(dev):
    """x / s = hi + lo with s a power of two, max|x| / s in [2^14, 2^15):  |x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 max|x|)."""
    from pytorch_quantize_impls_amd import synth
    for seed, spread in ((1, 1.0), (2, 1e-6), (3, 1e6)):
        x = np.concatenate([synth.normal(seed, (6000,)) * 3 * spread, synth.uniform(seed + 10, (3000,), -1e-4, 1e-4) * spread,
                            (synth.normal(seed + 20, (1000,)) * spread * 10.0 ** synth.uniform(seed + 30, (1000,), -6, 0)),
                            np.array([0.0, -0.0, 1.0, -1.0, 3.14159274, 1.0000001], np.float32) * spread]).astype(np.float32)
        x = x.reshape(2, -1)
        K = x.shape[1]
        before = dict(_lib.call_counts)
        with ops.float_split("f16x2"):
            tp = ops.split_bf16x3(torch.from_numpy(x).to(dev))
        # (round 6: a dense activation is split by qt_f16x2_absmax_pack_f32 — max|x| partials + fold-and-split, two launches)
        assert tp.terms == 2 and tp.elem == 3 and any(_lib.call_counts[k] > before.get(k, 0)
                                                      for k in ("qt_f16x2_pack_f32", "qt_f16x2_absmax_pack_f32"))
        terms, s = _decode_pairs(tp, K)
        amax = float(np.abs(x).max())
        assert s[0] * s[1] == 1.0 and np.log2(s[0]) == np.floor(np.log2(s[0])) and 2.0 ** 14 <= amax / s[0] < 2.0 ** 15
        rec = terms.sum(2) * s[0]
        err = np.abs(rec - x.astype(np.float64))
        bound = np.maximum(2.0 ** -22 * np.abs(x.astype(np.float64)), 2.0 ** -39 * amax)
        assert (err <= bound).all(), float((err / bound).max())
        assert not n(tp.data)[:, 2 * K:].any()                                     # zero pad
    w = synth.uniform(3, (4, 37), -1.5, 1.5)
    w[0, 0] = 0.0
    for kind, q in (("binary", np.where(w < 0, -1.0, 1.0)), ("sign", np.sign(w)),
                    ("ternary", np.where(w >= 0.5, 1.0, np.where(w < -0.5, -1.0, 0.0)))):
        wt = ops.weight_bf16x3(torch.from_numpy(w).to(dev), kind, terms=2)
        vals = n(wt.data).view(np.float16)[:, :2 * 37].astype(np.float32).reshape(4, 37, 2)
        assert np.array_equal(vals, np.repeat(q[:, :, None], 2, 2).astype(np.float32)), kind
    lv = torch.from_numpy((2 * np.arange(0, 128) - 127).astype(np.float32).reshape(1, -1)).to(dev)     # 7-bit DoReFa levels
    raw = n(ops.weight_bf16x3(lv, "raw", terms=2).data).view(np.float16)[:, :256].astype(np.float32).reshape(1, 128, 2)
    assert np.array_equal(raw[..., 0], n(lv)) and np.array_equal(raw[..., 1], n(lv))


@pytest.mark.parametrize("M,N,K", [(5, 7, 31), (130, 70, 363), (300, 260, 1000), (257, 129, 4096)])
@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_float_linear_both_splits_vs_fp64(dev, M, N, K, mode):
    from pytorch_quantize_impls_amd import synth
    x = synth.normal(M + K, (M, K)) * 2.0
    x[:, ::7] *= 1e-4                                            # columns spanning four decades
    w = synth.uniform(N + K, (N, K), -1.5, 1.5)
    b = synth.normal(N, (N,))
    q = np.where(w < 0, -1.0, 1.0)
    with ops.float_split(mode):
        y = n(ops.float_linear(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), "binary", torch.from_numpy(b).to(dev)))
    ref = x.astype(np.float64) @ q.T + b
    assert norm_err(y, ref) <= (2e-6 if mode == "f16x2" else TOL), norm_err(y, ref)


@pytest.mark.parametrize("N,Cin,Cout,H,k,s,p,cl", [(4, 3, 192, 224, 11, 4, 2, True), (3, 3, 64, 33, 3, 1, 1, False),
                                                    (2, 16, 40, 19, 5, 2, 2, True), (2, 64, 96, 14, 3, 1, 1, True)])
@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_real_input_conv_both_splits_vs_fp64(dev, N, Cin, Cout, H, k, s, p, cl, mode, s2d_first_layer):
    """BinConv2d on a REAL-valued image (the first layer: models/Alexnet/Alexnet_Bin.py:13) — space-to-depth and plain
    routes, both splits, against fp64 of F.conv2d(x, safeSign(W), b)."""
    torch.manual_seed(Cin + Cout + k)
    conv = BinConv2d(Cin, Cout, k, stride=s, padding=p).to(dev).eval()
    conv.binary_input = False
    x = torch.randn(N, Cin, H, H, device=dev) * 2.0
    x[:, :, ::3] *= 1e-3
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    from pytorch_quantize_impls_amd import lazy
    before = dict(_lib.call_counts)
    with torch.no_grad(), lazy.eager(), ops.float_split(mode):
        y = conv(x)
    pk = "qt_f16x2" if mode == "f16x2" else "qt_bf16x3"
    if mode == "f16x2" and (Cin, Cout, k, s, p) == (3, 64, 3, 1, 1):
        pk = "qt_conv3x3_first_f32"        # round 6: the one-pass first-layer kernel splits its patch in LDS (no pack entry point)
    assert any(k2.startswith(pk) and v > before.get(k2, 0) for k2, v in _lib.call_counts.items()), mode
    ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), s, p)
    assert norm_err(n(y), n(ref)) <= (2e-6 if mode == "f16x2" else TOL), norm_err(n(y), n(ref))


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_gradient_conv_with_channels_spanning_six_decades(dev, mode):
    """grad_x of a binarised conv (functions/binary_connect.py:141-143) for a gradient whose channels span six decades: the
    per-tensor fp16 scale loses the small channels' low bits, but only relative to the LARGE ones — the normalised error
    stays inside 1e-5 (and the exact three-term route stays selectable)."""
    g_ = torch.Generator(device=dev).manual_seed(3)
    Cout, Cin = 96, 64
    wq = torch.randint(-1, 2, (Cout, Cin, 3, 3), generator=g_, device=dev).float()
    go = torch.randn((4, Cout, 20, 20), device=dev, generator=g_)
    go = (go * (10.0 ** torch.linspace(-6, 0, Cout, device=dev)).view(1, -1, 1, 1)).contiguous(memory_format=torch.channels_last)
    with ops.float_split(mode):
        got = ops.conv2d_grad_input_q((4, Cin, 20, 20), wq, go, 1, 1, 1)
    ref = torch.nn.grad.conv2d_input((4, Cin, 20, 20), wq.double(), go.double(), padding=1)
    assert norm_err(n(got), n(ref)) <= (2e-6 if mode == "f16x2" else TOL)


def test_alexnet_first_block_bits_do_not_depend_on_the_epilogue_in_f16x2_mode(dev):
    """Fused conv1 block (threshold-bit epilogue on the fp16-pair conv) == sign(BatchNorm(pool(conv1(x)))) of the same
    route's fp32 output, bit for bit (the epilogue forms exactly the value the conv would have stored)."""
    import bench_models
    from pytorch_quantize_impls_amd import lazy
    torch.manual_seed(0)
    m = bench_models.AlexNetBin(num_classes=10)
    bench_models.randomize_bn(m, 0)
    m = m.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn(8, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad(), ops.float_split("f16x2"):
        y = m(x)
        with lazy.eager():
            e = m(x)
    assert torch.equal(y, e)


# ---- reference digests / fp64 vectors for the shapes whose only comparator was the same device's library -----------------

@pytest.fixture(scope="module")
def hashes_r3():
    import json
    import os
    from conftest import GOLDEN_DIR
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r3.json")) as fh:
        return json.load(fh)["cases"]


@pytest.fixture(scope="module")
def golden_r3():
    import os
    from conftest import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, "golden_r3_v1.npz"), allow_pickle=False)


@pytest.mark.parametrize("case", ["terconv_c5_128_256_56", "terconv_c5_512_512_14", "binconv_c4_64_128_32_s2",
                                  "binconv_c4_64_128_32_1x1s2", "binconv_c4_256_512_8_s2"])
def test_config_size_convs_reproduce_the_reference_digest(dev, hashes_r3, case):
    """TerConv2d at the two remaining VGG channel classes, BinConv2d at the ResNet stage transitions (3x3 / stride 2 and the
    1x1 / stride 2 shortcut): SHA-256 of the int32 result the REFERENCE layer produced on the same +-1 input
    (tests/golden/make_golden_r3.py), train and eval mode, tagged and un-tagged input."""
    import hashlib
    from pytorch_quantize_impls_amd import lazy, synth
    h = hashes_r3[case]
    x = synth.pm1(h["x_seed"], (h["B"], h["Cin"], h["H"], h["H"]))
    w = synth.uniform(h["w_seed"], (h["Cout"], h["Cin"], h["k"], h["k"]), h["w_lo"], h["w_hi"])
    cls = TerConv2d if case.startswith("terconv") else BinConv2d
    conv = cls(h["Cin"], h["Cout"], h["k"], stride=h["stride"], padding=h["pad"]).to(dev)
    wd = torch.from_numpy(w).to(dev)
    conv.bias.data.zero_()
    for tagged in (False, True):
        xd = torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last)
        if tagged:
            xd = BinaryConnectDeterministic.apply(xd)
        for training in (True, False):
            conv.train(True)
            conv.weight.data.copy_(wd)
            conv.train(training)
            before = dict(_lib.call_counts)
            with torch.no_grad():
                y = lazy.resolve(conv(xd))
            assert _lib.call_counts["qt_conv2d_implicit"] > before.get("qt_conv2d_implicit", 0)
            yi = n(y.contiguous()).astype(np.int32)
            assert hashlib.sha256(np.ascontiguousarray(yi).tobytes()).hexdigest() == h["sha256_int32"], (tagged, training)
            assert float(y.double().sum()) == h["sum"]


@pytest.mark.parametrize("case", ["w1a4_core_c4_64_128_32_s2", "w1a4_core_c4_64_128_32_1x1s2"])
def test_strided_w1a4_integer_core_reproduces_the_reference_digest(dev, hashes_r3, case):
    """4-bit activation codes x sign weights on the int8 matrix cores with unit scale at the strided C4 shapes."""
    import hashlib
    from pytorch_quantize_impls_amd import synth
    h = hashes_r3[case]
    B, C, H, k, s, p = h["B"], h["Cin"], h["H"], h["k"], h["stride"], h["pad"]
    codes = np.floor(synth.uniform(h["x_seed"], (B, C, H, H), 0.0, 16.0)).clip(0, 15).astype(np.float32)
    w = synth.uniform(h["w_seed"], (h["Cout"], C, k, k), h["w_lo"], h["w_hi"])
    xq = (torch.from_numpy(codes).to(dev) / 15.0).contiguous(memory_format=torch.channels_last)
    px, _ = ops.dorefa_codes(xq.permute(0, 2, 3, 1).reshape(B * H * H, C), 4, want_f32=False, ld_bytes=ops.code_ld_bytes(C, 16))
    wp = ops.pack_conv_weight_codes(torch.from_numpy(w).to(dev))
    y2 = ops.conv2d_codes(px, (B, C, H, H), wp, (k, k), 1.0, None, s, p, 1)
    Ho = (H + 2 * p - k) // s + 1
    y = y2.view(B, Ho, Ho, h["Cout"]).permute(0, 3, 1, 2).contiguous()
    assert hashlib.sha256(np.ascontiguousarray(n(y).astype(np.int32)).tobytes()).hexdigest() == h["sha256_int32"]


def test_strided_binconv_backward_vs_reference_fp64_vectors(dev, all_shapes_on_the_routes, golden_r3):
    """G15: forward + every gradient of a training-mode strided BinConv2d against the fp64 autograd of the REFERENCE layer."""
    for name in golden_r3["g15_cases"]:
        Cin, Cout, H, k, s, p = (int(v) for v in golden_r3[f"g15_{name}_geom"])
        t = {kk: torch.from_numpy(golden_r3[f"g15_{name}_{kk}"]).to(dev) for kk in ("x", "w", "b", "go")}
        conv = BinConv2d(Cin, Cout, k, stride=s, padding=p).to(dev).train()
        conv.weight.data.copy_(t["w"])
        conv.bias.data.copy_(t["b"])
        xr = t["x"].contiguous(memory_format=torch.channels_last).requires_grad_(True)
        xs = BinaryConnectDeterministic.apply(xr)
        xs.retain_grad()
        lib_before = dict(_fused.LIBRARY_PATHS)
        y = conv(xs)
        y.backward(t["go"])
        assert dict(_fused.LIBRARY_PATHS) == lib_before
        assert norm_err(n(y), golden_r3[f"g15_{name}_y"]) <= TOL
        assert norm_err(n(xs.grad), golden_r3[f"g15_{name}_gx"]) <= TOL
        assert norm_err(n(conv.weight.grad), golden_r3[f"g15_{name}_gw"]) <= TOL
        assert norm_err(n(conv.bias.grad), golden_r3[f"g15_{name}_gb"]) <= TOL


def test_alexnet_training_step_vs_fp64_of_the_reference_op_sequence(dev):
    """the whole-step comparator is the FP64 evaluation of the reference's op sequence (CPU), at 1e-5 —
    and the device's fp32 library run of the same sequence is measured against it too, so it is visible which side a
    difference comes from."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "bench_train_step", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_train_step.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ours, fp32_lib = mod.gradient_agreement_fp64(4)
    print(f"worst normalised gradient difference vs fp64: this backend {ours:.2e}, fp32 library ops {fp32_lib:.2e}")
    assert ours <= TOL, (ours, fp32_lib)


# ---- training-mode chain [MaxPool] -> BatchNorm(batch stats) -> [Hardtanh] -> sign on this backend's kernels --------------------

@pytest.mark.parametrize("shape,pool,ht,cl", [((8, 192, 27, 27), (3, 2), True, True), ((4, 96, 13, 13), None, True, True),
                                               ((3, 40, 9, 11), (2, 2), False, False), ((64, 300), None, True, False),
                                               ((5, 36, 7, 7), (3, 1), True, True), ((2, 8, 6, 6), (2, 2), True, True)])
def test_training_chain_vs_fp64_of_the_module_chain(dev, shape, pool, ht, cl):
    """layers.FusedTrainPoolBnSign against the reference's module chain MaxPool2d -> BatchNorm -> Hardtanh -> BinaryConnect
    (models/Alexnet/Alexnet_Bin.py:13-17) evaluated in fp64 on the CPU: the +-1 output, the gradient w.r.t. the input,
    dgamma / dbeta, and the running statistics."""
    import copy
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    from pytorch_quantize_impls_amd.layers import FusedTrainPoolBnSign
    torch.manual_seed(sum(shape))
    C = shape[1]
    bn = (torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d)(C, eps=1e-4, momentum=0.15)
    with torch.no_grad():
        bn.weight.uniform_(0.3, 1.5)
        bn.weight[::3] *= -1.0
        bn.bias.normal_(0, 0.4)
    pm = torch.nn.MaxPool2d(*pool) if pool else None
    hm = torch.nn.Hardtanh() if ht else None
    ref_bn = copy.deepcopy(bn).double().train()
    mod = FusedTrainPoolBnSign(bn.to(dev), pm, hm).train()
    x = torch.randn(shape) * 1.7 + 0.3
    xd = x.to(dev)
    if cl and len(shape) == 4:
        xd = xd.contiguous(memory_format=torch.channels_last)
    xd.requires_grad_(True)
    before = dict(_lib.call_counts)
    y = mod(xd)
    assert _lib.call_counts["qt_pool_bn_sign_train_f32"] > before.get("qt_pool_bn_sign_train_f32", 0)
    gout = torch.randn(y.shape)
    y.backward(gout.to(dev))
    assert _lib.call_counts["qt_pool_bn_sign_train_backward_f32"] > before.get("qt_pool_bn_sign_train_backward_f32", 0)
    xr = x.double().requires_grad_(True)
    h = pm(xr) if pm else xr
    h = ref_bn(h)
    if hm:
        h = hm(h)
    yr = BinaryConnect()(h)
    yr.backward(gout.double())
    assert torch.equal(y.detach().cpu().double(), yr.detach())
    assert norm_err(n(xd.grad), n(xr.grad)) <= TOL
    assert norm_err(n(bn.weight.grad), n(ref_bn.weight.grad)) <= TOL
    assert norm_err(n(bn.bias.grad), n(ref_bn.bias.grad)) <= TOL
    assert norm_err(n(bn.running_mean), n(ref_bn.running_mean)) <= 1e-6
    assert norm_err(n(bn.running_var), n(ref_bn.running_var)) <= 1e-6
    assert int(bn.num_batches_tracked) == 1
    # the +-1 output carries its sign planes like BinaryConnect's
    from pytorch_quantize_impls_amd import packed
    assert packed.lookup(y, packed.NHWC if len(shape) == 4 else packed.ROWS_LAST) is not None


def test_training_chain_pool_ties_route_the_gradient_like_torch(dev):
    """Conv outputs of +-1 nets are integers: equal maxima inside a window are the rule.  The gradient goes to the FIRST
    maximum in scan order (torch's max_pool2d rule on CPU and GPU)."""
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    from pytorch_quantize_impls_amd.layers import FusedTrainPoolBnSign
    torch.manual_seed(0)
    x = torch.randint(-3, 4, (4, 32, 9, 9)).float()
    bn = torch.nn.BatchNorm2d(32)
    import copy
    ref_bn = copy.deepcopy(bn).double()
    mod = FusedTrainPoolBnSign(bn.to(dev), torch.nn.MaxPool2d(3, 2), torch.nn.Hardtanh()).train()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = mod(xd)
    gout = torch.randn(y.shape)
    y.backward(gout.to(dev))
    xr = x.double().requires_grad_(True)
    yr = BinaryConnect()(torch.nn.functional.hardtanh(ref_bn(torch.nn.functional.max_pool2d(xr, 3, 2))))
    yr.backward(gout.double())
    assert torch.equal(y.detach().cpu().double(), yr.detach())
    assert norm_err(n(xd.grad), n(xr.grad)) <= TOL


def test_alexnet_training_step_with_the_fused_training_chain(dev):
    """bench_models.TrainFusedAlexNetBin (every pool / BatchNorm / Hardtanh / sign run on csrc/train_chain.hip) against the FP64
    evaluation of the reference's op sequence on the CPU, +-1 pixels: same loss, every parameter gradient within 1e-5.
    BatchNorm's gamma / beta are randomised: with the default gamma = 1, beta = 0 and integer conv sums, values sit EXACTLY on
    the batch mean (a mean of 8 or 288 integers is often an integer), and every fp32 evaluation — MIOpen's, this one — lands
    such a tie on one side or the other of the fp64 result (tools/probes/train_chain_debug.py: one element of 73 728 differs
    between MIOpen and these kernels in that configuration, and the logits of a binarised net follow any single flip)."""
    import copy
    import importlib.util
    import os
    import bench_models
    spec = importlib.util.spec_from_file_location(
        "bench_train_step", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_train_step.py"))
    bts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bts)
    torch.manual_seed(0)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model, 2)        # generic gamma / beta: no value sits exactly on a BatchNorm zero
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    x = torch.where(torch.randn(8, 3, 224, 224, device=dev) < 0, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
    t = torch.randint(0, 10, (8,), device=dev)
    ref = copy.deepcopy(model).cpu().double().train()
    loss_ref = torch.nn.functional.nll_loss(bts.ref_forward(ref, x.cpu().double()), t.cpu())
    loss_ref.backward()
    want = {k: p_.grad for k, p_ in ref.named_parameters()}
    fused = bench_models.TrainFusedAlexNetBin(model)
    _fused.LIBRARY_PATHS.clear()
    before = dict(_lib.call_counts)
    loss = torch.nn.functional.nll_loss(fused(x), t)
    loss.backward()
    assert not _fused.LIBRARY_PATHS, dict(_fused.LIBRARY_PATHS)
    assert _lib.call_counts["qt_pool_bn_sign_train_f32"] - before.get("qt_pool_bn_sign_train_f32", 0) == 7       # 5 conv + 2 FC blocks
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-5 * abs(float(loss_ref.detach()))
    top = max(float(v.abs().max()) for v in want.values())
    for k, p_ in model.named_parameters():
        if k.endswith(".bias") and float(want[k].abs().max()) < 1e-3 * top:
            continue          # a bias in front of a training-mode BatchNorm: mathematically zero gradient, rounding noise
        assert norm_err(n(p_.grad), n(want[k])) <= TOL, k


# ---- pixel-major weight gradient with the gradient as two fp16 planes ----------------------------------------------------------

@pytest.mark.parametrize("N,Cin,Cout,H,W,k,p,cl", [(3, 64, 128, 13, 13, 3, 1, True), (2, 96, 160, 27, 27, 5, 2, True),
                                                    (4, 64, 64, 20, 17, 3, 1, False), (5, 70, 40, 9, 11, 3, 0, True),
                                                    (2, 33, 65, 6, 5, 5, 2, True), (7, 128, 256, 8, 8, 3, 1, True)])
@pytest.mark.parametrize("terms", [2, 3])
def test_weight_gradient_pixel_major_both_splits_vs_fp64(dev, N, Cin, Cout, H, W, k, p, cl, terms):
    """ops.conv2d_grad_weight_pm with the gradient in two fp16 planes (per-channel power-of-two scales) and in three exact bf16
    planes against torch.nn.grad.conv2d_weight in fp64: +-1 / 0 activations and 4-bit codes, the STE mask, the bias by-product,
    a gradient whose channels span six decades, chunked batches with a ragged tail."""
    g_ = torch.Generator(device=dev).manual_seed(N * 1000 + Cin + k)
    x = torch.randint(-1, 2, (N, Cin, H, W), generator=g_, device=dev).float()
    Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    go = torch.randn((N, Cout, Ho, Wo), device=dev, generator=g_)
    go = go * (10.0 ** torch.linspace(-6, 0, Cout, device=dev)).view(1, -1, 1, 1)
    if cl:
        x, go = x.contiguous(memory_format=torch.channels_last), go.contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, k, k), device=dev, generator=g_) * 0.8
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), go.double(), stride=1, padding=p)
    before = dict(_lib.call_counts)
    bias = []
    got = ops.conv2d_grad_weight_pm(x, go, (k, k), p, weight=w, bias_grad=bias, terms=terms)
    kern = "qt_wgrad_pm_f16" if terms == 2 else "qt_wgrad_pm_f32"
    assert _lib.call_counts[kern] > before.get(kern, 0)
    refm = torch.where(w.abs() <= 1.001, ref, torch.zeros_like(ref))
    assert norm_err(n(got), n(refm)) <= (2e-6 if terms == 2 else TOL)
    # per output channel as well (the two-plane form scales every gradient channel by its own power of two: a row of dW only
    # sees its channel, so the small channels keep the 1e-5 bar relative to THEIR maximum, like an fp32 GEMM)
    per = ((got.double() - refm).abs().amax((1, 2, 3)) / (refm.abs().amax((1, 2, 3)) + 1e-300)).cpu().numpy()
    assert (per <= TOL).all(), per.max()
    if bias:
        assert norm_err(n(bias[0]), n(go.double().sum((0, 2, 3)))) <= TOL
    xq = nnDorefaQuant(4)(torch.rand((N, Cin, H, W), generator=g_, device=dev) * 1.2).detach()
    if cl:
        xq = xq.contiguous(memory_format=torch.channels_last)
    gq = ops.conv2d_grad_weight_pm(xq, go, (k, k), p, x_levels=15.0, terms=terms)
    refq = torch.nn.grad.conv2d_weight(xq.double(), (Cout, Cin, k, k), go.double(), stride=1, padding=p)
    assert norm_err(n(gq), n(refq)) <= (2e-6 if terms == 2 else TOL)
    old = ops.WGRAD_GEMM_BYTES
    try:       # chunked batches (ragged tail): partial gradients accumulate under the same scales
        ops.WGRAD_GEMM_BYTES = max(1 << 16, (3 * Ho * (W + 2 * p) * 2 * max(Cout, 64) * 2 + (H + 2 * p) * (W + 2 * p) * 2 * max(Cin, 64) * 2) * 2
                                   + 64 * k * k * max(Cout, 64) * max(Cin, 64) * 4)
        ch = ops.conv2d_grad_weight_pm(x, go, (k, k), p, terms=terms)
    finally:
        ops.WGRAD_GEMM_BYTES = old
    if ch is not None:
        assert norm_err(n(ch), n(ref)) <= (2e-6 if terms == 2 else TOL)


@pytest.mark.parametrize("N,C,H,W,cl", [(3, 192, 13, 13, True), (2, 64, 27, 27, True), (5, 70, 9, 11, True), (2, 33, 6, 5, True),
                                        (4, 64, 20, 17, False), (2, 1152, 5, 4, True), (1, 2052, 3, 3, True), (3, 8, 1, 1, True)])
def test_per_channel_scales_of_the_two_plane_gradient(dev, N, C, H, W, cl):
    """qt_f16x2_absmax_scale_ch_f32 (csrc/split_f16.hip): s[c] = the power of two with max|g[:, c]| / s[c] in [2^14, 2^15), 1 for
    an all-zero / non-finite channel and for the padding channels; 1 / s[c] behind it — dense channels-last (vector and scalar
    instances, more than 1024 channels), NCHW and a sliced (non-dense) view."""
    g_ = torch.Generator(device=dev).manual_seed(C + H)
    g = torch.randn((N, C, H, W), device=dev, generator=g_) * (10.0 ** torch.linspace(-9, 6, C, device=dev)).view(1, -1, 1, 1)
    g[:, C // 2] = 0.0
    if C > 4:
        g[0, 3, 0, 0] = float("inf")
    if cl:
        g = g.contiguous(memory_format=torch.channels_last)
    Cp = (C + 63) // 64 * 64
    for view in (g, g[:, :, : max(1, H - 1)]):
        n_, c_, h_, w_ = view.shape
        out = torch.full((2 * Cp,), -7.0, device=dev)
        work = torch.empty((int(_lib.load().qt_f16x2_absmax_ch_work_words(c_)),), dtype=torch.int32, device=dev)
        _lib.call("qt_f16x2_absmax_scale_ch_f32", view.data_ptr(), *(int(v) for v in view.stride()), n_, c_, h_, w_, Cp,
                  work.data_ptr(), out.data_ptr(), None)
        torch.cuda.synchronize()
        amax = view.abs().amax((0, 2, 3)).double().cpu().numpy()
        s, inv = out[:Cp].double().cpu().numpy(), out[Cp:].double().cpu().numpy()
        assert (s[C:] == 1.0).all() and (inv[C:] == 1.0).all()
        assert (s * inv == 1.0).all() and (np.log2(s) == np.round(np.log2(s))).all()
        for c in range(C):
            if amax[c] == 0.0 or not np.isfinite(amax[c]):
                assert s[c] == 1.0, (c, amax[c], s[c])
            else:
                assert 2.0 ** 14 <= amax[c] / s[c] < 2.0 ** 15, (c, amax[c], s[c])


# ---- training-mode chain BatchNorm(batch stats) [+ shortcut] -> ReLU -> nnDorefaQuant on this backend's kernels ------------------

@pytest.mark.parametrize("shape,bits,relu,res,cl", [((8, 64, 16, 16), 4, True, True, True), ((4, 128, 8, 8), 4, True, False, True),
                                                     ((3, 36, 9, 11), 2, True, True, False), ((64, 300), 4, True, False, False),
                                                     ((6, 256, 4, 4), 0, False, False, True), ((5, 40, 7, 7), 8, False, True, True)])
def test_dorefa_training_chain_vs_fp64_of_the_module_chain(dev, shape, bits, relu, res, cl):
    """layers.FusedTrainBnActQuant against the reference's module chain BatchNorm2d (batch statistics) [+ shortcut] -> ReLU ->
    nnDorefaQuant (models/Resnet/Resnet_bin.py:63-97, functions/dorefa_connect.py:28-45) evaluated in fp64 on the CPU: the
    quantised image (a value within rounding distance of a quantiser boundary may land on the neighbouring level: counted and
    bounded), the gradients w.r.t. the input, the shortcut, gamma / beta, the running statistics and the int8 codes the output
    carries."""
    import copy
    from pytorch_quantize_impls_amd import packed
    from pytorch_quantize_impls_amd.layers import FusedTrainBnActQuant
    torch.manual_seed(sum(shape) + bits)
    C = shape[1]
    bn = (torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d)(C, eps=1e-4, momentum=0.15)
    with torch.no_grad():
        bn.weight.uniform_(0.3, 1.5)
        bn.weight[::3] *= -1.0
        bn.bias.normal_(0, 0.4)
    ref_bn = copy.deepcopy(bn).double().train()
    mod = FusedTrainBnActQuant(bn.to(dev), bits, relu=relu).train()
    x = torch.randn(shape) * 1.7 + 0.3
    r = torch.randn(shape) * 0.8 if res else None
    xd = x.to(dev)
    rd = r.to(dev) if res else None
    if cl and len(shape) == 4:
        xd = xd.contiguous(memory_format=torch.channels_last)
        rd = rd.contiguous(memory_format=torch.channels_last) if res else None
    xd.requires_grad_(True)
    if res:
        rd.requires_grad_(True)
    before = dict(_lib.call_counts)
    y = mod(xd, residual=rd)
    assert _lib.call_counts["qt_bn_train_stats_f32"] > before.get("qt_bn_train_stats_f32", 0)
    gout = torch.randn(y.shape)
    y.backward(gout.to(dev))
    assert _lib.call_counts["qt_bn_act_train_backward_f32"] > before.get("qt_bn_act_train_backward_f32", 0)
    xr = x.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    h = ref_bn(xr)
    if res:
        h = h + rr
    if relu:
        h = torch.relu(h)
    nlev = float((1 << bits) - 1) if bits else 0.0
    hq = (torch.round(h * nlev) / nlev).detach() + (h - h.detach()) if bits else h          # identity STE
    hq.backward(gout.double())
    got = y.detach().cpu().double()
    if bits:
        off = (got - hq.detach()).abs()
        assert float(off.max()) <= 1.0 / nlev + 1e-6                                          # at most the neighbouring level
        flipped = off > 1e-6
        assert int(flipped.sum()) <= max(2, got.numel() // 20000), int(flipped.sum())
        dist = ((h.detach() * nlev) - torch.floor(h.detach() * nlev) - 0.5).abs()             # distance to the rounding boundary
        assert float(dist[flipped].max() if flipped.any() else 0.0) <= 1e-4
        tag = packed.lookup_codes(y, packed.NHWC if len(shape) == 4 else packed.ROWS_LAST)
        assert tag is not None and tag.bit_width == bits
        codes = tag.codes[:, :C].cpu().double().view(*( (shape[0], shape[2], shape[3], C) if len(shape) == 4 else shape))
        want_codes = torch.round(got * nlev)
        if len(shape) == 4:
            want_codes = want_codes.permute(0, 2, 3, 1)
        ok = want_codes.abs() <= 127
        assert torch.equal(codes[ok], want_codes[ok])
    else:
        assert norm_err(n(got), n(hq.detach())) <= 1e-6
    assert norm_err(n(xd.grad), n(xr.grad)) <= TOL
    if res:
        assert norm_err(n(rd.grad), n(rr.grad)) <= TOL
    assert norm_err(n(bn.weight.grad), n(ref_bn.weight.grad)) <= TOL
    assert norm_err(n(bn.bias.grad), n(ref_bn.bias.grad)) <= TOL
    assert norm_err(n(bn.running_mean), n(ref_bn.running_mean)) <= 1e-6
    assert norm_err(n(bn.running_var), n(ref_bn.running_var)) <= 1e-6
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("w_bits", [1, 3])
def test_dorefa_resnet18_training_step_with_the_fused_training_chain(dev, w_bits):
    """bench_models.TrainFusedDorefaResNet18 (BatchNorm + shortcut add + ReLU + quantiser of every block as one
    FusedTrainBnActQuant node) against the module graph of the same parameters on the same input: no MIOpen BatchNorm / torch
    relu / add kernel between the DoReFa layers, no dense-library contraction, the same loss to 1e-3.  Gradients: a 4-bit net
    follows every code that lands on the other side of a rounding boundary, so the yardstick is the module graph ITSELF on an
    input moved by one part in 10^6 (cosine ~0.85 per parameter at this depth; tools/probes/train_resnet_fused_agreement.py) —
    the fused chain has to agree with the module graph at least that well, and to 0.999 at the classifier."""
    import copy
    import bench_models
    torch.manual_seed(4)
    m = bench_models.DorefaResNet18(w_bits=w_bits, a_bits=4)
    bench_models.randomize_bn(m, seed=3)
    m = m.to(dev).to(memory_format=torch.channels_last).train()
    m2, m3 = copy.deepcopy(m), copy.deepcopy(m)
    x = torch.randn(64, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    t = torch.randint(0, 10, (64,), device=dev)
    from pytorch_quantize_impls_amd import lazy_train
    with lazy_train.eager():          # the yardstick is the module-by-module graph (MIOpen BatchNorm, torch add / relu)
        loss_ref = torch.nn.functional.cross_entropy(m2(x), t)
        loss_ref.backward()
        torch.nn.functional.cross_entropy(m3(x * (1.0 + 1e-6)), t).backward()
    fused = bench_models.TrainFusedDorefaResNet18(m)
    _fused.LIBRARY_PATHS.clear()
    before = dict(_lib.call_counts)
    loss = torch.nn.functional.cross_entropy(fused(x), t)
    loss.backward()
    torch.cuda.synchronize()
    assert not _fused.LIBRARY_PATHS, dict(_fused.LIBRARY_PATHS)
    assert _lib.call_counts["qt_bn_train_stats_f32"] - before.get("qt_bn_train_stats_f32", 0) == 20     # stem + 16 convs + 3 shortcuts
    assert _lib.call_counts["qt_bn_act_train_backward_f32"] - before.get("qt_bn_act_train_backward_f32", 0) == 20
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-3 * abs(float(loss_ref.detach()))

    def cos(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))

    for (k, p_), (_, q_), (_, r_) in zip(m.named_parameters(), m2.named_parameters(), m3.named_parameters()):
        assert p_.grad is not None and torch.isfinite(p_.grad).all(), k
        if float(q_.grad.norm()) <= 1e-6 * q_.grad.numel() ** 0.5:
            continue
        floor = cos(q_.grad, r_.grad)
        assert cos(p_.grad, q_.grad) >= min(0.999, floor - 0.02), (k, cos(p_.grad, q_.grad), floor)
    assert cos(m.linear.weight.grad, m2.linear.weight.grad) >= 0.999
    for (k, b1), (_, b2), (_, b3) in zip(m.named_buffers(), m2.named_buffers(), m3.named_buffers()):
        if b1.dtype == torch.float32:       # running statistics: same yardstick (flipped codes upstream move the batch moments a little)
            assert norm_err(n(b1), n(b2)) <= max(1e-5, 3.0 * norm_err(n(b3), n(b2))), k


def test_dorefa_training_codes_beyond_the_fp16_plane_are_flagged_not_wrong(dev, all_shapes_on_the_routes):
    """Codes beyond +-2047 (a 4-bit activation beyond 136) are not exact in the pixel-major kernel's fp16 activation plane any
    more: the quantiser's device flag (bit 1) turns the weight gradient of that route into NaN — never a silently wrong number —
    while the forward and grad_x (exact-split routes) stay right; with the exact three-term split selected the base-256 digits
    still give the fp64 result (up to 2^16)."""
    from pytorch_quantize_impls_amd import packed
    torch.manual_seed(5)
    conv = DorefaConv2d(64, 96, 3, padding=1, bias=True, bit_width=1).to(dev).train()
    raw = (torch.rand(4, 64, 12, 12, device=dev) * 200.0).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x = nnDorefaQuant(4)(raw)
    assert float(x.detach().max()) * 15 > 2100
    assert int(packed.lookup_codes(x, packed.NHWC).overflow.item()) == 3
    y = conv(x)
    gout = torch.randn_like(y)
    y.backward(gout)
    ry, rgx, rgw, _ = _fp64_layer_grads(conv, x, gout)
    assert norm_err(n(y), n(ry)) <= TOL and norm_err(n(raw.grad), n(rgx)) <= TOL
    assert torch.isnan(conv.weight.grad).all()
    conv.zero_grad()
    with ops.float_split("bf16x3"):
        y2 = conv(x)
        y2.backward(gout)
    assert norm_err(n(conv.weight.grad), n(rgw)) <= 2 * TOL
    small = nnDorefaQuant(4)((torch.rand(4, 64, 12, 12, device=dev) * 100.0).contiguous(memory_format=torch.channels_last))
    assert int(packed.lookup_codes(small, packed.NHWC).overflow.item()) == 1          # beyond int8, inside the fp16 plane


# ---- grouped binarised convs (VERDICT r2 missing #5) ---------------------------------------------------------------------------------

@pytest.mark.parametrize("cls_name,groups,Cin,Cout,k,s,p,pm1", [("BinConv2d", 2, 128, 192, 3, 1, 1, True), ("TerConv2d", 4, 128, 128, 3, 1, 1, True),
                                                                 ("BinConv2d", 2, 64, 64, 3, 2, 1, True), ("TerConv2d", 2, 6, 64, 5, 1, 2, False),
                                                                 ("BinConv2d", 32, 64, 64, 3, 1, 1, True)])      # 2-channel groups
def test_grouped_binarised_conv_runs_group_by_group_on_this_backend(dev, all_shapes_on_the_routes, cls_name, groups, Cin, Cout, k, s, p, pm1):
    """BinConv2d / TerConv2d with groups > 1 (layers/binary_layers.py:59-60,103-106): training-mode forward + backward and the
    eval-mode forward against the fp64 evaluation of F.conv2d(x, Q(W), b, groups=G) and the quantiser's STE — every group on the
    groups == 1 routes, no dense-library call while a group keeps >= 32 channels (the weight-gradient kernels' tile floor; the
    2-channel groups of the last case are still right, their weight gradient is counted in LIBRARY_PATHS)."""
    from pytorch_quantize_impls_amd import layers as L
    from pytorch_quantize_impls_amd.functions import BinaryConnectDeterministic
    torch.manual_seed(groups + Cin)
    conv = getattr(L, cls_name)(Cin, Cout, k, stride=s, padding=p, groups=groups).to(dev).train()
    conv.weight.data.uniform_(-1.3, 1.3)
    raw = torch.randn(6, Cin, 14, 14, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x = BinaryConnectDeterministic.apply(raw) if pm1 else raw * 1.0
    x.retain_grad()
    lib_before = dict(_fused.LIBRARY_PATHS)
    y = conv(x)
    gout = torch.randn_like(y)
    y.backward(gout)
    if min(Cin, Cout) // groups >= 32 or not pm1:
        assert dict(_fused.LIBRARY_PATHS) == lib_before
    else:
        assert set(_fused.LIBRARY_PATHS) - set(lib_before) <= {"conv grad_weight outside the matrix-core route"}
    lib_before = dict(_fused.LIBRARY_PATHS)
    w = conv.weight.detach().double().cpu()
    wq = (torch.where(w < 0, -torch.ones_like(w), torch.ones_like(w)) if cls_name == "BinConv2d"
          else ops.ternarize(conv.weight.detach()).double().cpu()).requires_grad_(True)
    xr = x.detach().double().cpu().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wq, conv.bias.detach().double().cpu(), s, p, 1, groups)
    yr.backward(gout.double().cpu())
    assert norm_err(n(y), n(yr)) <= TOL
    assert norm_err(n(x.grad), n(xr.grad)) <= TOL
    gw = torch.where(w.abs() <= 1.001, wq.grad, torch.zeros_like(wq.grad))            # STE of the quantiser
    assert norm_err(n(conv.weight.grad), n(gw)) <= TOL
    conv.eval()
    with torch.no_grad():
        ye = conv(x.detach())
        ye = ye if type(ye) is torch.Tensor else ye * 1.0
    assert dict(_fused.LIBRARY_PATHS) == lib_before
    assert norm_err(n(ye), n(yr)) <= TOL


# ---- small-map tile rule of the padded implicit-GEMM convs -----------------------------------------------------------------------------

@pytest.mark.parametrize("kind", ["f16x2", "bf16x3", "int8", "fp4"])
def test_small_map_conv_tiles_equal_the_standard_tiles(dev, kind):
    """Padded convs with fewer 256-row tiles than CUs run 128x128 (qt_conv2d_implicit_variant 5) or 64x64 / 512-byte-stage
    (variant 6) tiles since round 3 (csrc/mfma_gemm.hip: Conv128x128, ConvSkinny); same accumulation order, so the result must
    equal the standard double-buffered tiles (variant 1) and the dispatcher's choice (0) bit for bit — for every operand kind
    (two fp16 planes, three bf16 planes, int8 codes, fp4 nibbles), at shapes on both sides of the rule."""
    from pytorch_quantize_impls_amd.layers import BinConv2d
    torch.manual_seed(11)
    shapes = [(8, 256, 8, 256, 3, 1), (16, 512, 4, 512, 3, 1), (4, 64, 12, 96, 3, 1), (2, 128, 9, 130, 5, 2), (64, 128, 16, 128, 3, 1)]
    before = _lib.call_counts["qt_conv2d_implicit_variant"]
    try:
        for (B, C, H, Co, k, p) in shapes:
            outs = []
            if kind in ("f16x2", "bf16x3"):
                x = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
                w = torch.randn(Co, C, k, k, device=dev)
                with ops.float_split(kind):
                    for v in (0, 1, 5, 6):
                        ops.CONV_VARIANT = v
                        outs.append(ops.float_conv2d(x, w, "binary", None, 1, p, 1).clone())
            elif kind == "int8":
                xq = nnDorefaQuant(4)((torch.rand(B, C, H, H, device=dev) * 1.2).contiguous(memory_format=torch.channels_last))
                conv = DorefaConv2d(C, Co, k, padding=p, bias=True, bit_width=1).to(dev).train()
                for v in (0, 1, 5, 6):
                    ops.CONV_VARIANT = v
                    with torch.no_grad():
                        outs.append(conv(xq).clone())
            else:
                conv = BinConv2d(C, Co, k, padding=p).to(dev).train()
                x = torch.where(torch.randn(B, C, H, H, device=dev) < 0, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
                for v in (0, 1, 5, 6):
                    ops.CONV_VARIANT = v
                    with torch.no_grad():
                        outs.append(conv(x).clone())
            assert all(torch.equal(outs[0], o) for o in outs[1:]), (kind, B, C, H, Co, k)
        assert _lib.call_counts["qt_conv2d_implicit_variant"] - before >= 3 * len(shapes)      # the variants did run
    finally:
        ops.CONV_VARIANT = 0


# ---- a whole training step as one hipGraph -----------------------------------------------------------------------------------------------

@pytest.mark.parametrize("which", ["resnet18", "resnet18_fused", "alexnet"])
def test_graphed_train_step_replays_the_eager_step(dev, which):
    """utils.GraphedTrainStep: forward + loss + backward captured once, replayed on a NEW batch — the loss and the parameter
    gradients are the eager step's on that batch, a second replay reproduces the first bit for bit, BatchNorm's running statistics
    keep advancing, and no host synchronisation is needed inside (range verdicts remembered for the capture only)."""
    import copy
    import bench_models
    from pytorch_quantize_impls_amd import utils
    torch.manual_seed(12)
    if which == "alexnet":
        model = bench_models.AlexNetBin()
        shape, fwd = (8, 3, 224, 224), (lambda m: m)
        loss_fn = lambda out, t: torch.nn.functional.nll_loss(out, t)                          # noqa: E731
    else:
        model = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
        shape = (32, 3, 32, 32)
        fwd = (lambda m: bench_models.TrainFusedDorefaResNet18(m)) if which.endswith("fused") else (lambda m: m)
        loss_fn = lambda out, t: torch.nn.functional.cross_entropy(out, t)                     # noqa: E731
    bench_models.randomize_bn(model, 5)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    eager = copy.deepcopy(model)
    x0 = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    x1 = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    t0, t1 = torch.randint(0, 10, (shape[0],), device=dev), torch.randint(0, 10, (shape[0],), device=dev)
    mode_before = _fused.DETECT_MODE
    net = fwd(model)

    class _Wrap(torch.nn.Module):             # the step sees ONE module whose parameters are the model's
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, x):
            return self.net(x)

    step = utils.GraphedTrainStep(_Wrap(), loss_fn, x0, t0)
    assert _fused.DETECT_MODE == mode_before
    rm_before = {k: b.clone() for k, b in model.named_buffers() if k.endswith("running_mean")}
    loss = step(x1, t1).clone()
    grads = {k: p_.grad.clone() for k, p_ in model.named_parameters()}
    prev = _fused.DETECT_MODE
    _fused.DETECT_MODE = "remember"
    try:
        loss_e = loss_fn(fwd(eager)(x1), t1)
        loss_e.backward()
    finally:
        _fused.DETECT_MODE = prev
    # torch's own kernels in the step (the fp32 stem conv, MIOpen's BatchNorm) may pick other algorithms under capture, and a
    # quantised net follows every flipped code: the same loss to 1e-3, gradients that point the same way, the classifier's to 1e-3
    assert abs(float(loss) - float(loss_e)) <= 1e-3 * abs(float(loss_e))
    last = [k for k, _ in eager.named_parameters()][-2]
    for k, p_ in eager.named_parameters():
        a, b = grads[k].double().flatten(), p_.grad.double().flatten()
        if float(b.norm()) <= 1e-6 * b.numel() ** 0.5:
            continue
        c = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
        assert c >= (0.999 if k == last else 0.8), (k, c)
    assert any(not torch.equal(rm_before[k], b) for k, b in model.named_buffers() if k in rm_before)
    loss_again = step(x1, t1)
    assert torch.equal(loss_again, loss)                                                       # replay is reproducible
    with pytest.raises(ValueError):
        step(x1[:4], t1[:4])


@pytest.mark.parametrize("Cout,Cin,k", [(64, 64, 3), (96, 33, 5), (40, 130, 1), (192, 3, 11)])
@pytest.mark.parametrize("kind", ["binary", "ternary", "sign", "raw"])
def test_conv_weight_pair_plane_in_one_kernel_equals_the_composed_form(dev, Cout, Cin, k, kind):
    """qt_f16x2_pack_conv_weight_f32 (quantiser + [flip / transpose] + tap-major layout + row padding in one pass over the weight where
    it lies) against the composition it replaces: permute -> contiguous -> qt_f16x2_pack_f32 -> zero-padded rows, for the forward
    operand and for grad_x's flipped, transposed operand; odd channel counts exercise the 16-byte tap granule."""
    torch.manual_seed(Cout + Cin + k)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.8
    if kind == "raw":
        w = torch.round(w * 5)
    for tf in (False, True):
        src = w.flip(2, 3).transpose(0, 1).contiguous() if tf else w
        rows, chans = int(src.shape[0]), int(src.shape[1])
        Cb = ops.triple_ld_bytes(chans, 16, 2)
        taps = ops.weight_bf16x3(src.permute(0, 2, 3, 1).contiguous().view(rows * k * k, chans), kind, ld_bytes=Cb, terms=2)
        want = taps.data.view(rows, k * k * Cb // 2)
        for wsrc in (w, w.contiguous(memory_format=torch.channels_last)):      # the weight of a channels_last model is strided so
            before = _lib.call_counts["qt_f16x2_pack_conv_weight_f32"]
            got = ops.pack_conv_weight_bf16x3(wsrc, kind, terms=2, transpose_flip=tf)
            assert _lib.call_counts["qt_f16x2_pack_conv_weight_f32"] == before + 1
            assert got.rows == rows and got.terms == 2 and got.data.shape[1] % 64 == 0
            assert torch.equal(got.data[:, :want.shape[1]], want)
            assert not bool(got.data[:, want.shape[1]:].any())                   # padding is zero


@pytest.mark.parametrize("Cout,Cin,k", [(64, 64, 3), (96, 33, 5), (40, 130, 1), (512, 512, 3)])
@pytest.mark.parametrize("ternary", [False, True])
def test_conv_weight_code_plane_in_one_kernel_equals_the_composed_form(dev, Cout, Cin, k, ternary):
    """qt_pack_conv_weight_codes_i8 against the composition it replaces (permute -> contiguous -> qt_weight_codes_i8 -> zero-padded
    rows): same bytes, zero padding between the taps' channel granules and behind the row."""
    torch.manual_seed(Cout + Cin + k)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.8
    Cb = ops.code_ld_bytes(Cin, 16)
    taps = ops.weight_codes(w.permute(0, 2, 3, 1).contiguous().view(Cout * k * k, Cin), ternary, ld_bytes=Cb)
    want = taps.codes.view(Cout, k * k * Cb)
    for wsrc in (w, w.contiguous(memory_format=torch.channels_last)):
        before = _lib.call_counts["qt_pack_conv_weight_codes_i8"]
        got = ops.pack_conv_weight_codes(wsrc, ternary)
        assert _lib.call_counts["qt_pack_conv_weight_codes_i8"] == before + 1
        assert got.rows == Cout and got.K == k * k * Cb
        assert torch.equal(got.codes[:, :want.shape[1]], want)
        assert not bool(got.codes[:, want.shape[1]:].any())


def test_weight_gradient_takes_the_layout_of_a_channels_last_parameter(dev):
    """The pixel-major reduce addresses dW and the STE mask's weight by the parameter's own strides: a channels_last model's
    (channels-last) weight is neither copied to contiguous nor is its gradient re-laid-out by autograd — same values."""
    torch.manual_seed(3)
    x = torch.where(torch.randn(4, 64, 12, 12, device=dev) < 0, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
    go = torch.randn(4, 96, 12, 12, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(96, 64, 3, 3, device=dev) * 0.8
    a = ops.conv2d_grad_weight_pm(x, go, (3, 3), 1, weight=w)
    b = ops.conv2d_grad_weight_pm(x, go, (3, 3), 1, weight=w.contiguous(memory_format=torch.channels_last))
    assert a.is_contiguous() and b.is_contiguous(memory_format=torch.channels_last) and not b.is_contiguous()
    assert torch.equal(a, b)
    assert bool((b[w.abs() > 1.001] == 0).all()) and bool((b != 0).any())
