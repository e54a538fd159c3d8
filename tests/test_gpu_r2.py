"""Round-2 GPU parity: the rows and shapes VERDICT r1 found unpinned.

  * XNOR activation quantiser (a18) and the deprecated functional forms (a11 / a17) against reference-generated
    vectors (tests/golden/make_golden_r2.py) and the oracle;
  * reference digests of one C5 and one C4 conv layer at their configured spatial size;
  * the direct 3x3 kernel against the ORACLE's restatement of conv -> BatchNorm -> sign at the C5 shapes it is
    dispatched for (224 x 224 and 112 x 112, batch 8), binary, ternary and the real-valued first layer;
  * the fused C5 / C4 / C3 networks LAYER BY LAYER at the configured image size, every block fed with the CPU
    chain's own intermediate and compared bit for bit with the CPU evaluation of the same folded expression;

Integer / bit results: exact.  Float tails: max|a-b| / max|b| <= 1e-5."""
import copy
import hashlib
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import same, norm_err

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import _lib, lazy, ops, packed, synth  # noqa: E402
from pytorch_quantize_impls_amd.functions import BinaryConnectDeterministic, _fused  # noqa: E402
from pytorch_quantize_impls_amd.layers import (BinConv2d, TerConv2d, LinearBin, LinearTer, FusedConvPoolBnSign,  # noqa: E402
                                               PackedMaxPool, FusedFeatureClassifier, fold_batchnorm)

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


def g(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def n(t):
    return t.detach().cpu().numpy()


_PM_F16 = {"qt_wgrad_pm_f32": "qt_wgrad_pm_f16", "qt_wgrad_pm_pack_grad_f32": "qt_wgrad_pm_pack_grad_f16x2",
           "qt_wgrad_pm_pack_act_f32": "qt_wgrad_pm_pack_act_f16", "qt_wgrad_pm_pack_act_s2d_f32": "qt_wgrad_pm_pack_act_s2d_f16x2"}


def pm_entry(name):
    """The pixel-major weight-gradient entry the configured split dispatches for a quantised activation: three bf16 planes
    (ops.FLOAT_SPLIT == 'bf16x3') or two fp16 planes ('f16x2', the default since round 3)."""
    return _PM_F16.get(name, name) if ops.split_terms() == 2 else name


class used:
    def __init__(self, *names):
        self.names = names

    def __enter__(self):
        self.before = dict(_lib.call_counts)

    def __exit__(self, *exc):
        if exc[0] is None:
            for k in self.names:
                assert _lib.call_counts[k] > self.before.get(k, 0), f"{k} did not run"


# ---- a18: XNOR activation quantiser -------------------------------------------------------------------------------

@pytest.mark.parametrize("dim", [-1, 0, 1])
def test_xnor_act_golden(dev, golden_r2, dim):
    from pytorch_quantize_impls_amd.functions.xnor_connect import QuantXnor, nnQuantXnor
    for tag in golden_r2["g10_cases"]:
        x, gout = golden_r2[f"g10_{tag}_x"], golden_r2[f"g10_{tag}_g"]
        xi = g(x, dev).requires_grad_(True)
        with used("qt_xnor_act_f32", "qt_xnor_act_backward_f32"):
            y = QuantXnor(xi, dim=dim)
            y.backward(g(gout, dev))
        assert norm_err(n(y), golden_r2[f"g10_{tag}_d{dim}_y"]) <= TOL, (tag, dim)
        assert np.array_equal(n(y) == 0, golden_r2[f"g10_{tag}_d{dim}_y"] == 0)
        assert norm_err(n(xi.grad), golden_r2[f"g10_{tag}_d{dim}_gx"]) <= TOL, (tag, dim)
        with torch.no_grad():
            assert same(n(nnQuantXnor(dim)(g(x, dev))), n(y))          # module form = functional form


@pytest.mark.parametrize("R,C", [(4096, 4096), (300, 7), (3, 10000), (1, 1)])
@pytest.mark.parametrize("dim", [-1, 0, 1])
def test_xnor_act_vs_oracle(dev, oracle, R, C, dim):
    x = synth.normal(R + C + dim, (R, C)) + np.float32(0.25)     # a non-zero mean, so the signed mean has something to carry
    x.reshape(-1)[::7] = 0.0
    gout = synth.normal(R * 3 + C, (R, C))
    y, mean = ops.xnor_act(g(x, dev), dim)
    yo, mo = oracle.xnor_act(x, dim)
    assert norm_err(n(mean), mo) <= TOL
    assert norm_err(n(y), yo) <= TOL and np.array_equal(n(y) == 0, yo == 0)
    gin = ops.xnor_act_backward(g(gout, dev), g(x, dev), mean, dim)
    assert norm_err(n(gin), oracle.xnor_act_backward(gout, x, dim)) <= TOL
    # strided rows
    big = torch.zeros((R, C + 5), device=dev)
    big[:, :C] = g(x, dev)
    y2, _ = ops.xnor_act(big[:, :C], dim)
    assert torch.equal(y2, y)


def test_xnor_act_empty_and_errors(dev):
    y, m = ops.xnor_act(torch.zeros((0, 5), device=dev), 1)
    assert y.shape == (0, 5)
    with pytest.raises(ValueError):
        ops.xnor_act(torch.zeros((2, 3, 4), device=dev), 1)
    with pytest.raises(_lib.QtStatusError):
        ops.xnor_act(torch.zeros((2, 3), device=dev), 2)


# ---- a11 / a17: functional forms on the device -----------------------------------------------------------------------

def _functional_ops():
    from pytorch_quantize_impls_amd.functions.binary_connect import BinaryConv2d
    from pytorch_quantize_impls_amd.functions.dorefa_connect import QuantConv2d, QuantDense
    from pytorch_quantize_impls_amd.functions.terner_connect import TernaryConv2d, TernaryDense
    return BinaryConv2d, QuantConv2d, QuantDense, TernaryConv2d, TernaryDense


@pytest.mark.parametrize("name", ["ter", "q1", "q3", "q32"])
def test_functional_dense_forms_golden(dev, golden_r2, name):
    _, _, QuantDense, _, TernaryDense = _functional_ops()
    op = {"ter": lambda: TernaryDense(stochastic=False), "q1": lambda: QuantDense(1), "q3": lambda: QuantDense(3),
          "q32": lambda: QuantDense(32)}[name]()
    for tag in golden_r2["g11_lin_cases"]:
        x, w, gout = (golden_r2[f"g11_lin_{tag}_{k}"] for k in ("x", "w", "g"))
        has_b = f"g11_lin_{tag}_b" in golden_r2.files
        xi, wi = g(x, dev).requires_grad_(True), g(w, dev).requires_grad_(True)
        bi = g(golden_r2[f"g11_lin_{tag}_b"], dev).requires_grad_(True) if has_b else None
        _fused.LIBRARY_PATHS.clear()
        with used("qt_bf16x6_pack_f32", "qt_bf16_gemm"):        # the contraction runs on the matrix cores, not in a library
            y = op.apply(xi, wi, bi) if has_b else op.apply(xi, wi)
        y.backward(g(gout, dev))
        # round 4: the hand-written backward (terner_connect.py:97-108, dorefa_connect.py:140-155) runs on the same routes
        assert not _fused.LIBRARY_PATHS, dict(_fused.LIBRARY_PATHS)
        assert norm_err(n(y), golden_r2[f"g11_lin_{tag}_{name}_y"]) <= TOL, (tag, name)
        assert norm_err(n(xi.grad), golden_r2[f"g11_lin_{tag}_{name}_gx"]) <= TOL
        assert norm_err(n(wi.grad), golden_r2[f"g11_lin_{tag}_{name}_gw"]) <= TOL
        if has_b:
            assert norm_err(n(bi.grad), golden_r2[f"g11_lin_{tag}_{name}_gb"]) <= TOL


@pytest.mark.parametrize("name", ["ter", "q1", "q3", "bin"])
def test_functional_conv_forms_golden(dev, golden_r2, name):
    BinaryConv2d, QuantConv2d, _, TernaryConv2d, _ = _functional_ops()
    for tag in golden_r2["g11_conv_cases"]:
        parts = dict((p[0], int(p[1:])) for p in str(tag).split("_")[:6])
        st, pd = parts["s"], parts["p"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            op = {"ter": lambda: TernaryConv2d(stochastic=False, stride=st, padding=pd), "q1": lambda: QuantConv2d(st, pd, bit_width=1),
                  "q3": lambda: QuantConv2d(st, pd, bit_width=3), "bin": lambda: BinaryConv2d(st, pd)}[name]()
        x, w, gout = (golden_r2[f"g11_conv_{tag}_{k}"] for k in ("x", "w", "g"))
        has_b = f"g11_conv_{tag}_b" in golden_r2.files
        xi, wi = g(x, dev).requires_grad_(True), g(w, dev).requires_grad_(True)
        bi = g(golden_r2[f"g11_conv_{tag}_b"], dev).requires_grad_(True) if has_b else None
        _fused.LIBRARY_PATHS.clear()
        with used("qt_conv2d_implicit"):
            y = op.apply(xi, wi, bi) if has_b else op.apply(xi, wi)
        y.backward(g(gout, dev))
        # round 4: grad_input on the split-gradient conv; the weight gradient of a REAL-valued many-channel activation has no
        # route of its own (two real operands) — the only dense-library call these forms may still make, and it is counted
        assert set(_fused.LIBRARY_PATHS) <= {"conv grad_weight outside the matrix-core route"}, dict(_fused.LIBRARY_PATHS)
        assert norm_err(n(y), golden_r2[f"g11_conv_{tag}_{name}_y"]) <= TOL, (tag, name)
        assert norm_err(n(xi.grad), golden_r2[f"g11_conv_{tag}_{name}_gx"]) <= TOL
        assert norm_err(n(wi.grad), golden_r2[f"g11_conv_{tag}_{name}_gw"]) <= TOL
        if has_b:
            assert norm_err(n(bi.grad), golden_r2[f"g11_conv_{tag}_{name}_gb"]) <= TOL


# ---- reference digests at configured spatial sizes -----------------------------------------------------------------

def test_c5_conv_reference_digest(dev, golden_hashes_r2):
    """TerConv2d 64 -> 64, 3x3, padding 1, 224 x 224 (VGG-16 conv2 of config C5) on +-1 input: SHA-256 of the int32 result
    the REFERENCE layer produced (terner_layers.py:89-92), train and eval mode, tagged and un-tagged input."""
    h = golden_hashes_r2["terconv_c5_64_64_224"]
    x = synth.pm1(h["x_seed"], (h["B"], h["Cin"], h["H"], h["H"]))
    w = synth.uniform(h["w_seed"], (h["Cout"], h["Cin"], 3, 3), h["w_lo"], h["w_hi"])
    conv = TerConv2d(h["Cin"], h["Cout"], 3, padding=1).to(dev)
    conv.weight.data.copy_(g(w, dev))
    conv.bias.data.zero_()
    for tagged in (False, True):
        xd = g(x, dev).contiguous(memory_format=torch.channels_last)
        if tagged:
            xd = BinaryConnectDeterministic.apply(xd)
        for training in (True, False):
            conv.train(training)
            with torch.no_grad(), used("qt_conv2d_implicit"):
                y = lazy.resolve(conv(xd))
            yi = n(y.contiguous()).astype(np.int32)          # NCHW order, as the reference's tensor
            assert hashlib.sha256(np.ascontiguousarray(yi).tobytes()).hexdigest() == h["sha256_int32"], (tagged, training)
            assert float(y.double().sum()) == h["sum"]
        conv.train(True)
        conv.weight.data.copy_(g(w, dev))


def test_c4_conv_integer_core_reference_digest(dev, golden_hashes_r2):
    """Integer core of a C4 layer (64 -> 64, 3x3, 32 x 32, batch 64): 4-bit activation codes x sign weights on the int8
    matrix cores with unit scale, against the digest of the reference's F.conv2d on the same integers
    (binary_layers.py:105, what DorefaConv2d(bit_width=1) contracts up to its scale E/15)."""
    h = golden_hashes_r2["w1a4_core_c4_64_64_32"]
    B, C, H = h["B"], h["Cin"], h["H"]
    codes = np.floor(synth.uniform(h["x_seed"], (B, C, H, H), 0.0, 16.0)).clip(0, 15).astype(np.float32)
    w = synth.uniform(h["w_seed"], (h["Cout"], C, 3, 3), h["w_lo"], h["w_hi"])
    xq = (g(codes, dev) / 15.0).contiguous(memory_format=torch.channels_last)     # rint(15 * fl(c / 15)) == c
    px, _ = ops.dorefa_codes(xq.permute(0, 2, 3, 1).reshape(B * H * H, C), 4, want_f32=False, ld_bytes=ops.code_ld_bytes(C, 16))
    assert torch.equal(px.codes[:, :C].to(torch.float32).view(B, H, H, C).permute(0, 3, 1, 2), g(codes, dev))
    wp = ops.pack_conv_weight_codes(g(w, dev))
    with used("qt_conv2d_implicit"):
        y2 = ops.conv2d_codes(px, (B, C, H, H), wp, (3, 3), 1.0, None, 1, 1, 1)
    y = y2.view(B, H, H, h["Cout"]).permute(0, 3, 1, 2).contiguous()
    assert hashlib.sha256(np.ascontiguousarray(n(y).astype(np.int32)).tobytes()).hexdigest() == h["sha256_int32"]


# ---- direct 3x3 kernel against the oracle chain at the dispatched C5 shapes -----------------------------------------

def _decode(act):
    """PackedActivation -> +-1 (0 for nibble zeros) fp32 tensor [N, C, H, W] on the host."""
    N, C, H, W = act.shape
    if act.planes is not None:
        words = act.planes.sign.view(N, H, W, -1).cpu().numpy().view(np.uint32)
        bits = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(N, H, W, -1)[..., :C]
        return torch.from_numpy(np.where(bits == 1, -1.0, 1.0).astype(np.float32)).permute(0, 3, 1, 2)
    hy, hx = act.halo
    words = act.nib.words.view(N, H + 2 * hy, W + 2 * hx, -1).cpu().numpy().view(np.uint32)
    nib = ((words[..., None] >> (4 * np.arange(8, dtype=np.uint32))) & 0xF).reshape(N, H + 2 * hy, W + 2 * hx, -1)
    val = np.select([nib == 0x2, nib == 0xA, nib == 0], [1.0, -1.0, 0.0], default=np.nan).astype(np.float32)
    assert not np.isnan(val).any(), "a nibble outside {0x0, 0x2, 0xA}"
    border = np.ones(val.shape[:3], dtype=bool)
    border[:, hy:hy + H, hx:hx + W] = False
    assert (val[border] == 0).all(), "the halo of a nibble plane must be zero"
    assert (val[:, :, :, C:] == 0).all(), "pad channels must be zero"
    return torch.from_numpy(val[:, hy:hy + H, hx:hx + W, :C]).permute(0, 3, 1, 2)


def _as_input(x_pm1: torch.Tensor, producer, dev):
    """The +-1 host tensor in the form the producing fused block would have handed over: bit planes, or the consumer's
    nibble plane with its zero halo."""
    N, C, H, W = x_pm1.shape
    xd = x_pm1.to(dev).contiguous(memory_format=torch.channels_last)
    bits = ops.sign_pack(xd.permute(0, 2, 3, 1).reshape(N * H * W, C))[0]
    halo = getattr(producer, "out_nib_halo", None) if producer is not None else None
    if halo is None:
        return packed.PackedActivation(bits, (N, C, H, W))
    nib = ops.bits_to_nib_pad(bits, N, H, W, tuple(halo), ld=ops.pixel_ld_nib(C))
    return packed.PackedActivation(None, (N, C, H, W), nib=nib, halo=tuple(halo))


@pytest.mark.parametrize("C,Cout,H,kind", [(64, 64, 224, "ternary"), (64, 128, 112, "binary"), (128, 128, 112, "ternary")])
def test_direct_conv3x3_vs_oracle_chain_at_c5_shapes(dev, oracle, C, Cout, H, kind):
    """qt_conv3x3_direct_nib at the shapes it is dispatched for in the fused VGG-16 (batch 8), threshold bits and the
    next conv's nibble halo plane, against oracle.bin_conv_pool_bn_sign_planes on the first and the last image."""
    N = 8
    x = synth.pm1(C + H, (N, C, H, H))
    w = synth.uniform(Cout + H, (Cout, C, 3, 3), -1.4, 1.4)
    b = synth.normal(7, (Cout,)) * 3
    alpha = synth.uniform(8, (Cout,), -0.3, 0.3)
    beta = synth.uniform(9, (Cout,), -4, 4)
    xd = g(x, dev).contiguous(memory_format=torch.channels_last)
    bits = ops.sign_pack(xd.permute(0, 2, 3, 1).reshape(N * H * H, C))[0]
    px = ops.bits_to_nib_pad(bits, N, H, H, (1, 1), ld=ops.pixel_ld_nib(C))
    wq = oracle.safe_sign(w) if kind == "binary" else oracle.ternarize(w)
    wp = ops.pack_conv_weight_nib(g(wq, dev), kind)
    al, be, bd = g(alpha, dev), g(beta, dev), g(b, dev)
    outs = []
    for epi in ((al, be), ops.NibEpilogue(al, be, (1, 1))):
        if not ops.direct_conv3x3_applicable(C, Cout, (3, 3), 1, 1, 1, (1, 1), epi):
            continue
        with used("qt_conv3x3_direct_nib"):
            out = ops.conv3x3_direct_nib(px, N, C, H, H, wp, bd, epi)
        outs.append(packed.PackedActivation(out, (N, Cout, H, H)) if isinstance(out, ops.BitPlanes)
                    else packed.PackedActivation(None, (N, Cout, H, H), nib=out, halo=(1, 1)))
    assert outs, "the direct kernel must take at least one output form of this shape"
    for img in (0, N - 1):
        want, (Ho, Wo) = oracle.bin_conv_pool_bn_sign_planes(x[img:img + 1], wq, b, 1, 1, 1, alpha, beta, 1, 1)
        wbits = ((want.reshape(Ho, Wo, -1)[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(Ho, Wo, -1)[..., :Cout]
        want_pm1 = np.where(wbits == 1, -1.0, 1.0).astype(np.float32).transpose(2, 0, 1)
        for act in outs:
            assert np.array_equal(_decode(act)[img].numpy(), want_pm1), (img, "nib" if act.nib is not None else "bits")


def test_direct_first_layer_vs_oracle_chain(dev, oracle):
    """The real-valued 3 -> 64 first layer at 224 x 224 (bf16 triple planes, direct kernel): its accumulators are fp32
    sums of real products, so a bit may differ from the oracle's double-accumulated chain only where the folded value is
    within float rounding of zero."""
    N, C, Cout, H = 4, 3, 64, 224
    x = synth.normal(31, (N, C, H, H))
    w = synth.uniform(32, (Cout, C, 3, 3), -1.4, 1.4)
    conv = TerConv2d(C, Cout, 3, padding=1).to(dev)
    conv.weight.data.copy_(g(w, dev)); conv.bias.data.copy_(g(synth.normal(33, (Cout,)), dev))
    conv.eval()
    conv.binary_input = False
    bn = torch.nn.BatchNorm2d(Cout).to(dev).eval()
    bn.running_mean.copy_(g(synth.normal(34, (Cout,)), dev)); bn.running_var.copy_(g(synth.uniform(35, (Cout,), 2, 30), dev))
    bn.weight.data.copy_(g(synth.normal(36, (Cout,)), dev)); bn.bias.data.copy_(g(synth.normal(37, (Cout,)), dev))
    alpha, beta = (n(t) for t in fold_batchnorm(bn))
    wq = oracle.ternarize(w)
    # both splits of the image: fp16 pair pixels (the default since round 4: two taps per MFMA) and the exact bf16 triples
    # (round 6: under the two-term split a 3 -> 64 first layer takes the one-pass kernel qt_conv3x3_first_f32; the pair-plane
    #  direct kernel it replaced is pinned by switching the new route off)
    for mode, entry in (("f16x2", "qt_conv3x3_first_f32"), ("f16x2-pairs", "qt_conv3x3_direct_pairs"), ("bf16x3", "qt_conv3x3_direct_nib")):
        blk = FusedConvPoolBnSign(conv, bn)
        with torch.no_grad(), ops.float_split(mode.split("-")[0]), ops.scope(FIRST_3X3=(mode != "f16x2-pairs")), used(entry):
            act = blk(g(x, dev).contiguous(memory_format=torch.channels_last))
        got = _decode(act).numpy()
        for img in (0, N - 1):
            yconv = oracle.conv2d(x[img:img + 1], wq, n(conv.bias), 1, 1)[0].astype(np.float64)
            v = yconv * alpha.astype(np.float64)[:, None, None] + beta.astype(np.float64)[:, None, None]
            want = np.where(v < 0, -1.0, 1.0)
            diff = got[img] != want
            scale = np.abs(v).mean()
            assert diff.mean() <= 1e-4, (mode, diff.mean())
            assert np.all(np.abs(v[diff]) <= 1e-5 * scale), (mode, float(np.abs(v[diff]).max() / scale))   # ties only


# ---- fused networks, layer by layer at the configured image size ----------------------------------------------------

def _unit_scale_weights(model, seed):
    gen = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            m.weight.data.copy_(torch.empty_like(m.weight).uniform_(-1.2, 1.2, generator=gen))
            if m.bias is not None:
                m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen))


def _folded_sign(acc_plus_bias: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """The fused blocks' expression on the host, in fp32 with two roundings: fl(fl(t * alpha) + beta) < 0 -> -1."""
    shape = (1, -1, 1, 1) if acc_plus_bias.dim() == 4 else (1, -1)
    v = (acc_plus_bias * alpha.view(shape)) + beta.view(shape)
    return torch.where(v < 0, -1.0, 1.0), v


def _check_block_vs_modules(pm1_folded, v_folded, module_out_pm1, name):
    """Folded form vs the un-fused module chain (BatchNorm's own arithmetic): they may disagree only on ties."""
    diff = pm1_folded != module_out_pm1
    if diff.any():
        scale = float(v_folded.abs().mean())
        assert float(diff.float().mean()) <= 1e-5, (name, float(diff.float().mean()))
        assert float(v_folded[diff].abs().max()) <= 1e-5 * scale, (name, float(v_folded[diff].abs().max()) / scale)


def test_c5_fused_vgg16_layerwise_at_224(dev):
    """Config C5's network at 3 x 224 x 224: every fused block of FusedFeatureClassifier (threshold-bit convs incl. the
    direct 3x3 kernels, pools on bits, nibble hand-overs) on the GPU against the host evaluation of the same folded
    expression, each block fed with the HOST chain's intermediate in the form its producer would have handed over."""
    import bench_models
    torch.manual_seed(5)
    model = bench_models.TernaryVGG16(num_classes=1000, image=224, fc=4096)
    _unit_scale_weights(model, 8)
    bench_models.randomize_bn(model, seed=5)
    model.eval()                                          # TerConv2d / LinearTer now hold their ternarised weights
    gmodel = copy.deepcopy(model).to(dev).to(memory_format=torch.channels_last)
    gmodel.features[0].binary_input = False
    fused = FusedFeatureClassifier(gmodel.features, gmodel.classifier, (512, 7, 7))
    blocks = list(fused.features.children())
    assert sum(isinstance(b, FusedConvPoolBnSign) for b in blocks) == 13 and sum(isinstance(b, PackedMaxPool) for b in blocks) == 5
    N = 2
    x = torch.randn(N, 3, 224, 224)
    convs = [m for m in model.features if isinstance(m, TerConv2d)]
    bns = [m for m in model.features if isinstance(m, torch.nn.BatchNorm2d)]
    before = dict(_lib.call_counts)
    cur, prev, ci = x, None, 0
    with torch.no_grad():
        for bi, blk in enumerate(blocks):
            if isinstance(blk, PackedMaxPool):
                want = F.max_pool2d(cur, 2, 2)
                got = _decode(blk(_as_input(cur, prev, dev)))
                assert torch.equal(got, want), f"pool block {bi}"
                cur, prev = want, blk
                continue
            conv, bn = convs[ci], bns[ci]
            acc = F.conv2d(cur, conv.weight, conv.bias, padding=1)          # exact integers + bias for +-1 inputs
            alpha, beta = (t.cpu() for t in fold_batchnorm(blk.bn))
            want, v = _folded_sign(acc, alpha, beta)
            inp = x.to(dev).contiguous(memory_format=torch.channels_last) if ci == 0 else _as_input(cur, prev, dev)
            got = _decode(blk(inp))
            if ci == 0:      # real-valued input: fp32 accumulation order differs -> ties only
                diff = got != want
                assert float(diff.float().mean()) <= 1e-4 and (not diff.any() or float(v[diff].abs().max()) <= 1e-5 * float(v.abs().mean()))
            else:
                assert torch.equal(got, want), f"conv block {bi} ({conv.in_channels}->{conv.out_channels})"
            module_out = torch.where(F.hardtanh(bn(conv(cur))) < 0, -1.0, 1.0)
            _check_block_vs_modules(want, v, module_out, f"block {bi}")
            cur, prev, ci = want, blk, ci + 1
        # classifier: fused FC blocks on the flattened planes vs the host modules
        flat_in = _as_input(cur, None, dev).flatten_hwc()
        # block by block like the features: the FC sums are exact integers (+ bias), the BatchNorm1d -> Hardtanh -> sign blocks may
        # differ from the host modules only at ties (counted; |BatchNorm output| <= 1e-5 of its mean magnitude), and the device's
        # bits are carried forward on both sides, so the logits are compared on identical activations
        from pytorch_quantize_impls_amd.layers import FusedPoolBnSign
        cmods = list(model.classifier.children())
        h_cpu, h_gpu, ci, flips = cur.reshape(N, -1), flat_in, 0, 0
        for blk in fused.classifier.children():
            if isinstance(blk, FusedPoolBnSign):
                v = cmods[ci](h_cpu)                                   # BatchNorm1d on the host
                ci += 3                                                # (BatchNorm1d, Hardtanh, BinaryConnect)
                out = blk(h_gpu)
                words = out.planes.sign.cpu().numpy().view(np.uint32)
                bits = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(N, -1)[:, :v.shape[1]]
                got = torch.from_numpy(np.where(bits == 1, -1.0, 1.0).astype(np.float32))
                diff = got != torch.where(v < 0, -1.0, 1.0)
                flips += int(diff.sum())
                assert not diff.any() or float(v[diff].abs().max()) <= 1e-5 * float(v.abs().mean()), "a flip that is not a tie"
                h_cpu, h_gpu = got, out
            else:
                y_cpu, y_gpu = cmods[ci](h_cpu), blk(h_gpu).cpu()
                ci += 1
                assert norm_err(y_gpu.numpy(), y_cpu.numpy()) <= 1e-5
                h_cpu, h_gpu = y_cpu, blk(h_gpu)
        assert ci == len(cmods) and flips <= 4, flips
    used_ = {k: v - before.get(k, 0) for k, v in _lib.call_counts.items() if v - before.get(k, 0)}
    # conv1 (real input): the one-pass first-layer kernel; conv2, conv3: the direct 3 x 3 kernel on nibble halo planes
    assert used_.get("qt_conv3x3_first_f32", 0) >= 1 and used_.get("qt_conv3x3_direct_nib", 0) >= 2, used_


def test_c3_fused_alexnet_layerwise(dev):
    """Config C3's network (batch 4 of 3 x 224 x 224): the fused AlexNet-Bin feature blocks (conv -> MaxPool ->
    BatchNorm -> Hardtanh -> sign as threshold bits + pooling on bits) one by one against the host's folded chain."""
    import bench_models
    torch.manual_seed(6)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model, seed=6)
    model.eval()
    gmodel = copy.deepcopy(model).to(dev).to(memory_format=torch.channels_last)
    fusedm = bench_models.FusedAlexNetBin(gmodel)
    blocks = [b for b in fusedm.net.features.children()]
    convs = [m for m in model.features if isinstance(m, BinConv2d)]
    N = 4
    x = torch.randn(N, 3, 224, 224)
    cur, prev = x, None
    with torch.no_grad():
        for bi, blk in enumerate(blocks):
            assert isinstance(blk, FusedConvPoolBnSign), type(blk)
            conv = convs[bi]
            acc = F.conv2d(cur, conv.weight, conv.bias, conv.stride, conv.padding)
            if blk._pool.pool_k != 1 or blk._pool.pool_s != 1:
                # the reference order: MaxPool, then the (monotone) BatchNorm map, then the sign
                # (models/Alexnet/Alexnet_Bin.py:13-17); the kernel pools the threshold bits instead (AND / OR by the
                # sign of alpha), which is the same function of the window
                acc = F.max_pool2d(acc, blk._pool.pool_k, blk._pool.pool_s)
            alpha, beta = (t.cpu() for t in fold_batchnorm(blk.bn))
            want, v = _folded_sign(acc, alpha, beta)
            inp = x.to(dev).contiguous(memory_format=torch.channels_last) if bi == 0 else _as_input(cur, prev, dev)
            out = blk(inp)
            got = _decode(out) if len(out.shape) == 4 else None
            if got is None:        # the last block flattens for the classifier: undo (h, w, c)
                Cc, Hh, Ww = want.shape[1:]
                words = out.planes.sign.cpu().numpy().view(np.uint32)
                bits = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(N, -1)[:, :Hh * Ww * Cc]
                got = torch.from_numpy(np.where(bits == 1, -1.0, 1.0).astype(np.float32)).view(N, Hh, Ww, Cc).permute(0, 3, 1, 2)
            if bi == 0:
                diff = got != want
                assert float(diff.float().mean()) <= 1e-4 and (not diff.any() or float(v[diff].abs().max()) <= 1e-5 * float(v.abs().mean()))
            else:
                assert torch.equal(got, want), f"block {bi}"
            cur, prev = want, blk


def test_c4_fused_dorefa_resnet18_layerwise(dev, oracle):
    """Config C4's network (3 x 32 x 32, batch 32): every code-epilogue conv of the fused DoReFa ResNet-18 against the
    oracle's restatement of BatchNorm -> (+ shortcut) -> ReLU -> nnDorefaQuant(4) applied to the host's exact integer
    conv result, each conv fed with the HOST chain's codes (as the halo plane its producer writes)."""
    import bench_models
    torch.manual_seed(4)
    model = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(model, seed=3)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.mul_(4.0)
    model.eval()
    gmodel = copy.deepcopy(model).to(dev).to(memory_format=torch.channels_last)
    fusedm = bench_models.FusedDorefaResNet18(gmodel, a_bits=4)
    N = 32
    x = torch.randn(N, 3, 32, 32)

    inv_n = np.float32(ops.inv_levels(4))

    def codes_act(q: torch.Tensor, halo: int):
        """host integer codes [N, C, H, W] -> the CodeActivation its producer writes: NHWC int8 plane with a zero halo"""
        Nn, C, H, W = q.shape
        qd = (q.to(dev) * float(inv_n)).contiguous(memory_format=torch.channels_last)      # rint(15 * fl(c / 15)) == c
        planes, _ = ops.dorefa_codes(qd.permute(0, 2, 3, 1).reshape(Nn * H * W, C), 4, want_f32=False, ld_bytes=ops.code_ld_bytes(C, 16))
        ld = int(planes.codes.shape[1])
        full = torch.zeros((Nn, H + 2 * halo, W + 2 * halo, ld), dtype=torch.int8, device=dev)
        full[:, halo:halo + H, halo:halo + W] = planes.codes.view(Nn, H, W, ld)
        pl = ops.CodePlanes(codes=full.view(-1, ld), rows=Nn * (H + 2 * halo) * (W + 2 * halo), K=planes.K, inv_n=planes.inv_n,
                            bit_width=planes.bit_width, overflow=planes.overflow)
        return packed.CodeActivation(pl, (Nn, C, H, W), halo=(halo, halo))

    def decode_codes(act):
        a = act.without_halo()
        Nn, C, H, W = a.shape
        return a.codes.codes.view(Nn, H, W, -1)[..., :C].permute(0, 3, 1, 2).to(torch.float32).cpu()

    def expected_codes(conv_out_fp32, bn, residual_fp32=None):
        """oracle.affine_relu_dorefa_codes (fp32 steps, channel = dim 1) on the host's fp32 conv image, k = 4"""
        alpha, beta = (n(t) for t in fold_batchnorm(bn))
        q, _ = oracle.affine_relu_dorefa_codes(conv_out_fp32.numpy(), alpha, beta, 4, relu=True,
                                               res=None if residual_fp32 is None else residual_fp32.numpy())
        assert np.abs(q).max() <= 127
        return torch.from_numpy(q.astype(np.float32))

    def conv_image(codes, conv):
        """fl(acc * fl(inv_n * E)) — what the int8 conv stores: exact integer accumulators, ONE scale multiply"""
        E = np.float32(conv.weight.abs().amax().item())          # eval-mode weights hold sign(W) * E
        acc = F.conv2d(codes, torch.sign(conv.weight), None, conv.stride, conv.padding)
        assert float(acc.abs().max()) < 2 ** 24
        return acc * torch.tensor(inv_n * E)

    with torch.no_grad():
        # stem: fp32 library conv + one fused quantiser pass; MIOpen and ATen sum in different orders -> rint ties only
        q_host = expected_codes(model.stem(x), model.bn)
        q_gpu = decode_codes(fusedm.q0(gmodel.stem(x.to(dev).contiguous(memory_format=torch.channels_last))))
        d = q_gpu != q_host
        assert float(d.float().mean()) <= 1e-4 and (not d.any() or float((q_gpu - q_host).abs().max()) <= 1)
        cur = q_host
        before = dict(_lib.call_counts)
        for bi, (blk, fblk) in enumerate(zip(model.blocks, fusedm.blocks)):
            a_in = codes_act(cur, 1)
            q1_host = expected_codes(conv_image(cur, blk.conv1), blk.bn1)
            assert torch.equal(decode_codes(fblk.c1(a_in)), q1_host), f"block {bi} conv1"
            y2 = conv_image(q1_host, blk.conv2)
            if blk.shortcut is None:
                res_host = cur * torch.tensor(inv_n)             # identity shortcut: the block input's fp32 image fl(inv_n * q)
                res_gpu, sc_bn = a_in, None
            else:
                ys = conv_image(cur, blk.shortcut[0])
                al_s, be_s = (t.cpu() for t in fold_batchnorm(blk.shortcut[1]))
                res_host = ys * al_s.view(1, -1, 1, 1) + be_s.view(1, -1, 1, 1)
                res_gpu, sc_bn = fblk.sc_conv(a_in), fblk.sc_bn
                assert torch.equal(res_gpu.cpu(), ys), f"block {bi} shortcut conv"      # integer core x one scale: exact
            q2_host = expected_codes(y2, blk.bn2, res_host)
            q2_gpu = decode_codes(fblk.c2(codes_act(q1_host, 1), residual=res_gpu, residual_bn=sc_bn))
            assert torch.equal(q2_gpu, q2_host), f"block {bi} conv2"
            cur = q2_host
    ran = _lib.call_counts["qt_conv2d_implicit_codes"] - before.get("qt_conv2d_implicit_codes", 0)
    assert ran == 16, ran


# ---- one-launch linear forward ------------------------------------------------------------------------------------------


# ---- boundary / host-logic hardening ----------------------------------------------------

def test_inference_mode_binary_and_dorefa_chains(dev):
    """torch.inference_mode() tensors carry no version counter: the quantisers must not tag them (and not crash); the next
    layer then re-derives the planes from the fp32 values — same numbers as under no_grad."""
    from pytorch_quantize_impls_amd.functions import BinaryConnect, nnDorefaQuant
    from pytorch_quantize_impls_amd.layers import LinearDorefa
    torch.manual_seed(11)
    x = torch.randn(64, 256, device=dev)
    lin = LinearBin(256, 96).to(dev).eval()
    dlin = LinearDorefa(256, 40, bit_width=1).to(dev).eval()
    bc, dq = BinaryConnect(), nnDorefaQuant(4)
    with torch.no_grad():
        want_b = lin(bc(x))
        want_d = dlin(dq(torch.relu(x) * 0.3))
    with torch.inference_mode():
        got_b = lin(bc(x))
        got_d = dlin(dq(torch.relu(x) * 0.3))
    assert torch.equal(got_b, want_b)
    assert norm_err(n(got_d), n(want_d)) <= TOL


def test_eval_weight_off_grid_follows_the_reference(dev):
    """The reference's eval forward multiplies by whatever `weight` holds (binary_layers.py:46).  After
    model.eval(); load_state_dict(float checkpoint) the device layers must do the same instead of re-quantising."""
    from pytorch_quantize_impls_amd.functions import _fused
    from pytorch_quantize_impls_amd.layers import LinearDorefa, DorefaConv2d
    torch.manual_seed(12)
    x = torch.where(torch.rand(32, 128, device=dev) < 0.5, -1.0, 1.0)
    xc = torch.randn(2, 16, 9, 9, device=dev)
    for mk, xin in ((lambda: LinearBin(128, 24), x), (lambda: LinearTer(128, 24), x), (lambda: LinearDorefa(128, 24, bit_width=1), x),
                    (lambda: LinearDorefa(128, 24, bit_width=3), x), (lambda: BinConv2d(16, 8, 3, padding=1), xc),
                    (lambda: TerConv2d(16, 8, 3, padding=1), xc), (lambda: DorefaConv2d(16, 8, 3, padding=1, bit_width=1), xc)):
        layer = mk().to(dev).eval()
        with torch.no_grad():
            y_grid = lazy.resolve(layer(xin))                       # on-grid eval weight: packed / int8 / bf16 paths
        ckpt = {k: torch.randn_like(v) * 0.7 for k, v in layer.state_dict().items()}
        layer.load_state_dict(ckpt)                                 # float weights while in eval mode
        before = sum(_fused.LIBRARY_PATHS.values())
        with torch.no_grad():
            y = layer(xin)
            fn = torch.nn.functional.linear if xin.dim() == 2 else (lambda a, w, b: torch.nn.functional.conv2d(a, w, b, padding=1))
            want = fn(xin, layer.weight, layer.bias)
        assert torch.equal(y, want), type(layer).__name__
        assert not torch.equal(y, y_grid)
        assert sum(_fused.LIBRARY_PATHS.values()) == before + 1     # ... and the detour is counted, not silent


def test_half_precision_models_take_the_torch_expression(dev):
    """A device model in half precision is outside the fp32 kernels: forward and backward run the reference expression in
    torch (no TypeError), and the detour is counted."""
    from pytorch_quantize_impls_amd.functions import _fused
    lin = LinearBin(64, 16).to(dev).half()
    x = torch.randn(8, 64, device=dev, dtype=torch.float16, requires_grad=True)
    before = sum(_fused.LIBRARY_PATHS.values())
    y = lin(x)
    y.float().sum().backward()
    assert y.dtype == torch.float16 and lin.weight.grad is not None and x.grad is not None
    assert sum(_fused.LIBRARY_PATHS.values()) > before


def test_steady_state_forwards_do_not_synchronise(dev):
    """Content-dependent routing: a "not +-1" / "codes overflow" answer is remembered in every mode, so the un-fused eval
    forward of AlexNet-Bin (real-valued conv1, every other activation tagged by BinaryConnect) enqueues without a single
    host synchronisation by default (torch's sync debug mode raises on any); with DETECT_MODE = "remember" so do a
    pool-after-sign ternary stack (un-tagged +-1 inputs) and the module-by-module DoReFa ResNet-18 (code range flags)."""
    import bench_models
    from pytorch_quantize_impls_amd.functions import _fused
    torch.manual_seed(13)
    alex = bench_models.AlexNetBin()
    bench_models.randomize_bn(alex)
    alex = alex.to(dev).to(memory_format=torch.channels_last).eval()
    vgg = bench_models.TernaryVGG16(num_classes=10, image=32, fc=128)
    bench_models.randomize_bn(vgg, seed=2)
    vgg = vgg.to(dev).to(memory_format=torch.channels_last).eval()
    res = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(res, seed=3)
    res = res.to(dev).to(memory_format=torch.channels_last).eval()
    xa = torch.randn(8, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    xv = torch.randn(8, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        first = alex(xa)                                         # first forward: conv1 asks once (a host sync)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            second = alex(xa)                                    # default mode: nothing synchronises any more
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert torch.equal(second, first)
        _fused.DETECT_MODE = "remember"
        try:
            outs = [vgg(xv), res(xv)]                            # verdicts are asked (host syncs allowed)
            torch.cuda.synchronize()
            stats = dict(_fused.DETECT_STATS)
            torch.cuda.set_sync_debug_mode("error")
            try:
                again = [vgg(xv), res(xv)]
            finally:
                torch.cuda.set_sync_debug_mode("default")
        finally:
            _fused.DETECT_MODE = "verify"
    assert _fused.DETECT_STATS["sync"] == stats.get("sync", 0)       # nothing was asked again
    assert _fused.DETECT_STATS["cached"] > stats.get("cached", 0)    # ... the remembered verdicts were used
    for a, b in zip(outs, again):
        assert torch.equal(a, b)


def test_broken_pm1_assumption_poisons_instead_of_lying(dev):
    """DETECT_MODE = "remember": verdict "+-1" remembered, then a real-valued tensor arrives un-tagged: the packed route's
    result must be NaN (the device flag rides in through the bias), never a plausible wrong number; reset_detection()
    re-asks and recovers.  In the default mode the same sequence is simply correct (a sync per call)."""
    from pytorch_quantize_impls_amd.functions import _fused
    lin = LinearBin(256, 32).to(dev).eval()
    xpm = torch.where(torch.rand(16, 256, device=dev) < 0.5, -1.0, 1.0)
    xre = torch.randn(16, 256, device=dev)
    with torch.no_grad():
        want = torch.nn.functional.linear(xre, lin.weight, lin.bias)
        lin(xpm)
        assert norm_err(n(lin(xre)), n(want)) <= TOL           # default mode: re-verified, general route taken
        _fused.reset_detection(lin.weight)
    _fused.DETECT_MODE = "remember"
    try:
        _remember_mode_poison_body(lin, xpm, xre, _fused)
    finally:
        _fused.DETECT_MODE = "verify"


def _remember_mode_poison_body(lin, xpm, xre, _fused):
    with torch.no_grad():
        y0 = lin(xpm)                       # asks, remembers "+-1"
        y1 = lin(xpm.clone())               # cached verdict + device flag (clean)
        assert torch.equal(y0, y1) and not torch.isnan(y1).any()
        bad = lin(xre)
        assert torch.isnan(bad).all()
        _fused.reset_detection(lin.weight)
        good = lin(xre)
        assert norm_err(n(good), n(torch.nn.functional.linear(xre, lin.weight, lin.bias))) <= TOL


def test_fused_blocks_refold_after_load_state_dict(dev):
    """The folded BatchNorm of a fused block is keyed on the BatchNorm tensors' version counters: load_state_dict() after
    the first forward must change the output like a freshly built block."""
    conv = BinConv2d(32, 40, 3, padding=1).to(dev).eval()
    bn = torch.nn.BatchNorm2d(40).to(dev).eval()
    blk = FusedConvPoolBnSign(conv, bn)
    x = torch.where(torch.rand(2, 32, 10, 10, device=dev) < 0.5, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        a0 = blk(x).planes.sign.clone()
        sd = {k: (torch.randn_like(v) if v.dtype.is_floating_point else v) for k, v in bn.state_dict().items()}
        sd["running_var"] = sd["running_var"].abs() + 0.5
        bn.load_state_dict(sd)
        a1 = blk(x).planes.sign
        fresh = FusedConvPoolBnSign(conv, bn)(x).planes.sign
    assert torch.equal(a1, fresh) and not torch.equal(a1, a0)


def test_log_quantiser_outside_the_kernel_window(dev):
    """A 32-level-bit Log quantiser is outside qt_log_quantize_f32's parameter window: the torch expression runs on the
    device (as the reference would) instead of raising."""
    from pytorch_quantize_impls_amd.functions.log_lin_connect import LogQuant
    x = torch.randn(100, device=dev)
    y = LogQuant(fsr=3, bit_width=20).apply(x)
    assert torch.equal(y, LogQuant(fsr=3, bit_width=20).apply(x.cpu()).to(dev))


# ---- shift-based batch norm + backward convs on the matrix cores -----------------------------------------------------

def test_shift_batch_kernel_golden_and_oracle(dev, golden_r2, oracle):
    from pytorch_quantize_impls_amd.functions.binary_connect import ShiftBatch
    from pytorch_quantize_impls_amd.layers import ShiftNormBatch1d, ShiftNormBatch2d
    for tag in golden_r2["g13_cases"]:
        x, mean, var, w, b = (golden_r2[f"g13_{tag}_{k}"] for k in ("x", "mean", "var", "w", "b"))
        with used("qt_shift_batch_f32"):
            y = ShiftBatch.apply(g(x, dev), g(mean, dev), g(var, dev), g(w, dev), g(b, dev), 1e-4)
        assert norm_err(n(y), golden_r2[f"g13_{tag}_y"]) <= TOL, tag
        yo, normo = oracle.shift_batch(x, mean, var, w, b, 1e-4)
        assert norm_err(n(y), yo) <= TOL
    for name, cls in (("bn1d", ShiftNormBatch1d), ("bn2d", ShiftNormBatch2d)):
        x = golden_r2[f"g13_{name}_x"]
        m = cls(x.shape[1]).to(dev)
        m.weight.data.copy_(g(golden_r2[f"g13_{name}_w"], dev)); m.bias.data.copy_(g(golden_r2[f"g13_{name}_b"], dev))
        xi = g(x, dev).requires_grad_(True)
        with used("qt_shift_batch_f32"):
            y = m(xi)
        assert norm_err(n(y), golden_r2[f"g13_{name}_y"]) <= TOL, name
        y.sum().backward()                                     # the reference's backward expressions, on saved device tensors
        assert xi.grad is not None and torch.isfinite(xi.grad).all()
    big = torch.randn(4096, 4096, device=dev)
    p = [torch.randn(4096, device=dev), torch.rand(4096, device=dev) + 0.1, torch.randn(4096, device=dev), torch.randn(4096, device=dev)]
    yb = ShiftBatch.apply(big, *p, 1e-5)
    yo, _ = oracle.shift_batch(n(big)[:64], *(n(t) for t in p), 1e-5)
    assert norm_err(n(yb)[:64], yo) <= TOL


@pytest.mark.parametrize("Cin,Cout,k,pd,H,B,kind", [(192, 576, 5, 2, 27, 16, "binary"), (576, 1152, 3, 1, 13, 16, "ternary"),
                                                      (128, 96, 3, 1, 20, 5, "binary"), (120, 40, 3, 0, 9, 3, "ternary")])
def test_conv_backward_on_matrix_cores_vs_fp64(dev, Cin, Cout, k, pd, H, B, kind):
    """Backward of a training-mode BinConv2d / TerConv2d on a BinaryConnect-tagged activation at AlexNet conv2 / conv3
    shapes: grad_input = the exact-split conv of the gradient with the flipped quantised weight, grad_weight = the
    pixel contraction as batched K-major GEMMs (csrc/wgrad.hip), both on the bf16 matrix cores, against the fp64 evaluation of
    torch.nn.grad.conv2d_input / conv2d_weight (functions/binary_connect.py:141-143) and the STE mask."""
    from pytorch_quantize_impls_amd.functions import _fused
    torch.manual_seed(Cin + Cout)
    cls = BinConv2d if kind == "binary" else TerConv2d
    conv = cls(Cin, Cout, k, padding=pd).to(dev)
    conv.weight.data.uniform_(-1.3, 1.3)
    xr = torch.randn(B, Cin, H, H, device=dev, requires_grad=True)
    xs = BinaryConnectDeterministic.apply(xr.contiguous(memory_format=torch.channels_last))
    xs.retain_grad()
    before = dict(_lib.call_counts)
    lib_before = dict(_fused.LIBRARY_PATHS)
    y = conv(xs)
    gout = torch.randn_like(y)
    old_min, old_pix = _fused.BWD_MFMA_MIN_MACS, ops.WEIGHT_GRAD_MAX_PIXELS
    _fused.BWD_MFMA_MIN_MACS, ops.WEIGHT_GRAD_MAX_PIXELS = 0, 1 << 30      # exercise the routes at every test shape
    try:
        y.backward(gout)
    finally:
        _fused.BWD_MFMA_MIN_MACS, ops.WEIGHT_GRAD_MAX_PIXELS = old_min, old_pix
    used_now = {k: v - before.get(k, 0) for k, v in _lib.call_counts.items()}
    # forward + grad_input on the implicit conv; grad_weight on the pixel-major kernel (3 x 3 / 5 x 5), else as the batched
    # K-major GEMM, else the swapped conv
    pm_route = ops.wgrad_pm_applicable(xs.shape, y.shape, (k, k), 1, 1)
    gemm_route = not pm_route and ops.wgrad_gemm_applicable(xs.shape, y.shape, (k, k), 1, 1)
    assert pm_route
    assert used_now.get(pm_entry("qt_wgrad_pm_f32"), 0) == (1 if pm_route else 0)
    assert used_now.get("qt_bf16_gemm_taps", 0) == (1 if gemm_route else 0)
    assert used_now.get("qt_conv2d_implicit", 0) >= (2 if (gemm_route or pm_route) else 3)
    assert dict(_fused.LIBRARY_PATHS) == lib_before                                                 # no dense-library detour
    wq = (torch.where(conv.weight < 0, -1.0, 1.0) if kind == "binary" else ops.ternarize(conv.weight.detach())).double()
    gi = torch.nn.grad.conv2d_input(xs.shape, wq, gout.double(), padding=pd)
    gw = torch.nn.grad.conv2d_weight(xs.detach().double(), conv.weight.shape, gout.double(), padding=pd)
    gw = torch.where(conv.weight.detach().abs() > 1.001, torch.zeros_like(gw), gw)
    assert norm_err(n(xs.grad), gi.cpu().numpy()) <= TOL
    assert norm_err(n(conv.weight.grad), gw.cpu().numpy()) <= TOL
    assert norm_err(n(conv.bias.grad), gout.double().sum((0, 2, 3)).cpu().numpy()) <= TOL


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,p,cl", [(3, 5, 7, 9, 11, 3, 1, False), (4, 32, 48, 13, 15, 3, 1, True),
                                                    (2, 16, 8, 11, 13, 5, 2, True), (5, 64, 64, 17, 19, 3, 0, False),
                                                    (3, 24, 40, 8, 10, 1, 0, True), (2, 8, 8, 12, 14, 7, 3, False),
                                                    (7, 130, 70, 6, 33, 3, 1, True)])
def test_weight_gradient_gemm_vs_fp64(dev, N, Cin, Cout, H, W, k, p, cl):
    """ops.conv2d_grad_weight_gemm (exact bf16 split of the gradient x +-1 / 0 activation, K-major position planes, one
    batched GEMM launch + reduce with the STE mask) against torch.nn.grad.conv2d_weight in fp64: odd channel counts,
    both memory formats, kernels 1 .. 7, no padding, ternary activations, and batch chunking with a ragged last chunk."""
    g = torch.Generator(device=dev)
    g.manual_seed(N * 1000 + Cin)
    x = torch.randint(-1, 2, (N, Cin, H, W), generator=g, device=dev).float()             # -1 / 0 / +1
    Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    go = torch.randn((N, Cout, Ho, Wo), device=dev, generator=g)
    if cl:
        x, go = x.contiguous(memory_format=torch.channels_last), go.contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, k, k), device=dev, generator=g) * 0.8
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), go.double(), stride=1, padding=p)
    with used("qt_wgrad_pack_grad_f32", "qt_wgrad_pack_act_f32", "qt_bf16_gemm_taps", "qt_wgrad_reduce_f32"):
        got = ops.conv2d_grad_weight_gemm(x, go, (k, k), p)
    assert norm_err(n(got), ref.cpu().numpy()) <= TOL
    masked = ops.conv2d_grad_weight_gemm(x, go, (k, k), p, weight=w)
    refm = torch.where(w.abs() <= 1.001, ref, torch.zeros_like(ref))
    assert norm_err(n(masked), refm.cpu().numpy()) <= TOL
    assert np.array_equal(n(masked) == 0, (refm == 0).cpu().numpy() | (n(masked) == 0))
    old = ops.WGRAD_GEMM_BYTES
    try:       # force batch chunks (2 images each, N odd -> a ragged last chunk): partial gradients are accumulated
        ops.WGRAD_GEMM_BYTES = 1
        one = ops.conv2d_grad_weight_gemm(x[:1], go[:1], (k, k), p)
        assert one is None or norm_err(n(one), torch.nn.grad.conv2d_weight(x[:1].double(), (Cout, Cin, k, k), go[:1].double(),
                                                                            stride=1, padding=p).cpu().numpy()) <= TOL
    finally:
        ops.WGRAD_GEMM_BYTES = old
    nbytes = []
    real_plan_budget = ops.WGRAD_GEMM_BYTES
    # a budget that admits 2 images per launch but not N
    Wq, M = (W + 2 * p + 7) // 8 * 8, 3 * Cout
    per_img = (M * Ho * Wq * 2 + k * Cin * (H + 2 * p) * Wq * 2) * 2
    try:
        ops.WGRAD_GEMM_BYTES = max(per_img * 2 + k * k * 64 * M * ((Cin + 3) // 4 * 4) * 4 + (1 << 16), 1 << 16)
        chunked = ops.conv2d_grad_weight_gemm(x, go, (k, k), p)
    finally:
        ops.WGRAD_GEMM_BYTES = real_plan_budget
    if chunked is not None:
        assert norm_err(n(chunked), ref.cpu().numpy()) <= TOL


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,p,cl", [(3, 40, 70, 7, 9, 3, 1, False), (2, 33, 65, 6, 5, 5, 2, True),
                                                    (1, 64, 64, 5, 5, 3, 0, True), (5, 96, 32, 9, 7, 5, 0, False),
                                                    (17, 130, 200, 4, 4, 3, 1, True), (4, 64, 128, 19, 23, 3, 1, True),
                                                    (9, 256, 384, 13, 13, 3, 1, True), (6, 64, 96, 27, 27, 5, 2, True)])
def test_weight_gradient_pixel_major_vs_fp64(dev, N, Cin, Cout, H, W, k, p, cl):
    """ops.conv2d_grad_weight_pm (csrc/wgrad_pm.hip: [position][channel] operands, every tap of a 64 / 128 x 64 / 32 tile in one
    workgroup, transposing LDS reads, exact three-term bf16 split of the gradient) against torch.nn.grad.conv2d_weight in fp64:
    all three kernel instances (128 x 64 pipelined, 64 x 64, 64 x 32 for 5 x 5), channel counts that need padding, both memory
    formats, ternary activations, a gradient whose channels span six decades, the STE mask, DoReFa levels, and batch chunking
    with a ragged last chunk."""
    g = torch.Generator(device=dev)
    g.manual_seed(N * 1000 + Cin)
    x = torch.randint(-1, 2, (N, Cin, H, W), generator=g, device=dev).float()             # -1 / 0 / +1
    Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
    go = torch.randn((N, Cout, Ho, Wo), device=dev, generator=g) * torch.exp(torch.randn((1, Cout, 1, 1), device=dev, generator=g) * 3)
    if cl:
        x, go = x.contiguous(memory_format=torch.channels_last), go.contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, k, k), device=dev, generator=g) * 0.8
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), go.double(), stride=1, padding=p)
    per_channel = ref.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)

    def check(got, want):
        assert got.shape == want.shape and got.dtype == torch.float32
        assert float(((got.double() - want).abs() / per_channel).max()) <= TOL        # normalised PER OUTPUT CHANNEL

    with used(pm_entry("qt_wgrad_pm_pack_grad_f32"), pm_entry("qt_wgrad_pm_pack_act_f32"), pm_entry("qt_wgrad_pm_f32"),
              "qt_wgrad_pm_reduce_f32"):
        got = ops.conv2d_grad_weight_pm(x, go, (k, k), p)
    check(got, ref)
    masked = ops.conv2d_grad_weight_pm(x, go, (k, k), p, weight=w)
    check(masked, torch.where(w.abs() <= 1.001, ref, torch.zeros_like(ref)))
    assert bool((masked[w.abs() > 1.001] == 0).all())
    # k-bit DoReFa image: the codes enter the kernel, the result is scaled by fl(1 / n)
    xq = torch.randint(0, 16, (N, Cin, H, W), generator=g, device=dev).float() / 15.0
    refq = torch.nn.grad.conv2d_weight(xq.double(), (Cout, Cin, k, k), go.double(), stride=1, padding=p)
    gq = ops.conv2d_grad_weight_pm(xq, go, (k, k), p, x_levels=15.0)
    assert float(((gq.double() - refq).abs() / refq.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)).max()) <= TOL
    # more K slices than the planner would pick, and one workgroup slot (a single slice)
    for slots in (1, 4096):
        check(ops.conv2d_grad_weight_pm(x, go, (k, k), p, workgroups=slots), ref)
    if N > 2:
        old = ops.WGRAD_GEMM_BYTES
        try:       # a budget that admits ~2 images per launch: partial gradients are accumulated, the last chunk is ragged
            Cpo, Cpi = (Cout + 63) // 64 * 64, (Cin + 63) // 64 * 64
            per_img = 3 * Ho * (W + 2 * p) * Cpo * 2 + (H + 2 * p + k) * (W + 2 * p) * Cpi * 2
            ops.WGRAD_GEMM_BYTES = 2 * per_img + 64 * k * k * Cpo * Cpi * 4 + (1 << 16)
            chunked = ops.conv2d_grad_weight_pm(x, go, (k, k), p)
        finally:
            ops.WGRAD_GEMM_BYTES = old
        if chunked is not None:
            check(chunked, ref)
    # the bias gradient as a by-product of the gradient pack (channels-last gradients only; fixed summation order)
    got_db = []
    check(ops.conv2d_grad_weight_pm(x, go, (k, k), p, bias_grad=got_db), ref)
    if cl:
        ref_db = go.double().sum((0, 2, 3))
        assert len(got_db) == 1 and got_db[0].shape == (Cout,)
        assert float(((got_db[0].double() - ref_db).abs() / go.double().abs().sum((0, 2, 3)).clamp_min(1e-300)).max()) <= 1e-6
        again = []
        ops.conv2d_grad_weight_pm(x, go, (k, k), p, bias_grad=again)
        assert torch.equal(again[0], got_db[0])                                       # deterministic
        if N > 2:
            old = ops.WGRAD_GEMM_BYTES
            try:
                Cpo, Cpi = (Cout + 63) // 64 * 64, (Cin + 63) // 64 * 64
                per_img = 3 * Ho * (W + 2 * p) * Cpo * 2 + (H + 2 * p + k) * (W + 2 * p) * Cpi * 2
                ops.WGRAD_GEMM_BYTES = 2 * per_img + 64 * k * k * Cpo * Cpi * 4 + (1 << 16)
                chunked_db = []
                if ops.conv2d_grad_weight_pm(x, go, (k, k), p, bias_grad=chunked_db) is not None:
                    assert float(((chunked_db[0].double() - ref_db).abs() / go.double().abs().sum((0, 2, 3)).clamp_min(1e-300)).max()) <= 1e-6
            finally:
                ops.WGRAD_GEMM_BYTES = old
    else:
        assert got_db == []                                                            # NCHW gradient: the caller reduces it
    assert ops.conv2d_grad_weight_pm(x, go[:, :, :-1], (k, k), p) is None              # inconsistent shapes: not this route's
    assert ops.conv2d_grad_weight_pm(x, go, (7, 7), p) is None


@pytest.mark.parametrize("N,C,H,W,Cout,k,s,p,cl", [(4, 3, 224, 224, 192, 11, 4, 2, True), (3, 3, 37, 45, 64, 11, 4, 2, False),
                                                     (2, 1, 28, 28, 33, 5, 2, 2, False), (5, 3, 64, 64, 96, 9, 3, 0, True),
                                                     (3, 3, 32, 40, 64, 3, 1, 1, True), (2, 8, 17, 19, 40, 5, 1, 2, False)])
def test_weight_gradient_strided_first_layer_vs_fp64(dev, N, C, H, W, Cout, k, s, p, cl):
    """ops.conv2d_grad_weight_s2d — the weight gradient of a strided conv over a REAL-valued image with few channels (AlexNet's
    3 -> 192, k 11, stride 4): space-to-depth gather + exact three-term split of the image in one packer, the pixel-major kernel
    on the ceil(k / s)-tap stride-1 form, pixel_shuffle back — against torch.nn.grad.conv2d_weight in fp64, directly and as the
    backward of a training-mode BinConv2d (no library detour), incl. maps the stride does not divide."""
    from pytorch_quantize_impls_amd.functions import _fused
    g = torch.Generator(device=dev)
    g.manual_seed(N * 100 + k)
    x = torch.randn((N, C, H, W), device=dev, generator=g) * 3
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    go = torch.randn((N, Cout, Ho, Wo), device=dev, generator=g)
    if cl:
        x, go = x.contiguous(memory_format=torch.channels_last), go.contiguous(memory_format=torch.channels_last)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, C, k, k), go.double(), stride=s, padding=p)
    with used(pm_entry("qt_wgrad_pm_pack_act_s2d_f32"), pm_entry("qt_wgrad_pm_f32")):
        got = ops.conv2d_grad_weight_s2d(x, go, (Cout, C, k, k), s, p)
    with used("qt_wgrad_pm_pack_act_s2d_f32", "qt_wgrad_pm_f32"):               # the exact three-term form stays selectable
        assert norm_err(n(ops.conv2d_grad_weight_s2d(x, go, (Cout, C, k, k), s, p, terms=3)), ref.cpu().numpy()) <= TOL
    assert norm_err(n(got), ref.cpu().numpy()) <= TOL
    assert ops.conv2d_grad_weight_s2d(x, go, (Cout, C, k, k), s + 1, p) is None        # shapes that do not belong together
    conv = BinConv2d(C, Cout, k, stride=s, padding=p).to(dev)
    conv.weight.data.uniform_(-1.3, 1.3)
    old_min = _fused.BWD_MFMA_MIN_MACS
    _fused.BWD_MFMA_MIN_MACS = 0
    lib_before = dict(_fused.LIBRARY_PATHS)
    try:
        with used(pm_entry("qt_wgrad_pm_pack_act_s2d_f32")):
            conv(x).backward(go)
    finally:
        _fused.BWD_MFMA_MIN_MACS = old_min
    assert dict(_fused.LIBRARY_PATHS) == lib_before
    refm = torch.where(conv.weight.detach().abs() > 1.001, torch.zeros_like(ref), ref)
    assert norm_err(n(conv.weight.grad), refm.cpu().numpy()) <= TOL


def test_alexnet_training_step_matches_the_reference_op_sequence(dev):
    """One whole training step of BinaryNet-AlexNet (packed forward, STE masks, grad_input / grad_weight of every conv and
    linear layer on the matrix cores) against the reference's op sequence in torch on the same device (torch.sign,
    F.conv2d / F.linear fp32, the STE of functions/binary_connect.py:31-38), on +-1 pixels so that both forward passes
    are the same integers: every parameter gradient within 1e-5 normalised (2e-5: the reference side is MIOpen fp32)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "bench_train_step", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_train_step.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with used(pm_entry("qt_wgrad_pm_f32"), "qt_conv2d_implicit", "qt_nib_gemm"):
        worst = mod.gradient_agreement(16)
    assert worst <= 2e-5, worst


@pytest.mark.parametrize("Cin,Cout,k,pd,H,B,bits", [(128, 128, 3, 1, 16, 8, 4), (256, 128, 3, 1, 8, 16, 2), (128, 192, 1, 0, 12, 8, 8)])
def test_dorefa_conv_backward_on_matrix_cores_vs_fp64(dev, Cin, Cout, k, pd, H, B, bits):
    """Training-mode DorefaConv2d(bit_width=1) on a k-bit activation (nnDorefaQuant's tagged image): grad_input = E x the
    exact-split conv of the gradient with flipped sign(W), grad_weight = the K-major GEMMs over the INTEGER codes scaled
    by fl(1 / n) — against the fp64 evaluation of what autograd derives upstream (functions/dorefa_connect.py:66-79:
    scaled weight in the forward, UNscaled weight gradient, identity STE)."""
    from pytorch_quantize_impls_amd.functions import _fused, nnDorefaQuant
    from pytorch_quantize_impls_amd.layers import DorefaConv2d
    torch.manual_seed(Cin + bits)
    conv = DorefaConv2d(Cin, Cout, k, padding=pd, bit_width=1).to(dev)
    top = min(1.2, 120.0 / ((1 << bits) - 1))          # the un-clamped quantiser's codes must stay in int8
    xr = (torch.rand(B, Cin, H, H, device=dev) * top).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xq = nnDorefaQuant(bits)(xr)
    xq.retain_grad()
    old_min = _fused.BWD_MFMA_MIN_MACS
    _fused.BWD_MFMA_MIN_MACS = 0
    lib_before = dict(_fused.LIBRARY_PATHS)
    try:
        with used(pm_entry("qt_wgrad_pm_f32") if k == 3 else "qt_bf16_gemm_taps", "qt_conv2d_implicit"):
            y = conv(xq)
            gout = torch.randn_like(y)
            y.backward(gout)
    finally:
        _fused.BWD_MFMA_MIN_MACS = old_min
    assert dict(_fused.LIBRARY_PATHS) == lib_before
    w = conv.weight.detach().double()
    E = w.abs().mean()
    wq = torch.where(w < 0, -torch.ones_like(w), torch.ones_like(w)) * E
    gi = torch.nn.grad.conv2d_input(xq.shape, wq, gout.double(), padding=pd)
    gw = torch.nn.grad.conv2d_weight(xq.detach().double(), conv.weight.shape, gout.double(), padding=pd)
    assert norm_err(n(xq.grad), gi.cpu().numpy()) <= TOL
    assert norm_err(n(conv.weight.grad), gw.cpu().numpy()) <= TOL


def test_elementwise_ops_keep_channels_last_storage(dev):
    """The elementwise kernels walk dense storage as it lies: a channels-last activation / gradient comes back channels-last
    (no NCHW round trip), with the same VALUES as the NCHW evaluation; mixed layouts and non-dense views still work."""
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((5, 24, 7, 9), device=dev, generator=g) * 1.5
    go = torch.randn((5, 24, 7, 9), device=dev, generator=g)
    xcl, gcl = x.contiguous(memory_format=torch.channels_last), go.contiguous(memory_format=torch.channels_last)
    for fn in (ops.binarize, ops.ternarize, lambda t: ops.dorefa_quantize(t, 3), ops.ap2):
        a, b = fn(x), fn(xcl)
        assert b.is_contiguous(memory_format=torch.channels_last) and not b.is_contiguous()
        assert a.is_contiguous() and torch.equal(a, b)
    m_nchw = ops.ste_mask(go, x)
    m_cl = ops.ste_mask(gcl, xcl)
    assert m_cl.is_contiguous(memory_format=torch.channels_last) and torch.equal(m_nchw, m_cl)
    assert torch.equal(ops.ste_mask(gcl, x), m_nchw)                       # second operand brought into the first one's order
    assert torch.equal(ops.ste_mask(go, xcl), m_nchw)
    view = x[:, ::2]                                                       # not dense: copied, result still right
    assert torch.equal(ops.binarize(view), ops.binarize(view.contiguous()))
    one = torch.randn((4, 1, 6, 6), device=dev, generator=g)               # size-1 channel dim: ambiguous strides
    assert torch.equal(ops.binarize(one.contiguous(memory_format=torch.channels_last)), ops.binarize(one))
    # the module path: BinaryConnect's STE backward on a channels-last activation returns a channels-last gradient
    xr = xcl.clone().requires_grad_(True)
    BinaryConnectDeterministic.apply(xr).backward(gcl)
    assert xr.grad.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(xr.grad, torch.where(x.abs() > 1.001, torch.zeros_like(go), go))


def test_conv_behind_maxpool_keeps_the_matrix_core_weight_gradient(dev):
    """VGG pattern: BinaryConnect -> MaxPool2d -> TerConv2d.  The pool drops the quantiser's tag, the forward detects the +-1
    activation (verdict store + device poison flag) and the backward reads the same verdict: weight gradient on the pixel-major
    kernel, no dense-library detour, equal to the fp64 evaluation."""
    from pytorch_quantize_impls_amd.functions import _fused
    torch.manual_seed(4)
    conv = TerConv2d(64, 96, 3, padding=1).to(dev)
    conv.weight.data.uniform_(-1.2, 1.2)
    pool = torch.nn.MaxPool2d(2, 2)
    xr = torch.randn(8, 64, 24, 24, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    old_min = _fused.BWD_MFMA_MIN_MACS
    _fused.BWD_MFMA_MIN_MACS = 0
    lib_before = dict(_fused.LIBRARY_PATHS)
    try:
        xin = pool(BinaryConnectDeterministic.apply(xr))
        assert packed.lookup(xin, packed.NHWC) is None                     # the tag did not survive the pool
        xin.retain_grad()
        with used(pm_entry("qt_wgrad_pm_f32")):
            y = conv(xin)
            gout = torch.randn_like(y)
            y.backward(gout)
    finally:
        _fused.BWD_MFMA_MIN_MACS = old_min
    assert dict(_fused.LIBRARY_PATHS) == lib_before
    gw = torch.nn.grad.conv2d_weight(xin.detach().double(), conv.weight.shape, gout.double(), padding=1)
    gw = torch.where(conv.weight.detach().abs() > 1.001, torch.zeros_like(gw), gw)
    assert norm_err(n(conv.weight.grad), gw.cpu().numpy()) <= TOL
    wq = ops.ternarize(conv.weight.detach()).double()
    gi = torch.nn.grad.conv2d_input(xin.shape, wq, gout.double(), padding=1)
    assert norm_err(n(xin.grad), gi.cpu().numpy()) <= TOL
