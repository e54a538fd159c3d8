"""Round-5 GPU tests, part d: launches taken out of the C4 (DoReFa ResNet-18) forward and the deeper DMA ring of its late stages.

  * qt_codes_to_f32: the fp32 image of a code plane (any halo, ragged channel counts, raised range flag -> NaN), and with
    avg_pool2d in the same pass bit-identical to ATen's pooling of that image;
  * qt_affine_dorefa_codes_halo_i8: the BatchNorm + ReLU + quantiser pass writing straight into the consumer's halo plane equals the
    plain pass followed by qt_pad_pixel_plane, byte for byte;
  * qt_conv2d_implicit_halo_bn: the 1x1 shortcut conv with BatchNorm in its epilogue equals conv, then qt_bn_eval_device_f32;
  * the ring of 3 / 4 stage buffers on the 256- / 512-channel small-map convs produces the codes of the double-buffered loop;
  * the whole C4 net, fused form and un-modified module graph, still equals its module-by-module execution bit for bit."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import bench_models  # noqa: E402
from pytorch_quantize_impls_amd import _lib, lazy, ops, packed  # noqa: E402
from pytorch_quantize_impls_amd.layers import DorefaConv2d, FusedBnDorefaQuant, FusedDorefaConvBnQuant  # noqa: E402
from pytorch_quantize_impls_amd.layers import fused as fused_layers  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


def _plane(dev, N, C, H, W, halo, seed=0, bits=4):
    g = torch.Generator(device="cpu").manual_seed(seed)
    hy, hx = halo
    ld = ops.code_ld_bytes(C, 16)
    q = torch.zeros((N, H + 2 * hy, W + 2 * hx, ld), dtype=torch.int8)
    q[:, hy:hy + H, hx:hx + W, :C] = torch.randint(-20, 128, (N, H, W, C), generator=g, dtype=torch.int32).to(torch.int8)
    flag = torch.zeros((1,), dtype=torch.int32, device=dev)
    planes = ops.CodePlanes(codes=q.view(-1, ld).to(dev), rows=N * (H + 2 * hy) * (W + 2 * hx), K=C, inv_n=ops.inv_levels(bits),
                            bit_width=bits, overflow=flag)
    return packed.CodeActivation(planes, (N, C, H, W), halo=halo), q[:, hy:hy + H, hx:hx + W, :C]


@pytest.mark.parametrize("N,C,H,W,halo", [(3, 64, 8, 8, (1, 1)), (2, 20, 5, 7, (0, 0)), (2, 7, 4, 4, (2, 1)), (5, 512, 4, 4, (1, 1))])
def test_codes_to_f32_is_the_quantisers_image(dev, N, C, H, W, halo):
    act, q = _plane(dev, N, C, H, W, halo, seed=C)
    want = (q.to(torch.float32).to(dev) * act.codes.inv_n).permute(0, 3, 1, 2)         # fl(inv_n * q), the reference's image
    before = _lib.call_counts.get("qt_codes_to_f32", 0)
    y = act.float()
    assert _lib.call_counts.get("qt_codes_to_f32", 0) == before + 1
    assert y.shape == (N, C, H, W) and torch.equal(y, want)
    act.codes.overflow.fill_(1)                                                         # the chain left int8 somewhere
    assert bool(torch.isnan(act.float()).all())
    with pytest.raises(RuntimeError):
        act.float(check=True)


@pytest.mark.parametrize("N,C,H,W,halo,k", [(256, 512, 4, 4, (1, 1), 4), (3, 64, 8, 8, (1, 1), 2), (2, 20, 6, 9, (0, 0), 3), (2, 36, 7, 7, (0, 0), 7),
                                            (4, 128, 8, 8, (0, 0), 4)])
def test_avg_pool_on_codes_equals_atens_pooling_of_the_image(dev, N, C, H, W, halo, k):
    act, _ = _plane(dev, N, C, H, W, halo, seed=7 * C + k)
    img = act.float()
    want_cl = torch.nn.functional.avg_pool2d(img, k)                                    # channels_last image (what the head sees)
    want_nchw = torch.nn.functional.avg_pool2d(img.contiguous(), k)                     # ... and the NCHW kernel
    got = act.avg_pool2d(k)
    assert got.shape == want_cl.shape
    assert torch.equal(got, want_cl) and torch.equal(got, want_nchw)


@pytest.mark.parametrize("N,C,H,W,halo", [(4, 64, 32, 32, (1, 1)), (3, 24, 5, 7, (1, 2)), (2, 128, 16, 16, (1, 1))])
def test_affine_codes_into_a_halo_plane_equals_pass_plus_padding(dev, N, C, H, W, halo):
    torch.manual_seed(C + H)
    x = (torch.randn(N, C, H, W, device=dev) * 0.6).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(C).to(dev).eval()
    bench_models.randomize_bn(bn, seed=2)
    with torch.no_grad():
        plain = FusedBnDorefaQuant(bn, 4, fold="device")(x)
        padded = ops.pad_pixel_plane(plain.codes.codes, N, H, W, halo)
        before = _lib.call_counts.get("qt_affine_dorefa_codes_halo_i8", 0)
        got = FusedBnDorefaQuant(bn, 4, out_halo=halo, fold="device")(x)
    assert _lib.call_counts.get("qt_affine_dorefa_codes_halo_i8", 0) == before + 1
    assert got.halo == tuple(halo) and torch.equal(got.codes.codes, padded)
    assert int(got.codes.overflow.item()) == int(plain.codes.overflow.item())


@pytest.mark.parametrize("cin,cout,hw", [(64, 128, 32), (128, 256, 16), (256, 512, 8)])
def test_shortcut_conv_with_batchnorm_in_its_epilogue(dev, cin, cout, hw):
    torch.manual_seed(cin)
    N = 8
    act, _ = _plane(dev, N, cin, hw, hw, (1, 1), seed=cin)
    act.codes.codes.clamp_(min=0, max=15)
    sc = DorefaConv2d(cin, cout, 1, stride=2, bias=False, bit_width=1).to(dev).eval()
    main = DorefaConv2d(cin, cout, 3, stride=2, padding=1, bias=False, bit_width=1).to(dev).eval()
    bn_main, bn_sc = torch.nn.BatchNorm2d(cout).to(dev).eval(), torch.nn.BatchNorm2d(cout).to(dev).eval()
    bench_models.randomize_bn(bn_main, seed=3)
    bench_models.randomize_bn(bn_sc, seed=4)
    for b in (bn_main, bn_sc):
        b.running_var.mul_(4.0)
    blk = FusedDorefaConvBnQuant(main, bn_main, 4, out_halo=1, fold="device")
    with torch.no_grad():
        two = blk(act, residual=sc(act), residual_bn=bn_sc)                              # conv, then qt_bn_eval_device_f32
        before = _lib.call_counts.get("qt_conv2d_implicit_halo_bn", 0), _lib.call_counts.get("qt_bn_eval_device_f32", 0)
        one = blk(act, residual_conv=(sc, act), residual_bn=bn_sc)
    assert _lib.call_counts.get("qt_conv2d_implicit_halo_bn", 0) == before[0] + 1
    assert _lib.call_counts.get("qt_bn_eval_device_f32", 0) == before[1]
    assert torch.equal(one.codes.codes, two.codes.codes)
    # ... and the branch value itself against the library's eval-mode BatchNorm of the conv output
    with torch.no_grad():
        y2, none = blk._shortcut(sc, act, bn_sc)
        want = torch.nn.functional.batch_norm(sc(act), bn_sc.running_mean, bn_sc.running_var, bn_sc.weight, bn_sc.bias, False, 0.0, bn_sc.eps)
    assert none is None
    assert torch.equal(y2.view(N, hw // 2, hw // 2, cout).permute(0, 3, 1, 2), want)


_RING_AB = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
import bench_models
from pytorch_quantize_impls_amd import _lib, ops, packed
from pytorch_quantize_impls_amd.layers import DorefaConv2d, FusedDorefaConvBnQuant
dev = torch.device("cuda:0")
h = hashlib.sha256()
for cin, hw, stride in ((256, 8, 1), (512, 4, 1), (256, 8, 2), (128, 16, 2)):
    cout = cin * stride
    torch.manual_seed(cin + stride)
    N = 256
    ld = ops.code_ld_bytes(cin, 16)
    q = torch.zeros((N, hw + 2, hw + 2, ld), dtype=torch.int8)
    q[:, 1:-1, 1:-1, :cin] = torch.randint(0, 16, (N, hw, hw, cin)).to(torch.int8)
    planes = ops.CodePlanes(codes=q.view(-1, ld).to(dev), rows=N * (hw + 2) ** 2, K=cin, inv_n=ops.inv_levels(4), bit_width=4,
                            overflow=torch.zeros((1,), dtype=torch.int32, device=dev))
    act = packed.CodeActivation(planes, (N, cin, hw, hw), halo=(1, 1))
    conv = DorefaConv2d(cin, cout, 3, stride=stride, padding=1, bias=False, bit_width=1).to(dev).eval()
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    bench_models.randomize_bn(bn, seed=1)
    bn.running_var.mul_(4.0)
    with torch.no_grad():
        out = FusedDorefaConvBnQuant(conv, bn, 4, out_halo=1, fold="device")(act, residual=act if stride == 1 else None)
    torch.cuda.synchronize()
    h.update(out.codes.codes.cpu().numpy().tobytes())
    h.update(bytes([int(out.codes.overflow.item())]))
print("DIGEST", h.hexdigest())
"""


def test_deep_dma_ring_produces_the_double_buffered_codes(dev):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for ring in ("0", "1"):
        env = dict(os.environ)
        env.pop("QT_NO_CONV_DEEP_RING", None)
        if ring == "0":
            env["QT_NO_CONV_DEEP_RING"] = "1"
        out = subprocess.run([sys.executable, "-c", _RING_AB % root], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert digests[0] == digests[1]


def test_c4_forms_equal_the_module_by_module_graph_and_count_their_launches(dev):
    torch.manual_seed(4)
    m4 = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(m4, seed=3)
    for m in m4.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.mul_(4.0)
    m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn((32, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
    f4 = bench_models.FusedDorefaResNet18(m4, fold="device")
    with torch.no_grad():
        with lazy.eager():
            ye = m4(x)
        c0 = dict(_lib.call_counts)
        yf = f4(x)
        c1 = dict(_lib.call_counts)
        lazy.STATS.clear()
        ym = m4(x)
        c2 = dict(_lib.call_counts)

    def delta(a, b, name):
        return b.get(name, 0) - a.get(name, 0)
    assert torch.equal(yf, ye) and torch.equal(ym, ye)
    for a, b, pads in ((c0, c1, 0), (c1, c2, 1)):
        assert delta(a, b, "qt_conv2d_implicit_halo_bn") == 3          # the three conv + BatchNorm shortcut branches, one launch each
        assert delta(a, b, "qt_bn_eval_device_f32") == 0
        assert delta(a, b, "qt_codes_to_f32") == 1                     # the head: codes -> image -> avg_pool2d in one pass
        # fused form: the stem quantiser writes the halo plane itself; module graph: the chain's first conv reads the code TAG of the
        # stem's nnDorefaQuant result (no halo) and pads it once to run the direct kernel like every later conv
        assert delta(a, b, "qt_pad_pixel_plane") == pads
        assert delta(a, b, "qt_conv2d_implicit_codes") == 16
    assert lazy.STATS["avg_pool_on_codes"] == 1


@pytest.mark.parametrize("cin,cout,hw,stride,k,batch", [
    (64, 256, 16, 1, 1, 128),      # ONE 64-byte stage on the ring of four (fewer stages than buffers)
    (128, 256, 16, 1, 1, 128),     # one 128-byte stage
    (192, 256, 16, 1, 1, 128),     # two stages, the second half empty
    (64, 256, 16, 1, 3, 128),      # 4.5 stages: exactly the ring depth + a tail
    (128, 256, 32, 2, 3, 128),     # wide stride-2 conv on a small map (the 128 x 128 deep-ring rule)
    (256, 256, 8, 1, 3, 256),      # 18 stages (stage 3 of the ResNet)
    (512, 512, 4, 1, 3, 199),      # ragged M: 3184 rows of 128 x 64 tiles, ring of three
    (144, 256, 16, 2, 3, 250),     # 1296-byte rows: the wide stride-2 rule with a partial last stage and ragged M
])
def test_deep_ring_short_and_ragged_k_loops(dev, cin, cout, hw, stride, k, batch):
    """Whatever configuration the rules pick with and without the ring (the switch is read per call, so ONE process): short K loops
    (these stay on the double-buffered small-K tiles), partial last stages, row tiles past M on the ring configurations."""
    torch.manual_seed(cin + cout + k)
    pad = k // 2
    act, _ = _plane(dev, batch, cin, hw, hw, (1, 1), seed=cin + k)
    act.codes.codes.clamp_(min=0, max=15)
    conv = DorefaConv2d(cin, cout, k, stride=stride, padding=pad, bias=False, bit_width=1).to(dev).eval()
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    bench_models.randomize_bn(bn, seed=5)
    bn.running_var.mul_(4.0)
    blk = FusedDorefaConvBnQuant(conv, bn, 4, out_halo=1, fold="device")
    old = os.environ.pop("QT_NO_CONV_DEEP_RING", None)
    try:
        with torch.no_grad():
            ring = blk(act)
            os.environ["QT_NO_CONV_DEEP_RING"] = "1"
            plain = blk(act)
    finally:
        os.environ.pop("QT_NO_CONV_DEEP_RING", None)
        if old is not None:
            os.environ["QT_NO_CONV_DEEP_RING"] = old
    torch.cuda.synchronize()
    assert torch.equal(ring.codes.codes, plain.codes.codes)
    assert int(ring.codes.overflow.item()) == 0
    # ... and against the integer conv evaluated by the library on the code values (exact in fp32 for these sizes), pushed through the
    # same BatchNorm / ReLU / quantiser modules
    with torch.no_grad():
        xq = act.float()
        want = torch.nn.functional.conv2d(xq, conv.weight.sign() * conv.weight.abs().amax(), None, stride, pad)
        ref = FusedBnDorefaQuant(bn, 4, fold="device")(want.contiguous(memory_format=torch.channels_last))
    got = ring.without_halo().codes.codes[:, :cout]
    flips = (got != ref.codes.codes[:, :cout])
    assert float(flips.float().mean()) < 1e-3                      # the library's fp32 conv rounds differently: ties only


@pytest.mark.parametrize("halo,k,pad,stride", [((1, 1), 1, 0, 2), ((1, 1), 3, 1, 1), ((0, 0), 3, 1, 1), ((0, 0), 1, 0, 1), ((2, 2), 3, 1, 2)])
def test_conv_with_batchnorm_epilogue_every_route(dev, halo, k, pad, stride):
    """ops.conv2d_codes(epi=BnEpilogue): one launch on halo planes / un-padded convs, conv + qt_bn_eval_device_f32 where the plane
    is padded first — the same bits either way."""
    torch.manual_seed(k + pad + stride)
    N, cin, cout, hw = 4, 32, 48, 12
    act, _ = _plane(dev, N, cin, hw, hw, halo, seed=3)
    act.codes.codes.clamp_(min=0, max=15)
    conv = DorefaConv2d(cin, cout, k, stride=stride, padding=pad, bias=True, bit_width=1).to(dev).eval()
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    bench_models.randomize_bn(bn, seed=6)
    Ho, Wo = ops.conv_out_hw(hw, hw, k, k, stride, pad, 1)
    with torch.no_grad():
        w, b, stats = fused_layers.device_bn_fold(bn, (N, cout, Ho, Wo), True)
        wc = ops.pack_conv_weight_codes(conv.weight.detach())
        E = conv.weight.abs().amax()
        args = (act.codes, act.shape, wc, (k, k), act.codes.inv_n, conv.bias, stride, pad, 1)
        plain = ops.conv2d_codes(*args, scale_dev=E, in_halo=halo)
        want = ops.bn_eval_device(plain, w, b, stats)
        got = ops.conv2d_codes(*args, scale_dev=E, epi=ops.BnEpilogue(w, b, stats), in_halo=halo)
        lib = torch.nn.functional.batch_norm(plain.view(N, Ho, Wo, cout).permute(0, 3, 1, 2), bn.running_mean, bn.running_var, bn.weight,
                                             bn.bias, False, 0.0, bn.eps)
    assert torch.equal(got, want)
    assert torch.equal(got.view(N, Ho, Wo, cout).permute(0, 3, 1, 2), lib)


def test_shortcut_batchnorm_written_before_the_chain_is_used_is_caught(dev):
    """The one-launch shortcut branch reads its BatchNorm when the chain is USED: an in-place write in between is reported like any
    other producer of a deferred activation (version counters of the branch's BatchNorm node, not only of its conv)."""
    from pytorch_quantize_impls_amd.functions import nnDorefaQuant
    torch.manual_seed(2)
    m = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(m, seed=3)
    m = m.to(dev).to(memory_format=torch.channels_last).eval()
    blk = m.blocks[2]                                               # 64 -> 128, stride 2: conv + BatchNorm shortcut
    x = torch.randn(4, 64, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        q = nnDorefaQuant(4)(torch.relu(x))
        y = blk(q)
        assert isinstance(y, lazy.LazyActivation)
        ref = y + 0
        y2 = blk(q)
        blk.shortcut[1].running_mean.add_(0.25)
        with pytest.raises(RuntimeError, match="modified in place"):
            y2 + 0
        y3 = blk(q)                                                 # recorded after the write: the new statistics
        with lazy.eager():
            want = blk(q)
        assert torch.equal(y3 + 0, want) and not torch.equal(want, ref)


def test_graphed_module_replays_from_its_static_input_without_a_copy(dev):
    from pytorch_quantize_impls_amd import utils
    torch.manual_seed(9)
    m = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(m, seed=3)
    m = m.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn((8, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
    x2 = torch.randn_like(x)
    with torch.no_grad():
        g = utils.graphed(m, x)
        y1 = g(x).clone()
        g.static_input.copy_(x2)                       # the caller fills the captured buffer itself ...
        y2 = g(g.static_input).clone()                 # ... and replays without the device-to-device copy
        assert torch.equal(y2, g(x2)) and not torch.equal(y1, y2)
        with lazy.eager():
            assert torch.equal(y1, m(x))
