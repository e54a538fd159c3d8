"""Deferred activations (pytorch_quantize_impls_amd/lazy.py): the un-modified module-by-module model must
  * run the fused kernels (lazy.STATS / _lib.call_counts), bit-identical to the opt-in fused form built from the same
    modules (whose own parity against the oracle / the CPU chain is pinned in test_gpu_parity.py / test_gpu_r2.py),
  * hand out exactly the module-by-module value whenever a deferred activation is used by anything else."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import _lib, lazy  # noqa: E402
from pytorch_quantize_impls_amd.functions import BinaryConnect, _fused  # noqa: E402
from pytorch_quantize_impls_amd.functions.binary_connect import BinaryConnectDeterministic  # noqa: E402
from pytorch_quantize_impls_amd.layers import BinConv2d, TerConv2d, LinearBin, FusedFeatureClassifier  # noqa: E402
import bench_models  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _fresh_stats():
    """Explicit fused forms in this file are built with the device fold (what lazy.py uses)."""
    from pytorch_quantize_impls_amd.layers import fused as fused_mod
    lazy.STATS.clear()
    _fused.LIBRARY_PATHS.clear()
    prev = fused_mod.DEFAULT_FOLD
    fused_mod.DEFAULT_FOLD = "device"
    yield
    fused_mod.DEFAULT_FOLD = prev
    assert lazy.ENABLED and lazy.DEFER_CODES


def _alexnet(dev, seed=0):
    torch.manual_seed(seed)
    m = bench_models.AlexNetBin(num_classes=10)
    bench_models.randomize_bn(m, seed)
    return m.to(dev).to(memory_format=torch.channels_last).eval()


def test_alexnet_module_graph_runs_fused_and_equals_the_fused_form(dev):
    m = _alexnet(dev)
    x = torch.randn(16, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    fused = bench_models.FusedAlexNetBin(m)
    with torch.no_grad():
        ref = fused(x)
        before = dict(_lib.call_counts)
        lazy.STATS.clear()
        y = m(x)
        with lazy.eager():
            e = m(x)
    assert type(y) is torch.Tensor and type(e) is torch.Tensor          # LogSoftmax reads the last LinearBin's (computed) result
    assert lazy.STATS["dense_deferred"] == 3 and lazy.STATS["dense_fused"] == 2, lazy.STATS   # BatchNorm1d / Hardtanh / sign: one pass
    assert torch.equal(y, ref)                                          # same kernels, same bits
    assert lazy.STATS["deferred"] == 5 and lazy.STATS["fused"] == 5 and lazy.STATS["materialised"] == 0, lazy.STATS
    assert not any(k.startswith("fallback") for k in lazy.STATS), lazy.STATS
    assert _lib.call_counts["qt_conv2d_implicit_bits"] + _lib.call_counts["qt_conv2d_implicit_nib"] \
        > before.get("qt_conv2d_implicit_bits", 0) + before.get("qt_conv2d_implicit_nib", 0)
    assert not _fused.LIBRARY_PATHS, _fused.LIBRARY_PATHS
    # against the module-by-module evaluation on this device: bit-identical (thresholds bisected on its own F.batch_norm)
    assert torch.equal(y, e)
    assert torch.isfinite(y).all()


def test_ternary_vgg_module_graph_equals_the_fused_form(dev):
    torch.manual_seed(1)
    m = bench_models.TernaryVGG16(num_classes=10, image=64, fc=256)
    bench_models.randomize_bn(m, 1)
    m = m.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn(4, 3, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
    fused = FusedFeatureClassifier(m.features, m.classifier, (512, 2, 2))
    with torch.no_grad():
        ref = fused(x)
        lazy.STATS.clear()
        y = m(x)
        st = dict(lazy.STATS)
        with lazy.eager():
            e = m(x)
    assert torch.equal(y, ref)
    assert torch.equal(y, e)                     # ... and the module-by-module execution, bit for bit
    lazy.STATS.clear()
    lazy.STATS.update(st)
    assert lazy.STATS["deferred"] == 13 and lazy.STATS["fused"] == 13 and lazy.STATS["materialised"] == 0, lazy.STATS


def _block(dev, kind=BinConv2d, cin=64, cout=128, pool=True):
    torch.manual_seed(3)
    mods = [kind(cin, cout, 3, padding=1)]
    if pool:
        mods.append(nn.MaxPool2d(2, 2))
    mods += [nn.BatchNorm2d(cout), nn.Hardtanh(inplace=True), BinaryConnect()]
    seq = nn.Sequential(*mods)
    bench_models.randomize_bn(seq, 3)
    return seq.to(dev).eval()


def _pm1(shape, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(0, 2, shape, generator=g).float() * 2 - 1).to(dev)


@pytest.mark.parametrize("kind", [BinConv2d, TerConv2d])
def test_escaping_activations_equal_the_module_by_module_value(dev, kind):
    seq = _block(dev, kind)
    x = BinaryConnectDeterministic.apply(_pm1((4, 64, 20, 20), dev))
    with torch.no_grad():
        with lazy.eager():
            e_conv = seq[0](x)
            e_bn = seq[2](seq[1](e_conv))
            e_all = seq(x)
        y = seq[0](x)
        assert isinstance(y, lazy.LazyActivation) and isinstance(y, torch.Tensor)
        assert y.shape == e_conv.shape and y.dtype == torch.float32 and y.device == e_conv.device and y.dim() == 4
        assert torch.equal(y + 0, e_conv)                       # arithmetic materialises: the conv's own fp32 result
        assert torch.equal(y.cpu(), e_conv.cpu())
        assert torch.equal(torch.relu(seq[2](seq[1](seq[0](x)))), torch.relu(e_bn))      # un-recordable op after BN
        out = seq(x)                                             # ends with BinaryConnect: still deferred
        assert isinstance(out, lazy.LazyActivation) and out._qt.signed
        assert torch.equal(out.value(), e_all)                   # replayed module by module
        assert torch.equal(out * 1.0, e_all)
        s = repr(out)
        assert s.startswith("tensor(")
        assert float(out.abs().min()) == 1.0
    assert lazy.STATS["fused"] == 0 and lazy.STATS["materialised"] > 0


def test_chain_feeding_a_second_conv_and_a_linear(dev):
    a, b = _block(dev, BinConv2d, 64, 128, pool=True), _block(dev, TerConv2d, 128, 128, pool=False)
    fc = LinearBin(128 * 10 * 10, 32).to(dev).eval()
    x = _pm1((3, 64, 20, 20), dev)
    with torch.no_grad():
        with lazy.eager():
            e = fc(b(a(x)).reshape(3, -1))
        lazy.STATS.clear()
        y = fc(torch.flatten(b(a(x)), 1))
        y2 = fc(nn.Flatten()(b(a(x))))
        y3 = fc(b(a(x)).view(3, -1))
    # an eval-mode LinearBin hands its (computed) result out as a "dense" deferred activation: BatchNorm1d -> Hardtanh -> BinaryConnect
    # would be recorded on it, anything else reads the value
    assert type(y) is lazy.LazyDense and y._qt.kind == "dense" and type(y.value()) is torch.Tensor
    assert torch.equal(y, e) and torch.equal(y2, e) and torch.equal(y3, e)      # +-1 / 0 operands: exact integers
    assert lazy.STATS["fused"] == 6 and lazy.STATS["materialised"] == 0, lazy.STATS


def test_nothing_is_deferred_with_autograd_training_or_cpu(dev):
    seq = _block(dev)
    x = _pm1((2, 64, 12, 12), dev)
    assert type(seq(x)) is torch.Tensor                          # grad mode: autograd graph needed
    seq.train()
    with torch.no_grad():
        assert type(seq[0](x)) is torch.Tensor
    seq.eval()
    with torch.no_grad():
        assert isinstance(seq[0](x), lazy.LazyActivation)
        assert type(copy.deepcopy(seq).cpu()[0](x.cpu())) is torch.Tensor
        with lazy.eager():
            assert type(seq[0](x)) is torch.Tensor
        g = BinConv2d(64, 64, 3, padding=1, groups=2).to(dev).eval()
        assert type(g(x)) is torch.Tensor
    assert lazy.STATS["deferred"] == 1


def test_first_use_outside_no_grad(dev):
    """Deferred under no_grad, used after the block ended (autograd on again): evaluated as deferred, by the HIP path."""
    from pytorch_quantize_impls_amd.functions import nnDorefaQuant
    from pytorch_quantize_impls_amd.layers import DorefaConv2d
    seq = _block(dev, BinConv2d, 64, 128, pool=False)
    x = _pm1((2, 64, 12, 12), dev)
    dconv = DorefaConv2d(16, 32, 3, padding=1, bit_width=1).to(dev).eval()
    xq = torch.rand(2, 16, 8, 8, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        with lazy.eager():
            e, q = seq[0](x), nnDorefaQuant(4)(xq)
            ed = dconv(q)
        y, yd = seq[0](x), dconv(q)
    assert isinstance(yd, lazy.LazyActivation)
    assert torch.is_grad_enabled()
    v, vd = y * 1.0, yd * 1.0
    assert torch.equal(v, e) and torch.equal(vd, ed) and not v.requires_grad and not vd.requires_grad
    assert not _fused.LIBRARY_PATHS, _fused.LIBRARY_PATHS


def test_inference_mode_and_hooks(dev):
    a, b = _block(dev, BinConv2d, 64, 128, pool=True), _block(dev, BinConv2d, 128, 64, pool=False)
    x = _pm1((2, 64, 16, 16), dev)
    seen = []
    h = a[2].register_forward_hook(lambda mod, inp, out: seen.append((type(inp[0]), tuple(out.shape))))
    with torch.no_grad(), lazy.eager():
        e = b(a(x)) + 0
    with torch.inference_mode():
        y = b(a(x))
        assert isinstance(y, lazy.LazyActivation)
        assert torch.equal(y + 0, e)
    h.remove()
    assert seen[-1][1] == (2, 128, 8, 8)


def test_batchnorm_update_is_picked_up(dev):
    a, b = _block(dev, BinConv2d, 64, 128, pool=False), _block(dev, BinConv2d, 128, 64, pool=False)
    x = _pm1((2, 64, 16, 16), dev)
    with torch.no_grad():
        y0 = b(a(x)) + 0
        a[1].running_mean.add_(40.0)                              # in place: same storage, new version counter
        with lazy.eager():
            e1 = b(a(x)) + 0
        y1 = b(a(x)) + 0
        sd = {k: v.clone() for k, v in a[1].state_dict().items()}
        sd["running_mean"] -= 40.0
        a[1].load_state_dict(sd)
        y2 = b(a(x)) + 0
    assert torch.equal(y1, e1) and not torch.equal(y0, y1) and torch.equal(y2, y0)


def test_in_place_modification_before_use_is_an_error_not_a_wrong_value(dev):
    a = _block(dev, BinConv2d, 64, 128, pool=False)
    x = _pm1((2, 64, 16, 16), dev)
    with torch.no_grad():
        y = a(x)
        a[0].load_state_dict({k: torch.sign(torch.randn_like(v)) for k, v in a[0].state_dict().items()})
        with pytest.raises(RuntimeError, match="modified in place"):
            y + 0
        y = a[0](x)
        x.neg_()
        with pytest.raises(RuntimeError, match="modified in place"):
            y.cpu()
        y = a(x)
        a[1].running_var.mul_(2.0)
        with pytest.raises(RuntimeError, match="modified in place"):
            y.value()
        y = a(x)
        v = y + 0                          # used first: later writes do not matter
        a[1].running_var.mul_(0.5)
        assert torch.equal(y + 0, v)


def test_same_deferred_activation_used_twice(dev):
    a = _block(dev, BinConv2d, 64, 128, pool=True)
    c1 = BinConv2d(128, 64, 3, padding=1).to(dev).eval()
    c2 = BinConv2d(128, 32, 1).to(dev).eval()
    x = _pm1((2, 64, 16, 16), dev)
    with torch.no_grad():
        with lazy.eager():
            t = a(x)
            e1, e2 = c1(t), c2(t)
        t = a(x)
        y1, y2 = c1(t) + 0, c2(t) + 0        # different paddings: the producer writes two operands
    assert torch.equal(y1, e1) and torch.equal(y2, e2)


def test_writes_into_a_deferred_activation_go_to_a_private_copy(dev):
    """``act[mask] = v``, ``act.mul_(2)`` and ``torch.add(a, b, out=act)`` on a deferred activation: the wrapper moves on to the
    written result, a chain recorded on the activation BEFORE the write still replays from the unwritten value."""
    conv = BinConv2d(32, 48, 3, padding=1).to(dev).eval()
    bn = nn.BatchNorm2d(48).to(dev).eval()
    bn.running_var.uniform_(0.5, 4.0)
    x = _pm1((2, 32, 9, 9), dev)
    with torch.no_grad():
        with lazy.eager():
            e = conv(x)
            e_bn = bn(e)
        for kind in ("setitem", "mul_", "out"):
            t = conv(x)
            assert isinstance(t, lazy.LazyActivation)
            later = bn(t)                         # recorded before the write
            want = e.clone()
            if kind == "setitem":
                t[:, :3] = 7.0
                want[:, :3] = 7.0
            elif kind == "mul_":
                t.mul_(2.0)
                want.mul_(2.0)
            else:
                torch.add(e, 1.0, out=t)
                want = e + 1.0
            assert torch.equal(t + 0, want), kind
            assert torch.equal(later + 0, e_bn), kind


def test_module_graph_in_a_hipgraph(dev):
    from pytorch_quantize_impls_amd import utils
    m = _alexnet(dev, 5)
    x = torch.randn(8, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = m(x)
        graphed = utils.graphed(m, x)
        y = graphed(x).clone()          # the graph owns one output buffer
        x2 = torch.randn_like(x)
        assert torch.equal(graphed(x2), m(x2))
    assert torch.equal(y, ref)


# ---- DoReFa chains: conv -> BatchNorm [-> + shortcut] -> ReLU -> nnDorefaQuant(k) ------------------------------------

def _resnet(dev, seed=4):
    torch.manual_seed(seed)
    m = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(m, seed=3)
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_var.mul_(4.0)
    return m.to(dev).to(memory_format=torch.channels_last).eval()


def test_dorefa_resnet_blocks_run_in_the_code_epilogue_and_equal_the_fused_form(dev):
    from pytorch_quantize_impls_amd import packed
    from pytorch_quantize_impls_amd.functions import nnDorefaQuant
    m = _resnet(dev)
    fused_blocks = nn.Sequential(*[bench_models._FusedDorefaBlock(b, 4, True, 1) for b in m.blocks])
    x = torch.randn(8, 64, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        q = nnDorefaQuant(4)(torch.relu(x))                       # fp32 image + int8 code tag
        codes = packed.lookup_codes(q, packed.NHWC)
        assert codes is not None
        ref = fused_blocks(packed.CodeActivation(codes, tuple(q.shape))).float()
        before = dict(_lib.call_counts)
        lazy.STATS.clear()
        y = m.blocks(q)
        assert isinstance(y, lazy.LazyActivation) and y._qt.quant == 4
        got = y + 0
        with lazy.eager():
            e = m.blocks(q)
    assert torch.equal(got, ref)
    # 16 block convs run with the code epilogue; the 3 shortcut convs run with their BatchNorm in the epilogue (one launch each, never
    # materialised as a conv output) and give the fp32 residual; the only value produced is the one ``y + 0`` asks for
    assert lazy.STATS["deferred"] == 19 and lazy.STATS["fused"] == 16 and lazy.STATS["materialised"] == 1, lazy.STATS
    assert _lib.call_counts["qt_conv2d_implicit_halo_bn"] - before.get("qt_conv2d_implicit_halo_bn", 0) == 3
    assert _lib.call_counts["qt_conv2d_implicit_codes"] - before.get("qt_conv2d_implicit_codes", 0) >= 13
    # against the module-by-module evaluation (MIOpen BatchNorm, separate add / ReLU / quantiser passes): the code epilogue
    # evaluates BatchNorm in this device's own arithmetic (layers.fused.device_bn_fold), so the codes are the same, bit for bit
    assert torch.equal(got, e)


def test_dorefa_resnet_whole_model_and_escapes(dev):
    m = _resnet(dev, 6)
    x = torch.randn(16, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = m(x)
        with lazy.eager():
            e = m(x)
        blk = m.blocks[2]                                         # stride-2 block with a conv shortcut
        q = m.quant(torch.relu(m.bn(m.stem(x))))
        a = m.blocks[1](m.blocks[0](q))
        with lazy.eager():
            ae = m.blocks[1](m.blocks[0](q))
            t_e = blk.bn1(blk.conv1(ae))
        t = blk.bn1(blk.conv1(a))                                 # conv -> BatchNorm, then something off the grammar
        assert isinstance(t, lazy.LazyActivation)
        assert torch.equal(torch.sigmoid(t), torch.sigmoid(t_e))
        assert type(blk.conv1(a) * 2.0) is torch.Tensor
    assert type(y) is torch.Tensor and torch.isfinite(y).all()
    assert torch.equal(y, e)                     # whole network: the deferred graph IS the module-by-module graph


def test_deferred_output_and_stream_capture(dev):
    """A graph whose output is a deferred activation: utils.graphed turns it into its value inside the capture; a manual
    capture that lets it escape is an error, not a tensor that replays would never update."""
    from pytorch_quantize_impls_amd import utils
    seq = _block(dev, BinConv2d, 64, 128, pool=True)[:3]          # conv -> pool -> BatchNorm: the output stays deferred
    seq[0].binary_input = True                                    # no +-1 detection (a host sync) inside the capture
    x = _pm1((2, 64, 16, 16), dev)
    with torch.no_grad():
        with lazy.eager():
            e = seq(x)
        assert isinstance(seq(x), lazy.LazyActivation)
        g = utils.graphed(seq, x)
        assert type(g(x)) is torch.Tensor and torch.equal(g(x), e)
        x2 = -x
        with lazy.eager():
            e2 = seq(x2)
        assert torch.equal(g(x2), e2)
        s = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        xs = x.clone()
        with torch.cuda.stream(s):
            seq(xs) + 0
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=s):
                out = seq(xs)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="captured"):
            out + 0


def test_results_do_not_depend_on_concurrent_load(dev):
    """The C2 routes and the module-graph forwards repeated while another stream keeps CUs busy (workgroups start staggered):
    bit-identical to the quiet device.  (The experiment that failed exactly this — a one-launch linear forward with an
    intra-launch hand-off — lives in tools/experiments and is not in the library.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "stress_concurrent", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_concurrent.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.run(6, verbose=False)
    assert not any(bad.values()), bad


@pytest.mark.parametrize("seed", list(range(20)))
def test_fuzz_random_stacks_deferred_equals_fuse_sequential(dev, seed):
    """Random binarised / ternarised conv stacks (kernel, stride, padding, pooling before the BatchNorm or after the sign,
    channel counts that are not multiples of 32, optional Hardtanh / Dropout) closed by a conv that consumes the last
    activation: the un-modified nn.Sequential (deferred execution) against layers.fuse_sequential built from the same
    modules — exact integers, so bit for bit — and against the module-by-module evaluation up to BatchNorm ties."""
    import random
    from pytorch_quantize_impls_amd.layers import fuse_sequential
    rng = random.Random(seed)
    torch.manual_seed(seed)
    cin = rng.choice([3, 16, 40, 64])
    H = rng.choice([24, 33, 40, 48])
    mods, c = [], cin
    first = True
    for _ in range(rng.randint(2, 4)):
        cout = rng.choice([24, 32, 64, 72, 128])
        k = rng.choice([1, 3, 3, 5])
        conv = rng.choice([BinConv2d, TerConv2d])(c, cout, k, stride=rng.choice([1, 1, 2]), padding=rng.choice([0, k // 2]))
        if first:
            conv.binary_input = False                     # real-valued pixels
            first = False
        mods.append(conv)
        pre_pool = rng.random() < 0.3
        if pre_pool:
            mods.append(nn.MaxPool2d(2, 2))
        mods.append(nn.BatchNorm2d(cout))
        if rng.random() < 0.7:
            mods.append(nn.Hardtanh(inplace=rng.random() < 0.5))
        mods.append(BinaryConnect())
        if not pre_pool and rng.random() < 0.3:
            mods.append(nn.MaxPool2d(2, 2))
        if rng.random() < 0.2:
            mods.append(nn.Dropout(0.3))
        c = cout
    mods.append(BinConv2d(c, 16, 1))
    seq = nn.Sequential(*mods)
    bench_models.randomize_bn(seq, seed)
    seq = seq.to(dev).eval()
    x = torch.randn(3, cin, H, H, device=dev)
    if rng.random() < 0.5:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        with lazy.eager():
            try:
                e = seq(x)
            except RuntimeError:                          # a map shrank below a kernel / pooling window
                pytest.skip("degenerate geometry")
        # (the explicit fused form has no Dropout module for packed activations; in eval mode it is the identity)
        fused = fuse_sequential(nn.Sequential(*[m for m in seq if not isinstance(m, nn.Dropout)]), fuse_conv=True,
                                packed_pool=True)
        ref = fused(x)
        lazy.STATS.clear()
        y = seq(x)
        assert isinstance(y, lazy.LazyActivation)
        got = y + 0
    assert lazy.STATS["fused"] >= 1
    assert torch.equal(got, ref), (seed, lazy.STATS)
    assert torch.equal(got, e), (seed, float((got - e).abs().max()))      # the eager graph on this device, bit for bit


# ---- dense roots: Linear -> BatchNorm1d -> [Hardtanh] -> BinaryConnect as one pass (lazy.DEFER_DENSE) -------------------------------------

@pytest.mark.parametrize("cls_name", ["LinearBin", "LinearTer", "LinearXNOR"])
@pytest.mark.parametrize("hardtanh,width", [(True, 256), (False, 256), (True, 257)])
def test_linear_bn_sign_chain_runs_as_one_pass_and_equals_the_module_chain(dev, cls_name, hardtanh, width):
    """The classifier pattern of the BinaryNet models (models/Alexnet/Alexnet_Bin.py:40-54, benchmark/BinaryNet/MLPBin.py): the fp32 result
    of an eval-mode quantised Linear is handed out deferred; BatchNorm1d -> [Hardtanh] -> BinaryConnect recorded on it run as ONE
    launch with this device's own BatchNorm thresholds and feed the next layer's packed GEMM — the same bits and the same logits as
    the module-by-module execution; a use outside the grammar reads the value."""
    from pytorch_quantize_impls_amd import layers as L
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    torch.manual_seed(5)
    cls = getattr(L, cls_name)
    mods = [BinaryConnect(), cls(300, width), nn.BatchNorm1d(width)] + ([nn.Hardtanh()] if hardtanh else []) + [BinaryConnect(), cls(width, 40)]
    seq = nn.Sequential(*mods).to(dev)
    bn = seq[2]
    bn.running_mean.normal_(0, 3.0)
    bn.running_var.uniform_(0.5, 2.0)
    bn.weight.data.normal_(0, 1.0)                      # negative slopes included
    bn.bias.data.normal_(0, 1.0)
    for m in seq:
        if isinstance(m, cls):
            m.weight.data.normal_(0, 0.3)
    seq.eval()
    x = torch.randn(64, 300, device=dev)
    with torch.no_grad():
        with lazy.eager():
            e = seq(x)
            e_mid = seq[:-1](x)
        lazy.STATS.clear()
        before = dict(_lib.call_counts)
        y = seq(x)
        mid = seq[:-1](x)
    assert type(y) is lazy.LazyDense and y._qt.kind == "dense"
    if width % 4 == 0:
        assert lazy.STATS["dense_fused"] == 1 and lazy.STATS["materialised"] == 0, lazy.STATS       # (`mid` was recorded, never used)
        one_pass = sum(_lib.call_counts.get(k, 0) - before.get(k, 0)
                       for k in ("qt_pool_affine_sign_pack_nhwc", "qt_pool_affine_sign_pack_nib_nhwc"))
        assert one_pass == 1                                         # (batch > 32: the variant that also writes the next GEMM's nibble rows)
    else:                                 # a width the one-pass kernel does not take (C % 4): the recorded chain runs module by module
        assert lazy.STATS["dense_fused"] == 0 and lazy.STATS["materialised"] > 0, lazy.STATS
    assert torch.equal(y, e)
    assert torch.equal(mid.value(), e_mid)                # the recorded chain's own value: sign(hardtanh(BatchNorm(.)))
    # outside the grammar: the value is simply there
    with torch.no_grad():
        z = seq[1](BinaryConnect()(x))
        assert torch.equal(z + 1.0, seq[1]._forward_impl(BinaryConnect()(x)) + 1.0)
        assert type(torch.relu(z)) is torch.Tensor


def test_graph_capture_of_a_model_that_ends_in_a_quantised_linear(dev):
    """The last layer's dense deferred activation is resolved inside the captured region (utils.graphed / utils.auto_graphed):
    replays return the eager logits."""
    from pytorch_quantize_impls_amd import utils
    torch.manual_seed(3)
    seq = nn.Sequential(BinaryConnect(), LinearBin(256, 128), nn.BatchNorm1d(128), nn.Hardtanh(), BinaryConnect(), LinearBin(128, 10)).to(dev)
    seq[2].running_mean.normal_(0, 2.0)
    seq.eval()
    x = torch.randn(32, 256, device=dev)
    with torch.no_grad():
        with lazy.eager():
            e = seq(x)
        assert type(seq(x)) is lazy.LazyDense
        g = utils.graphed(seq, x)
        assert type(g(x)) is torch.Tensor and torch.equal(g(x), e)
        a = utils.auto_graphed(seq)
        for _ in range(3):
            assert torch.equal(a(x), e)
