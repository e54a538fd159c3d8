"""Round-3 (late) GPU parity: kernels added after tests/test_gpu_r3.py was closed.

  * the space-to-depth fp16-pair pack of the first layer (AlexNet: models/Alexnet/Alexnet_Bin.py:13): the row-staging kernel
    (channels-last images) and the generic gather (NCHW storage) write identical planes, and the planes meet the stated bound;
  * the streaming popcount GEMM (csrc/popc_stream.hip: K along the lanes, DPP wavefront reduction) for batch <= 32 / classifier
    heads (layers/binary_layers.py:42-46, terner_layers.py:47-51): bit-exact against the oracle and the tiled kernel."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import _lib, ops  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


@pytest.mark.parametrize("N,C,H,W,s,pad", [(2, 3, 224, 224, 4, 2), (3, 3, 30, 34, 4, 2), (2, 3, 17, 22, 4, 0), (1, 3, 9, 10, 4, (3, 4)),
                                           (5, 3, 4, 2, 4, (0, 2)),
                                           (2, 3, 31, 33, 4, 2), (2, 3, 32, 32, 4, 1), (2, 4, 32, 32, 4, 2), (2, 3, 32, 32, 2, 2)])
def test_f16x2_s2d_pack_row_staged_kernel_equals_generic(dev, N, C, H, W, s, pad):
    gen = torch.Generator(device=dev).manual_seed(N * 1000 + H * 10 + W)
    x = torch.randn((N, C, H, W), device=dev, generator=gen) * 3.0
    x[0, 0, 0, 0] = 37.5                                    # the per-tensor scale is not 1
    xl = x.contiguous(memory_format=torch.channels_last)
    before = _lib.call_counts.get("qt_f16x2_s2d_pack_f32", 0)
    a, hw_a = ops.s2d_triple_pack(xl, s, pad, terms=2)
    b, hw_b = ops.s2d_triple_pack(x.contiguous(), s, pad, terms=2)
    assert _lib.call_counts.get("qt_f16x2_s2d_pack_f32", 0) == before + 2
    assert hw_a == hw_b and a.data.shape == b.data.shape and a.terms == b.terms == 2
    assert torch.equal(a.scale, b.scale)
    assert torch.equal(a.data, b.data), (N, C, H, W, s, pad)
    # and the planes mean what the header says: s * (hi + lo) reproduces the image to 2^-22 relative
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    Hs, Ws = hw_a
    xp = torch.zeros((N, C, Hs * s, Ws * s), device=dev, dtype=torch.float64)
    xp[:, :, ph:ph + H, pw:pw + W] = x.double()
    ref = xp.view(N, C, Hs, s, Ws, s).permute(0, 2, 4, 1, 3, 5).reshape(N * Hs * Ws, C * s * s)
    pairs = a.data.view(torch.float16).double().view(N * Hs * Ws, -1, 2)[:, :C * s * s]
    got = (pairs[..., 0] + pairs[..., 1]) * float(a.scale[0])
    assert float((got - ref).abs().max()) <= 2.0 ** -22 * float(ref.abs().max())


# ---- streaming popcount GEMM (csrc/popc_stream.hip): K along the lanes + DPP wavefront reduction -------------------------------

import numpy as np  # noqa: E402
from pytorch_quantize_impls_amd import synth  # noqa: E402

STREAM_SHAPES = [(1, 1, 1), (1, 4096, 4096), (1, 4096, 9216), (2, 1000, 4096), (3, 70, 100), (5, 7, 31), (5, 7, 33), (8, 4097, 64),
                 (9, 300, 1000), (16, 4096, 4096), (17, 33, 25088), (32, 4096, 9216), (32, 32, 96),           # few = the batch
                 (256, 10, 4096), (64, 10, 4096), (33, 1, 100), (200, 2, 9216), (1000, 8, 33), (70, 9, 10000), (130, 17, 784),
                 (4099, 32, 4096)]                                                                               # few = the head


def _forced(variant, fn):
    ops.POPC_VARIANT = variant
    try:
        return fn()
    finally:
        ops.POPC_VARIANT = 0


@pytest.mark.parametrize("M,N,K", STREAM_SHAPES)
@pytest.mark.parametrize("with_bias", [False, True])
def test_streaming_xnor_gemm_vs_oracle_and_tiled_kernel(dev, oracle, M, N, K, with_bias):
    x = synth.pm1(M * 7 + K, (M, K))
    w = synth.uniform(N * 5 + K, (N, K), -1, 1)
    b = synth.normal(N, (N,)) if with_bias else None
    xp, wp = ops.sign_pack(torch.from_numpy(x).to(dev))[0], ops.sign_pack(torch.from_numpy(w).to(dev))[0]
    bd = None if b is None else torch.from_numpy(b).to(dev)
    before = _lib.call_counts.get("qt_xnor_gemm_variant", 0)
    y3 = _forced(3, lambda: ops.xnor_gemm(xp, wp, bd))
    y1 = _forced(1, lambda: ops.xnor_gemm(xp, wp, bd))
    assert _lib.call_counts.get("qt_xnor_gemm_variant", 0) == before + 2
    y0 = ops.xnor_gemm(xp, wp, bd)                       # the automatic choice takes the streaming kernel at these shapes
    assert torch.equal(y3, y1) and torch.equal(y0, y1)
    want = oracle.xnor_gemm(oracle.sign_pack(x), oracle.sign_pack(w), K)
    if b is not None:
        want = want + b[None, :]
    assert np.array_equal(y3.cpu().numpy(), want.astype(np.float32))


@pytest.mark.parametrize("M,N,K", STREAM_SHAPES)
def test_streaming_tern_gemm_vs_oracle_and_tiled_kernel(dev, oracle, M, N, K):
    x = synth.pm1(M * 3 + K, (M, K))
    w = synth.uniform(N * 9 + K, (N, K), -1.5, 1.5)
    b = synth.normal(N + 1, (N,))
    xp, wp = ops.sign_pack(torch.from_numpy(x).to(dev))[0], ops.ternary_pack(torch.from_numpy(w).to(dev))
    bd = torch.from_numpy(b).to(dev)
    y3 = _forced(3, lambda: ops.tern_gemm(xp, wp, bd))
    y1 = _forced(1, lambda: ops.tern_gemm(xp, wp, bd))
    assert torch.equal(y3, y1) and torch.equal(ops.tern_gemm(xp, wp, bd), y1)
    want = oracle.linear(x, oracle.ternarize(w)) + b[None, :]
    assert np.array_equal(y3.cpu().numpy(), want.astype(np.float32))


def test_streaming_kernel_refuses_shapes_with_two_large_dimensions(dev):
    xp = ops.sign_pack(torch.ones((40, 64), device=dev))[0]
    wp = ops.sign_pack(torch.ones((33, 64), device=dev))[0]
    with pytest.raises(Exception):
        _forced(3, lambda: ops.xnor_gemm(xp, wp))


# ---- utils.auto_graphed: the un-modified model with its inference forwards replayed as hipGraphs ---------------------------------

def test_auto_graphed_replays_the_unmodified_model_and_follows_weight_updates(dev):
    import copy, os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench_models
    from pytorch_quantize_impls_amd import utils
    torch.manual_seed(3)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    auto = utils.auto_graphed(model)
    gen = torch.Generator(device=dev).manual_seed(11)
    xs = [torch.randn((2, 3, 224, 224), device=dev, generator=gen).contiguous(memory_format=torch.channels_last) for _ in range(4)]
    with torch.no_grad():
        want = [model(x).clone() for x in xs]
        got = [auto(x) for x in xs]
        assert all(torch.equal(a, b) for a, b in zip(got, want))
        assert auto.replays == 3 and auto.eager_calls == 1            # first call eager, captured on the second
        # another signature starts eagerly again, the first one keeps replaying
        x8 = torch.randn((8, 3, 224, 224), device=dev, generator=gen).contiguous(memory_format=torch.channels_last)
        assert torch.equal(auto(x8), model(x8)) and auto.eager_calls == 2
        assert torch.equal(auto(xs[0]), want[0]) and auto.replays == 4
        # a weight update (load_state_dict bumps the version counters) drops the graphs: results follow the new weights
        sd = copy.deepcopy(model.state_dict())
        for k in sd:
            if k.endswith("bias") and sd[k].dtype == torch.float32:
                sd[k] = sd[k] + 0.25
        model.load_state_dict(sd)
        new = model(xs[1]).clone()
        assert not torch.equal(new, want[1])
        assert torch.equal(auto(xs[1]), new) and torch.equal(auto(xs[1]), new) and torch.equal(auto(xs[2]), model(xs[2]))
    # autograd enabled / training mode: the eager module
    before = auto.eager_calls
    y = auto(xs[0])
    assert auto.eager_calls == before + 1 and torch.equal(y.detach(), model(xs[0]).detach())
