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


def _check_planes(a, x, N, C, H, W, s, pad, floor_exp):
    """The plane means what the header says: a.scale[0] * (hi + lo) reproduces the space-to-depth image to
    max(2^-22 |x|, 2^floor_exp max|x|)."""
    dev = x.device
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    Hs, Ws = (H + 2 * ph + s - 1) // s, (W + 2 * pw + s - 1) // s
    xp = torch.zeros((N, C, Hs * s, Ws * s), device=dev, dtype=torch.float64)
    xp[:, :, ph:ph + H, pw:pw + W] = x.double()
    ref = xp.view(N, C, Hs, s, Ws, s).permute(0, 2, 4, 1, 3, 5).reshape(N * Hs * Ws, C * s * s)
    pairs = a.data.view(torch.float16).double().view(N * Hs * Ws, -1, 2)[:, :C * s * s]
    got = (pairs[..., 0] + pairs[..., 1]) * float(a.scale[0])
    amax = float(ref.abs().max())
    bound = torch.maximum(2.0 ** -22 * ref.abs(), torch.full_like(ref, 2.0 ** floor_exp * amax))
    assert bool(((got - ref).abs() <= bound).all()), float(((got - ref).abs() / bound.clamp_min(1e-300)).max())
    assert float(a.scale[0]) * float(a.scale[1]) == 1.0


@pytest.mark.parametrize("N,C,H,W,s,pad", [(2, 3, 224, 224, 4, 2), (3, 3, 30, 34, 4, 2), (2, 3, 17, 22, 4, 0), (1, 3, 9, 10, 4, (3, 4)),
                                           (5, 3, 4, 2, 4, (0, 2)), (40, 3, 112, 112, 4, 2),      # 1160 rows: the repack launch walks rows
                                           (2, 3, 31, 33, 4, 2), (2, 3, 32, 32, 4, 1), (2, 4, 32, 32, 4, 2), (2, 3, 32, 32, 2, 2)])
@pytest.mark.parametrize("amp", [3.0, 40.0, 1e-4, 3e5, 15.999, 16.0, 31.995])
def test_f16x2_s2d_pack_channels_last_vs_generic(dev, N, C, H, W, s, pad, amp):
    """Channels-last images take the speculative row-staging pack (fixed scale 2^-11 unless max|x| falls outside [2^-3, 2^4): then
    the device rewrites the plane with the exact-binade scale); NCHW storage takes max|x| pass + generic gather.  Same scale ->
    identical planes; either way the plane meets its bound."""
    gen = torch.Generator(device=dev).manual_seed(N * 1000 + H * 10 + W)
    x = torch.randn((N, C, H, W), device=dev, generator=gen).clamp_(-4, 4) * (amp / 4.0)
    x[0, 0, 0, 0] = amp                                     # max|x| = amp exactly
    xl = x.contiguous(memory_format=torch.channels_last)
    before = dict(_lib.call_counts)
    a, hw_a = ops.s2d_triple_pack(xl, s, pad, terms=2)
    b, hw_b = ops.s2d_triple_pack(x.contiguous(), s, pad, terms=2)
    assert _lib.call_counts["qt_f16x2_s2d_pack_spec_f32"] == before.get("qt_f16x2_s2d_pack_spec_f32", 0) + 1
    assert _lib.call_counts["qt_f16x2_s2d_pack_f32"] == before.get("qt_f16x2_s2d_pack_f32", 0) + 1
    assert hw_a == hw_b and a.data.shape == b.data.shape and a.terms == b.terms == 2
    in_window = 2.0 ** -3 <= amp < 2.0 ** 4     # r = max|x| / 2^-11 in [2^8, 2^15): fp16 overflows from 65520
    assert float(a.scale[0]) == (2.0 ** -11 if in_window else float(b.scale[0]))
    assert 2.0 ** 14 <= amp / float(b.scale[0]) < 2.0 ** 15
    if not in_window:
        assert torch.equal(a.data, b.data), (N, C, H, W, s, pad)
    _check_planes(a, x, N, C, H, W, s, pad, -33)
    _check_planes(b, x, N, C, H, W, s, pad, -39)
    assert bool(torch.isfinite(a.data.view(torch.float16)).all()) and bool(torch.isfinite(b.data.view(torch.float16)).all())
    # the speculation switched off: the two-pass form, identical to the generic kernel's planes
    old = ops.S2D_SPEC_SCALE
    ops.S2D_SPEC_SCALE = None
    try:
        c, _ = ops.s2d_triple_pack(xl, s, pad, terms=2)
    finally:
        ops.S2D_SPEC_SCALE = old
    assert torch.equal(c.data, b.data) and torch.equal(c.scale, b.scale)


def test_f16x2_s2d_speculative_pack_corner_images(dev):
    """All-zero images keep the fixed scale (nothing to represent); a NaN / inf anywhere makes the device fall back to scale 1,
    like the max|x| pass does; both without a host round trip."""
    z = torch.zeros((2, 3, 16, 16), device=dev).contiguous(memory_format=torch.channels_last)
    a, _ = ops.s2d_triple_pack(z, 4, 2, terms=2)
    assert float(a.scale[0]) == 2.0 ** -11 and not bool(a.data.any())
    for bad in (float("nan"), float("inf")):
        x = torch.randn((2, 3, 16, 16), device=dev)
        x[1, 2, 3, 4] = bad
        a, _ = ops.s2d_triple_pack(x.contiguous(memory_format=torch.channels_last), 4, 2, terms=2)
        b, _ = ops.s2d_triple_pack(x.contiguous(), 4, 2, terms=2)
        assert float(a.scale[0]) == float(b.scale[0]) == 1.0 and torch.equal(a.data, b.data)


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


def test_auto_graphed_replays_the_dorefa_resnet(dev):
    """DoReFa layers keep a ``weight_op`` child whose training flag never changes (upstream's train() / eval()): the eval-mode
    ResNet-18 is still recognised as replayable, and the replayed logits are the module graph's."""
    import bench_models
    from pytorch_quantize_impls_amd import utils
    torch.manual_seed(5)
    m = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(m, 4)
    m = m.to(dev).to(memory_format=torch.channels_last).eval()
    auto = utils.auto_graphed(m)
    x = torch.randn(32, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        want = m(x).clone()
        for _ in range(4):
            assert torch.equal(auto(x), want)
    assert auto.replays == 3 and not auto.capture_failures, (auto.replays, auto.capture_failures)


def test_bench_two_ranks_self_launched_on_one_device():
    """The whole N > 1 code path of bench.py on a 1-GPU box: `python bench.py --gpus 2 --share-device` with no launcher around it
    self-launches two ranks (gloo rendezvous, both on cuda:0), each runs its batch shard, rank 0 prints the single JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-device",
                        "--steps", "5", "--warmup", "2", "--no-extras", "--alexnet-batch", "0", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = lines[0]
    assert r["n_gpus"] == 2 and r["dist"]["world_size_seen_by_the_process_group"] == 2 and len(r["per_rank_ms_per_step"]) == 2
    assert r["config"]["global_batch"] == 2 * r["config"]["batch_per_gpu"] and r["value"] > 0
