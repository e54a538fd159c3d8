"""Converters, flat_net and the packed checkpoint — host logic, CPU only."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from pytorch_quantize_impls_amd import layers as L
from pytorch_quantize_impls_amd.layers.common import QLayer
from pytorch_quantize_impls_amd.utils import (binary_net_convert, convert, dorefa_net_convert, flat_net,
                                              load_packed_state_dict, packed_state_dict,
                                              packed_state_nbytes, ternary_net_convert, xnor_net_convert)
from pytorch_quantize_impls_amd.utils import packed_state as ps


def float_net():
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1, bias=False), nn.BatchNorm2d(8), nn.Hardtanh(),
                         nn.Sequential(nn.Conv2d(8, 5, 3, stride=2, padding=1), nn.ReLU()),
                         nn.Flatten(), nn.Linear(5 * 3 * 3, 11), nn.Linear(11, 7, bias=False))


def test_binary_convert_replaces_every_linear_and_conv_and_keeps_the_rest():
    net = float_net()
    b = binary_net_convert(net, deterministic=False)
    assert b is not net and isinstance(net[0], nn.Conv2d) and type(net[0]) is nn.Conv2d   # deep copy
    assert isinstance(b[0], L.BinConv2d) and isinstance(b[3][0], L.BinConv2d)
    assert isinstance(b[5], L.LinearBin) and isinstance(b[6], L.LinearBin)
    assert type(b[1]) is nn.BatchNorm2d and type(b[2]) is nn.Hardtanh
    assert b[0].bias is None and b[3][0].bias is not None and b[6].bias is None
    assert b[3][0].stride == (2, 2) and b[3][0].padding == (1, 1) and b[0].deterministic is False
    assert b[5].in_features == 45 and b[5].out_features == 11


def test_family_converters_and_kwargs():
    net = float_net()
    t = ternary_net_convert(net)
    assert isinstance(t[0], L.TerConv2d) and isinstance(t[5], L.LinearTer)
    d = dorefa_net_convert(net, weight_bit=2)       # upstream dies here (kwarg name), see convertor.py
    assert isinstance(d[0], L.DorefaConv2d) and isinstance(d[5], L.LinearDorefa)
    assert d[0].bit_width == 2 and d[5].bit_width == 2
    from pytorch_quantize_impls_amd.utils import log_lin_net_convert
    ll = log_lin_net_convert(net, fsr=2, bitwight=4, dtype="log")    # upstream's (misspelt) keyword, which crashes there
    assert isinstance(ll[0], L.QuantConv2d) and isinstance(ll[5], L.LinearQuant) and ll[5].bit_width == 4 and ll[0].qdtype == "log"
    x = xnor_net_convert(net, dim=[0, 1])
    assert isinstance(x[0], L.XNORConv2d) and isinstance(x[5], L.LinearXNOR)


def test_convert_matches_exact_class_only_and_layer_convert_type_errors():
    class MyLinear(nn.Linear):
        pass
    net = nn.Sequential(MyLinear(4, 4), nn.Linear(4, 2))
    b = binary_net_convert(net)
    assert type(b[0]) is MyLinear and isinstance(b[1], L.LinearBin)
    c = convert(net, {MyLinear: L.LinearTer})
    assert isinstance(c[0], L.LinearTer) and type(c[1]) is nn.Linear
    with pytest.raises(TypeError, match="Expected a torch.nn.Linear"):
        L.LinearBin.convert(nn.Conv2d(1, 1, 1))
    with pytest.raises(TypeError, match="Expected a torch.nn.Conv2d"):
        L.BinConv2d.convert(nn.Linear(1, 1))


def test_flat_net_order_and_whole_module_match():
    b = binary_net_convert(float_net())
    got = flat_net(b, QLayer)
    assert got == [b[0], b[3][0], b[5], b[6]]
    assert flat_net(b[5], QLayer) == [b[5]]
    assert flat_net(b, nn.Sequential) == [b]


def test_numpy_packer_equals_the_oracle_planes(oracle):
    rng = np.random.default_rng(3)
    for rows, K in ((1, 1), (3, 31), (5, 33), (2, 128), (4, 200)):
        x = rng.standard_normal((rows, K)).astype(np.float32)
        x[rng.random((rows, K)) < 0.2] = 0.0
        sign = ps._pack_bits_cpu(x < 0).numpy().view(np.uint32)
        assert np.array_equal(sign, oracle.sign_pack(x))
        t = oracle.ternarize(x)
        m_o, s_o = oracle.ternary_pack(x)
        assert np.array_equal(ps._pack_bits_cpu(t != 0).numpy().view(np.uint32), m_o)
        assert np.array_equal(ps._pack_bits_cpu(t < 0).numpy().view(np.uint32), s_o)
        back = ps._unpack_bits(torch.from_numpy(sign.view(np.int32)), K).numpy()
        assert np.array_equal(back, x < 0)


@pytest.mark.parametrize("family", ["binary", "ternary", "dorefa1"])
@pytest.mark.parametrize("from_mode", ["train", "eval"])
def test_packed_state_round_trip(family, from_mode):
    torch.manual_seed(7)
    conv = {"binary": binary_net_convert, "ternary": ternary_net_convert,
            "dorefa1": lambda n: dorefa_net_convert(n, weight_bit=1)}[family]
    m = conv(float_net())
    for p in m.parameters():
        p.data.uniform_(-1.2, 1.2)
    m[1].running_mean.normal_(); m[1].running_var.uniform_(0.5, 2.0)
    m.train(from_mode == "train")
    st = packed_state_dict(m)
    assert st["format"] == "qt-packed-v1" and set(st["layers"]) == {"0", "3.0", "5", "6"}
    assert all(e["kind"] == family for e in st["layers"].values())
    assert "0.weight" not in st["rest"] and "1.running_mean" in st["rest"]
    n_w = sum(p.numel() for n, p in m.named_parameters() if n.endswith("weight") and not n.startswith("1."))
    assert packed_state_nbytes(st) < 4 * n_w      # smaller than the fp32 weights alone (tiny net: row padding dominates)
    fresh = conv(float_net())
    load_packed_state_dict(fresh, st)
    assert not fresh.training
    m.eval()
    x = torch.randn(4, 3, 6, 6)
    assert torch.equal(fresh(x), m(x))
    for a, b in zip(flat_net(fresh, QLayer), flat_net(m, QLayer)):
        assert torch.equal(a.weight.data, b.weight.data)
    # toggling train()/eval() on the loaded model must not resurrect other weights
    ref = fresh(x).clone()
    fresh.train(); fresh.eval()
    if family == "dorefa1":   # re-quantising sign(W)*E recomputes E = mean|+-E|: equal up to fp32 summation rounding
        assert torch.allclose(fresh(x), ref, rtol=1e-5, atol=1e-5)
    else:
        assert torch.equal(fresh(x), ref)


def test_packed_state_errors():
    m = binary_net_convert(float_net())
    st = packed_state_dict(m)
    with pytest.raises(ValueError, match="qt-packed-v1"):
        load_packed_state_dict(m, {"format": "other"})
    with pytest.raises(TypeError, match="checkpoint holds a binary layer"):
        load_packed_state_dict(ternary_net_convert(float_net()), st)
    small = binary_net_convert(nn.Sequential(nn.Conv2d(3, 4, 3)))
    with pytest.raises(KeyError, match="lacks"):
        load_packed_state_dict(small, st)
    st2 = copy.deepcopy(st); st2["layers"]["5"]["shape"] = [11, 44]
    with pytest.raises(ValueError, match="weight shape"):
        load_packed_state_dict(binary_net_convert(float_net()), st2)


def test_packing_ratio_on_a_realistic_layer():
    m = nn.Sequential(L.LinearBin(1024, 512), L.LinearTer(1024, 512))
    st = packed_state_dict(m)
    assert st["layers"]["0"]["sign"].shape == (512, 32) and st["layers"]["1"]["mask"].shape == (512, 32)
    w_bytes = 512 * 1024 * 4   # per layer
    assert packed_state_nbytes(st) - 2 * 512 * 4 == w_bytes // 32 + w_bytes // 16   # 1 bit + 2 bits per weight


def test_auto_graphed_host_logic_on_cpu():
    """utils.auto_graphed keeps everything that cannot be replayed on the eager module: CPU tensors, training mode, autograd."""
    from pytorch_quantize_impls_amd.utils import AutoGraphed, auto_graphed
    m = binary_net_convert(float_net()).eval()
    a = auto_graphed(m, capture_after=2, max_graphs=3)
    assert isinstance(a, AutoGraphed) and a.module is m and (a.capture_after, a.max_graphs) == (2, 3)
    x = torch.randn(2, 3, 6, 6)
    with torch.no_grad():
        ref = m(x)
        for _ in range(4):
            assert torch.equal(a(x), ref)
    assert a.replays == 0 and a.eager_calls == 4 and not a._graphs          # a CPU tensor is never captured
    y = a(x)                                                                 # autograd enabled: the module itself
    assert y.requires_grad and a.eager_calls == 5
    a.train()
    assert m.training and a(x).requires_grad
    # the state signature follows version counters and storage; reset() forgets everything
    a.eval()
    s0 = a._state_sig()
    with torch.no_grad():
        next(m.parameters()).add_(1.0)
    assert a._state_sig() != s0
    a._seen[("k",)] = 3
    a.reset()
    assert not a._seen and a._state is None


def test_auto_graphed_training_check_ignores_the_layers_own_children():
    """The quantised layers' train() / eval() set the flag of the layer alone, like upstream (layers/binary_layers.py:30-40): a
    DoReFa layer's ``weight_op`` child reads training=True for ever.  AutoGraphed's "every module in eval mode" test looks at
    the layers and at everything around them (BatchNorm, Dropout, function modules), not inside the layers."""
    import bench_models
    from pytorch_quantize_impls_amd.utils.graphs import _any_training
    m = bench_models.DorefaResNet18(w_bits=1, a_bits=4).eval()
    assert any(mod.training for mod in m.modules())            # the weight_op children
    assert not _any_training(m)
    m.blocks[3].bn2.train()
    assert _any_training(m)
    m.eval()
    m.blocks[5].conv1.train()
    assert _any_training(m)
    assert not _any_training(m.eval()) and _any_training(m.train())


def test_detect_scope_and_float_split_are_thread_local():
    """the switches a capturing / exact-split thread opens are invisible to the thread beside it."""
    import threading
    from pytorch_quantize_impls_amd import ops
    from pytorch_quantize_impls_amd.functions import _fused
    seen = {}

    def other():
        seen["detect"], seen["split"] = _fused.detect_mode(), ops.current_float_split()

    with _fused.detect_scope("remember"), ops.float_split("bf16x3"):
        assert _fused.detect_mode() == "remember" and ops.current_float_split() == "bf16x3"
        t = threading.Thread(target=other)
        t.start()
        t.join()
        with _fused.detect_scope(None), ops.float_split(None):          # None = leave as is
            assert _fused.detect_mode() == "remember" and ops.current_float_split() == "bf16x3"
    assert seen == {"detect": _fused.DETECT_MODE, "split": ops.FLOAT_SPLIT}
    assert _fused.detect_mode() == _fused.DETECT_MODE and _fused.detect_mode_override() is None
    import pytest
    with pytest.raises(ValueError):
        with _fused.detect_scope("sometimes"):
            pass


def test_route_switch_scopes_are_thread_local_and_validated():
    """ops.scope / functions._fused.scope: per-thread overrides of the module-level route switches; the module
    attributes stay the process-wide defaults; an unknown name is an error; a second thread does not see the first one's scope."""
    import threading
    from pytorch_quantize_impls_amd import ops
    from pytorch_quantize_impls_amd.functions import _fused
    assert ops._cfg("FIRST_DIRECT") is True and _fused._cfg("GEMM_IMPL") == "auto"
    seen = {}
    with ops.scope(FIRST_DIRECT=False, POPC_VARIANT=3), _fused.scope(GEMM_IMPL="valu"):
        assert ops._cfg("FIRST_DIRECT") is False and ops._cfg("POPC_VARIANT") == 3 and _fused._cfg("GEMM_IMPL") == "valu"
        with ops.scope(FIRST_DIRECT=True):
            assert ops._cfg("FIRST_DIRECT") is True and ops._cfg("POPC_VARIANT") == 3          # nests
        assert ops._cfg("FIRST_DIRECT") is False
        t = threading.Thread(target=lambda: seen.update(a=ops._cfg("FIRST_DIRECT"), b=_fused._cfg("GEMM_IMPL")))
        t.start()
        t.join()
        assert ops.FIRST_DIRECT is True and _fused.GEMM_IMPL == "auto"                          # the defaults were not touched
    assert seen == {"a": True, "b": "auto"}
    assert ops._cfg("FIRST_DIRECT") is True and ops.scope_overrides() is None
    import pytest as _pt
    with _pt.raises(KeyError):
        with ops.scope(NOT_A_SWITCH=1):
            pass
    prev = ops.FIRST_DIRECT
    try:
        ops.FIRST_DIRECT = False                                                                 # the process-wide default still works
        assert ops._cfg("FIRST_DIRECT") is False
    finally:
        ops.FIRST_DIRECT = prev
