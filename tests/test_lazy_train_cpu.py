"""Host logic of lazy_train.py (the un-modified TRAINING graph reaching the fused training nodes): which calls are recorded,
what the consumers hand to the fused nodes, what happens outside the grammar.  The fused nodes themselves are HIP kernels
(tests/test_gpu_r3.py, test_gpu_r4.py); here they are replaced by stand-ins built from torch's own ops with the same call
signature, and the deferral condition (a HIP tensor) is lifted, so the recording can be followed on the CPU."""
import copy

import pytest
import torch
import torch.nn.functional as F

import bench_models
from pytorch_quantize_impls_amd import lazy_train
from pytorch_quantize_impls_amd.functions import BinaryConnect, nnDorefaQuant
from pytorch_quantize_impls_amd.functions.binary_connect import BinaryConnectDeterministic
from pytorch_quantize_impls_amd.layers import BinConv2d, DorefaConv2d, LinearBin, fused


@pytest.fixture()
def nodes(monkeypatch):
    calls = []

    class Sign:
        @staticmethod
        def apply(x, w, b, rm, rv, eps, mom, k, s, lo, hi):
            assert type(x) is torch.Tensor
            calls.append(("sign", k, s, lo, hi))
            h = F.max_pool2d(x, k, s) if k > 1 else x
            h = F.batch_norm(h, rm, rv, w, b, True, mom, eps)
            if lo > -1e30:
                h = F.hardtanh(h, lo, hi)
            return BinaryConnectDeterministic.apply(h)

    class Quant:
        @staticmethod
        def apply(x, res, w, b, rm, rv, eps, mom, relu, bits):
            assert type(x) is torch.Tensor and (res is None or type(res) is torch.Tensor)
            calls.append(("quant", bits, bool(relu), res is not None))
            h = F.batch_norm(x, rm, rv, w, b, True, mom, eps)
            if bits == 0:
                return h
            if res is not None:
                h = h + res
            if relu:
                h = torch.relu(h)
            return nnDorefaQuant(bits)(h)

    def wrap(layer, out):
        if lazy_train.enabled() and layer.training and torch.is_grad_enabled() and type(out) is torch.Tensor and out.requires_grad:
            return out.as_subclass(lazy_train.TrainOut)
        return out

    monkeypatch.setattr(fused, "_TrainPoolBnSignFn", Sign)
    monkeypatch.setattr(fused, "_TrainBnActQuantFn", Quant)
    monkeypatch.setattr(lazy_train, "wrap", wrap)
    lazy_train.STATS.clear()
    return calls


def _same_step(m, x, loss):
    m2 = copy.deepcopy(m)
    y = m(x)
    loss(y).backward()
    with lazy_train.eager():
        y2 = m2(x)
        loss(y2).backward()
    assert torch.equal(torch.as_tensor(y), y2)
    for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert (p.grad is None) == (q.grad is None), k
        if p.grad is not None:
            assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-6), k
    for (k, p), (_, q) in zip(m.named_buffers(), m2.named_buffers()):
        assert torch.equal(p, q), k


def test_alexnet_training_graph_records_seven_sign_chains(nodes):
    torch.manual_seed(0)
    m = bench_models.AlexNetBin(10, 1).train()
    _same_step(m, torch.randn(3, 3, 224, 224), lambda y: y.square().sum())
    kinds = [c for c in nodes if c[0] == "sign"]
    assert len(kinds) == 7 and len(nodes) == 7
    assert [c[1:3] for c in kinds] == [(3, 2), (3, 2), (1, 1), (1, 1), (3, 2), (1, 1), (1, 1)]
    assert all(c[3:] == (-1.0, 1.0) for c in kinds)
    assert lazy_train.STATS["fused:sign"] == 7 and not any(k.startswith("replayed") for k in lazy_train.STATS)


def test_resnet_training_graph_records_the_block_chains(nodes):
    torch.manual_seed(1)
    m = bench_models.DorefaResNet18().train()
    _same_step(m, torch.randn(4, 3, 32, 32), lambda y: y.square().sum())
    q = [c for c in nodes if c[0] == "quant"]
    assert sum(1 for c in q if c[1] == 4 and c[2] and not c[3]) == 8      # conv1 of every block: BatchNorm -> ReLU -> quantiser
    assert sum(1 for c in q if c[1] == 4 and c[2] and c[3]) == 8          # conv2: BatchNorm -> + shortcut -> ReLU -> quantiser
    assert sum(1 for c in q if c[1] == 0) == 3                            # the three shortcut BatchNorms, no quantiser
    assert lazy_train.STATS["fused:quant"] == 16 and lazy_train.STATS["replayed:bn"] == 3


def test_outside_the_grammar_the_recorded_calls_are_replayed(nodes):
    torch.manual_seed(2)
    conv, bn = BinConv2d(8, 8, 3, padding=1).train(), torch.nn.BatchNorm2d(8).train()
    bn2 = copy.deepcopy(bn)
    x = torch.randn(2, 8, 6, 6)
    h = bn(conv(x))
    assert type(h) is lazy_train.TrainChain and h.shape == (2, 8, 6, 6) and h.dim() == 4
    y = torch.sigmoid(F.hardtanh(h))              # no BinaryConnect: BatchNorm by the quantiser-less node, the rest by torch
    assert type(y) is torch.Tensor
    with lazy_train.eager():
        y2 = torch.sigmoid(F.hardtanh(bn2(conv(x))))
    assert torch.equal(y, y2) and torch.equal(bn.running_mean, bn2.running_mean)
    assert nodes == [("quant", 0, False, False)]
    # a chain used twice: the BatchNorm statistics are updated once
    nodes.clear()
    h = bn(conv(x))
    a, b = BinaryConnect()(h), h * 2.0
    with lazy_train.eager():
        h2 = bn2(conv(x))
    assert torch.equal(a, BinaryConnect()(h2)) and torch.allclose(b, h2 * 2.0, atol=1e-6)
    assert torch.allclose(bn.running_mean, bn2.running_mean) and torch.allclose(bn.running_var, bn2.running_var)
    assert int(bn.num_batches_tracked) == int(bn2.num_batches_tracked) == 2


def test_relu_before_batchnorm_and_eval_mode_are_left_alone(nodes):
    torch.manual_seed(3)
    lin, bn = LinearBin(16, 8).train(), torch.nn.BatchNorm1d(8).train()
    x = torch.randn(5, 16)
    h = torch.relu(lin(x))                         # benchmark/BinaryNet/MLPBin.py's order: ReLU first -> an ordinary tensor
    assert type(h) is torch.Tensor
    out = lin(x)
    assert type(out) is lazy_train.TrainOut and out.data_ptr() == lazy_train.resolve(out).data_ptr()
    assert type(BinaryConnect()(out)) is torch.Tensor       # nothing recorded: the plain op
    bn.eval()
    assert type(bn(lin(x))) is torch.Tensor        # eval-mode BatchNorm is not part of a training chain
    lin.eval()
    assert type(lin(x)) is torch.Tensor
    with torch.no_grad():
        assert type(lin.train()(x)) is torch.Tensor
    assert not nodes


def test_in_place_forms_and_the_residual_on_either_side(nodes):
    torch.manual_seed(4)
    conv = DorefaConv2d(8, 8, 3, padding=1, bias=False, bit_width=1).train()
    bn = torch.nn.BatchNorm2d(8).train()
    x, r = torch.rand(2, 8, 5, 5), torch.randn(2, 8, 5, 5)
    for form in ("iadd", "radd", "relu_"):
        bn2 = copy.deepcopy(bn)
        nodes.clear()
        h = bn(conv(x))
        if form == "iadd":
            h += r
            h = torch.relu(h)
        elif form == "radd":
            h = F.relu(r + h, inplace=True)
        else:
            h = h + r
            h.relu_()
        y = nnDorefaQuant(4)(h)
        assert nodes == [("quant", 4, True, True)], form
        with lazy_train.eager():
            y2 = nnDorefaQuant(4)(torch.relu(bn2(conv(x)) + r))
        assert torch.equal(y, y2), form


def _upstream_grads(consume, seed=5):
    """Gradients of conv weight / BatchNorm affine / input for  consume(bn(conv(x)))  with the recording on and off."""
    res = []
    for eager in (False, True):
        torch.manual_seed(seed)
        conv, bn = BinConv2d(8, 8, 3, padding=1).train(), torch.nn.BatchNorm2d(8).train()
        x = torch.randn(2, 8, 6, 6, requires_grad=True)
        torch.manual_seed(seed + 1)                    # the stochastic ops draw the same uniforms both times
        if eager:
            with lazy_train.eager():
                y = consume(bn(conv(x)))
        else:
            h = bn(conv(x))
            assert type(h) is lazy_train.TrainChain
            y = consume(h)
        assert type(y) is torch.Tensor and y.grad_fn is not None
        (y * torch.arange(y.numel(), dtype=torch.float32).reshape(y.shape)).sum().backward()
        res.append((y.detach(), conv.weight.grad, bn.weight.grad, bn.bias.grad, x.grad))
    return res


@pytest.mark.parametrize("name", ["bin_stochastic", "ter_det", "ter_stochastic", "quant_xnor", "lin_quant", "log_quant",
                                  "dorefa_quant_fn", "binary_dense", "xnor_dense"])
def test_every_package_function_keeps_the_graph_to_the_batchnorm(nodes, name):
    """Function.apply bypasses __torch_function__; a stand-in handed to any package Function other than
    BinaryConnectDeterministic / nnDorefaQuant used to enter forward as a no-grad leaf (grads upstream silently None)."""
    from pytorch_quantize_impls_amd import functions as Fn
    from pytorch_quantize_impls_amd.functions import binary_connect, terner_connect, xnor_connect, log_lin_connect, dorefa_connect
    w = torch.randn(5, 8 * 6 * 6)
    consume = {
        "bin_stochastic": lambda h: binary_connect.BinaryConnectStochastic.apply(h),
        "ter_det": lambda h: terner_connect.TernaryConnectDeterministic.apply(h),
        "ter_stochastic": lambda h: terner_connect.TernaryConnectStochastic.apply(h),
        "quant_xnor": lambda h: xnor_connect.QuantXnor(h.reshape(2, -1), 1),
        "lin_quant": lambda h: log_lin_connect.Quant(h, "lin", bit_width=4),
        "log_quant": lambda h: log_lin_connect.Quant(h, "log", bit_width=4),
        "dorefa_quant_fn": lambda h: dorefa_connect.DorefaQuant(h, 3),
        "binary_dense": lambda h: binary_connect.BinaryDense.apply(h.reshape(2, -1), w, None),
        "xnor_dense": lambda h: xnor_connect.XNORDense().apply(h.reshape(2, -1), w, None),
    }[name]
    (y, gw, gbw, gbb, gx), (y2, gw2, gbw2, gbb2, gx2) = _upstream_grads(consume)
    assert torch.equal(y, y2)
    for a, b in ((gw, gw2), (gbw, gbw2), (gbb, gbb2), (gx, gx2)):
        assert a is not None and b is not None, name
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), name


def test_a_foreign_function_consuming_a_stand_in_fails_loudly(nodes):
    class Sq(torch.autograd.Function):                 # a user Function: not derived from QtFunction
        @staticmethod
        def forward(ctx, t):
            ctx.save_for_backward(t)
            return t * t

        @staticmethod
        def backward(ctx, g):
            t, = ctx.saved_tensors
            return 2 * t * g

    torch.manual_seed(7)
    conv, bn = BinConv2d(8, 8, 3, padding=1).train(), torch.nn.BatchNorm2d(8).train()
    x = torch.randn(2, 8, 6, 6)
    y = Sq.apply(bn(conv(x)))
    assert y.requires_grad                           # the stand-in is a graph leaf now, not a constant
    with pytest.raises(RuntimeError, match="lazy_train.resolve"):
        y.sum().backward()
    # the two documented ways out
    y = Sq.apply(lazy_train.resolve(bn(conv(x))))
    y.sum().backward()
    g = conv.weight.grad.clone()
    conv.weight.grad = None
    with lazy_train.eager():
        Sq.apply(bn(conv(x))).sum().backward()
    assert torch.allclose(conv.weight.grad, g, rtol=1e-4, atol=1e-6)


def test_in_place_op_outside_the_grammar_does_not_touch_recorded_parents(nodes):
    """a = y + r recorded, then y.mul_(2): `a` must replay on the un-mutated y (eager computed it before)."""
    torch.manual_seed(8)
    conv, bn = BinConv2d(8, 8, 3, padding=1).train(), torch.nn.BatchNorm2d(8).train()
    bn2 = copy.deepcopy(bn)
    x, r = torch.randn(2, 8, 6, 6), torch.randn(2, 8, 6, 6)
    y = bn(conv(x))
    yv = torch.sigmoid(y)                              # forces (and caches) the BatchNorm value
    a = y + r                                          # recorded on the cached step
    y.mul_(2.0)                                        # out of grammar, in place
    with lazy_train.eager():
        y2 = bn2(conv(x))
        yv2 = torch.sigmoid(y2)
        a2 = y2 + r
        y2 = y2 * 2.0
    assert torch.equal(yv, yv2)
    assert torch.allclose(lazy_train.resolve(a) if type(a) is lazy_train.TrainChain else a, a2, atol=1e-6)
    assert torch.allclose(lazy_train.resolve(y), y2, atol=1e-6)


def test_recorded_add_keeps_the_residual_it_saw(nodes):
    """out = bn2(c2) + shortcut recorded; a later shortcut.relu_() must not change what the add resolves to."""
    torch.manual_seed(9)
    conv = DorefaConv2d(8, 8, 3, padding=1, bias=False, bit_width=1).train()
    conv_s = DorefaConv2d(8, 8, 1, bias=False, bit_width=1).train()
    bn, bn_s = torch.nn.BatchNorm2d(8).train(), torch.nn.BatchNorm2d(8).train()
    bnc, bn_sc = copy.deepcopy(bn), copy.deepcopy(bn_s)
    x = torch.rand(2, 8, 5, 5)
    short = bn_s(conv_s(x))
    out = bn(conv(x)) + short
    short.relu_()                                      # rebinds the stand-in after the add was recorded
    got = torch.sigmoid(out)
    with lazy_train.eager():
        s2 = bn_sc(conv_s(x))
        want = torch.sigmoid(bnc(conv(x)) + s2)
    assert torch.allclose(got, want, atol=1e-6)
