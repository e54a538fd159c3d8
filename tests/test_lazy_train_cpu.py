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
        if lazy_train.ENABLED and layer.training and torch.is_grad_enabled() and type(out) is torch.Tensor and out.requires_grad:
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
