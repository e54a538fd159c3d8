"""Host logic of the opt-in training-chain modules (layers.FusedTrainPoolBnSign / FusedTrainBnActQuant): on CPU tensors, in eval
mode or outside the kernels' shapes they ARE the reference's module chain (models/Alexnet/Alexnet_Bin.py:13-17,
models/Resnet/Resnet_bin.py:63-97) — same values, same gradients, same running statistics."""
import copy

import pytest
import torch

import bench_models
from pytorch_quantize_impls_amd.functions import nnDorefaQuant
from pytorch_quantize_impls_amd.layers import FusedTrainBnActQuant


@pytest.mark.parametrize("bits,relu,res", [(4, True, True), (2, True, False), (0, False, False)])
def test_bn_act_quant_falls_back_to_the_module_chain_on_cpu(bits, relu, res):
    torch.manual_seed(bits)
    bn = torch.nn.BatchNorm2d(8)
    ref_bn = copy.deepcopy(bn)
    mod = FusedTrainBnActQuant(bn, bits, relu=relu).train()
    x = torch.randn(3, 8, 5, 5, requires_grad=True)
    r = torch.randn(3, 8, 5, 5, requires_grad=True) if res else None
    y = mod(x, residual=r)
    y.sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    h = ref_bn(x2)
    if res:
        h = h + r.detach()
    if relu:
        h = torch.relu(h)
    if bits:
        h = nnDorefaQuant(bits)(h)
    h.sum().backward()
    assert torch.equal(y, h) and torch.equal(x.grad, x2.grad)
    assert torch.equal(bn.running_mean, ref_bn.running_mean) and torch.equal(bn.running_var, ref_bn.running_var)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1


def test_bn_act_quant_rejects_bit_widths_without_int8_codes():
    with pytest.raises(ValueError):
        FusedTrainBnActQuant(torch.nn.BatchNorm2d(4), 9)


def test_train_fused_resnet_shares_parameters_and_matches_the_model_on_cpu():
    torch.manual_seed(0)
    m = bench_models.DorefaResNet18(w_bits=1, a_bits=4).train()
    f = bench_models.TrainFusedDorefaResNet18(m)
    assert {id(p) for p in f.parameters()} == {id(p) for p in m.parameters()}
    x = torch.randn(2, 3, 32, 32)
    m2 = copy.deepcopy(m)
    assert torch.equal(f(x), m2(x))
    for (k, a), (_, b) in zip(m.named_buffers(), m2.named_buffers()):
        assert torch.equal(a, b), k
