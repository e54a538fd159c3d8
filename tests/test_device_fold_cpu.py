"""layers.fused.device_sign_fold: per-channel thresholds bisected on the F.batch_norm the device really evaluates
(here: the CPU's) reproduce its sign for every fp32 input — the property that makes deferred activations bit-identical
to the module-by-module graph.  The GPU twin is tests/test_gpu_lazy.py (torch.equal against lazy.eager())."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pytorch_quantize_impls_amd.layers import fused


def _bn(C, seed, dims=2):
    g = torch.Generator().manual_seed(seed)
    bn = (torch.nn.BatchNorm2d if dims == 4 else torch.nn.BatchNorm1d)(C, eps=1e-4).eval()
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(C, generator=g) * 3)
        bn.running_var.copy_(torch.rand(C, generator=g) * 4 + 0.01)
        bn.weight.copy_(torch.randn(C, generator=g))
        bn.bias.copy_(torch.randn(C, generator=g))
        bn.weight[0] = 0.0                    # constant channels of either sign
        bn.bias[0] = -0.25
        bn.weight[1] = 0.0
        bn.bias[1] = 0.5
        bn.weight[2] = -1e-3                  # shallow negative slope
    return bn


@pytest.mark.parametrize("shape,cl", [((5, 24), False), ((3, 24, 6, 6), False), ((3, 24, 6, 6), True)])
def test_threshold_form_equals_batchnorm_sign_around_every_threshold(shape, cl):
    C = shape[1]
    bn = _bn(C, 7, len(shape))
    alpha, beta = fused.device_sign_fold(bn, shape, cl)
    assert set(alpha.tolist()) <= {-1.0, 0.0, 1.0}
    # candidates: the thresholds themselves, +-4 ulps around them, and random values
    theta = torch.where(alpha > 0, -beta, beta)
    keys = fused._float_key(theta)
    cands = [fused._key_float(keys + d) for d in range(-4, 5)]
    g = torch.Generator().manual_seed(1)
    cands += [torch.randn(C, generator=g) * s for s in (0.01, 1.0, 30.0, 1e4)]
    cands += [torch.zeros(C), -torch.zeros(C)]
    for v in cands:
        if len(shape) == 2:
            inp = v.unsqueeze(0).expand(shape).contiguous()
            ref = F.batch_norm(inp, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)[0] < 0
        else:
            inp = v.view(1, C, 1, 1).expand(shape).contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
            ref = F.batch_norm(inp, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)[1, :, 3, 2] < 0
        got = (v * alpha + beta) < 0          # alpha is +-1 / 0: the product is exact, the kernel's two-rounding form
        assert torch.equal(got, ref), (v[got != ref], theta[got != ref])
    assert bool(((torch.zeros(C) * alpha + beta) < 0)[0]) and not bool(((torch.zeros(C) * alpha + beta) < 0)[1])


def test_integer_thresholds_follow_the_device_fold():
    """+-1 operands: the conv accumulator is an integer, the fused epilogue compares it with ops.integer_thresholds of the
    fold — for the device fold that is exactly [F.batch_norm(acc + bias) < 0]."""
    from pytorch_quantize_impls_amd import ops
    C, K = 16, 300
    bn = _bn(C, 3, 4)
    bias = torch.randn(C)
    alpha, beta = fused.device_sign_fold(bn, (2, C, 5, 5), True)
    thr = ops.integer_thresholds(bias, alpha, beta, K)
    acc = torch.arange(-K, K + 1, dtype=torch.float32)
    x = (acc.view(-1, 1) + bias.view(1, -1))                                 # what the conv stores: fl(acc + bias)
    ref = F.batch_norm(x.view(-1, C, 1, 1).contiguous(memory_format=torch.channels_last), bn.running_mean, bn.running_var,
                       bn.weight, bn.bias, False, 0.0, bn.eps).view(-1, C) < 0
    got = (acc.view(-1, 1) < thr.view(1, -1)) ^ (alpha < 0).view(1, -1)
    const = alpha == 0
    got[:, const] = (beta[const] < 0).view(1, -1).expand(acc.numel(), -1)
    assert torch.equal(got, ref)


def test_float_key_is_monotone_and_invertible():
    v = torch.tensor([-np.inf, -3e38, -1.0, -1e-45, -0.0, 0.0, 1e-45, 1.0, 3e38, np.inf], dtype=torch.float32)
    k = fused._float_key(v)
    assert torch.all(k[1:] > k[:-1])
    assert torch.equal(fused._key_float(k).view(torch.int32), v.view(torch.int32))
