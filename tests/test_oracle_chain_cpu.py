"""CPU checks of the oracle's restatement of the eval-mode chain between binarised layers
(MaxPool2d -> BatchNorm2d(eval) -> Hardtanh -> BinaryConnect, models/Alexnet/Alexnet_Bin.py:14-17) and of
the identity the threshold-bit fusion rests on: max-pooling commutes with x -> fl(fl(x*a) + b)."""
import numpy as np
import torch

from pytorch_quantize_impls_amd.layers import fold_batchnorm


def test_oracle_chain_matches_torch_modules(oracle):
    torch.manual_seed(1)
    N, C, H = 2, 37, 11
    y = (torch.randn(N, C, H, H) * 25).round()
    bn = torch.nn.BatchNorm2d(C).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 40); bn.weight.data.normal_(); bn.bias.data.normal_()
    alpha, beta = fold_batchnorm(bn)
    for (k, s) in ((3, 2), (2, 2), (1, 1)):
        plane, (Ho, Wo) = oracle.pool_bn_sign_planes(y.numpy(), alpha.numpy(), beta.numpy(), k, s)
        with torch.no_grad():
            p = torch.nn.functional.max_pool2d(y, k, s) if k > 1 else y
            t = p * alpha.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
            ht = torch.nn.functional.hardtanh(t)
        assert (Ho, Wo) == tuple(p.shape[2:])
        want = oracle.sign_pack(ht.permute(0, 2, 3, 1).reshape(-1, C).numpy())
        assert np.array_equal(plane, want)
        assert np.array_equal(oracle.maxpool2d(y.numpy(), k, s), p.numpy())


def test_maxpool_commutes_with_the_folded_batchnorm():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((64, 9)) * 300).round().astype(np.float32)      # 64 windows of 9 conv outputs
    x[:8] += rng.standard_normal((8, 9)).astype(np.float32)                      # and some non-integers
    for a in np.float32([0.0371, -0.0371, 1.0, -2.5, 0.0, 3e-8, -7e5]):
        for b in np.float32([0.0, 0.41, -11.3, 5e4]):
            f = lambda v: (v * a).astype(np.float32) + b
            pooled_first = f(x.max(1)) < 0
            bits = f(x) < 0
            on_bits = bits.any(1) if a < 0 else bits.all(1)
            assert np.array_equal(pooled_first, on_bits), (a, b)


def test_oracle_dorefa_block_chain_matches_torch_modules(oracle):
    """oracle.affine_relu_dorefa_codes = folded BatchNorm + shortcut + ReLU + the reference quantiser
    (functions/dorefa_connect.py:24-25, restated by the package's CPU nnDorefaQuant, itself pinned to golden G3)."""
    from pytorch_quantize_impls_amd.functions import nnDorefaQuant
    torch.manual_seed(2)
    N, C, H = 2, 19, 5
    x = torch.randn(N, C, H, H) * 2
    r = torch.randn(N, C, H, H)
    bn, bn_r = torch.nn.BatchNorm2d(C).eval(), torch.nn.BatchNorm2d(C).eval()
    for b in (bn, bn_r):
        b.running_mean.normal_(); b.running_var.uniform_(0.5, 4); b.weight.data.normal_(); b.bias.data.normal_()
    (a, be), (ra, rb) = fold_batchnorm(bn), fold_batchnorm(bn_r)
    v = lambda p: p.view(1, -1, 1, 1)
    for k in (2, 4, 8):
        quant = nnDorefaQuant(k)
        n = float((1 << k) - 1)
        for relu in (True, False, "pre"):
            for res, aff in ((None, None), (r, None), (r, (ra, rb))):
                with torch.no_grad():
                    t = (torch.relu(x) if relu == "pre" else x) * v(a) + v(be)
                    if res is not None:
                        t = t + (res * v(ra) + v(rb) if aff is not None else res)
                    t = torch.relu(t) if relu is True else t
                    want = quant(t)
                q, y = oracle.affine_relu_dorefa_codes(x.numpy(), a.numpy(), be.numpy(), k, relu,
                                                       None if res is None else res.numpy(),
                                                       None if aff is None else (ra.numpy(), rb.numpy()))
                assert np.array_equal(y, want.numpy())
                assert np.array_equal(q, torch.round(n * t).numpy())
