"""Host logic of the deferred activations (lazy.py), on CPU: recording grammar, shapes, replay == torch.
A stand-in conv layer supplies the root value; the device-only gate (conv_forward) is checked to stay closed on CPU."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from pytorch_quantize_impls_amd import lazy
from pytorch_quantize_impls_amd.functions import BinaryConnect
from pytorch_quantize_impls_amd.layers import BinConv2d


class _Stub:
    training = False

    def _forward_impl(self, inp):
        return inp * 2


def _lazy(x):
    return lazy.LazyActivation(lazy._Node(None, None, x.shape, layer=_Stub(), kind="binary", input=x), x.device)


def _tail(C):
    seq = nn.Sequential(nn.MaxPool2d(2, 2), nn.BatchNorm2d(C), nn.Hardtanh(inplace=True), BinaryConnect()).eval()
    seq[1].running_mean.normal_()
    seq[1].running_var.uniform_(0.5, 2)
    return seq


def test_cpu_layers_never_defer():
    m = nn.Sequential(BinConv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8)).eval()
    with torch.no_grad():
        assert type(m(torch.randn(2, 3, 8, 8))) is torch.Tensor


def test_recording_and_replay():
    x = torch.randn(2, 3, 8, 8)
    seq = _tail(3)
    with torch.no_grad():
        out = seq(_lazy(x))
        assert isinstance(out, lazy.LazyActivation) and out.shape == (2, 3, 4, 4)
        n = out._qt
        assert n.signed and n.pool == (2, 2) and n.hardtanh == (-1.0, 1.0) and n.bn is not None and not n.flat
        ref = seq(x * 2)
        assert torch.equal(out.value(), ref)
        flat = out.reshape(out.size(0), -1)
        assert isinstance(flat, lazy.LazyActivation) and flat.shape == (2, 48) and flat._qt.chw == (3, 4, 4)
        assert torch.equal(flat.value(), ref.reshape(2, -1))
        assert torch.equal(torch.flatten(out, 1).value(), ref.reshape(2, -1))
        assert isinstance(F.max_pool2d(out, 2), lazy.LazyActivation)            # pool after the sign
        assert torch.equal(F.max_pool2d(out, 2, 2) + 0, F.max_pool2d(ref, 2, 2))
        assert BinaryConnect()(out) is out                                        # sign of a sign
        assert F.dropout(out, training=False) is out


def test_everything_else_materialises():
    x = torch.randn(2, 3, 8, 8)
    seq = _tail(3)
    with torch.no_grad():
        lz = _lazy(x)
        assert type(lz + 1) is torch.Tensor and torch.equal(lz + 1, x * 2 + 1)
        assert type(F.relu(lz)) is torch.Tensor
        assert type(lz.reshape(2, -1)) is torch.Tensor                            # flatten before BatchNorm: not in the grammar
        assert type(F.max_pool2d(lz, 3, 2, padding=1)) is torch.Tensor            # padded pooling
        bn = seq[1]
        assert type(F.batch_norm(lz, None, None, training=True)) is torch.Tensor
        t = seq[1](seq[0](_lazy(x)))
        assert isinstance(t, lazy.LazyActivation)
        assert type(seq[1](t)) is torch.Tensor                                    # a second BatchNorm
        assert type(F.hardtanh(t, 0.0, 1.0)) is torch.Tensor                      # does not keep the sign
        assert type(t.reshape(4, -1)) is torch.Tensor                             # not the (N, C*H*W) flattening
        assert torch.equal(torch.cat([lz, lz]), torch.cat([x * 2, x * 2]))
        assert lz.sum().item() == (x * 2).sum().item()
        assert len(lz) == 2 and lz.numel() == x.numel() and lz.ndim == 4 and not lz.is_cuda
        lz2 = _lazy(x)
        lz2 += 1
        assert torch.equal(lz2, x * 2 + 1)
    assert lazy.STATS["materialised"] > 0
