"""Round-6 GPU tests.

  * batch-256 reference digests of the remaining AlexNet conv layers of config 3 (tests/golden/make_golden_r6.py, G24): conv2
    (K = 4800, the 384 x 192 tiles), conv4, conv5 — with `c3_binconv3_b256` (round 4) every binarised conv of
    models/Alexnet/Alexnet_Bin.py:12-39 is now digest-pinned at the config's own batch."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu

from pytorch_quantize_impls_amd import lazy, synth  # noqa: E402
from pytorch_quantize_impls_amd.layers import BinConv2d  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


def t32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


@pytest.mark.parametrize("name", ["c3_binconv2_b256", "c3_binconv4_b256", "c3_binconv5_b256"])
@pytest.mark.parametrize("mode", ["eval_deferred", "eval_eager", "train"])
def test_alexnet_conv_layers_reproduce_the_reference_digest_at_batch_256(dev, name, mode):
    """BinConv2d at AlexNet's conv2 / conv4 / conv5 (layers/binary_layers.py:103-106), batch 256, +-1 inputs: the SHA-256 of the
    fp32 result is the reference layer's (exact integer sums) through the deferred inference graph (materialised), the
    module-by-module eval path and the training-mode forward, channels-last storage."""
    with open(os.path.join(GOLDEN_DIR, "golden_hashes_r6.json")) as fh:
        c = json.load(fh)["cases"][name]
    B, Cin, Cout, H, k = c["B"], c["Cin"], c["Cout"], c["H"], c["k"]
    conv = BinConv2d(Cin, Cout, k, stride=c["stride"], padding=c["pad"]).to(dev)
    conv.weight.data.copy_(t32(synth.uniform(c["w_seed"], (Cout, Cin, k, k), -1.0, 1.0), dev))
    conv.bias.data.copy_(t32(np.round(synth.normal(c["b_seed"], (Cout,)) * 4), dev))
    x = t32(synth.pm1(c["x_seed"], (B, Cin, H, H)), dev).contiguous(memory_format=torch.channels_last)
    conv.train(mode == "train")
    with torch.no_grad():
        if mode == "eval_deferred":
            y = conv(x)
            assert isinstance(y, lazy.LazyActivation)
            y = y.value()
        else:
            with lazy.eager():
                y = conv(x)
    a = np.ascontiguousarray(y.detach().float().contiguous().cpu().numpy(), dtype=np.float32)
    assert a.shape == (B, Cout, H, H)
    assert hashlib.sha256(a.tobytes()).hexdigest() == c["sha256_f32_nchw"], (float(a.astype(np.float64).sum()), c["sum"])
