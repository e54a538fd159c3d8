#!/usr/bin/env python
"""Driver for rocprofv3 passes over the direct first-layer conv (AlexNet conv1, batch 256): a handful of launches of each epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
from pytorch_quantize_impls_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(os.environ.get("B", "256"))
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(192, 3, 11, 11, device=dev).sign()
fw = ops.pack_first_layer_weight(w, 4)
fwr = ops.pack_first_layer_weight(w * 0.037, 4, real=True)
al, be = torch.ones(192, device=dev), torch.zeros(192, device=dev)
for _ in range(5):
    ops.conv_first_direct(x, fw, None, 4, 2, epi=(al, be))
for _ in range(5):
    ops.conv_first_direct(x, fwr, None, 4, 2, epi=(al, be))
torch.cuda.synchronize()
