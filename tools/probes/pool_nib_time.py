#!/usr/bin/env python
"""qt_pool_bits_nib at the pooling layers of AlexNet (k3 s2) and VGG-16 (k2 s2), batch 256: run under tools/probes/kt_any.sh."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for C, H, k, s, halo in ((192, 55, 3, 2, 2), (576, 27, 3, 2, 1), (64, 224, 2, 2, 1), (128, 112, 2, 2, 1), (256, 56, 2, 2, 1), (512, 28, 2, 2, 1)):
    N = 256
    ld = ops.packed_ld(C)
    planes = ops.BitPlanes(sign=torch.randint(-2**31, 2**31 - 1, (N * H * H, ld), dtype=torch.int32, device=dev), rows=N * H * H, K=C)
    na = torch.zeros((ld,), dtype=torch.int32, device=dev)
    for _ in range(12):
        out, _ = ops.pool_bits_nib(planes, N, H, H, k, s, na, (halo, halo))
    torch.cuda.synchronize()
    print(C, H, "in MB", planes.sign.numel() * 4 / 1e6, "out MB", out.words.numel() * 4 / 1e6)
