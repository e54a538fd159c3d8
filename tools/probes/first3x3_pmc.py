#!/usr/bin/env python
"""20 launches of the one-pass 3x3 first-layer kernel (VGG conv1_1 shape, batch 256, nibble epilogue) for rocprofv3 passes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
wq = torch.randint(-1, 2, (64, 3, 3, 3), device=dev).float()
frag = ops.pack_first3x3_weight(wq)
alpha, beta = torch.randn(64, device=dev), torch.randn(64, device=dev)
mode = os.environ.get("MODE", "nib")
epi = ops.NibEpilogue(alpha, beta, (1, 1)) if mode == "nib" else ((alpha, beta) if mode == "bits" else None)
for _ in range(20):
    y = ops.conv_first3x3(x, frag, 64, None, epi=epi)
torch.cuda.synchronize()
