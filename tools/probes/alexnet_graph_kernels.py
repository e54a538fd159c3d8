#!/usr/bin/env python
"""Kernels inside the captured hipGraph of the un-modified BinaryNet-AlexNet forward (batch 256): under rocprofv3 --kernel-trace,
50 replays.  XNOR=1: the XNOR-Net flavour."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import utils
dev = torch.device("cuda:0")
torch.manual_seed(1234)
if os.environ.get("XNOR") == "1":
    from pytorch_quantize_impls_amd.layers import XNORConv2d, LinearXNOR
    m = bench_models.alexnet_xnor()
    for mod in m.modules():
        if isinstance(mod, (XNORConv2d, LinearXNOR)):
            mod.weight.data.normal_(0, 0.05); mod.bias.data.zero_()
else:
    m = bench_models.AlexNetBin()
bench_models.randomize_bn(m)
m = m.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    g = utils.graphed(m, x)
    torch.cuda.synchronize()
    for _ in range(50): g(x)
torch.cuda.synchronize()
