#!/usr/bin/env python
"""Kernels of XNORConv2d forward at AlexNet conv2 / conv3, quant_input False (MODE=0) or True (MODE=1): run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pytorch_quantize_impls_amd.functions import xnor_connect, BinaryConnectDeterministic
dev = torch.device("cuda:0")
mode = os.environ.get("MODE", "1") == "1"
for name, Cin, Cout, H, k, p in (("conv2", 192, 576, 27, 5, 2), ("conv3", 576, 1152, 13, 3, 1)):
    torch.manual_seed(0)
    x = torch.randn(256, Cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    f = xnor_connect.XNORConv2d([0, 1], mode, 1, p, 1, 1)
    inp = x if mode else BinaryConnectDeterministic.apply(x)
    with torch.no_grad():
        for _ in range(20):
            f.apply(inp, w)
    torch.cuda.synchronize()
