#!/usr/bin/env python
"""C4 (DoReFa ResNet-18 W1A4, batch 256) eval forward (FUSED=1: fused inference form) under rocprofv3 --kernel-trace --stats: where the time goes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import ops
ops.PAD_PIXEL_PLANES = os.environ.get("PAD", "0") == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
m4 = bench_models.DorefaResNet18(); bench_models.randomize_bn(m4, seed=3)
m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
x4 = torch.randn((256, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
for m in m4.modules():
    if isinstance(m, torch.nn.BatchNorm2d): m.running_var.mul_(4.0)
net = bench_models.FusedDorefaResNet18(m4, fold=os.environ.get("FOLD") or None) if os.environ.get("FUSED", "0") == "1" else m4
with torch.no_grad():
    for _ in range(int(os.environ.get("ITERS", "13"))): net(x4)
torch.cuda.synchronize()
