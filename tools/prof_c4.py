#!/usr/bin/env python
"""C4 (DoReFa ResNet-18 W1A4, batch 256) eval forward under rocprofv3 --kernel-trace --stats: where the time goes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
m4 = bench_models.DorefaResNet18(); bench_models.randomize_bn(m4, seed=3)
m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
x4 = torch.randn((256, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(13): m4(x4)
torch.cuda.synchronize()
