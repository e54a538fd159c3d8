#!/usr/bin/env python
"""Fused AlexNet-Bin forward only (batch 256, 10 forwards) — target for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.AlexNetBin(); bench_models.randomize_bn(model)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
fused = bench_models.FusedAlexNetBin(model)
with torch.no_grad():
    for _ in range(10): fused(x)
torch.cuda.synchronize()
