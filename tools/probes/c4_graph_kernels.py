#!/usr/bin/env python
"""Kernels inside the captured hipGraph of the fused C4 forward: run under rocprofv3 --kernel-trace; 50 replays after capture, so
per-replay counts = calls / 50 for everything launched only by replays.  MODULE=1: the un-modified module graph instead."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import utils
dev = torch.device("cuda:0")
torch.manual_seed(4)
m4 = bench_models.DorefaResNet18(w_bits=1, a_bits=4); bench_models.randomize_bn(m4, seed=3)
for m in m4.modules():
    if isinstance(m, torch.nn.BatchNorm2d): m.running_var.mul_(4.0)
m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
x4 = torch.randn((256, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
net = m4 if os.environ.get("MODULE") == "1" else bench_models.FusedDorefaResNet18(m4, fold="device")
with torch.no_grad():
    g = utils.graphed(net, x4)
    torch.cuda.synchronize()
    for _ in range(int(os.environ.get("REPLAYS", "50"))): g(x4)
torch.cuda.synchronize()
