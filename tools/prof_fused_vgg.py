#!/usr/bin/env python
"""Fused ternary VGG-16 forward only (batch C5_BATCH, default 64) — target for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
dev = torch.device("cuda:0")
B = int(os.environ.get("C5_BATCH", "64"))
torch.manual_seed(0)
m5 = bench_models.TernaryVGG16(num_classes=1000, image=224); bench_models.randomize_bn(m5, seed=5)
m5 = m5.to(dev).to(memory_format=torch.channels_last).eval()
m5.features[0].binary_input = False
x5 = torch.randn((B, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
f5 = FusedFeatureClassifier(m5.features, m5.classifier, (512, 7, 7))
with torch.no_grad():
    for _ in range(int(os.environ.get("ITERS", "10"))): f5(x5)
torch.cuda.synchronize()
