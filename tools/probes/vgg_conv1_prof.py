#!/usr/bin/env python
"""The first block of the fused ternary VGG-16 (TerConv2d 3 -> 64, 3 x 3, padding 1 on the fp32 image + BatchNorm + sign), batch 256,
10 calls — target for rocprofv3 --kernel-trace --stats (which kernels make up its 0.45 ms)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench_models
from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = bench_models.TernaryVGG16(num_classes=1000, image=224); bench_models.randomize_bn(m, seed=5)
m = m.to(dev).to(memory_format=torch.channels_last).eval()
m.features[0].binary_input = False
f = FusedFeatureClassifier(m.features, m.classifier, (512, 7, 7))
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
blk = list(f.features.children())[0]
with torch.no_grad():
    for _ in range(3): blk(x)
    torch.cuda.synchronize()
    for _ in range(10): blk(x)
torch.cuda.synchronize()
