#!/usr/bin/env python
"""The batch splits ops.BATCH_SPLIT measures for the fused AlexNet / VGG-16 forwards at batch 256 (two eager forwards each)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(1)
for name, ctor in (("alexnet", bench_models.AlexNetBin), ("vgg16", bench_models.TernaryVGG16)):
    if ctor is None:
        continue
    m = ctor()
    bench_models.randomize_bn(m)
    m = m.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
    before = set(ops.batch_split_choices())
    with torch.no_grad():
        for _ in range(2):
            m(x)
    torch.cuda.synchronize()
    for k, v in ops.batch_split_choices().items():
        if k not in before:
            print(name, k[0], "N", k[3], "HxW", k[4], k[5], "k", k[7], "Cout", k[15], "->", v)
