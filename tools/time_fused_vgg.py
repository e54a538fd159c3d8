#!/usr/bin/env python
"""Per-module wall time of the fused ternary VGG-16 forward (C5), batch from argv (default 64)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
m = bench_models.TernaryVGG16(num_classes=1000, image=224); bench_models.randomize_bn(m, seed=5)
m = m.to(dev).to(memory_format=torch.channels_last).eval()
m.features[0].binary_input = False
f = FusedFeatureClassifier(m.features, m.classifier, (512, 7, 7))
x = torch.randn((B, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3): f(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f(x)
    torch.cuda.synchronize(); print(f"fused VGG-16, batch {B}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms / forward")
    h = x
    mods = list(f.features.children()) + list(f.classifier.children())
    for i, mod in enumerate(mods):
        if i == len(f.features): h = h.flatten_hwc()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out = mod(h)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
        desc = type(mod).__name__
        if hasattr(mod, "conv"): desc += f" {mod.conv.in_channels}->{mod.conv.out_channels} @{h.shape[2]}"
        print(f"   {desc:44s} {dt:7.3f} ms")
        h = out
