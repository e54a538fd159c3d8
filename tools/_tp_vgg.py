#!/usr/bin/env python
"""torch.profiler view of ONE fused AlexNet-Bin forward: which torch ops (copies, casts, fills) still run beside the
C-ABI kernels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.TernaryVGG16(num_classes=1000, image=224); bench_models.randomize_bn(model, seed=5)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((64, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
model.features[0].binary_input = False
fused = FusedFeatureClassifier(model.features, model.classifier, (512, 7, 7))
with torch.no_grad():
    for _ in range(5): fused(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fused(x)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
