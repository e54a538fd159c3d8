#!/usr/bin/env python
"""A/B of ops.PAD_PIXEL_PLANES (physical zero border + un-padded conv kernels) on the un-fused networks."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    best = 1e9
    for _ in range(3):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
torch.manual_seed(0)
m4 = bench_models.DorefaResNet18(); bench_models.randomize_bn(m4, seed=3)
m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
x4 = torch.randn((256, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
ma = bench_models.AlexNetBin(); bench_models.randomize_bn(ma, seed=1)
ma = ma.to(dev).to(memory_format=torch.channels_last).eval(); ma.features[0].binary_input = False
xa = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
m5 = bench_models.TernaryVGG16(num_classes=1000, image=224); bench_models.randomize_bn(m5, seed=5)
m5 = m5.to(dev).to(memory_format=torch.channels_last).eval(); m5.features[0].binary_input = False
x5 = torch.randn((64, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for flag in (False, True, False, True):
        ops.PAD_PIXEL_PLANES = flag
        print(f"PAD_PIXEL_PLANES={flag}: C4 {t(lambda: m4(x4)):.3f} ms  AlexNet-Bin {t(lambda: ma(xa)):.3f} ms  "
              f"VGG16 {t(lambda: m5(x5), n=5):.3f} ms", flush=True)
