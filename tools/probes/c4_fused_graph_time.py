"""Fused C4 forward (batch 256) as a hipGraph: ms per forward (kernel-bound), for A/B of conv tile rules
(round 3: 128x128 tiles on the 8 x 8 stage instead of 64x64: 0.737 -> 0.672 ms, same digest)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
from pytorch_quantize_impls_amd import utils
dev = torch.device("cuda:0")
torch.manual_seed(0)
m4 = bench_models.DorefaResNet18(); bench_models.randomize_bn(m4, seed=3)
for m in m4.modules():
    if isinstance(m, torch.nn.BatchNorm2d): m.running_var.mul_(4.0)
m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
x4 = torch.randn((256, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
f4 = bench_models.FusedDorefaResNet18(m4, fold="device")
with torch.no_grad():
    y = f4(x4).clone()
g = utils.graphed(f4, x4)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ms = t(lambda: g(x4))
print(f"fused C4 as hipGraph: {ms:.4f} ms = {256 / ms * 1e3:.0f} img/s; digest {float(y.double().sum()):.6f} {float(y.double().abs().max()):.6f}")
