#!/usr/bin/env python
"""Informative throughput of the C4 (DoReFa ResNet-18 W1A4, 3x32x32) and C5 (ternary VGG-16, 3x224x224) eval forwards."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import _lib
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
torch.manual_seed(0)
m4 = bench_models.DorefaResNet18(); bench_models.randomize_bn(m4, seed=3)
for m in m4.modules():
    if isinstance(m, torch.nn.BatchNorm2d): m.running_var.mul_(4.0)
m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
x4 = torch.randn((256, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    b = dict(_lib.call_counts); m4(x4); used = {k: v - b.get(k, 0) for k, v in _lib.call_counts.items() if v - b.get(k, 0)}
    ms = t(lambda: m4(x4))
print(f"C4 DoReFa ResNet-18 W1A4, batch 256: {ms:.3f} ms/forward = {256 / ms * 1e3:.0f} img/s   calls {used}")
B5 = int(os.environ.get("C5_BATCH", "64"))
m5 = bench_models.TernaryVGG16(num_classes=1000, image=224); bench_models.randomize_bn(m5, seed=5)
m5 = m5.to(dev).to(memory_format=torch.channels_last).eval()
m5.features[0].binary_input = False
x5 = torch.randn((B5, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    ms = t(lambda: m5(x5), n=5)
print(f"C5 ternary VGG-16, batch {B5}: {ms:.3f} ms/forward = {B5 / ms * 1e3:.0f} img/s (un-fused reference graph)")
from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
f5 = FusedFeatureClassifier(m5.features, m5.classifier, (512, 7, 7))
with torch.no_grad():
    same = torch.equal(f5(x5).argmax(1), m5(x5).argmax(1))
    ms = t(lambda: f5(x5), n=5)
print(f"C5 ternary VGG-16, batch {B5}, fused (threshold-bit convs, pools on bits): {ms:.3f} ms/forward = {B5 / ms * 1e3:.0f} img/s, same argmax {same}")
# how much of C4 is the per-activation overflow check (one host sync per quantised tensor)?
from pytorch_quantize_impls_amd import ops
orig = ops.CodePlanes.usable
ops.CodePlanes.usable = lambda self: True
with torch.no_grad():
    ms2 = t(lambda: m4(x4))
ops.CodePlanes.usable = orig
print(f"C4 without the int8-overflow host check: {ms2:.3f} ms/forward = {256 / ms2 * 1e3:.0f} img/s")
ops.CodePlanes.usable = orig
f4 = bench_models.FusedDorefaResNet18(m4, fuse_conv=False)
with torch.no_grad():
    ms = t(lambda: f4(x4), n=20)
print(f"C4 fused, fp32 conv outputs + one BatchNorm/shortcut/ReLU/quantiser pass each: {ms:.3f} ms/forward = {256 / ms * 1e3:.0f} img/s")
f4 = bench_models.FusedDorefaResNet18(m4)
with torch.no_grad():
    same = (f4(x4).argmax(1) == m4(x4).argmax(1)).float().mean().item()
    ms = t(lambda: f4(x4), n=20)
print(f"C4 fused (BatchNorm + shortcut + ReLU + quantiser in the conv epilogue, code planes between layers): "
      f"{ms:.3f} ms/forward = {256 / ms * 1e3:.0f} img/s, argmax agreement {same:.3f}")
# launch-bound? replay the fused C4 forward as a hipGraph (needs the range check waived: it is a host sync)
from pytorch_quantize_impls_amd import utils
ops.ASSUME_CODES_FIT = True
with torch.no_grad():
    g4 = utils.graphed(f4, x4)
    same = torch.equal(g4(x4), f4(x4))
    ms = t(lambda: g4(x4), n=20)
    ms_e = t(lambda: f4(x4), n=20)
ops.ASSUME_CODES_FIT = False
print(f"C4 fused as a hipGraph replay: {ms:.3f} ms/forward = {256 / ms * 1e3:.0f} img/s (eager with the same waiver {ms_e:.3f} ms), identical {same}")
