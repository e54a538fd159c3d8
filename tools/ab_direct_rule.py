#!/usr/bin/env python
"""A/B on one box: fused VGG-16 (batch 256) with the direct 3x3 kernel (a) as dispatched, (b) also for 128-channel
inputs with bit-plane output, (c) off."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import ops
from pytorch_quantize_impls_amd.layers import FusedFeatureClassifier
dev = torch.device("cuda:0")
m5 = bench_models.TernaryVGG16(num_classes=1000, image=224); bench_models.randomize_bn(m5, seed=5)
m5 = m5.to(dev).to(memory_format=torch.channels_last).eval(); m5.features[0].binary_input = False
x5 = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
f5 = FusedFeatureClassifier(m5.features, m5.classifier, (512, 7, 7))
def t(fn, n=8):
    best = 1e9
    for _ in range(3):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best
orig = ops.direct_conv3x3_applicable
def wide(C, Cout, k, s, p, d, h, epi):
    return (ops.DIRECT_CONV3X3 and tuple(k) == (3, 3) and ops._pairs(s) == (1, 1) and ops._pairs(p) == (1, 1) and ops._pairs(d) == (1, 1)
            and tuple(h) == (1, 1) and ops.pixel_ld_nib(C) in (8, 16) and Cout <= 128
            and ((isinstance(epi, ops.NibEpilogue) and tuple(epi.out_halo) == (1, 1) and not epi.d2s_cout) or isinstance(epi, tuple)))
with torch.no_grad():
    for rep in range(2):
        for name, fn in (("dispatched", orig), ("+128ch bits", wide), ("off", lambda *a: False)):
            ops.direct_conv3x3_applicable = fn
            print(f"{name:12s} {t(lambda: f5(x5)):.3f} ms", flush=True)
