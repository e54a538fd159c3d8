#!/usr/bin/env python
"""C4 (DoReFa ResNet-18 W1A4, batch 256) fused inference form: eager and hipGraph ms / forward, logits torch.equal to the
module-by-module graph.  A/B tool for the small-map int8 conv kernels.   python tools/time_c4.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_models  # noqa: E402
from pytorch_quantize_impls_amd import lazy, utils  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(4)
m4 = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
bench_models.randomize_bn(m4, seed=3)
for m in m4.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_var.mul_(4.0)
m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
x4 = torch.randn((256, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
f4 = bench_models.FusedDorefaResNet18(m4, fold="device")


def timed(fn, n=20, reps=3):
    best = None
    with torch.no_grad():
        for _ in range(3):
            fn()
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            e = (time.perf_counter() - t0) / n * 1e3
            best = e if best is None else min(best, e)
    return best


with torch.no_grad():
    with lazy.eager():
        ye = m4(x4)
    yf = f4(x4)
    gf = utils.graphed(f4, x4)
    gm = utils.graphed(m4, x4)
    print("fused == module-by-module:", bool(torch.equal(yf, ye)), " graph == fused:", bool(torch.equal(gf(x4), yf)),
          " module graph (deferred, hipGraph) == module-by-module:", bool(torch.equal(gm(x4), ye)))
print(f"fused eager {timed(lambda: f4(x4)):.3f} ms   fused hipGraph {timed(lambda: gf(x4)):.3f} ms   "
      f"module graph hipGraph {timed(lambda: gm(x4)):.3f} ms   module graph eager {timed(lambda: m4(x4)):.3f} ms")
