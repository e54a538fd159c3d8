#!/usr/bin/env python
"""Fused AlexNet-Bin at small batch: eager launches vs hipGraph replay (launch-bound regime)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.AlexNetBin(); bench_models.randomize_bn(model)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
fused = bench_models.FusedAlexNetBin(model)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (1, 8, 32, 256):
    x = torch.randn((B, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ye = fused(x)
        te = t(lambda: fused(x))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): fused(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                yg = fused(x)
            torch.cuda.synchronize()
            tg = t(g.replay)
        g.replay(); torch.cuda.synchronize()
    print(f"batch {B:4d}: eager {te:7.3f} ms ({B / te * 1e3:9.0f} img/s)   graph {tg:7.3f} ms ({B / tg * 1e3:9.0f} img/s)   equal {torch.equal(ye, yg)}")
