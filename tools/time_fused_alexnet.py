#!/usr/bin/env python
"""Per-module wall time of the fused AlexNet-Bin forward (host + device), batch 256."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench_models
dev = torch.device("cuda:0")
if os.environ.get("CONV_FORCE"):
    from pytorch_quantize_impls_amd import ops
    for v in os.environ["CONV_FORCE"].split(","):     # plain-conv variant (an argument of qt_conv2d_implicit_variant)
        ops.CONV_VARIANT = int(v)
torch.manual_seed(0)
model = bench_models.AlexNetBin(); bench_models.randomize_bn(model)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
for fc in (True, False):
    fused = bench_models.FusedAlexNetBin(model, fuse_conv=fc)
    nfeat = len(fused.net.features)
    mods = list(fused.net.features.children()) + list(fused.net.classifier.children())
    with torch.no_grad():
        for _ in range(3): fused(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fused(x)
        torch.cuda.synchronize()
        print(f"fuse_conv={fc}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms / forward")
        h = x
        for mi, m in enumerate(mods):
            if mi == nfeat:
                h = h.flatten_hwc()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): out = m(h)
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print(f"   {type(m).__name__:22s} host {1e3 * (t1 - t0) / 5:7.3f} ms   total {1e3 * (t2 - t0) / 5:7.3f} ms")
            h = out
