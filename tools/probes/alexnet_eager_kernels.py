#!/usr/bin/env python
"""The un-modified BinaryNet-AlexNet eval forward (batch 256), ITERS eager forwards: the target of tools/probes/c3_pmc.sh
(rocprofv3 --kernel-trace / --pmc passes; per-dispatch counters want ordinary launches, not graph replays)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench_models
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = bench_models.AlexNetBin()
bench_models.randomize_bn(m)
m = m.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(int(os.environ.get("ITERS", "12"))):
        y = m(x)
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(y).all()))
