#!/usr/bin/env python
"""Time of the stem quantiser pass of C4 (BatchNorm + ReLU + 4-bit quantiser over a 256 x 64 x 32 x 32 fp32 conv output, written into the
1-pixel halo plane): HIP events over 50 launches, plain and halo form."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench_models
from pytorch_quantize_impls_amd.layers import FusedBnDorefaQuant
dev = torch.device("cuda:0")
torch.manual_seed(0)
for N, C, H in ((256, 64, 32), (256, 128, 16)):
    x = torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(C).to(dev).eval(); bench_models.randomize_bn(bn, seed=1)
    for halo in (0, 1):
        q = FusedBnDorefaQuant(bn, 4, out_halo=halo, fold="device")
        with torch.no_grad():
            for _ in range(5): q(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): q(x)
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{N}x{C}x{H}x{H} halo {halo}: {us:.1f} us / call  ({N*C*H*H*4/us/1e6:.2f} TB/s of fp32 read)")
