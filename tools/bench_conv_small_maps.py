#!/usr/bin/env python
"""Padded 3x3 convs on small maps (the late stages of a CIFAR ResNet, batch 256) per conv-kernel variant:
0 = the dispatcher's choice, 5 = 128x128 tiles, 6 = 64x64 tiles with 512-byte stages (bounds-checked skinny).
Real-valued activation (two fp16 planes: the backward's grad_x and the forward of activations beyond int8) and int8 codes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for (B, C, H, Co) in ((256, 64, 32, 64), (256, 128, 16, 128), (256, 256, 8, 256), (256, 512, 4, 512), (256, 512, 8, 512)):
    x = torch.randn(B, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, C, 3, 3, device=dev)
    wt = ops.pack_conv_weight_bf16x3(w, "binary")
    want = None
    line = [f"B{B} {C}->{Co} @{H}x{H}:"]
    for v in (0, 5, 6):
        ops.CONV_VARIANT = v
        try:
            y = ops.float_conv2d(x, w, "binary", None, 1, 1, 1, weight_triples=wt)
            if want is None: want = y.clone()
            same = torch.equal(y, want)
            us = timeit(lambda: ops.float_conv2d(x, w, "binary", None, 1, 1, 1, weight_triples=wt))
            line.append(f"v{v} {us:7.1f} us{'' if same else ' (differs %.1e)' % float((y - want).abs().max() / want.abs().max())}")
        except Exception as e:
            line.append(f"v{v} {type(e).__name__}")
    ops.CONV_VARIANT = 0
    print(" ".join(line), flush=True)
