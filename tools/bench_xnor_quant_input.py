#!/usr/bin/env python
"""XNORConv2d(quant_input=True) vs the quant_input=False conv at the AlexNet conv2 / conv3 shapes, batch 256:
forward only (no_grad) and forward + backward, HIP-event timed.   python tools/bench_xnor_quant_input.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_quantize_impls_amd.functions import xnor_connect, _fused  # noqa: E402
from pytorch_quantize_impls_amd.functions import BinaryConnectDeterministic  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dev = torch.device("cuda:0")
for name, Cin, Cout, H, k, p in (("conv2", 192, 576, 27, 5, 2), ("conv3", 576, 1152, 13, 3, 1), ("conv4", 1152, 768, 13, 3, 1)):
    B = 256
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05)
    xs = BinaryConnectDeterministic.apply(x)
    f_false = xnor_connect.XNORConv2d([0, 1], False, 1, p, 1, 1)
    f_true = xnor_connect.XNORConv2d([0, 1], True, 1, p, 1, 1)
    before = dict(_fused.LIBRARY_PATHS)
    with torch.no_grad():
        t0 = timed(lambda: f_false.apply(xs, w))
        t1 = timed(lambda: f_true.apply(x, w))
    xr = x.clone().requires_grad_(True)
    xsr = xs.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    g = torch.randn(B, Cout, H, H, device=dev).contiguous(memory_format=torch.channels_last)

    def step(f, inp):
        inp.grad = None
        wr.grad = None
        f.apply(inp, wr).backward(g)
    tb0 = timed(lambda: step(f_false, xsr), 5)
    tb1 = timed(lambda: step(f_true, xr), 5)
    lib = {k_: v - before.get(k_, 0) for k_, v in _fused.LIBRARY_PATHS.items() if v != before.get(k_, 0)}
    print(f"{name}: forward quant_input=False {t0:.0f} us, True {t1:.0f} us ({t1 / t0:.2f}x); fwd+bwd False {tb0:.0f} us, True {tb1:.0f} us "
          f"({tb1 / tb0:.2f}x); library calls {lib}")
