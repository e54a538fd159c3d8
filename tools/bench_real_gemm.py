#!/usr/bin/env python
"""REAL x REAL GEMM: fp32 library vs the six-term bf16 route (ops.real_linear), accuracy vs fp64 and time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (M, N, K) in [(4096, 4096, 4096), (1024, 1024, 1024), (256, 4096, 9216), (8192, 512, 512)]:
    x = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev) * 0.1
    ref = x.double() @ w.double().t()
    a = torch.nn.functional.linear(x, w); b = ops.real_linear(x, w)
    e = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"{M}x{N}x{K}: fp32 library {t(lambda: torch.nn.functional.linear(x, w)):.3f} ms (err {e(a):.1e})   six-term bf16 {t(lambda: ops.real_linear(x, w)):.3f} ms (err {e(b):.1e})")
