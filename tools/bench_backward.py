#!/usr/bin/env python
"""Backward GEMMs of LinearBin at C2 size: fp32 library vs the bf16x3 matrix-core route (accuracy + time)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd.functions import _fused
dev = torch.device("cuda:0")
M = N = K = 4096
g = torch.randn((M, N), device=dev)
wq = torch.randn((N, K), device=dev).sign()
x = torch.randn((M, K), device=dev).sign()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ref = (g.double() @ wq.double())
a = g.mm(wq); b = _fused.pm1_matmul(g, wq)
e = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
print(f"grad_x  : fp32 library {t(lambda: g.mm(wq)):.3f} ms (err {e(a):.2e})   bf16x3 MFMA {t(lambda: _fused.pm1_matmul(g, wq)):.3f} ms (err {e(b):.2e})")
ref2 = g.t().double() @ x.double()
a2 = g.t().mm(x); b2 = _fused.pm1_matmul(g.t(), x)
e2 = lambda y: float((y.double() - ref2).abs().max() / ref2.abs().max())
print(f"grad_W  : fp32 library {t(lambda: g.t().mm(x)):.3f} ms (err {e2(a2):.2e})   bf16x3 MFMA {t(lambda: _fused.pm1_matmul(g.t(), x)):.3f} ms (err {e2(b2):.2e})")
