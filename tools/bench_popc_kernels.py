#!/usr/bin/env python
"""Tiled vs skinny popcount GEMM over small-M / small-N shapes (run on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
for (M, N, K) in [(1, 4096, 4096), (16, 4096, 4096), (64, 4096, 4096), (128, 4096, 4096), (256, 4096, 4096),
                  (256, 10, 4096), (256, 1000, 4096), (64, 1000, 25088), (1024, 64, 4096)]:
    x = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev)
    xp, wp = ops.sign_pack(x)[0], ops.sign_pack(w)[0]
    out = {}
    for which, name in ((1, "tiled"), (2, "skinny")):
        ops.POPC_VARIANT = which
        y = ops.xnor_gemm(xp, wp)
        for _ in range(3): ops.xnor_gemm(xp, wp, out=y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(20):
            e0.record(); ops.xnor_gemm(xp, wp, out=y); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort(); out[name] = (ts[10], y.clone())
    ops.POPC_VARIANT = 0
    print(f"M={M:5d} N={N:5d} K={K:6d}  tiled {out['tiled'][0]:8.1f} us  skinny {out['skinny'][0]:8.1f} us  equal={torch.equal(out['tiled'][1], out['skinny'][1])}")
