#!/usr/bin/env python
"""popcount (auto tiled/skinny) vs matrix-core GEMM incl. operand conversion from bit planes, to set the
'auto' thresholds in ops.select_gemm_impl (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]

for (M, N, K) in [(4096, 32, 4096), (4096, 64, 4096), (4096, 128, 4096), (4096, 192, 4096), (64, 4096, 4096),
                  (128, 4096, 4096), (192, 4096, 4096), (256, 4096, 4096), (512, 4096, 4096), (1024, 1024, 1024),
                  (256, 256, 4096), (4096, 4096, 128), (4096, 4096, 256), (4096, 4096, 512), (256, 4096, 9216),
                  (256, 10, 4096), (256, 1000, 4096), (32, 4096, 4096), (8, 4096, 4096), (1024, 4096, 1024), (2048, 2048, 2048), (512, 512, 512), (128, 512, 784)]:
    x = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev)
    xb, wb = ops.sign_pack(x)[0], ops.sign_pack(w)[0]
    wn = ops.sign_pack_nib(w)
    y = torch.empty((M, N), device=dev)
    tp = t(lambda: ops.xnor_gemm(xb, wb, out=y))
    tm = t(lambda: ops.nib_gemm(ops.bits_to_nib(xb), wn, out=y))      # activation planes arrive as bits (tag path)
    print(f"M={M:5d} N={N:5d} K={K:5d}  popcount {tp:8.1f} us   mfma(+bits->nib) {tm:8.1f} us   -> {'mfma' if tm < tp else 'popc'}")
