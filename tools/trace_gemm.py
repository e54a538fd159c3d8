#!/usr/bin/env python
"""Phase timeline of the nibble GEMM (variant 205): s_memtime stamps of block (0,0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import _lib
M = N = K = 4096
dev = torch.device("cuda:0")
x = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev)
ld = K // 8
xn = torch.empty((M, ld), dtype=torch.int32, device=dev); wn = torch.empty((N, ld), dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr()); I = ctypes.c_int64
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.call("qt_sign_pack_nib_f32", P(x), I(K), P(xn), I(ld), I(M), I(K), st)
_lib.call("qt_sign_pack_nib_f32", P(w), I(K), P(wn), I(ld), I(N), I(K), st)
y = torch.empty((M, N), device=dev)
tr = torch.zeros((8, 8, 8), dtype=torch.int64, device=dev)
for _ in range(3):
    _lib.call("qt_nib_gemm_variant", ctypes.c_int(205), P(xn), I(ld), P(wn), I(ld), P(tr), P(y), I(N), I(M), I(N), I(K), st)
torch.cuda.synchronize()
t = tr.cpu()
names = ["start", "ds_issued", "dma_issued", "lds_landed", "mfma_issued", "dma_landed", "after_barrier"]
base = int(t[:, 0, 0].min())
print("s_memtime ticks relative to first stamp; per stage: " + ", ".join(names))
for wv in (0, 1, 4, 7):
    for s in range(4):
        print(f"wave {wv} stage {s}: " + " ".join(f"{int(t[wv, s, p]) - base:7d}" for p in range(7)))
