#!/usr/bin/env python
"""Cycle / wall-clock stamps of the ping-pong GEMM (variant 165: needs a library built with -DQT_PROFILING_VARIANTS,
selected through QT_HIP_LIB; Y is garbage)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pytorch_quantize_impls_amd import _lib
M = N = K = 4096
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 165
dev = torch.device("cuda:0")
x = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev)
ld = K // 8
xn = torch.empty((M, ld), dtype=torch.int32, device=dev); wn = torch.empty((N, ld), dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr()); I = ctypes.c_int64
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.call("qt_sign_pack_nib_f32", P(x), I(K), P(xn), I(ld), I(M), I(K), st)
_lib.call("qt_sign_pack_nib_f32", P(w), I(K), P(wn), I(ld), I(N), I(K), st)
y = torch.zeros((M, N), device=dev)
for rep in range(3):
    _lib.call("qt_nib_gemm_variant", ctypes.c_int(VAR), P(xn), I(ld), P(wn), I(ld), ctypes.c_void_p(0), P(y), I(N), I(M), I(N), I(K), st)
    torch.cuda.synchronize()
yi = y.view(torch.int32).cpu().numpy()
rows = []
for tm in range(M // 256):
    for tn in range(N // 256):
        for half in range(2):
            r = np.ascontiguousarray(yi[tm * 256 + half * 128, tn * 256: tn * 256 + 32]).view(np.int64)
            rows.append(r)
t = np.array(rows)            # [512][14]
names = ["top", "frags read", "dma issued", "waitcnt done", "barrier1", "mfma issued", "barrier2"]
a, b = t[0], t[1]
print("stage stamps of tile (0,0), cycles from the leading wave's load start:")
base = min(a[0], b[0])
for nm, r in (("wave 0 (A)", a), ("wave 4 (B)", b)):
    print(f"  simd {int(r[14])}", end="")
    print(f"  {nm}: " + "  ".join(f"{n}={int(r[i] - base)}" for i, n in enumerate(names)) + f"")
wall = t[:, 9:14].astype(np.float64) / 100.0   # us (100 MHz)
w0 = wall[:, 0].min()
lab = ["entry", "prologue done", "loop end", "stores issued", "stores done"]
print("SIMD ids of waves 0 / 4 over tiles:", sorted(set((int(t[2*i][14]), int(t[2*i+1][14])) for i in range(256))))
print("wall clock (us from first workgroup entry), over 512 (tile, half) records:")
for i, l in enumerate(lab):
    c = wall[:, i] - w0
    print(f"  {l:14s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f}")
d = wall[:, 1:] - wall[:, :-1]
for i in range(4):
    print(f"  {lab[i]} -> {lab[i+1]}: median {np.median(d[:, i]):6.2f} us  max {d[:, i].max():6.2f}")
loop_us = wall[:, 2] - wall[:, 1]
print(f"main loop: median {np.median(t[:, 15]):.0f} cycles in {np.median(loop_us):.2f} us -> shader clock {np.median(t[:, 15] / loop_us) / 1e3:.3f} GHz")
