#!/usr/bin/env python
"""xnor / ternary popcount GEMM at the C2 size: bit-exactness against the matrix-core route + median time."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
M = N = K = 4096
x = torch.where(torch.rand((M, K), device=dev) < 0.5, -1.0, 1.0); w = torch.randn((N, K), device=dev)
xp, wp, tp = ops.sign_pack(x)[0], ops.sign_pack(w)[0], ops.ternary_pack(w)
ref = ops.nib_gemm(ops.sign_pack_nib(x), ops.sign_pack_nib(w))
reft = ops.nib_gemm(ops.sign_pack_nib(x), ops.ternary_pack_nib(w))
y = torch.empty((M, N), device=dev)
for name, fn, r in (("xnor", lambda: ops.xnor_gemm(xp, wp, out=y), ref), ("tern", lambda: ops.tern_gemm(xp, tp, out=y), reft)):
    ops.POPC_VARIANT = 1
    for _ in range(3): fn()
    ok = torch.equal(y, r)
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    pairs = M * N * K / 32 / (ts[10] * 1e-6) / 1e12
    print(f"{name}: exact={ok} median {ts[10]:.1f} us  {2*M*N*K/ts[10]/1e6:.0f} TOPS  {pairs:.1f} T pairs/s ({pairs/23.3:.0%} of the 23.3 T xor+bcnt ceiling)")
