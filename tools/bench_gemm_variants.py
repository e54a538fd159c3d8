#!/usr/bin/env python
"""A/B of the nibble-plane MFMA GEMM tile configurations against the (validated) popcount GEMM.
Run on the GPU box:  python tools/bench_gemm_variants.py [M N K]"""
import ctypes
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import _lib, ops

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 4096, 4096)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((M, K), device=dev, generator=g)
w = torch.randn((N, K), device=dev, generator=g)
ref = ops.xnor_gemm(ops.sign_pack(x)[0], ops.sign_pack(w)[0])
ld = max(4, ((K + 7) // 8 + 3) // 4 * 4)
xn = torch.empty((M, ld), dtype=torch.int32, device=dev)
wn = torch.empty((N, ld), dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
I = ctypes.c_int64
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.call("qt_sign_pack_nib_f32", P(x), I(K), P(xn), I(ld), I(M), I(K), st)
_lib.call("qt_sign_pack_nib_f32", P(w), I(K), P(wn), I(ld), I(N), I(K), st)
y = torch.empty((M, N), device=dev)
ops_ = 2.0 * M * N * K
for variant in [int(v) for v in os.environ.get("VARIANTS", "0,5,6,7,8").split(",")]:
    y.fill_(float("nan"))
    def run():
        _lib.call("qt_nib_gemm_variant", ctypes.c_int(variant), P(xn), I(ld), P(wn), I(ld), ctypes.c_void_p(0),
                  P(y), I(N), I(M), I(N), I(K), st)
    run(); torch.cuda.synchronize()
    ok = torch.equal(y, ref)
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); med = ts[len(ts)//2]
    print(f"variant {variant}: exact={ok}  median {med*1e3:8.1f} us  min {ts[0]*1e3:8.1f} us  {ops_/med/1e9:8.1f} TOPS")
# pack kernels
for name, fn in (("sign_pack bits", lambda: ops.sign_pack(x)), ("sign_pack nib", lambda: _lib.call("qt_sign_pack_nib_f32", P(x), I(K), P(xn), I(ld), I(M), I(K), st))):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(20):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); print(f"{name}: median {ts[10]*1e3:.1f} us  ({M*K*4/ts[10]/1e6:.0f} GB/s read)")
