#!/usr/bin/env python
"""C2 step time: eager with per-GEMM events / eager without / hipGraph replay (launch-gap audit)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
B = K = N = 4096
x = torch.randn((B, K), device=dev).sign_(); x[x == 0] = 1
w = torch.randn((N, K), device=dev) / 64
y = torch.empty((B, N), device=dev)
impl = "mfma"
def step(ev=None):
    xp = ops.pack_activations(x, impl); wp = ops.pack_weights(w, "binary", impl)
    if ev: ev[0].record()
    ops.packed_gemm(xp, wp, None, out=y, impl=impl)
    if ev: ev[1].record()
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(256)]
it = iter(range(10**9))
print(f"eager + events : {timeit(lambda: step(evs[next(it) % 256])):7.1f} us/step")
print(f"eager          : {timeit(step):7.1f} us/step")
t0 = time.perf_counter()
for _ in range(2000): pass
cpu0 = time.perf_counter()
for _ in range(200): step()
cpu = (time.perf_counter() - cpu0) / 200 * 1e6
torch.cuda.synchronize()
print(f"host time to enqueue one step: {cpu:7.1f} us")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    print(f"graph replay   : {timeit(g.replay):7.1f} us/step")
    g10 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g10, stream=s):
        for _ in range(10): step()
    print(f"graph x10      : {timeit(g10.replay, 50) / 10:7.1f} us/step")
