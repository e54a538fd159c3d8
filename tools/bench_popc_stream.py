"""Popcount GEMM in the regime with a handful of rows on one side: streaming kernel (variant 3: K along the lanes, DPP reduction) vs
the skinny kernel (variant 2: lane <-> batch row) vs the tiled kernel (1).  Times are per call inside a captured hipGraph of 20
calls (no launch gaps); GB/s = packed bytes of both operands + the fp32 result over that time."""
import sys
import torch
sys.path.insert(0, ".")
from pytorch_quantize_impls_amd import ops

dev = torch.device("cuda:0")


def timed(fn, reps=20, it=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (it * reps) * 1e3


print(f"{'M':>5} {'N':>5} {'K':>6} | {'stream us':>9} {'GB/s':>7} | {'skinny us':>9} | {'tiled us':>8} | {'bits->nib + fp4 MFMA us':>22}")
for (M, N, K) in [(1, 4096, 4096), (1, 4096, 9216), (1, 4096, 25088), (8, 4096, 9216), (16, 4096, 9216), (24, 4096, 9216), (32, 4096, 9216), (32, 4096, 4096),
                  (1, 1000, 4096), (256, 10, 4096), (64, 10, 4096), (2048, 10, 4096), (1, 16384, 16384)]:
    x = torch.randn((M, K), device=dev)
    w = torch.randn((N, K), device=dev)
    xp, wp = ops.sign_pack(x)[0], ops.sign_pack(w)[0]
    out = torch.empty((M, N), device=dev)
    res = {}
    for v in (3, 2, 1):
        ops.POPC_VARIANT = v
        try:
            res[v] = timed(lambda: ops.xnor_gemm(xp, wp, out=out))
        finally:
            ops.POPC_VARIANT = 0
    byts = (M + N) * K / 8 + 4 * M * N
    wn = ops.bits_to_nib(wp)
    mf = timed(lambda: ops.nib_gemm(ops.bits_to_nib(xp), wn, out=out)) if K % 1024 == 0 else float("nan")
    print(f"{M:5d} {N:5d} {K:6d} | {res[3]:9.2f} {byts / res[3] * 1e-3:7.1f} | {res[2]:9.2f} | {res[1]:8.2f} | {mf:22.2f}")
