"""Times the LinearXNOR operand routes at the XNOR-AlexNet FC shapes (batch 256): today's fp16 pair GEMM against the int8 GEMM
with the alpha digits stacked along M (3 x 256 rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pytorch_quantize_impls_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (M, N, K) in [(256, 4096, 9216), (256, 4096, 4096), (256, 10, 4096), (256, 1000, 4096)]:
    w = torch.randn(N, K, device=dev)
    x = torch.sign(torch.randn(M, K, device=dev))
    alpha = w.abs().mean(0)
    wt = ops.weight_bf16x3(w, "sign", terms=2)
    xp = ops.split_bf16x3(x * alpha, terms=2)
    t_pair = timeit(lambda: ops.bf16_gemm(xp, wt))
    wc = ops.weight_codes(w)
    res = {}
    for D in (3, 4):
        q = torch.randint(-127, 128, (D * M, K), device=dev).float() / 127.0
        cp, _ = ops.dorefa_codes(q, 7, want_f32=False)
        cp.overflow = None
        res[D] = timeit(lambda: ops.i8_gemm(cp, wc, 1.0))
    print(f"{M}x{N}x{K}: fp16 pairs {t_pair:.1f} us | i8 digits x3 {res[3]:.1f} us, x4 {res[4]:.1f} us", flush=True)
