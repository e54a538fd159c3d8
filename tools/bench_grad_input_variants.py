#!/usr/bin/env python
"""grad_input of the AlexNet conv2-5 / VGG shapes: which conv kernel variant (ops.CONV_VARIANT: 0 automatic, 1 double-buffered, 2 ping-pong)
serves the exact-split conv of the gradient best."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (Cin, Cout, k, pd, H, B) in [(192, 576, 5, 2, 27, 256), (576, 1152, 3, 1, 13, 256), (1152, 768, 3, 1, 13, 256), (768, 256, 3, 1, 13, 256),
                                 (512, 512, 3, 1, 28, 64), (256, 256, 3, 1, 56, 64), (128, 128, 3, 1, 112, 64)]:
    wq = torch.where(torch.rand(Cout, Cin, k, k, device=dev) < 0.5, -1.0, 1.0)
    Ho = H + 2 * pd - k + 1
    g = torch.randn(B, Cout, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
    res = []
    for v in (0, 1, 2):
        ops.CONV_VARIANT = v
        try:
            res.append(t(lambda: ops.conv2d_grad_input_q((B, Cin, H, H), wq, g, 1, pd, 1)))
        except Exception as e:
            res.append(float("nan"))
    ops.CONV_VARIANT = 0
    flops = ops.split_terms() * 2.0 * B * H * H * Cin * Cout * k * k
    print(f"grad_input {Cout}->{Cin} k{k} {H}x{H} B{B}: auto {res[0]:.3f} ms, double-buffered {res[1]:.3f}, ping-pong {res[2]:.3f}  (matrix floor at 1.9 PF {flops / 1.9e15 * 1e3:.3f} ms)", flush=True)

# (the physically padded variant of this comparison: tools/bench_grad_input_padded.py)
