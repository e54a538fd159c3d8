#!/usr/bin/env python
"""Pixel-major weight-gradient kernel (csrc/wgrad_pm.hip): agreement with an fp64 reference, time beside the K-major GEMM route and MIOpen."""
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
torch.manual_seed(0)
small = [(40, 70, 3, 1, 7, 9, 3), (33, 65, 5, 2, 6, 5, 2), (64, 64, 3, 0, 5, 5, 1), (96, 32, 5, 0, 9, 7, 5), (130, 200, 3, 1, 4, 4, 17)]
for (Cin, Cout, k, pd, H, W, B) in small:
    x = torch.where(torch.rand(B, Cin, H, W, device=dev) < 0.5, -1.0, 1.0)
    g = torch.randn(B, Cout, H + 2 * pd - k + 1, W + 2 * pd - k + 1, device=dev) * torch.exp(torch.randn(B, Cout, 1, 1, device=dev) * 3)
    for cl in (False, True):
        xs = x.contiguous(memory_format=torch.channels_last) if cl else x
        gs = g.contiguous(memory_format=torch.channels_last) if cl else g
        got = ops.conv2d_grad_weight_pm(xs, gs, (k, k), pd)
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), g.double(), padding=pd)
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        print(f"small {Cin}->{Cout} k{k} p{pd} {H}x{W} B{B} cl={cl}: rel err {err:.2e}", flush=True)
        assert err < 2e-6, err
big = [(192, 576, 5, 2, 27, 256), (576, 1152, 3, 1, 13, 256), (1152, 768, 3, 1, 13, 256), (512, 512, 3, 1, 28, 256),
       (512, 512, 3, 1, 14, 256), (256, 256, 3, 1, 56, 256), (128, 128, 3, 1, 112, 64), (64, 64, 3, 1, 224, 32)]
for (Cin, Cout, k, pd, H, B) in big:
    x = torch.where(torch.rand(B, Cin, H, H, device=dev) < 0.5, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
    Ho = H + 2 * pd - k + 1
    g = torch.randn(B, Cout, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
    got = ops.conv2d_grad_weight_pm(x, g, (k, k), pd)
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), g, padding=pd)
    err = float((got - ref).abs().max() / ref.abs().max())
    a = t(lambda: ops.conv2d_grad_weight_pm(x, g, (k, k), pd))
    b = t(lambda: ops.conv2d_grad_weight_gemm(x, g, (k, k), pd)) if ops.wgrad_gemm_applicable(x.shape, g.shape, (k, k), 1, 1) else float("nan")
    c = t(lambda: torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), g, padding=pd))
    flops = 2.0 * B * Ho * Ho * Cin * Cout * k * k
    print(f"{Cin}->{Cout} k{k} {H}x{H} B{B}: pixel-major {a:.3f} ms ({3 * flops / a / 1e9:.0f} TFLOP/s bf16-equivalent incl. packs), "
          f"K-major GEMMs {b:.3f} ms, MIOpen {c:.3f} ms; |diff vs fp32 lib| {err:.1e}", flush=True)
