#!/usr/bin/env python
"""Backward of a quantised conv (AlexNet conv2 / conv3 shapes, batch 256): matrix-core routes vs torch.nn.grad (MIOpen)."""
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
for (Cin, Cout, k, pd, H, B) in [(192, 576, 5, 2, 27, 256), (576, 1152, 3, 1, 13, 256), (1152, 768, 3, 1, 13, 256), (64, 64, 3, 1, 112, 32)]:
    x = torch.where(torch.rand(B, Cin, H, H, device=dev) < 0.5, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
    wq = torch.where(torch.rand(Cout, Cin, k, k, device=dev) < 0.5, -1.0, 1.0)
    Ho = H + 2 * pd - k + 1
    g = torch.randn(B, Cout, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
    a = t(lambda: ops.conv2d_grad_input_q(x.shape, wq, g, 1, pd, 1))
    b = t(lambda: torch.nn.grad.conv2d_input(x.shape, wq, g, padding=pd))
    swapped = ops.conv2d_grad_weight_pm1(x, g, (k, k), 1, pd, 1)
    c = t(lambda: ops.conv2d_grad_weight_pm1(x, g, (k, k), 1, pd, 1)) if swapped is not None else float("nan")
    c2 = t(lambda: ops.conv2d_grad_weight_gemm(x, g, (k, k), pd))
    c3 = t(lambda: ops.conv2d_grad_weight_pm(x, g, (k, k), pd)) if ops.wgrad_pm_applicable(x.shape, g.shape, (k, k), 1, 1) else float("nan")
    d = t(lambda: torch.nn.grad.conv2d_weight(x, wq.shape, g, padding=pd))
    gi, gi_ref = ops.conv2d_grad_input_q(x.shape, wq, g, 1, pd, 1), torch.nn.grad.conv2d_input(x.shape, wq, g, padding=pd)
    pm = ops.wgrad_pm_applicable(x.shape, g.shape, (k, k), 1, 1)
    gw = ops.conv2d_grad_weight_pm(x, g, (k, k), pd) if pm else ops.conv2d_grad_weight_gemm(x, g, (k, k), pd)
    gw_ref = torch.nn.grad.conv2d_weight(x, wq.shape, g, padding=pd)
    e1 = float((gi - gi_ref).abs().max() / gi_ref.abs().max()); e2 = float((gw - gw_ref).abs().max() / gw_ref.abs().max())
    print(f"{Cin}->{Cout} k{k} {H}x{H} B{B}: grad_input {a:.3f} ms (MIOpen {b:.3f})  grad_weight: pixel-major {c3:.3f} ms, K-major GEMMs {c2:.3f} ms, swapped conv {c:.3f} ms "
          f"(MIOpen {d:.3f}; dispatched: {'pixel-major' if pm else 'gemm' if ops.wgrad_gemm_applicable(x.shape, g.shape, (k, k), 1, 1) else 'other'})  |diff| {e1:.1e} {e2:.1e}")
