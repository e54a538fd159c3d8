#!/usr/bin/env python
"""DoReFa conv (int8 code planes, 1-bit weights) with fp32 output vs the code epilogue (BatchNorm + residual + ReLU +
quantiser in the kernel), with / without a residual, on the C4 ResNet-18 stage shapes (batch 256)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import ops, packed
from pytorch_quantize_impls_amd.layers import DorefaConv2d, FusedBnDorefaQuant, FusedDorefaConvBnQuant
dev = torch.device("cuda:0")
ops.PAD_PIXEL_PLANES = os.environ.get("PAD", "0") == "1"
def t(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
torch.manual_seed(0)
for (C, HW) in ((64, 32), (128, 16), (256, 8), (512, 4)):
    N = 256
    conv = DorefaConv2d(C, C, 3, padding=1, bias=False, bit_width=1).to(dev).eval()
    bn = torch.nn.BatchNorm2d(C).to(dev).eval()
    codes, _ = ops.dorefa_codes(torch.rand((N * HW * HW, C), device=dev), 4, want_f32=False, ld_bytes=ops.code_ld_bytes(C, 16))
    act = packed.CodeActivation(codes, (N, C, HW, HW))
    rf = torch.randn((N, C, HW, HW), device=dev).contiguous(memory_format=torch.channels_last)
    f = FusedDorefaConvBnQuant(conv, bn, 4)
    q = FusedBnDorefaQuant(bn, 4)
    with torch.no_grad():
        y = conv(act)
        print(f"C={C:3d} {HW}x{HW}: conv fp32 out {t(lambda: conv(act)):6.1f} us | quantiser pass {t(lambda: q(y)):5.1f} us | "
              f"code epilogue {t(lambda: f(act)):6.1f} us, + code residual {t(lambda: f(act, residual=act)):6.1f} us, "
              f"+ fp32 residual {t(lambda: f(act, residual=rf)):6.1f} us", flush=True)
