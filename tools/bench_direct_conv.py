#!/usr/bin/env python
"""Direct 3x3 conv on halo planes (qt_conv3x3_direct_nib) against the implicit-GEMM kernel on the early VGG-16 layer
shapes (batch 256): nibble-plane output (feeds a conv) and bit-plane output (feeds a pool)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
N = int(os.environ.get("BATCH", "256"))
for (C, Cout, HW) in ((64, 64, 224), (64, 128, 112), (128, 128, 112)):
    x = torch.randn((N, C, HW, HW), device=dev).sign().contiguous(memory_format=torch.channels_last)
    bits = ops.sign_pack(x.permute(0, 2, 3, 1).contiguous())[0]
    px = ops.bits_to_nib_pad(bits, N, HW, HW, (1, 1), ld=ops.pixel_ld_nib(C))
    del x, bits
    wp = ops.pack_conv_weight_nib(torch.randn((Cout, C, 3, 3), device=dev), "ternary")
    alpha, beta = torch.rand(Cout, device=dev) + 0.1, torch.randn(Cout, device=dev)
    nib = ops.NibEpilogue(alpha, beta, (1, 1))
    args = (px, (N, C, HW + 2, HW + 2), wp, (3, 3), None, 1, 0, 1)
    same = torch.equal(ops.conv3x3_direct_nib(px, N, C, HW, HW, wp, None, nib).words, ops.conv2d_nib(*args, epi=nib).words)
    print(f"{C:3d}->{Cout:3d} @{HW}: implicit nib {t(lambda: ops.conv2d_nib(*args, epi=nib)):6.1f} us  bits {t(lambda: ops.conv2d_nib(*args, epi=(alpha, beta))):6.1f} us | "
          f"direct nib {t(lambda: ops.conv3x3_direct_nib(px, N, C, HW, HW, wp, None, nib)):6.1f} us  bits {t(lambda: ops.conv3x3_direct_nib(px, N, C, HW, HW, wp, None, (alpha, beta))):6.1f} us   identical {same}", flush=True)
