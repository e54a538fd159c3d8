#!/usr/bin/env python
"""grad_x of the AlexNet conv2-5 / ResNet shapes (two-term split): the bounds-checked conv kernels on the gradient's pair plane
(ops.conv2d_grad_input_q today) vs the un-padded ("valid") kernels on a PHYSICALLY padded pair plane written by the split pass
itself (qt_f16x2_s2d_pack_f32 with s = 1: the padding as a zero border)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (Cin, Cout, k, pd, H, B) in [(192, 576, 5, 2, 27, 256), (576, 1152, 3, 1, 13, 256), (1152, 768, 3, 1, 13, 256), (768, 256, 3, 1, 13, 256),
                                 (64, 64, 3, 1, 32, 256), (128, 128, 3, 1, 16, 256), (256, 256, 3, 1, 8, 256)]:
    wq = torch.where(torch.rand(Cout, Cin, k, k, device=dev) < 0.5, -1.0, 1.0)
    Ho = H + 2 * pd - k + 1
    g = torch.randn(B, Cout, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
    wT = wq.flip(2, 3).transpose(0, 1).contiguous()
    p2 = k - 1 - pd
    wt = ops.pack_conv_weight_bf16x3(wT, "sign")
    def now():
        return ops.float_conv2d(g, wT, "sign", None, 1, (p2, p2), 1, weight_triples=wt)
    def padded():
        px, (Hp, Wp) = ops.s2d_triple_pack(g, 1, (p2, p2))
        return ops.float_conv2d(None, wT, "sign", None, 1, 0, 1, weight_triples=wt, pixels=px, in_shape=(B, Cout, Hp, Wp))
    same = bool(torch.equal(now(), padded()))
    print(f"grad_x {Cout}->{Cin} k{k} {H}x{H} B{B}: bounds-checked {t(now):.3f} ms, padded plane + valid kernels {t(padded):.3f} ms (same result: {same})", flush=True)
