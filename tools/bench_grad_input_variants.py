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
    flops = 3 * 2.0 * B * H * H * Cin * Cout * k * k
    print(f"grad_input {Cout}->{Cin} k{k} {H}x{H} B{B}: auto {res[0]:.3f} ms, double-buffered {res[1]:.3f}, ping-pong {res[2]:.3f}  (matrix floor at 1.9 PF {flops / 1.9e15 * 1e3:.3f} ms)", flush=True)

# the same conv on a PHYSICALLY padded gradient plane (pad pass + un-padded "valid" conv kernels): is the valid mode worth a halo-writing split?
print()
for (Cin, Cout, k, pd, H, B) in [(192, 576, 5, 2, 27, 256), (576, 1152, 3, 1, 13, 256), (1152, 768, 3, 1, 13, 256), (256, 256, 3, 1, 56, 64)]:
    wq = torch.where(torch.rand(Cout, Cin, k, k, device=dev) < 0.5, -1.0, 1.0)
    Ho = H + 2 * pd - k + 1
    g = torch.randn(B, Cout, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
    wT = wq.flip(2, 3).transpose(0, 1).contiguous()
    p2 = k - 1 - pd
    wt = ops.pack_conv_weight_bf16x3(wT, "sign")
    Cb = ops.triple_ld_bytes(Cout, 16)
    nhwc = g.permute(0, 2, 3, 1)
    px = ops.split_bf16x3(nhwc.reshape(B * Ho * Ho, Cout), ld_bytes=Cb)
    padded = ops.pad_pixel_plane(px.data, B, Ho, Ho, (p2, p2))
    geom0 = ((1, 1), (0, 0), (1, 1))
    geomp = ((1, 1), (p2, p2), (1, 1))
    a = t(lambda: ops._conv_implicit(2, px.data, B, Ho, Ho, Cb // 4, k, k, geomp, wt.data, wt.ld_words, None, 1.0, None, Cin))
    b = t(lambda: ops._conv_implicit(2, padded, B, Ho + 2 * p2, Ho + 2 * p2, Cb // 4, k, k, geom0, wt.data, wt.ld_words, None, 1.0, None, Cin))
    c = t(lambda: ops.split_bf16x3(nhwc.reshape(B * Ho * Ho, Cout), ld_bytes=Cb))
    d = t(lambda: ops.pad_pixel_plane(px.data, B, Ho, Ho, (p2, p2)))
    y0 = ops._conv_implicit(2, px.data, B, Ho, Ho, Cb // 4, k, k, geomp, wt.data, wt.ld_words, None, 1.0, None, Cin)
    y1 = ops._conv_implicit(2, padded, B, Ho + 2 * p2, Ho + 2 * p2, Cb // 4, k, k, geom0, wt.data, wt.ld_words, None, 1.0, None, Cin)
    print(f"grad_input conv {Cout}->{Cin} k{k} {H}x{H} B{B}: bounds-checked kernel {a:.3f} ms, valid kernel on the padded plane {b:.3f} ms "
          f"(split {c:.3f} ms, pad pass {d:.3f} ms; same result: {bool(torch.equal(y0, y1))})", flush=True)
