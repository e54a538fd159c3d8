#!/usr/bin/env python
"""AlexNet conv1 (3 -> 192, k 11, s 4, p 2, batch 256) piece by piece in both split modes: scale pass, s2d pack, conv."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(192, 3, 11, 11, device=dev)
ws = ops.s2d_weight(torch.sign(w), 4)


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("aminmax us", t(lambda: torch.aminmax(x)))
print("pow2_scale us", t(lambda: ops.pow2_scale(x)))
for mode in ("bf16x3", "f16x2"):
    with ops.float_split(mode):
        px, (Hs, Ws) = ops.s2d_triple_pack(x, 4, 2)
        wt = ops.pack_conv_weight_bf16x3(ws, "sign")
        meta = torch.empty(tuple(ws.shape), device="meta")
        print(mode, "s2d pack us", t(lambda: ops.s2d_triple_pack(x, 4, 2)), "plane MB", px.data.numel() * 2 / 1e6)
        print(mode, "conv us", t(lambda: ops.float_conv2d(None, meta, "sign", None, 1, 0, 1, weight_triples=wt, pixels=px,
                                                             in_shape=(256, 48, Hs, Ws))))
