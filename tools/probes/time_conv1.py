#!/usr/bin/env python
"""Kernel time of the direct first-layer conv at AlexNet conv1, batch 256: four variants (+-1 / real weights x fp32 / bits epilogue)."""
import os, sys, torch
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
B = int(os.environ.get("B", "256"))
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(192, 3, 11, 11, device=dev).sign()
fw = ops.pack_first_layer_weight(w, 4)
fwr = ops.pack_first_layer_weight(w * 0.037, 4, real=True)
al, be = torch.ones(192, device=dev), torch.zeros(192, device=dev)
def timed(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
res = {}
res["pm1_fp32"] = timed(lambda: ops.conv_first_direct(x, fw, None, 4, 2))
res["pm1_bits"] = timed(lambda: ops.conv_first_direct(x, fw, None, 4, 2, epi=(al, be)))
res["real_fp32"] = timed(lambda: ops.conv_first_direct(x, fwr, None, 4, 2))
res["real_bits"] = timed(lambda: ops.conv_first_direct(x, fwr, None, 4, 2, epi=(al, be)))
print({k: round(v, 1) for k, v in res.items()}, flush=True)
y = ops.conv_first_direct(x[:8], fw, None, 4, 2)
ref = torch.nn.functional.conv2d(x[:8].double(), w.double(), None, 4, 2)
y = y.reshape(8, 55, 55, 192).permute(0, 3, 1, 2)
print("err", float((y.double() - ref).abs().max() / ref.abs().max()))
