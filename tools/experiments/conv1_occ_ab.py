#!/usr/bin/env python
"""A/B of the direct first-layer conv's workgroup shapes at AlexNet conv1, batch 256: 4 waves (2 x 3 accumulator tiles per wave, two
waves per SIMD) against 6 waves (2 x 2 tiles, three waves per SIMD), every epilogue; the outputs must be identical."""
import os, sys, torch
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
from pytorch_quantize_impls_amd import ops, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
B = int(os.environ.get("B", "256"))
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(192, 3, 11, 11, device=dev).sign()
fw = ops.pack_first_layer_weight(w, 4)
fwr = ops.pack_first_layer_weight(w * 0.037, 4, real=True)
al, be = torch.randn(192, device=dev), torch.randn(192, device=dev)
def timed(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
legs = {"pm1_fp32": lambda: ops.conv_first_direct(x, fw, None, 4, 2),
        "pm1_bits": lambda: ops.conv_first_direct(x, fw, None, 4, 2, epi=(al, be)),
        "real_fp32": lambda: ops.conv_first_direct(x, fwr, None, 4, 2),
        "real_bits": lambda: ops.conv_first_direct(x, fwr, None, 4, 2, epi=(al, be))}
outs = {}
for waves in (2, 3, 2, 3):
    _lib.call("qt_conv_first_direct_config", waves)
    res = {}
    for k, fn in legs.items():
        res[k] = round(timed(fn), 1)
        o = fn()
        o = o.sign if isinstance(o, ops.BitPlanes) else o
        if k in outs:
            assert torch.equal(outs[k], o), (k, waves)
        outs[k] = o.clone()
    print(waves, "workgroups per CU:", res, flush=True)
_lib.call("qt_conv_first_direct_config", 0)
