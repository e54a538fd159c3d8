#!/usr/bin/env python
"""Per-launch time of the one-pass 3x3 first-layer kernel at VGG conv1_1's shape (batch 256), the three epilogues, inside one event pair."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
wq = torch.randint(-1, 2, (64, 3, 3, 3), device=dev).float()
frag = ops.pack_first3x3_weight(wq)
alpha, beta = torch.randn(64, device=dev), torch.randn(64, device=dev)
for name, epi in (("nib", ops.NibEpilogue(alpha, beta, (1, 1))), ("bits", (alpha, beta)), ("f32", None)):
    for _ in range(3):
        ops.conv_first3x3(x, frag, 64, None, epi=epi)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv_first3x3(x, frag, 64, None, epi=epi)
    e1.record()
    torch.cuda.synchronize()
    print(f"QT_F3_OCC={os.environ.get('QT_F3_OCC', '2')} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
