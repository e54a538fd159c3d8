#!/usr/bin/env python
"""AlexNet conv1 (batch 256) through the direct first-layer kernel, 10 launches per variant: the workload of the rocprofv3 --pmc passes
that name what the kernel waits for (profiles/r4_conv1_pmc.md)."""
import os, sys, torch
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(192, 3, 11, 11, device=dev).sign()
fw = ops.pack_first_layer_weight(w, 4)
al, be = torch.randn(192, device=dev), torch.randn(192, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "bits"
for _ in range(10):
    if which == "bits":
        ops.conv_first_direct(x, fw, None, 4, 2, epi=(al, be))
    else:
        ops.conv_first_direct(x, fw, None, 4, 2)
torch.cuda.synchronize()
