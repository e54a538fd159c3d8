#!/usr/bin/env python
"""Kernels inside the captured hipGraph of the un-modified ternary VGG-16 forward (C5, batch 256 at 224 x 224): under rocprofv3
--kernel-trace (tools/probes/c4_graph_seq.sh with SCRIPT=c5_graph_kernels.py), REPLAYS replays."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import utils
dev = torch.device("cuda:0")
torch.manual_seed(5)
m = bench_models.TernaryVGG16(num_classes=1000, image=224)
bench_models.randomize_bn(m, seed=4)
m = m.to(dev).to(memory_format=torch.channels_last).eval()
m.features[0].binary_input = False
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    g = utils.graphed(m, x)
    torch.cuda.synchronize()
    for _ in range(int(os.environ.get("REPLAYS", "20"))): g(x)
torch.cuda.synchronize()
