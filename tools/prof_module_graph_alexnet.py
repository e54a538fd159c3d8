#!/usr/bin/env python
"""The UN-MODIFIED AlexNet module graph (batch 256, 10 forwards; argument `xnor`: the XNOR-Net flavour) — target for
rocprofv3 --kernel-trace --stats: the kernels the deferred graph really launches, classifier included."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "xnor":
    from pytorch_quantize_impls_amd.layers import XNORConv2d, LinearXNOR
    model = bench_models.alexnet_xnor()
    for mod in model.modules():
        if isinstance(mod, (XNORConv2d, LinearXNOR)):
            mod.weight.data.normal_(0, 0.05)
else:
    model = bench_models.AlexNetBin()
bench_models.randomize_bn(model)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3): model(x)
    torch.cuda.synchronize()
    for _ in range(10): model(x)
torch.cuda.synchronize()
