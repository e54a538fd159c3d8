#!/usr/bin/env python
"""Which aten ops of the BinaryNet-AlexNet eval forward copy / allocate more than 1 MB (TorchDispatchMode; innermost frame of this package)."""
import os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench_models
from torch.utils._python_dispatch import TorchDispatchMode

dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = bench_models.AlexNetBin()
bench_models.randomize_bn(m)
m = m.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        y = m(x)

class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        big = [tuple(a.shape) + (str(a.dtype),) for a in list(args) + ([out] if isinstance(out, torch.Tensor) else [])
               if isinstance(a, torch.Tensor) and a.numel() * a.element_size() > (1 << 20)]
        if big and any(k in name for k in ("copy", "clone", "contiguous", "cat", "pad", "to", "fill", "zero", "empty_like", "index", "where")):
            fr = [f for f in traceback.extract_stack() if "pytorch_quantize_impls_amd" in f.filename or "bench_models" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"
            print(f"{name:40s} {big}  <- {where}")
        return out

with torch.no_grad(), Mode():
    y = m(x)
torch.cuda.synchronize()
