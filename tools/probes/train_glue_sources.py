#!/usr/bin/env python
"""Where do the torch elementwise / copy / fill launches of a training step come from?  One step under torch.profiler (CPU side,
with_stack); every aten op that launches a kernel is attributed to the innermost frame inside this package (or bench_models / the
caller) and counted.  MODEL=resnet18 (default) | alexnet."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import bench_models
from pytorch_quantize_impls_amd.functions import _fused
dev = torch.device("cuda:0")
torch.manual_seed(0)
if os.environ.get("MODEL", "resnet18") == "alexnet":
    model = bench_models.AlexNetBin().to(dev).to(memory_format=torch.channels_last).train()
    x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    loss_fn = lambda o, t: F.nll_loss(o, t)
else:
    model = bench_models.DorefaResNet18(w_bits=1, a_bits=4).to(dev).to(memory_format=torch.channels_last).train()
    x = torch.randn(256, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    loss_fn = lambda o, t: F.nll_loss(F.log_softmax(o, 1), t)
t = torch.randint(0, 10, (256,), device=dev)
_fused.DETECT_MODE = "remember"
def step():
    model.zero_grad(set_to_none=True)
    loss_fn(model(x), t).backward()
for _ in range(4):
    step()
torch.cuda.synchronize()
import sys as _sys
from torch.utils._python_dispatch import TorchDispatchMode
SKIP = ("aten.view", "aten.reshape", "aten._unsafe_view", "aten.permute", "aten.transpose", "aten.detach", "aten.empty", "aten.as_strided",
        "aten.slice", "aten.select", "aten.expand", "aten.t.", "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.empty_like",
        "aten.new_empty", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten._local_scalar", "aten.lift_fresh", "aten.unbind",
        "aten.narrow", "aten.split", "aten.chunk", "aten.flatten", "aten.view_as", "aten.result_type", "aten.empty_strided")
cnt = collections.Counter()
class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "(no package frame: autograd engine / torch module)"
            f = _sys._getframe(0).f_back
            chain = []
            while f is not None:
                fn = f.f_code.co_filename
                if "pytorch_quantize_impls_amd" in fn or "bench_models" in fn:
                    chain.append(f"{fn.split('pytorch_quantize_impls_amd/')[-1].split('/')[-1] if 'bench_models' in fn else fn.split('pytorch_quantize_impls_amd/')[-1]}:{f.f_lineno}({f.f_code.co_name})")
                    if len(chain) == 2:
                        break
                f = f.f_back
            if chain:
                where = " <- ".join(chain)
            shape = ""
            for a in args:
                if isinstance(a, torch.Tensor):
                    shape = str(tuple(a.shape))
                    break
            cnt[(name, where, shape if name.startswith(("aten.copy_", "aten.clone", "aten.add.", "aten.fill_", "aten.zero_")) else "")] += 1
        return func(*args, **(kwargs or {}))
with Rec():
    step()
torch.cuda.synchronize()
print("aten ops (views / metadata skipped) in one step:", sum(cnt.values()))
for k, v in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{v:4d} x {k[0]:34s} {k[2]:24s} {k[1]}")
