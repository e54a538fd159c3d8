#!/usr/bin/env python
"""Which Python lines launch strided copies (aten::copy_ / contiguous / clone) during one AlexNet-Bin training step, and their device time."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench_models
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = bench_models.AlexNetBin().to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
target = torch.randint(0, 10, (B,), device=dev)
def one():
    model.zero_grad(set_to_none=True)
    F.nll_loss(model(x), target).backward()
for _ in range(3): one()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    one(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name in ("aten::copy_",) and ev.device_time_total > 0:
        here = [s for s in ev.stack if "/root/repo" in s or "repo/" in s or "pytorch_quantize" in s]
        key = (here[0] if here else (ev.stack[0] if ev.stack else "?")) + "  " + str(ev.input_shapes[:1])
        agg[key][0] += 1; agg[key][1] += ev.device_time_total
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{t / 1e3:8.3f} ms {n:4d}  {k}")
print("total copy ms", sum(v[1] for v in agg.values()) / 1e3)
