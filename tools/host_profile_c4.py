#!/usr/bin/env python
"""cProfile of the host side of the fused C4 forward (eager launches are host-bound at this size)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
m4 = bench_models.DorefaResNet18(); bench_models.randomize_bn(m4, seed=3)
for m in m4.modules():
    if isinstance(m, torch.nn.BatchNorm2d): m.running_var.mul_(4.0)
m4 = m4.to(dev).to(memory_format=torch.channels_last).eval()
x4 = torch.randn((256, 3, 32, 32), device=dev).contiguous(memory_format=torch.channels_last)
f4 = bench_models.FusedDorefaResNet18(m4)
ops.ASSUME_CODES_FIT = True
with torch.no_grad():
    for _ in range(5): f4(x4)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): f4(x4)
    pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
