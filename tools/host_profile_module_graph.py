#!/usr/bin/env python
"""cProfile of the host side of the UN-MODIFIED module graph in eval mode (deferred activations, lazy.py): C4 (DoReFa ResNet-18, batch 256)
and AlexNet-Bin at batch 1 — the eager paths VERDICT r4 item 7 measures (1.96 ms / 0.655 ms).  MODEL=c4|alexnet"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
which = os.environ.get("MODEL", "c4")
if which == "c4":
    m = bench_models.DorefaResNet18(); bench_models.randomize_bn(m, seed=3)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d): mod.running_var.mul_(4.0)
    x = torch.randn((256, 3, 32, 32), device=dev)
else:
    m = bench_models.AlexNetBin(); bench_models.randomize_bn(m)
    x = torch.randn((int(os.environ.get("B", "1")), 3, 224, 224), device=dev)
m = m.to(dev).to(memory_format=torch.channels_last).eval()
x = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(5): m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): y = m(x)
    torch.cuda.synchronize()
    print(f"{which}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms / forward (eager, un-modified module graph)")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): y = m(x)
    float(y.sum())
    pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "35")))
