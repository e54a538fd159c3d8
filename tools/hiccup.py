import os, sys, time, gc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
# mimic bench: C2 phase first
from pytorch_quantize_impls_amd import ops
xx = torch.randn((4096, 4096), device=dev).sign_(); ww = torch.randn((4096, 4096), device=dev); yy = torch.empty((4096, 4096), device=dev)
for _ in range(60):
    ops.packed_gemm(ops.pack_activations(xx, "mfma"), ops.pack_weights(ww, "binary", "mfma"), None, out=yy, impl="mfma")
torch.cuda.synchronize()
model = bench_models.AlexNetBin(); bench_models.randomize_bn(model)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(7): model(x)
    fused = bench_models.FusedAlexNetBin(model)
    ts = []
    for i in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fused(x)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
print("iter: host ms / total ms")
for i, (a, b) in enumerate(ts):
    if i < 6 or b > 2.5: print(i, round(a, 2), round(b, 2))
print("alloc stats:", torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_reserved() / 1e9)
