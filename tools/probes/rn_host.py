import os, sys, time, cProfile, pstats, io
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
from pytorch_quantize_impls_amd.functions import _fused
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.DorefaResNet18().to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(256, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (256,), device=dev)
def one():
    model.zero_grad(set_to_none=True)
    loss = F.nll_loss(F.log_softmax(model(x), 1), t)
    loss.backward()
def timeit(n=5):
    for _ in range(2): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): one()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("verify  ", timeit())
_fused.DETECT_MODE = "remember"
print("remember", timeit())
# host-only time: enqueue without waiting
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): one()
h = (time.perf_counter() - t0) / 5 * 1e3
torch.cuda.synchronize()
print("host enqueue per step (remember)", h)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): one()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:6000])
