"""Can a whole training step (forward + backward through this backend's autograd Functions) be captured as a hipGraph?"""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
from pytorch_quantize_impls_amd.functions import _fused
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
torch.manual_seed(0)
if which == "resnet18":
    model = bench_models.DorefaResNet18(w_bits=1, a_bits=4); xs = (256, 3, 32, 32); fwd = lambda m, t: F.log_softmax(m(t), 1)
else:
    model = bench_models.AlexNetBin(); xs = (256, 3, 224, 224); fwd = lambda m, t: m(t)
model = model.to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(xs, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (xs[0],), device=dev)
_fused.DETECT_MODE = "remember"
def one():
    model.zero_grad(set_to_none=False)
    loss = F.nll_loss(fwd(model, x), t); loss.backward(); return loss
for p in model.parameters():
    p.grad = torch.zeros_like(p)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): one()
torch.cuda.synchronize()
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager (remember) ms", timeit(one))
g = torch.cuda.CUDAGraph()
try:
    torch.cuda.set_sync_debug_mode("warn")
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            loss = one()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in model.named_parameters()}
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(ref[k], p.grad) for k, p in model.named_parameters())
    print("captured; replay reproduces the gradients:", same, "loss", float(loss))
    print("graph replay ms", timeit(g.replay))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:400])
