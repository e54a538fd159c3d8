"""cProfile of the host side of (a) the deferred C4 forward, (b) the ResNet-18 training step: where the Python time goes."""
import cProfile, pstats, sys, io, time, torch
sys.path.insert(0, ".")
import bench_models
from pytorch_quantize_impls_amd import lazy, lazy_train, _lib
from pytorch_quantize_impls_amd.functions import _fused
dev = torch.device("cuda:0"); torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "infer"
B = 256
m = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
bench_models.randomize_bn(m, 5)
m = m.to(dev).to(memory_format=torch.channels_last)
x = torch.randn(B, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (B,), device=dev)
if which == "infer":
    m.eval()
    def step():
        with torch.no_grad():
            return m(x)
else:
    m.train()
    if which == "train_remember":
        _fused.DETECT_MODE = "remember"
    def step():
        m.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(x), t)
        loss.backward()
        return loss
for _ in range(5):
    step()
torch.cuda.synchronize()
c0 = sum(_lib.call_counts.values())
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(which, "ms/step", (time.perf_counter() - t0) / 20 * 1e3, "C-ABI calls/step", (sum(_lib.call_counts.values()) - c0) / 20)
t0 = time.perf_counter()
for _ in range(20):
    step()
host = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print("host-only ms/step", host)
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
print("kernels/step", sum(e.count for e in rows) / 5, "device ms/step", sum(e.device_time_total for e in rows) / 5e3)
for e in rows[:25]:
    print("   %8.1f us x%5.1f  %s" % (e.device_time_total / 5, e.count / 5, e.key[:110]))
