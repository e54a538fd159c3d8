import sys, time, torch, importlib.util, os
sys.path.insert(0, ".")
import bench_models
from pytorch_quantize_impls_amd import lazy_train
spec = importlib.util.spec_from_file_location("bts", "tools/bench_train_step.py"); bts = importlib.util.module_from_spec(spec); spec.loader.exec_module(bts)
dev = torch.device("cuda:0"); torch.manual_seed(0)
B = 256
mt = bench_models.AlexNetBin().to(dev).to(memory_format=torch.channels_last).train()
xt = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
tt = torch.randint(0, 10, (B,), device=dev)
lazy_train.STATS.clear()
print("graph", bts.step_time(mt, mt, xt, tt), dict(lazy_train.STATS))
mf = bench_models.TrainFusedAlexNetBin(mt)
print("explicit", bts.step_time(mf, mt, xt, tt))
print("graph", bts.step_time(mt, mt, xt, tt))
with lazy_train.eager():
    print("mbm", bts.step_time(mt, mt, xt, tt))
from torch.profiler import ProfilerActivity, profile
for name, f in (("graph", mt), ("explicit", mf)):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        bts.step_time(f, mt, xt, tt, n=3)
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
    print(name, "total ms/step", sum(e.device_time_total for e in rows) / 5 / 1e3, "launches/step", sum(e.count for e in rows) / 5)
    for e in rows[:14]:
        print("   %8.1f us x%3d  %s" % (e.device_time_total / 5, e.count // 5, e.key[:100]))
