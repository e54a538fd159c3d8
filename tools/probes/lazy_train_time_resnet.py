import sys, time, torch, importlib.util, os
sys.path.insert(0, ".")
import bench_models
from pytorch_quantize_impls_amd import lazy_train
from pytorch_quantize_impls_amd.functions import _fused
spec = importlib.util.spec_from_file_location("bts", "tools/bench_train_step.py"); bts = importlib.util.module_from_spec(spec); spec.loader.exec_module(bts)
dev = torch.device("cuda:0"); torch.manual_seed(0)
B = 256
mr = bench_models.DorefaResNet18(w_bits=1, a_bits=4).to(dev).to(memory_format=torch.channels_last).train()
xr = torch.randn(B, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
tt = torch.randint(0, 10, (B,), device=dev)
lsm = lambda net: (lambda t: torch.nn.functional.log_softmax(net(t), 1))
ex = bench_models.TrainFusedDorefaResNet18(mr)
def mbm(t):
    with lazy_train.eager():
        return torch.nn.functional.log_softmax(mr(t), 1)
for mode in ("sync", "remember"):
    _fused.DETECT_MODE = mode if mode == "remember" else _fused.DETECT_MODE
    for rep in range(2):
        for name, f in (("graph", lsm(mr)), ("explicit", lsm(ex)), ("mbm", mbm)):
            print(mode, name, "%.2f ms" % bts.step_time(f, mr, xr, tt, n=10)[0])
from torch.profiler import ProfilerActivity, profile
for name, f in (("graph", lsm(mr)), ("explicit", lsm(ex))):
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        bts.step_time(f, mr, xr, tt, n=3)
    rows = [e for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
    k = [e for e in prof.key_averages()]
    print(name, "kernel ms/step", sum(e.self_device_time_total for e in k) / 5 / 1e3, "cpu self ms/step", sum(e.self_cpu_time_total for e in k) / 5 / 1e3)
