"""Which aten::copy_ / mul / add calls of a ResNet-18 training step are big?  (torch.profiler, shapes + python stacks)"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
from pytorch_quantize_impls_amd.functions import _fused
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
torch.manual_seed(0)
alex = os.environ.get("MODEL") == "alexnet"
model = (bench_models.AlexNetBin() if alex else bench_models.DorefaResNet18(w_bits=1, a_bits=4)).to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(*((256, 3, 224, 224) if alex else (256, 3, 32, 32)), device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (256,), device=dev)
_fused.DETECT_MODE = "remember"
net = ((bench_models.TrainFusedAlexNetBin if alex else bench_models.TrainFusedDorefaResNet18)(model)) if os.environ.get("FUSED") else model
def step():
    model.zero_grad(set_to_none=True)
    F.nll_loss(net(x) if alex else F.log_softmax(net(x), 1), t).backward()
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
want = sys.argv[1:] or ["aten::copy_", "aten::mul", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::abs", "aten::mean", "aten::fill_", "aten::zero_", "aten::zeros", "aten::mul_"]
rows = []
for e in prof.events():
    if e.name in want and e.device_time > 0:
        st = [s for s in (e.stack or []) if "pytorch_quantize_impls_amd" in s or "bench_models" in s]
        rows.append((e.device_time, e.name, str(e.input_shapes)[:70], (st[0] if st else "(autograd / torch)")[-90:]))
import collections
agg = collections.defaultdict(lambda: [0.0, 0])
for d, n, s, st in rows:
    key = s if os.environ.get("BY") == "shape" else st
    agg[(n, key)][0] += d; agg[(n, key)][1] += 1
for (n, st), (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{d:9.1f} us {c:4d}x {n:18s} {st}")
