import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
from pytorch_quantize_impls_amd.functions import _fused
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.DorefaResNet18(w_bits=1, a_bits=4).to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(256, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (256,), device=dev)
_fused.DETECT_MODE = "remember"
if os.environ.get("FUSED"):
    net = bench_models.TrainFusedDorefaResNet18(model)
    model_fwd = net
else:
    model_fwd = model
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.zero_grad(set_to_none=True)
    F.nll_loss(F.log_softmax(model_fwd(x), 1), t).backward()
    torch.cuda.synchronize(); print(i, round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
# the stem alone
stem = model.stem
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = stem(x); y.sum().backward()
    torch.cuda.synchronize(); print("stem fwd+bwd", round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
