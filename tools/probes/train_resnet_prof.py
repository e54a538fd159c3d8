"""ResNet-18 (DoReFa W{W_BITS}A4, CIFAR shape, batch 256) training steps for `rocprofv3 --kernel-trace --stats`."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
from pytorch_quantize_impls_amd.functions import _fused
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.DorefaResNet18(w_bits=int(os.environ.get("W_BITS", "1")), a_bits=4).to(dev).to(memory_format=torch.channels_last).train()
x = torch.randn(256, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (256,), device=dev)
_fused.DETECT_MODE = "remember"
net = bench_models.TrainFusedDorefaResNet18(model) if os.environ.get("FUSED") else model
for _ in range(int(os.environ.get("STEPS", "8"))):
    model.zero_grad(set_to_none=True)
    F.nll_loss(F.log_softmax(net(x), 1), t).backward()
torch.cuda.synchronize()
print("library paths:", dict(_fused.LIBRARY_PATHS))
