import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.AlexNetBin().to(dev).to(memory_format=torch.channels_last).train()
net = bench_models.TrainFusedAlexNetBin(model) if os.environ.get("FUSED") else model
x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (256,), device=dev)
for _ in range(6):
    model.zero_grad(set_to_none=True)
    F.nll_loss(net(x), t).backward()
torch.cuda.synchronize()
