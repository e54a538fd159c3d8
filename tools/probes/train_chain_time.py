import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.AlexNetBin().to(dev).to(memory_format=torch.channels_last).train()
fused = bench_models.TrainFusedAlexNetBin(model)
x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
t = torch.randint(0, 10, (256,), device=dev)
def step_time(net, n=5):
    def one():
        model.zero_grad(set_to_none=True)
        F.nll_loss(net(x), t).backward()
    for _ in range(2): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): one()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("module graph ms", step_time(model)); print("fused training chain ms", step_time(fused))
