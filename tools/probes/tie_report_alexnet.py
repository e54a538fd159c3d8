import os, sys, importlib.util, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
spec = importlib.util.spec_from_file_location("bts", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_train_step.py")); bts = importlib.util.module_from_spec(spec); spec.loader.exec_module(bts)
dev = torch.device("cuda:0"); torch.manual_seed(0)
mt = bench_models.AlexNetBin().to(dev).to(memory_format=torch.channels_last).train()
xt = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
tt = torch.randint(0, 10, (256,), device=dev)
print(bts.tie_report(mt, xt, tt))
