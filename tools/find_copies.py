import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models, traceback
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.AlexNetBin(); bench_models.randomize_bn(model)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
fused = bench_models.FusedAlexNetBin(model)
with torch.no_grad():
    for _ in range(3): fused(x)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        fused(x)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::fill_", "aten::to", "aten::_to_copy") and ev.input_shapes:
        st = [f for f in (ev.stack or []) if "pytorch_quantize_impls_amd" in f or "bench_models" in f][:3]
        print(ev.name, ev.input_shapes[:2], " <- ", " | ".join(s.split("/")[-1] for s in st))
