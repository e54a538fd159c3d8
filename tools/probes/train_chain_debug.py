import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.AlexNetBin().to(dev).to(memory_format=torch.channels_last).train()
fused = bench_models.TrainFusedAlexNetBin(model)
x = torch.where(torch.randn(8, 3, 224, 224, device=dev) < 0, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
state = copy.deepcopy(model.state_dict())
with torch.no_grad():
    a = x; b = x
    mods = list(model.features.children()) + [list(model.classifieur.children())[0]]
    fm = list(fused.features.children())
    i = 0
    for blk in fm:
        name = type(blk).__name__
        if name == "FusedTrainPoolBnSign":
            n_mods = 2 + (blk.pool is not None) + (blk.hardtanh is not None)
        else:
            n_mods = 1
        for m in mods[i:i + n_mods]:
            a = m(a)
        i += n_mods
        model.load_state_dict(state)
        b = blk(b)
        d = (a - b).abs()
        print(name, tuple(a.shape), "max diff", float(d.max()), "frac diff", float((d > 0).float().mean()), a.is_contiguous(memory_format=torch.channels_last), b.is_contiguous(memory_format=torch.channels_last))
        b = a.clone() if False else b
