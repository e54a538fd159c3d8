"""Fused BinaryNet-AlexNet forward, batch 256, with and without the speculative first-layer pack (ops.S2D_SPEC_SCALE) in one process."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench_models
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench_models.AlexNetBin(); bench_models.randomize_bn(model)
model = model.to(dev).to(memory_format=torch.channels_last).eval()
fused = bench_models.FusedAlexNetBin(model)
first = list(fused.net.features.children())[0]
x = torch.randn((256, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)


def t(fn, it=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


spec = ops.S2D_SPEC_SCALE
with torch.no_grad():
    outs = {}
    for rep in range(2):
        for mode in (spec, None):
            ops.S2D_SPEC_SCALE = mode
            outs[mode] = fused(x).clone()
            print(f"S2D_SPEC_SCALE={mode!s:24s} first block {t(lambda: first(x)):7.1f} us   fused forward {t(lambda: fused(x)):7.1f} us")
    ops.S2D_SPEC_SCALE = spec
    d = (outs[spec] - outs[None]).abs().max().item()
    print("max |logit difference| between the two scales:", d, " argmax equal:", bool((outs[spec].argmax(1) == outs[None].argmax(1)).all()))
