import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_quantize_impls_amd import ops, _lib
dev = torch.device("cuda:0")
x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
work = torch.zeros(2048, dtype=torch.int32, device=dev); out = torch.empty(2, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(fn, n=50):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("kernel only us", t(lambda: _lib.call("qt_f16x2_absmax_scale_f32", x.data_ptr(), x.numel(), work.data_ptr(), out.data_ptr(), st)))
print("scale", out.tolist(), float(x.abs().max()))
print("pow2_scale() us", t(lambda: ops.pow2_scale(x)))
y = torch.empty_like(x)
print("copy us (308 MB moved)", t(lambda: y.copy_(x)))
