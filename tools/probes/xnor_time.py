"""XNOR-Net AlexNet (batch 256): training-step time, and the eval-mode LinearXNOR layers on packed bits with the digit-plane int8
form against the fp16 pair form.  Under rocprofv3 --kernel-trace --stats the kernel split of both is in the stats file.
  python tools/probes/xnor_time.py [train] [fc]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench_models
import bench_train_step as bts
from pytorch_quantize_impls_amd import ops, packed
from pytorch_quantize_impls_amd.functions import _fused, BinaryConnectDeterministic
from pytorch_quantize_impls_amd.layers import XNORConv2d, LinearXNOR

dev = torch.device("cuda:0")
what = sys.argv[1:] or ["train", "fc"]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


if "train" in what:
    torch.manual_seed(0)
    mx = bench_models.alexnet_xnor()
    for mod in mx.modules():
        if isinstance(mod, (XNORConv2d, LinearXNOR)):
            mod.weight.data.normal_(0, 0.05)
            mod.bias.data.zero_()
    mx = mx.to(dev).to(memory_format=torch.channels_last).train()
    xt = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    tt = torch.randint(0, 10, (256,), device=dev)
    for rep in range(3):
        t, loss = bts.step_time(mx, mx, xt, tt)
        print(f"xnor alexnet training step: {t:.2f} ms (loss {loss:.4f}), library calls {dict(_fused.LIBRARY_PATHS)}", flush=True)
    del mx, xt

if "fc" in what:
    for (rows, N, K) in [(256, 4096, 9216), (256, 4096, 4096), (256, 10, 4096)]:
        lin = LinearXNOR(K, N, bias=True).to(dev)
        lin.weight.data.normal_(0, 0.05)
        lin.eval()
        x = BinaryConnectDeterministic.apply(torch.randn(rows, K, device=dev))
        act = packed.PackedActivation(packed.lookup(x, packed.ROWS_LAST), (rows, K))
        res = {}
        with torch.no_grad():
            for flag in (True, False):
                _fused.XNOR_LINEAR_DIGITS = flag
                res[flag] = timeit(lambda: _fused.packed_xnor_linear(lin, act))
            _fused.XNOR_LINEAR_DIGITS = True
            dg = _fused.xnor_linear_digits(lin)
            ld = int(dg[0].codes.shape[1])
            t_dig = timeit(lambda: ops.bits_alpha_digits(act.planes, dg[1], ld_bytes=ld))
        print(f"{rows}x{N}x{K}: digits {res[True]:.1f} us (digit planes alone {t_dig:.1f}), pairs {res[False]:.1f} us; plan {ops.splitk_plan(3 * rows, N, ld)}", flush=True)
