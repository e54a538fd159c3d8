#!/usr/bin/env python
"""Results must not depend on what else the device is doing: the C2 step (pack + matrix-core GEMM, popcount GEMM), the
fused AlexNet forward and the DoReFa ResNet-18 forward are repeated while a long library GEMM runs on ANOTHER stream (the
workgroups of the kernels under test then start staggered, on whatever CUs are free) and compared bit for bit with the
results obtained on the quiet device."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_models  # noqa: E402
from pytorch_quantize_impls_amd import ops, synth  # noqa: E402


def run(iters: int = 60, verbose: bool = True) -> dict:
    dev = torch.device("cuda:0")
    B = K = N = 4096
    x = torch.from_numpy(synth.pm1(1, (B, K))).to(dev)
    w = torch.from_numpy(synth.uniform(2, (N, K), -1.0, 1.0)).to(dev)
    torch.manual_seed(0)
    alex = bench_models.AlexNetBin()
    bench_models.randomize_bn(alex)
    alex = alex.to(dev).to(memory_format=torch.channels_last).eval()
    xa = torch.randn(64, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    res = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
    bench_models.randomize_bn(res, seed=3)
    res = res.to(dev).to(memory_format=torch.channels_last).eval()
    xr = torch.randn(128, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)

    def c2(kind, impl):
        xp, wp = ops.pack_linear_operands(x, w, kind, impl)
        return ops.packed_gemm(xp, wp, None, impl=impl)

    # weight gradients on the pixel-major kernel (hand-counted vmcnt / lgkmcnt, mid-stage barrier): all three instances
    gen = torch.Generator(device=dev).manual_seed(11)
    wg = []
    for (n_, ci, co, hw, k, p) in ((32, 256, 384, 13, 3, 1), (16, 64, 64, 56, 3, 1), (16, 96, 160, 27, 5, 2)):
        xw = torch.where(torch.rand((n_, ci, hw, hw), device=dev, generator=gen) < 0.5, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
        gw = torch.randn((n_, co, hw, hw), device=dev, generator=gen).contiguous(memory_format=torch.channels_last)
        wg.append((xw, gw, k, p))
    ximg = torch.randn(16, 3, 224, 224, device=dev)
    gimg = torch.randn(16, 192, 55, 55, device=dev).contiguous(memory_format=torch.channels_last)

    # round 4: XNOR-Net AlexNet (per-tap conv, direct first layer with real-valued weights, digit-plane split-K int8 GEMM + head
    # kernel) and the first blocks of the ternary VGG-16 (first layer on fp16 pair pixels)
    from pytorch_quantize_impls_amd.layers import XNORConv2d, LinearXNOR
    axn = bench_models.alexnet_xnor()
    for mod in axn.modules():
        if isinstance(mod, (XNORConv2d, LinearXNOR)):
            mod.weight.data.normal_(0, 0.05)
    bench_models.randomize_bn(axn, seed=7)
    axn = axn.to(dev).to(memory_format=torch.channels_last).eval()
    vgg = bench_models.TernaryVGG16(num_classes=100, image=64, fc=512)
    bench_models.randomize_bn(vgg, seed=5)
    vgg = vgg.to(dev).to(memory_format=torch.channels_last).eval()
    vgg.features[0].binary_input = False
    xv = torch.randn(32, 3, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)

    cases = {"xnor alexnet module graph": lambda: axn(xa) * 1.0, "ternary vgg16 module graph": lambda: vgg(xv) * 1.0,
             "c2 binary mfma": lambda: c2("binary", "mfma"), "c2 ternary mfma": lambda: c2("ternary", "mfma"),
             "c2 binary popcount": lambda: c2("binary", "valu"), "alexnet module graph": lambda: alex(xa) * 1.0,
             "dorefa resnet18 module graph": lambda: res(xr) * 1.0,
             "grad_W 3x3 128x64 tile": lambda: ops.conv2d_grad_weight_pm(wg[0][0], wg[0][1], (3, 3), 1),
             "grad_W 3x3 64x64 tile": lambda: ops.conv2d_grad_weight_pm(wg[1][0], wg[1][1], (3, 3), 1),
             "grad_W 5x5": lambda: ops.conv2d_grad_weight_pm(wg[2][0], wg[2][1], (5, 5), 2),
             "grad_W strided first layer": lambda: ops.conv2d_grad_weight_s2d(ximg, gimg, (192, 3, 11, 11), 4, 2)}
    side = torch.cuda.Stream(device=dev)
    big = torch.randn(8192, 8192, device=dev)
    bad = {k: 0 for k in cases}
    with torch.no_grad():
        quiet = {k: f().clone() for k, f in cases.items()}
        torch.cuda.synchronize()
        for it in range(iters):
            with torch.cuda.stream(side):
                for _ in range(2):
                    big @ big
            for k, f in cases.items():
                y = f()
                if not torch.equal(y, quiet[k]):
                    bad[k] += 1
                    if verbose:
                        print(f"it {it} {k}: {int((y != quiet[k]).sum())} mismatches", flush=True)
        torch.cuda.synchronize()
    return bad


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = run(iters)
    print(f"{iters} rounds under concurrent load:", {k: f"{v} bad" for k, v in bad.items()})
    sys.exit(1 if any(bad.values()) else 0)


if __name__ == "__main__":
    main()
