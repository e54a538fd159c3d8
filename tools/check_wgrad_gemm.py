#!/usr/bin/env python
"""Weight gradient of stride-1 binarised convs on the matrix cores (ops.conv2d_grad_weight_gemm, csrc/wgrad.hip) against
fp64 on small shapes and against MIOpen's fp32 weight gradient (time + error) at the AlexNet / VGG shapes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_quantize_impls_amd import ops  # noqa: E402


def pm1(shape, dev, g):
    return (torch.randint(0, 2, shape, generator=g, device=dev).float() * 2 - 1)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    ok = True
    for (N, Cin, Cout, H, k, p, cl) in [(3, 5, 7, 9, 3, 1, False), (4, 32, 48, 13, 3, 1, True), (2, 16, 8, 11, 5, 2, True),
                                        (5, 64, 64, 17, 3, 0, False), (3, 24, 40, 8, 1, 0, True), (2, 8, 8, 12, 7, 3, False)]:
        x = pm1((N, Cin, H, H + 2), dev, g)
        Ho, Wo = H + 2 * p - k + 1, H + 2 + 2 * p - k + 1
        go = torch.randn((N, Cout, Ho, Wo), device=dev, generator=g)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
            go = go.contiguous(memory_format=torch.channels_last)
        w = torch.randn((Cout, Cin, k, k), device=dev, generator=g) * 0.8
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), go.double(), stride=1, padding=p)
        got = ops.conv2d_grad_weight_gemm(x, go, (k, k), p)
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        gotm = ops.conv2d_grad_weight_gemm(x, go, (k, k), p, weight=w)
        refm = torch.where(w.abs() <= 1.001, ref, torch.zeros_like(ref))
        errm = float((gotm.double() - refm).abs().max() / ref.abs().max())
        print(f"N={N} {Cin}->{Cout} {H}x{H + 2} k{k} p{p} cl={cl}: err {err:.2e}  masked {errm:.2e}")
        ok &= err <= 1e-5 and errm <= 1e-5
    shapes = [("alex conv2", 256, 192, 576, 27, 5, 2), ("alex conv3", 256, 576, 1152, 13, 3, 1), ("alex conv4", 256, 1152, 768, 13, 3, 1),
              ("alex conv5", 256, 768, 256, 13, 3, 1), ("vgg 64->64@224 (b32)", 32, 64, 64, 224, 3, 1),
              ("vgg 128->128@112 (b64)", 64, 128, 128, 112, 3, 1), ("vgg 256->256@56", 256, 256, 256, 56, 3, 1),
              ("vgg 512->512@28", 256, 512, 512, 28, 3, 1), ("vgg 512->512@14", 256, 512, 512, 14, 3, 1)]
    for name, N, Cin, Cout, H, k, p in shapes:
        x = pm1((N, Cin, H, H), dev, g).contiguous(memory_format=torch.channels_last)
        Ho = H + 2 * p - k + 1
        go = torch.randn((N, Cout, Ho, Ho), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        t_lib = timeit(lambda: torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), go, stride=1, padding=p), 3)
        ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), go, stride=1, padding=p)
        got = ops.conv2d_grad_weight_gemm(x, go, (k, k), p)
        if got is None:
            print(f"{name}: route declined; MIOpen {t_lib:.2f} ms")
            continue
        t_gemm = timeit(lambda: ops.conv2d_grad_weight_gemm(x, go, (k, k), p), 3)
        err = float((got - ref).abs().max() / ref.abs().max())
        flop = 2.0 * N * Ho * Ho * Cout * Cin * k * k
        print(f"{name}: gemm {t_gemm:.2f} ms ({flop / t_gemm / 1e9:.0f} useful TFLOP/s)  MIOpen {t_lib:.2f} ms  diff {err:.1e}")
    print("RESULT", "PASS" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
