#!/usr/bin/env python
"""AlexNet conv1 (3 -> 192, 11 x 11, stride 4, padding 2, batch 256): the direct first-layer kernel (csrc/conv_first_direct.hip)
against the round-3 route (space-to-depth pack + implicit GEMM), fp32 output and threshold-bit output, BinConv2d and XNORConv2d.

    python tools/bench_conv1.py [--batch 256]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from pytorch_quantize_impls_amd import lazy, ops  # noqa: E402
from pytorch_quantize_impls_amd.functions import BinaryConnect  # noqa: E402
from pytorch_quantize_impls_amd.layers import BinConv2d, XNORConv2d  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(args.batch, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    macs = args.batch * 55 * 55 * 192 * 363
    out = {}
    for name, cls in (("bin", BinConv2d), ("xnor", XNORConv2d)):
        conv = cls(3, 192, 11, stride=4, padding=2).to(dev)
        conv.weight.data.normal_(0, 0.3)
        conv.binary_input = False
        bn = torch.nn.BatchNorm2d(192).to(dev)
        bn.running_var.uniform_(5, 50)
        block = torch.nn.Sequential(conv, torch.nn.MaxPool2d(3, 2), bn, torch.nn.Hardtanh(), BinaryConnect()).eval()
        conv.eval()
        row = {}
        for direct in (True, False):
            ops.FIRST_DIRECT = direct
            with torch.no_grad():
                with lazy.eager():
                    t_f32 = timed(lambda: conv(x))
                t_blk = timed(lambda: block(x)._qt.force())
            if direct:     # the kernel alone, threshold epilogue / fp32 epilogue
                from pytorch_quantize_impls_amd.layers.fused import fold_batchnorm
                al, be = fold_batchnorm(bn)
                if name == "bin":
                    fw = ops.pack_first_layer_weight(conv.weight.detach(), 4)
                else:
                    fw = ops.pack_first_layer_weight(ops.xnor_weight(conv.weight.detach(), 2)[0], 4, real=True)
                row["kernel_bits_us"] = timed(lambda: ops.conv_first_direct(x, fw, conv.bias, 4, 2, epi=(al, be)))
                row["kernel_fp32_us"] = timed(lambda: ops.conv_first_direct(x, fw, conv.bias, 4, 2))
            tag = "direct" if direct else "r3_route"
            row[tag] = {"fp32_out_us": t_f32, "fused_block_us": t_blk,
                        "fused_block_frac_f16_peak_2term": 2 * 2 * macs / (t_blk * 1e-6) / 2.5e15}
        ops.FIRST_DIRECT = True
        out[name] = row
        print(name, json.dumps(row))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
