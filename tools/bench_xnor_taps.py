#!/usr/bin/env python
"""Per-tap scaled conv (XNORConv2d on +-1 activations, csrc/conv_taps.hip) against the un-scaled fp4 conv (BinConv2d) on the same
pre-packed operands at the AlexNet shapes, batch 256: what the Horner multiplies on the accumulators cost.

    python tools/bench_xnor_taps.py [--batch 256] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
from pytorch_quantize_impls_amd import ops  # noqa: E402

SHAPES = [("conv2", 192, 576, 27, 5, 2), ("conv3", 576, 1152, 13, 3, 1), ("conv4", 1152, 768, 13, 3, 1),
          ("conv5", 768, 256, 13, 3, 1)]


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    for name, Cin, Cout, H, k, p in SHAPES:
        g = torch.Generator(device=dev).manual_seed(Cin)
        x = torch.randn((args.batch, Cin, H, H), device=dev, generator=g).sign_().contiguous(memory_format=torch.channels_last)
        w = torch.randn((Cout, Cin, k, k), device=dev, generator=g) * 0.05
        px = ops.pack_pixels_nib(x, ld=ops.pixel_ld_nib_taps(Cin))
        wb = ops.pack_conv_weight_nib(w, "binary", cw=px.ld)
        ws = ops.pack_conv_weight_nib(w, "sign", cw=px.ld)
        ts = ops.xnor_tap_prep(w)
        shape = (args.batch, Cin, H, H)
        macs = args.batch * H * H * Cout * Cin * k * k
        t_bin = timed(lambda: ops.conv2d_nib(px, shape, wb, (k, k), None, 1, p, 1), args.iters)
        t_tap = timed(lambda: ops.conv2d_nib_taps(px, shape, ws, (k, k), ts.fwd, None, 1, p, 1), args.iters)
        # ... and on the physically padded plane (what the fused inference path feeds: un-padded kernels)
        xp = torch.nn.functional.pad(x, (p, p, p, p))
        pxp = ops.pack_pixels_nib(xp.contiguous(memory_format=torch.channels_last), ld=ops.pixel_ld_nib_taps(Cin))
        # (padding pixels must be fp4 zeros, not +1: zero their words)
        v = pxp.words.view(args.batch, H + 2 * p, H + 2 * p, -1)
        v[:, :p] = 0; v[:, -p:] = 0; v[:, :, :p] = 0; v[:, :, -p:] = 0
        shp = (args.batch, Cin, H + 2 * p, H + 2 * p)
        t_bin_v = timed(lambda: ops.conv2d_nib(pxp, shp, wb, (k, k), None, 1, 0, 1), args.iters)
        t_tap_v = timed(lambda: ops.conv2d_nib_taps(pxp, shp, ws, (k, k), ts.fwd, None, 1, 0, 1), args.iters)
        y0 = ops.conv2d_nib_taps(px, shape, ws, (k, k), ts.fwd, None, 1, p, 1)
        y1 = ops.conv2d_nib_taps(pxp, shp, ws, (k, k), ts.fwd, None, 1, 0, 1)
        out[name] = {"bin_us": t_bin, "taps_us": t_tap, "ratio": t_tap / t_bin, "bin_valid_us": t_bin_v, "taps_valid_us": t_tap_v,
                     "ratio_valid": t_tap_v / t_bin_v, "taps_frac_fp4_peak": 2 * macs / (t_tap * 1e-6) / 1e16,
                     "taps_valid_frac_fp4_peak": 2 * macs / (t_tap_v * 1e-6) / 1e16,
                     "bin_valid_frac_fp4_peak": 2 * macs / (t_bin_v * 1e-6) / 1e16,
                     "padded_equals_unpadded": bool(torch.equal(y0, y1))}
        print(name, json.dumps(out[name]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
