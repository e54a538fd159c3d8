#!/usr/bin/env python
"""Stress of the one-launch linear forward (qt_linear_fused_f32): many launches on one workspace, interleaved with other
work on the same and on a side stream, every result compared bit for bit with the two-launch route."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lf  # noqa: E402  (also puts the repo root on sys.path)
from pytorch_quantize_impls_amd import ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    B = K = N = 4096
    x = torch.from_numpy(synth.pm1(1, (B, K))).to(dev)
    ws = [torch.from_numpy(synth.uniform(s, (N, K), -1.0, 1.0)).to(dev) for s in (2, 3)]
    refs = {}
    for kind in ("binary", "ternary"):
        for i, w in enumerate(ws):
            xp, wp = ops.pack_linear_operands(x, w, kind, "mfma")
            refs[(kind, i)] = ops.packed_gemm(xp, wp, None, impl="mfma")
    side = torch.cuda.Stream(device=dev)
    big = torch.randn(8192, 8192, device=dev)
    bad = 0
    for it in range(iters):
        kind = ("binary", "ternary")[it & 1]
        i = (it >> 1) & 1
        mode = it % 5
        if mode == 1:
            with torch.cuda.stream(side):
                for _ in range(3):
                    big @ big                      # a long library GEMM occupying CUs on another stream
        elif mode == 2:
            big.mul_(1.0)                          # memory-bound work queued right in front on the same stream
        elif mode == 3:
            torch.cuda.synchronize()
        y = lf.linear_fused(x, ws[i], None, kind)
        ok = bool(torch.equal(y, refs[(kind, i)]))
        err = lf.linear_fused_error(dev, B, N, K)
        if not ok or err:
            bad += 1
            diff = int((y != refs[(kind, i)]).sum())
            print(f"it {it} mode {mode} kind {kind}: equal={ok} error_word={err:#x} mismatches={diff}", flush=True)
    torch.cuda.synchronize()
    print(f"{iters} launches, {bad} bad")


if __name__ == "__main__":
    main()
