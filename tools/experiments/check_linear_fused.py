"""GPU check + timing of qt_linear_fused_f32 against the two-launch route (qt_pack_pair_nib_f32 + qt_nib_gemm).

    python tools/experiments/check_linear_fused.py [--iters 200] [--K 4096]

Bit-exact comparison on fresh data every launch (a stale hand-off would show up as a mismatch), edge values in the
weights, ternary weights, bias; then stream-time per call of both routes (HIP events around `iters` back-to-back calls).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lf  # noqa: E402  (also puts the repo root on sys.path)
from pytorch_quantize_impls_amd import ops  # noqa: E402


def two_launch(x, w, bias, kind):
    xp, wp = ops.pack_linear_operands(x, w, kind, "mfma")
    return ops.packed_gemm(xp, wp, bias, impl="mfma")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M = N = 4096
    K = a.K
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    ok = True
    for r in range(a.rounds):
        kind = "ternary" if r % 3 == 2 else "binary"
        x = torch.where(torch.rand((M, K), device=dev, generator=g) < 0.5, -1.0, 1.0)
        w = torch.randn((N, K), device=dev, generator=g) * (0.6 if kind == "ternary" else 1.0 / 64)
        if r == 1:   # edge values: +-0, NaN, subnormals, thresholds
            edge = torch.tensor([0.0, -0.0, float("nan"), 1e-45, -1e-45, 0.5, -0.5, float("inf"), -float("inf")], device=dev)
            w.view(-1)[: edge.numel() * 1000] = edge.repeat(1000)
        bias = torch.randn(N, device=dev, generator=g) if r % 2 else None
        y = lf.linear_fused(x, w, bias, kind)
        ref = two_launch(x, w, bias, kind)
        torch.cuda.synchronize()
        err = lf.linear_fused_error(dev, M, N, K)
        same = torch.equal(y, ref)
        nbad = int((y != ref).sum().item()) if not same else 0
        print(f"round {r} kind={kind} bias={bias is not None}: equal={same} mismatches={nbad} error_word={err}", flush=True)
        if not same and r == 0:
            bad = (y != ref)
            tiles = bad.view(16, 256, 16, 256).sum(dim=(1, 3))
            print("mismatches per 256x256 tile (rows = tile_m):")
            print(tiles.cpu().numpy())
            d = (y - ref)[bad]
            print("diff stats: min", d.min().item(), "max", d.max().item(), "mean |d|", d.abs().mean().item())
            rows = bad.sum(dim=1).nonzero().flatten()
            cols = bad.sum(dim=0).nonzero().flatten()
            print("bad rows:", rows.numel(), rows[:40].tolist())
            print("bad cols:", cols.numel(), cols[:40].tolist())
        ok = ok and same and err == 0
    # back-to-back launches on changing data (the counter sets alternate; a stale panel would be caught at the end)
    xs = [torch.where(torch.rand((M, K), device=dev, generator=g) < 0.5, -1.0, 1.0) for _ in range(3)]
    ws = [torch.randn((N, K), device=dev, generator=g) / 64 for _ in range(3)]
    outs = [lf.linear_fused(xs[i % 3], ws[(i * 2) % 3], None, "binary").clone() for i in range(9)]
    torch.cuda.synchronize()
    for i in range(9):
        ref = two_launch(xs[i % 3], ws[(i * 2) % 3], None, "binary")
        same = torch.equal(outs[i], ref)
        ok = ok and same
        if not same:
            print(f"back-to-back launch {i}: MISMATCH {(outs[i] != ref).sum().item()}")
    print("back-to-back launches on changing data:", "ok" if ok else "FAILED", flush=True)

    x, w = xs[0], ws[0]
    y = torch.empty((M, N), device=dev)

    def timeit(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters

    t_f = timeit(lambda: lf.linear_fused(x, w, None, "binary", out=y))
    t_2 = timeit(lambda: two_launch(x, w, None, "binary"))
    byt = 4.0 * (M * K + N * K + M * N)
    print(f"fused: {t_f:.1f} us/call ({byt / t_f / 1e6:.2f} TB/s = {byt / t_f / 8e6:.3f} of 8 TB/s)   "
          f"two-launch: {t_2:.1f} us/call ({byt / t_2 / 8e6:.3f})   error_word={lf.linear_fused_error(dev, M, N, K)}")
    print("RESULT", "PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
