"""CPU baseline for bench.py — TEST/BENCH INFRASTRUCTURE ONLY (never imported by the product).

What the reference executes for the benchmarked configs, restated op for op in torch on CPU
tensors (the reference's Python cannot travel to the GPU box):

  LinearBin.forward, training mode (layers/binary_layers.py:42-44):
      w_b = safeSign(W)            -> torch.sign + masked write   (functions/common.py:4-7)
      y   = F.linear(x, w_b, bias) -> ATen addmm -> MKL sgemm
  BinConv2d.forward (layers/binary_layers.py:103-105): the same with F.conv2d.

Equality of this port with the imported reference is frozen in tests/golden (G4, G7 digests).
"""
import time

import torch
import torch.nn.functional as F


def safe_sign(w: torch.Tensor) -> torch.Tensor:
    r = torch.sign(w)
    r[r == 0] = 1
    return r


def linear_bin_forward(x, weight, bias=None):
    return F.linear(x, safe_sign(weight), bias)


def bin_conv2d_forward(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return F.conv2d(x, safe_sign(weight), bias, stride, padding, dilation, groups)


def sequential_forward(seq, x):
    """An nn.Sequential of this package's binarised layers + torch modules, evaluated with the REFERENCE's op sequence
    only (torch.sign + masked write, F.conv2d / F.linear in fp32, the torch modules as they are) on whatever device the
    tensors live on: what the un-modified reference computes in eval mode (layers/binary_layers.py:46,106; the eval-mode
    weight already holds sign(W), functions/binary_connect.py:22-28 for the BinaryConnect modules)."""
    for m in seq:
        name = type(m).__name__
        if name in ("BinConv2d", "TerConv2d"):
            x = F.conv2d(x, m.weight, m.bias, m.stride, m.padding, m.dilation, m.groups)
        elif name in ("LinearBin", "LinearTer"):
            x = F.linear(x, m.weight, m.bias)
        elif name == "_FunctionModule":               # BinaryConnect(): deterministic sign
            x = safe_sign(x)
        else:
            x = m(x)
    return x


def time_callable(fn, budget_s: float = 12.0, warmup: int = 2, min_iters: int = 3, max_iters: int = 50):
    """Median wall time of fn() over a bounded sample (about ``budget_s`` seconds of CPU work)."""
    with torch.no_grad():
        for _ in range(warmup):
            fn()
        times = []
        t_end = time.perf_counter() + budget_s
        while len(times) < max_iters and (len(times) < min_iters or time.perf_counter() < t_end):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], len(times)


def best_thread_count(fn, candidates=None, probe_iters: int = 2):
    """The reference path is torch CPU ops; on a many-core host the default (all logical CPUs) is often
    far from the fastest.  Probe a few thread counts with ``fn`` and return the best (threads, time)."""
    import os
    ncpu = os.cpu_count() or 1
    if candidates is None:
        candidates = sorted({c for c in (8, 16, 32, 64, 128, ncpu // 2, ncpu) if 1 <= c <= ncpu})
    best = None
    with torch.no_grad():
        for c in candidates:
            torch.set_num_threads(c)
            fn()
            t0 = time.perf_counter()
            for _ in range(probe_iters):
                fn()
            dt = (time.perf_counter() - t0) / probe_iters
            if best is None or dt < best[1]:
                best = (c, dt)
    torch.set_num_threads(best[0])
    return best
