#!/usr/bin/env python
"""Randomised parity soak of the weight-gradient routes (pixel-major kernel incl. the strided first-layer form, K-major GEMMs) against
torch.nn.grad.conv2d_weight in float64: random shapes, paddings, memory formats, ternary / DoReFa activations, STE masks, K-slice counts.
python tools/soak_wgrad.py [seed] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pytorch_quantize_impls_amd import ops

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
dev = torch.device("cuda:0")
bad, t0, worst = 0, time.time(), 0.0
ran = {}
for it in range(iters):
    kind = rng.choice(["pm", "pm", "pm", "s2d", "gemm"])
    N = int(rng.integers(1, 10))
    if kind == "s2d":
        s = int(rng.choice([1, 2, 3, 4]))
        k2 = int(rng.choice([3, 5]))
        k = int(rng.integers((k2 - 1) * s + 1, k2 * s + 1)) if s > 1 else k2
        C = int(rng.integers(1, 4)) if s > 1 else int(rng.integers(1, 9))
        if 3 * C * s * s > 256 or k < s:
            continue
        H, W = int(rng.integers(k, k + 40)), int(rng.integers(k, k + 40))
        p = int(rng.integers(0, k // 2 + 1))
        Cout = int(rng.integers(32, 200))
        x = torch.randn(N, C, H, W, device=dev) * float(rng.choice([0.01, 1.0, 30.0]))
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        if Ho <= 0 or Wo <= 0:
            continue
        g = torch.randn(N, Cout, Ho, Wo, device=dev)
        if rng.random() < 0.5:
            x, g = x.contiguous(memory_format=torch.channels_last), g.contiguous(memory_format=torch.channels_last)
        got = ops.conv2d_grad_weight_s2d(x, g, (Cout, C, k, k), s, p)
        if got is None:
            continue
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, C, k, k), g.double(), stride=s, padding=p)
    else:
        k = int(rng.choice([3, 5])) if kind == "pm" else int(rng.choice([1, 2, 3, 4, 7]))
        p = int(rng.integers(0, k))
        Cin, Cout = int(rng.integers(32, 260)), int(rng.integers(32, 300))
        H, W = int(rng.integers(max(1, k - 2 * p), 34)), int(rng.integers(max(1, k - 2 * p), 34))
        Ho, Wo = H + 2 * p - k + 1, W + 2 * p - k + 1
        if Ho <= 0 or Wo <= 0:
            continue
        levels = float(rng.choice([1.0, 1.0, 3.0, 15.0, 127.0]))
        if levels == 1.0:
            x = torch.randint(-1, 2, (N, Cin, H, W), device=dev).float()
        else:
            x = torch.randint(0, int(levels) + 1, (N, Cin, H, W), device=dev).float() / levels
        g = torch.randn(N, Cout, Ho, Wo, device=dev) * torch.exp(torch.randn(1, Cout, 1, 1, device=dev) * 2)
        if rng.random() < 0.6:
            x, g = x.contiguous(memory_format=torch.channels_last), g.contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, k, k, device=dev) if rng.random() < 0.5 else None
        if kind == "pm":
            got = ops.conv2d_grad_weight_pm(x, g, (k, k), p, weight=w, x_levels=levels, workgroups=int(rng.choice([0, 0, 1, 7, 100, 5000])))
        else:
            got = ops.conv2d_grad_weight_gemm(x, g, (k, k), p, weight=w, x_levels=levels)
        if got is None:
            continue
        ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, k, k), g.double(), padding=p)
        if w is not None:
            ref = torch.where(w.abs() <= 1.001, ref, torch.zeros_like(ref))
    ran[str(kind)] = ran.get(str(kind), 0) + 1
    scale = ref.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)
    err = float(((got.double() - ref).abs() / scale).max())
    worst = max(worst, err)
    if not (err <= 1e-5) or not bool(torch.isfinite(got).all()):
        bad += 1
        print(f"MISMATCH it {it} kind {kind} shape x {tuple(x.shape)} g {tuple(g.shape)} k {k} p {p}: err {err:.3e}", flush=True)
print(f"seed {seed}: {iters} iterations ({ran} checked), {bad} mismatches, worst per-channel normalised error {worst:.2e}, {time.time() - t0:.0f} s")
