#!/usr/bin/env python
"""Cycle / wall-clock stamps of the 384x192 ping-pong conv kernel (qt_conv2d_implicit_variant 3: a -DQT_PROFILING_VARIANTS build of the library, QT_HIP_LIB=...) on an
AlexNet conv2-shaped problem: x [256, 192, 27, 27] +-1, W [576, 192, 5, 5], padding 2."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from pytorch_quantize_impls_amd import _lib, ops
dev = torch.device("cuda:0")
CONV1 = len(sys.argv) > 1 and sys.argv[1] == "conv1"     # AlexNet conv1: real input, bf16 triples, space-to-depth form
N, C, H, Cout, k, pad = 256, 192, 27, 576, 5, 2
x = torch.randn((N, C, H, H), device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn((Cout, C, k, k), device=dev)
px = ops.pack_pixels_nib(x)
wp = ops.pack_conv_weight_nib(w, "binary")
def run():
    return ops.conv2d_nib(px, (N, C, H, H), wp, (k, k), None, 1, pad, 1)
if CONV1:
    x1 = torch.randn((N, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
    w1 = torch.randn((192, 3, 11, 11), device=dev).sign()
    px1, (Hs, Ws) = ops.s2d_triple_pack(x1, 4, 2)
    ws1 = ops.s2d_weight(w1, 4)
    wt1 = ops.pack_conv_weight_bf16x3(ws1, "sign")
    meta = torch.empty(tuple(ws1.shape), device="meta")
    run = lambda: ops.float_conv2d(None, meta, "sign", None, 1, 0, 1, weight_triples=wt1, pixels=px1, in_shape=(N, 48, Hs, Ws))
    OUT_GRID = (192, 55)       # Cout, Ho for the decode below
    STAGES = (9 * 288 + 63) // 64
for which in (1, 0):
    ops.CONV_VARIANT = which
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"force {which}: median {sorted(ts)[5]:.1f} us")
xpad = torch.nn.functional.pad(x, (pad, pad, pad, pad)).contiguous(memory_format=torch.channels_last)   # zeros are not +-1:
px2 = ops.pack_pixels_nib(torch.where(xpad == 0, torch.ones_like(xpad), xpad))                           # timing only
for which in (1, 0):
    ops.CONV_VARIANT = which
    f = lambda: ops.conv2d_nib(px2, (N, C, H + 2 * pad, H + 2 * pad), wp, (k, k), None, 1, 0, 1)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"physically padded plane, conv padding 0, force {which}: median {sorted(ts)[5]:.1f} us")
STAGES = globals().get("STAGES", 192 * 25 // 2 // 64)
ops.CONV_VARIANT = 3
PADDED = len(sys.argv) > 1 and sys.argv[1] == "padded"
if PADDED:
    run = lambda: ops.conv2d_nib(px2, (N, C, H + 2 * pad, H + 2 * pad), wp, (k, k), None, 1, 0, 1)
    print("stamps below: un-padded (VALID) kernel on the physically padded plane")
for _ in range(2):
    y = run()
torch.cuda.synchronize()
ops.CONV_VARIANT = 0
if CONV1:
    Cout, H = OUT_GRID
M = N * H * H
yi = y.view(torch.int32).cpu().numpy()
rows = []
for tm in range((M + 383) // 384):
    for tn in range(Cout // 192):
        for half in range(4):           # wave_m = 0..3 (WM = 4), wave_n == 0 writes
            r0 = tm * 384 + half * 96
            if r0 < M:
                rows.append(np.ascontiguousarray(yi[r0, tn * 192: tn * 192 + 32]).view(np.int64))
t = np.array(rows)
names = ["top", "frags read", "dma issued", "waitcnt done", "barrier1", "mfma issued", "barrier2"]
print("per-segment cycle stamps (last steady-state stage) of the first tile's waves 0/2/4/6:")
base = t[:4, 0].min()
for i in range(4):
    r = t[i]
    print(f"  simd {int(r[14])} wave_m {i}: " + "  ".join(f"{nm}={int(r[j] - base)}" for j, nm in enumerate(names)))
wall = t[:, 9:14].astype(np.float64) / 100.0
w0 = wall[:, 0].min()
lab = ["entry", "prologue done", "loop end", "stores issued", "stores done"]
d = wall[:, 1:] - wall[:, :-1]
for i in range(4):
    print(f"  {lab[i]} -> {lab[i+1]}: median {np.median(d[:, i]):6.2f} us  max {d[:, i].max():6.2f}")
loop_us = wall[:, 2] - wall[:, 1]
print(f"main loop: median {np.median(t[:, 15]):.0f} cycles in {np.median(loop_us):.2f} us -> {np.median(t[:, 15] / loop_us) / 1e3:.3f} GHz; "
      f"stages = {STAGES} -> {np.median(t[:, 15]) / STAGES:.0f} cycles per 64-byte stage (18 MFMAs per wave = 576 cycles of pipe per slot, 2 slots)")
print(f"kernel span: {wall[:, 4].max() - w0:.1f} us; workgroups {len(t) // 4}")
