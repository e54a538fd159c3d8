#!/usr/bin/env python
"""Per-phase shader-clock cycles of the direct first-layer conv kernel (profiling build: csrc built with -DQT_PROFILING_VARIANTS,
QT_HIP_LIB=pytorch_quantize_impls_amd/lib/libqt_hip_prof.so).  Wave 0 of every workgroup accumulates the cycles between phase
boundaries; printed per tile (AlexNet conv1, batch 256)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
from pytorch_quantize_impls_amd import _lib, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
B = int(os.environ.get("B", "256"))
x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(192, 3, 11, 11, device=dev).sign()
fw = ops.pack_first_layer_weight(w, 4)
fwr = ops.pack_first_layer_weight(w * 0.037, 4, real=True)
al, be = torch.ones(192, device=dev), torch.zeros(192, device=dev)
names = ["issue_patch", "mfma prologue (zero acc, first frags)", "mfma loop", "epilogue", "finish: max + to barrier 1", "finish: barrier 1 wait",
         "finish: convert + LDS writes", "finish: barrier 2 wait"]
lib = _lib.load()
h = (ctypes.c_ulonglong * 40)()
def fetch(reset):
    fn = lib.qt_first_stamps_fetch
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert fn(h, reset) == 0
    return np.array(list(h), dtype=np.float64)
for label, f in (("+-1 weights, bits", lambda: ops.conv_first_direct(x, fw, None, 4, 2, epi=(al, be))),
                 ("+-1 weights, fp32", lambda: ops.conv_first_direct(x, fw, None, 4, 2)),
                 ("real weights, bits", lambda: ops.conv_first_direct(x, fwr, None, 4, 2, epi=(al, be)))):
    for _ in range(3):
        f()
    torch.cuda.synchronize(); fetch(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    s = fetch(1)
    wgs = s[32]
    tiles = B * 25
    print(f"{label}: {e0.elapsed_time(e1) * 1e3:.0f} us, {int(wgs)} workgroups, {tiles / wgs:.1f} tiles each; cycles per tile, waves 0-3:")
    w4 = s[:32].reshape(4, 8)
    for i, n in enumerate(names):
        print(f"    {n:40s} " + " ".join(f"{w4[k, i] / tiles:8.0f}" for k in range(4)))
    tot = w4.sum(1)
    print(f"    {'total':40s} " + " ".join(f"{t / tiles:8.0f}" for t in tot) + f"   -> {tot[0] / wgs / (e0.elapsed_time(e1) * 1e3) / 1e3:.2f} GHz")
