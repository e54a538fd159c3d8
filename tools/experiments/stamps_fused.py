"""Phase stamps of the fused linear kernel.  Needs a library built with the stamps variant (the product build has none):

    python tools/experiments/lf.py -DQT_LF_STAMPS && python tools/experiments/stamps_fused.py
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lf  # noqa: E402  (also puts the repo root on sys.path)
from pytorch_quantize_impls_amd import ops
dev = torch.device("cuda:0")
M = N = K = 4096
x = torch.where(torch.rand((M, K), device=dev) < 0.5, -1.0, 1.0)
w = torch.randn((N, K), device=dev) / 64
for _ in range(5):
    y = lf.linear_fused(x, w, None, "binary")
torch.cuda.synchronize()
yb = y.cpu().numpy()
names = ["start", "fill issued", "fill drained", "chunk0 ready", "loop start", "phase1 end", "loop end", "epi issued", "epi drained"]
rows = []
for tm in range(16):
    for tn in range(16):
        for wm in range(2):
            v = yb[tm * 256 + wm * 128, tn * 256: tn * 256 + 20].view(np.uint64)
            rows.append(v[:10].astype(np.float64))
a = np.array(rows)
t0 = a[:, 0].min()
us = (a[:, :9] - t0) / 100.0     # wall_clock64: 100 MHz
print("phase            min     median   max   (us since the first workgroup's start)")
for i, n in enumerate(names):
    print(f"{n:14s} {us[:, i].min():7.2f} {np.median(us[:, i]):7.2f} {us[:, i].max():7.2f}")
print("spin iterations per wave: median", np.median(a[:, 9]), "max", a[:, 9].max(), "total", a[:, 9].sum())
