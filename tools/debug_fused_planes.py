"""Bring-up aid: after one qt_linear_fused_f32 launch compare the nibble workspace with the expected planes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_quantize_impls_amd import ops

def expected_plane(t):
    R, K = t.shape
    nch = K // 512
    neg = (t < 0).view(R, nch, 2, 64, 4)            # [row][chunk][half a/b][lane][elem]
    # element e = half*4 + i  -> nibble 7 - e of dword `lane`
    nib = torch.where(neg, 0xA, 0x2).to(torch.int64)
    word = torch.zeros((R, nch, 64), dtype=torch.int64, device=t.device)
    for half in range(2):
        for i in range(4):
            e = half * 4 + i
            word |= nib[:, :, half, :, i] << (4 * (7 - e))
    return word.to(torch.int32).view(R, nch * 64) if False else word.view(R, nch * 64)

dev = torch.device("cuda:0")
M = N = K = 4096
g = torch.Generator(device=dev); g.manual_seed(7)
for it in range(3):
    x = torch.where(torch.rand((M, K), device=dev, generator=g) < 0.5, -1.0, 1.0)
    w = torch.randn((N, K), device=dev, generator=g) / 64
    y = ops.linear_fused(x, w, None, "binary")
    torch.cuda.synchronize()
    ws = ops.linear_fused_workspace(dev, M, N, K)
    planes = ws[36864:].view(torch.int32).view(2, M, K // 8).to(torch.int64) & 0xFFFFFFFF
    for name, t, p in (("X", x, planes[0]), ("W", w, planes[1])):
        exp = expected_plane(t)
        bad = (p != exp)
        print(f"launch {it} plane {name}: wrong dwords {int(bad.sum())}")
        if bad.any():
            idx = bad.nonzero()
            rows = idx[:, 0].unique()
            print("  rows:", rows[:24].tolist(), "n rows", rows.numel())
            r0 = int(rows[0])
            cols = idx[idx[:, 0] == r0][:, 1]
            print(f"  row {r0}: bad dwords {cols.tolist()[:40]} (chunk = dword // 64)")
            c0 = int(cols[0])
            print(f"   got {int(p[r0, c0]):08x} exp {int(exp[r0, c0]):08x}")
    ref = (torch.where(x < 0, -1.0, 1.0) @ torch.where(w < 0, -1.0, 1.0).t())
    print("  y == dense ref:", bool(torch.equal(y, ref)), "mismatches", int((y != ref).sum()))
