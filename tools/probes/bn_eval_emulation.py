#!/usr/bin/env python
"""Can eval-mode F.batch_norm of this device be reproduced bit for bit as fma(fl(fl(x - mean) * rs), weight, bias) with rs read
back from the device's own kernel (F.batch_norm(ones, 0, var, 1, 0) = rs)?  Decides whether the DoReFa code epilogue can take
the device's BatchNorm arithmetic (lazy.DEFER_CODES by default again)."""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)
dev = torch.device("cuda:0")
for name, shape, cl in (("4d_nhwc", (8, 96, 13, 13), True), ("4d_nchw", (8, 96, 13, 13), False), ("4d_nhwc_big", (64, 64, 32, 32), True)):
    C = shape[1]
    x = (torch.randn(shape, device=dev) * 5)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    m = torch.randn(C, device=dev); v = torch.rand(C, device=dev) * 1.5 + 0.5
    w = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    eps = 1e-5
    y = F.batch_norm(x, m, v, w, b, False, 0.0, eps)
    ones = torch.ones((1, C, 2, 2), device=dev)
    if cl:
        ones = ones.contiguous(memory_format=torch.channels_last)
    rs = F.batch_norm(ones, torch.zeros(C, device=dev), v, torch.ones(C, device=dev), torch.zeros(C, device=dev), False, 0.0, eps)[0, :, 0, 0]
    rs_t = torch.rsqrt(v + eps)
    rs_d = 1.0 / torch.sqrt(v + eps)
    xx = x.permute(0, 2, 3, 1).double().cpu().numpy(); yy = y.permute(0, 2, 3, 1).cpu().numpy()
    f32 = np.float32
    d = (x.permute(0, 2, 3, 1) - m).cpu().numpy().astype(f32)
    res = {}
    for nm, r in (("rs from the kernel", rs), ("torch.rsqrt", rs_t), ("1/torch.sqrt", rs_d)):
        r = r.cpu().numpy().astype(f32)
        t = (d * r).astype(f32)
        fma = (t.astype(np.float64) * w.cpu().numpy().astype(np.float64) + b.cpu().numpy().astype(np.float64)).astype(f32)
        two = ((t * w.cpu().numpy()).astype(f32) + b.cpu().numpy()).astype(f32)
        res[nm] = (float((fma == yy).mean()), float((two == yy).mean()))
    print(name, {k: tuple(round(z, 5) for z in v_) for k, v_ in res.items()}, "(fma match, mul+add match)")
