#!/usr/bin/env python
"""Which fp32 expression does torch's eval-mode F.batch_norm evaluate on this device, per layout?  (the fused
threshold must reproduce the eager predicate bit for bit.)  Compares against candidate formulas evaluated in numpy."""
import json
import numpy as np
import torch

torch.manual_seed(0)
dev = torch.device("cuda:0")
C = 96
res = {}
for name, shape, cl in (("4d_nchw", (8, C, 13, 13), False), ("4d_nhwc", (8, C, 13, 13), True), ("2d", (512, C), False),
                        ("4d_nchw_small", (1, C, 2, 2), False), ("4d_nhwc_small", (1, C, 2, 2), True)):
    x = torch.randn(shape) * 5
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    bn = (torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d)(C, eps=1e-5).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2); bn.weight.data.normal_(); bn.bias.data.normal_()
    y_cpu = bn(x).detach()
    y = bn.to(dev)(x.to(dev)).detach().cpu()
    perm = (0, 2, 3, 1) if len(shape) == 4 else (0, 1)
    yy, xx = y.permute(*perm).numpy(), x.permute(*perm).numpy()
    m, v, w, b = (t.detach().cpu().numpy() for t in (bn.running_mean, bn.running_var, bn.weight, bn.bias))
    f32, f64 = np.float32, np.float64
    inv = (f32(1) / np.sqrt(v + f32(1e-5))).astype(f32)
    inv_r = (f32(1) / np.sqrt((v + f32(1e-5)).astype(f64))).astype(f32)       # correctly rounded rsqrt
    alpha = (inv * w).astype(f32); beta = (b - m * alpha).astype(f32)
    cands = {}
    cands["fl(fl(x*alpha)+beta)"] = ((xx * alpha).astype(f32) + beta).astype(f32)
    cands["fma(x,alpha,beta)"] = (xx.astype(f64) * alpha.astype(f64) + beta.astype(f64)).astype(f32)
    for iname, iv in (("inv", inv), ("rsqrt", inv_r)):
        d = (xx - m).astype(f32)
        cands[f"((x-m)*{iname})*w+b"] = (((d * iv).astype(f32) * w).astype(f32) + b).astype(f32)
        cands[f"fma((x-m)*{iname},w,b)"] = ((d * iv).astype(f32).astype(f64) * w.astype(f64) + b.astype(f64)).astype(f32)
        cands[f"(x-m)*({iname}*w)+b"] = ((d * (iv * w).astype(f32)).astype(f32) + b).astype(f32)
        cands[f"fma(x-m,{iname}*w,b)"] = (d.astype(f64) * (iv * w).astype(f32).astype(f64) + b.astype(f64)).astype(f32)
        cands[f"w*(x-m)*{iname}+b"] = (((w * d).astype(f32) * iv).astype(f32) + b).astype(f32)
        cands[f"fma(w*(x-m),{iname},b)"] = ((w * d).astype(f32).astype(f64) * iv.astype(f64) + b.astype(f64)).astype(f32)
    r = {k: float((c == yy).mean()) for k, c in cands.items()}
    r["== cpu"] = float((y == y_cpu).float().mean())
    res[name] = dict(sorted(r.items(), key=lambda kv: -kv[1])[:5])
    print(name, res[name], flush=True)
json.dump(res, open("gpurun_out/bn_eval_arith.json", "w"), indent=1)
