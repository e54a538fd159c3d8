#!/usr/bin/env python
"""Per-image time of the threshold convs as a function of the images per launch (is a batch of 256 better run as a few launches?)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pytorch_quantize_impls_amd import ops, synth

dev = torch.device("cuda:0")
t32 = lambda a: torch.from_numpy(a).to(dev)

def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

def case(name, Cin, Cout, k, N, H, out_halo=(1, 1)):
    pd = k // 2
    x = t32(synth.pm1(1, (N, Cin, H, H))).contiguous(memory_format=torch.channels_last)
    bits = ops.sign_pack(x.permute(0, 2, 3, 1).contiguous())[0]
    px = ops.bits_to_nib_pad(bits, N, H, H, (pd, pd), ld=ops.pixel_ld_nib(Cin))
    wp = ops.pack_conv_weight_nib(t32(synth.uniform(2, (Cout, Cin, k, k), -1, 1)), "binary")
    alpha, beta = t32(synth.uniform(4, (Cout,), -1, 1)), t32(synth.uniform(5, (Cout,), -8, 8))
    thr = ops.integer_thresholds(None, alpha, beta, Cin * k * k)
    Hp = H + 2 * pd
    per_img = px.words.shape[0] // N
    def run(n0, n1):
        sub = ops.NibPlanes(px.words[n0 * per_img:n1 * per_img], (n1 - n0) * per_img, px.K)
        return ops.conv2d_nib(sub, (n1 - n0, Cin, Hp, Hp), wp, (k, k), None, 1, 0, 1, epi=ops.NibEpilogue(alpha, beta, out_halo, thr=thr))
    out = [name]
    for chunk in (256, 128, 86, 64, 43, 32):
        parts = [(i, min(N, i + chunk)) for i in range(0, N, chunk)]
        t = timed(lambda: [run(a, b) for a, b in parts])
        out.append(f"{len(parts)} x {chunk}: {t:.1f}")
    print("  ".join(out))

case("alexnet conv2 192->576 k5 @27", 192, 576, 5, 256, 27)
case("alexnet conv3 576->1152 @13", 576, 1152, 3, 256, 13)
case("alexnet conv4 1152->768 @13", 1152, 768, 3, 256, 13)
case("alexnet conv5 768->256 @13", 768, 256, 3, 256, 13)
case("vgg conv3_2 256->256 @56", 256, 256, 3, 256, 56)
case("vgg conv4_1 256->512 @28", 256, 512, 3, 256, 28)
case("vgg conv4_2 512->512 @28", 512, 512, 3, 256, 28)
case("vgg conv5_1 512->512 @14", 512, 512, 3, 256, 14)
