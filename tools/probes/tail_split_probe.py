#!/usr/bin/env python
"""Does splitting the batch so that the big-tile launch fills whole rounds of the 256 CUs pay?  Times conv(N) against conv(k) + conv(N - k)
for the layer shapes whose tile count leaves a mostly empty last round (profiles/r6_threshold_epilogue.md)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from pytorch_quantize_impls_amd import ops, synth

dev = torch.device("cuda:0")

def t32(a):
    return torch.from_numpy(a).to(dev)

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

def case(name, Cin, Cout, k, N, H, tn, tm=256):
    pd = k // 2
    M = N * H * H
    x = t32(synth.pm1(1, (N, Cin, H, H))).contiguous(memory_format=torch.channels_last)
    bits = ops.sign_pack(x.permute(0, 2, 3, 1).contiguous())[0]
    px = ops.bits_to_nib_pad(bits, N, H, H, (pd, pd), ld=ops.pixel_ld_nib(Cin))
    wp = ops.pack_conv_weight_nib(t32(synth.uniform(2, (Cout, Cin, k, k), -1, 1)), "binary")
    alpha, beta = t32(synth.uniform(4, (Cout,), -1, 1)), t32(synth.uniform(5, (Cout,), -8, 8))
    thr = ops.integer_thresholds(None, alpha, beta, Cin * k * k)
    Hp = H + 2 * pd
    per_img = px.words.shape[0] // N
    def run(n0, n1):
        sub = ops.NibPlanes(px.words[n0 * per_img:n1 * per_img], (n1 - n0) * per_img, px.K) if (n0, n1) != (0, N) else px
        return ops.conv2d_nib(sub, (n1 - n0, Cin, Hp, Hp), wp, (k, k), None, 1, 0, 1, epi=ops.NibEpilogue(alpha, beta, (1, 1), thr=thr))
    cols = (Cout + tn - 1) // tn
    tiles = ((M + tm - 1) // tm) * cols
    full = tiles // 256 * 256
    # largest image count whose big tiles fit the whole rounds
    kimg = N
    while kimg > 0 and ((kimg * H * H + tm - 1) // tm) * cols > full:
        kimg -= 1
    t_all = timed(lambda: run(0, N))
    t_a = timed(lambda: run(0, kimg)) if 0 < kimg < N else float("nan")
    t_b = timed(lambda: run(kimg, N)) if 0 < kimg < N else float("nan")
    print(f"{name}: tiles {tiles} = {tiles / 256:.2f} rounds; all {t_all:.1f} us; first {kimg} images {t_a:.1f} + last {N - kimg} {t_b:.1f} = {t_a + t_b:.1f} us")

case("vgg conv4_2 512->512 @28", 512, 512, 3, 256, 28, 256)
case("vgg conv4_1 256->512 @28", 256, 512, 3, 256, 28, 256)
case("vgg conv3_2 256->256 @56", 256, 256, 3, 256, 56, 256)
case("alexnet conv3 576->1152 @13", 576, 1152, 3, 256, 13, 192)
case("alexnet conv2 192->576 k5 @27", 192, 576, 5, 256, 27, 192, 384)
