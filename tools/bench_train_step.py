#!/usr/bin/env python
"""Training step of BinaryNet-AlexNet (forward + backward, batch 256, channels_last) on one MI355X:
  a) this backend, all matrix-core routes on;  b) backward convs left to MIOpen (_fused.BWD_CONV_MFMA = False);
  c) the reference's op sequence in torch on the same GPU (sign via torch ops, F.conv2d / F.linear fp32, STE by autograd
     Functions restated from functions/binary_connect.py:14-38) — what the un-modified package does here."""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_models  # noqa: E402
from pytorch_quantize_impls_amd.functions import _fused  # noqa: E402


class _RefSign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        r = torch.sign(x)
        r[r == 0] = 1
        return r

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        g = g.clone()
        g[x.abs() > 1.001] = 0
        return g


def ref_forward(model, x):
    def run(seq, h):
        for m in seq:
            name = type(m).__name__
            if name == "BinConv2d":
                h = F.conv2d(h, _RefSign.apply(m.weight), m.bias, m.stride, m.padding)
            elif name == "LinearBin":
                h = F.linear(h, _RefSign.apply(m.weight), m.bias)
            elif name == "_FunctionModule":
                h = _RefSign.apply(h)
            else:
                h = m(h)
        return h
    h = run(model.features, x)
    return run(model.classifieur, h.reshape(h.size(0), 256 * 6 * 6))


def step_time(fwd, model, x, target, n=5):
    def one():
        model.zero_grad(set_to_none=True)
        loss = F.nll_loss(fwd(x), target)
        loss.backward()
        return loss
    for _ in range(2):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loss = one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(loss.detach())


def tie_report(model, x, target):
    """Why the loss of this backend's step and of the reference op sequence differ on real-valued pixels: the
    two conv1 routes round differently in fp32, so behind the training-mode BatchNorm a few activations sit on the other side of
    BinaryConnect's boundary (sign ties), and every flip moves the integer sums downstream by whole steps.  Forward only: the
    reference op sequence records the output of every activation quantiser; this backend's forward then runs with those outputs
    FORCED in (forward hooks on the BinaryConnect modules), counting the elements that differed.  With the reference's codes forced
    the two losses must agree to the float tail; the flip count is the whole explanation of the gap."""
    from pytorch_quantize_impls_amd.functions.common import _FunctionModule
    rec = []
    orig = _RefSign.apply

    def run(seq, h, act):
        for m in seq:
            name = type(m).__name__
            if name == "BinConv2d":
                h = F.conv2d(h, orig(m.weight), m.bias, m.stride, m.padding)
            elif name == "LinearBin":
                h = F.linear(h, orig(m.weight), m.bias)
            elif name == "_FunctionModule":
                h = orig(h)
                act(h)
            else:
                h = m(h)
        return h

    was = {m: m.momentum for m in model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)}
    for m in was:
        m.momentum = 0.0                                    # the report does not move the running statistics
    try:
        with torch.no_grad():
            h = run(model.features, x, lambda t: rec.append(t.detach()))
            loss_ref = float(F.nll_loss(run(model.classifieur, h.reshape(h.size(0), 256 * 6 * 6), lambda t: rec.append(t.detach())), target))
            it = iter(rec)
            stats = {"flips": 0, "elements": 0}

            def force(mod, inp, out):
                forced = next(it)
                o = torch.as_tensor(out).detach()
                stats["elements"] += o.numel()
                stats["flips"] += int((forced.reshape(o.shape) != o).sum())
                return forced.reshape(o.shape).to(o.dtype)
            hooks = [m.register_forward_hook(force) for m in model.modules() if isinstance(m, _FunctionModule)]
            try:
                loss_forced = float(F.nll_loss(model(x), target))
            finally:
                for hk in hooks:
                    hk.remove()
            loss_free = float(F.nll_loss(model(x), target))
    finally:
        for m, mom in was.items():
            m.momentum = mom
    return {"sign_flips_vs_reference_ops": stats["flips"], "activation_elements": stats["elements"],
            "loss_reference_ops": loss_ref, "loss_with_the_reference_codes_forced": loss_forced, "loss_unforced": loss_free,
            "rel_gap_forced": abs(loss_forced - loss_ref) / max(abs(loss_ref), 1e-30),
            "rel_gap_unforced": abs(loss_free - loss_ref) / max(abs(loss_ref), 1e-30)}


def gradient_agreement(B: int = 16, seed: int = 0) -> float:
    """Worst normalised difference between this backend's parameter gradients and the reference op sequence's for one
    AlexNet-Bin training step on +-1 pixels (identical forward passes); zero-gradient biases excluded."""
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    model = bench_models.AlexNetBin()
    # generic gamma / beta: with gamma = 1, beta = 0 and integer conv sums, values sit EXACTLY on the batch mean and any two fp32
    # evaluations of BatchNorm (MIOpen's, this backend's fused chain) land such a tie on either side of the fp64 result
    bench_models.randomize_bn(model, 2)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    x = torch.where(torch.randn(B, 3, 224, 224, device=dev) < 0, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
    target = torch.randint(0, 10, (B,), device=dev)
    grads = []
    for fwd in (model, lambda t: ref_forward(model, t)):
        model.zero_grad(set_to_none=True)
        F.nll_loss(fwd(x), target).backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    ours, ref = grads
    top = max(float(g.abs().max()) for g in ref.values())
    return max(float((ours[k] - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)) for k in ours
               if not (k.endswith(".bias") and ref[k].abs().max() < 1e-3 * top))


def gradient_agreement_fp64(B: int = 4, seed: int = 0):
    """(worst normalised difference of this backend's parameter gradients, of the fp32 reference op sequence's on this device)
    against the FP64 evaluation of the reference op sequence on the CPU, for one AlexNet-Bin training step on +-1 pixels:
    the comparator is neither this backend nor the device's fp32 library."""
    import copy
    dev = torch.device("cuda:0")
    torch.manual_seed(seed)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model, 2)              # generic gamma / beta (see gradient_agreement)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    x = torch.where(torch.randn(B, 3, 224, 224, device=dev) < 0, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
    target = torch.randint(0, 10, (B,), device=dev)
    ref = copy.deepcopy(model).cpu().double().train()
    ref.zero_grad(set_to_none=True)
    F.nll_loss(ref_forward(ref, x.cpu().double()), target.cpu()).backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    worst = []
    for fwd in (model, lambda t: ref_forward(model, t)):
        m2 = copy.deepcopy(model)                       # BatchNorm running statistics must not move between the two runs
        m2.zero_grad(set_to_none=True)
        F.nll_loss((m2 if fwd is model else (lambda t: ref_forward(m2, t)))(x), target).backward()
        got = {k: p.grad.double().cpu() for k, p in m2.named_parameters()}
        top = max(float(g.abs().max()) for g in want.values())
        worst.append(max(float((got[k] - want[k]).abs().max() / (want[k].abs().max() + 1e-300)) for k in got
                         if not (k.endswith(".bias") and want[k].abs().max() < 1e-3 * top)))
    return tuple(worst)


def other_models(name):
    """MODEL=vgg16 | resnet18: the C5 / C4 nets in training mode — this backend vs the same graph with the backward convs left to
    MIOpen, the dense-library paths taken, and the per-step strided-copy count."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if name == "vgg16":
        B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
        model = bench_models.TernaryVGG16(num_classes=1000, image=224, fc=4096)
        x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
        target = torch.randint(0, 1000, (B,), device=dev)
        fwd = lambda t: F.log_softmax(model(t), 1)
    else:
        B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
        model = bench_models.DorefaResNet18(w_bits=int(os.environ.get("W_BITS", "1")), a_bits=4)
        x = torch.randn(B, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
        target = torch.randint(0, 10, (B,), device=dev)
        fwd = lambda t: F.log_softmax(model(t), 1)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    before = dict(_fused.LIBRARY_PATHS)
    ta, la = step_time(fwd, model, x, target)
    paths = {k: v - before.get(k, 0) for k, v in _fused.LIBRARY_PATHS.items() if v != before.get(k, 0)}
    if os.environ.get("ONLY_OURS"):
        print(f"this backend {ta:.2f} ms")
        return
    _fused.BWD_CONV_MFMA = False
    tb, lb = step_time(fwd, model, x, target)
    _fused.BWD_CONV_MFMA = True
    print(f"{name} training step, batch {B}: this backend {ta:.2f} ms ({B / ta * 1e3:.0f} img/s), with MIOpen backward convs {tb:.2f} ms; "
          f"loss {la:.5f} / {lb:.5f}; dense-library paths in 7 steps: {paths}")


def main():
    if os.environ.get("MODEL", "alexnet") != "alexnet":
        return other_models(os.environ["MODEL"])
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    torch.manual_seed(0)
    model = bench_models.AlexNetBin().to(dev).to(memory_format=torch.channels_last).train()
    x = torch.randn(B, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    if os.environ.get("PM1_INPUT"):      # +-1 pixels: every conv sum is an exact integer on both sides, so no sign can flip
        x = torch.where(x < 0, -1.0, 1.0).contiguous(memory_format=torch.channels_last)
    target = torch.randint(0, 10, (B,), device=dev)
    ta, la = step_time(model, model, x, target)
    if os.environ.get("ONLY_OURS"):      # profiling runs: just this backend's steps
        print(f"this backend {ta:.2f} ms")
        return
    grads_a = {k: p.grad.clone() for k, p in model.named_parameters()}
    _fused.BWD_CONV_MFMA = False
    tb, lb = step_time(model, model, x, target)
    _fused.BWD_CONV_MFMA = True
    tc, lc = step_time(lambda t: ref_forward(model, t), model, x, target)
    grads_c = {k: p.grad.clone() for k, p in model.named_parameters()}
    diffs = {k: float((grads_a[k] - grads_c[k]).abs().max() / (grads_c[k].abs().max() + 1e-30)) for k in grads_a}
    if os.environ.get("VERBOSE"):
        for k, v in diffs.items():
            print(f"  {k:28s} diff {v:.2e}   |grad|max ours {float(grads_a[k].abs().max()):.3e}  reference ops {float(grads_c[k].abs().max()):.3e}")
    # a bias in front of a training-mode BatchNorm has a mathematically zero gradient (the batch mean is subtracted): what
    # both sides report for it is rounding noise, so those entries are left out of the comparison
    worst = max(v for k, v in diffs.items() if not (k.endswith(".bias") and grads_c[k].abs().max() < 1e-3 * max(
        float(g.abs().max()) for g in grads_c.values())))
    print(f"AlexNet-Bin training step, batch {B}: this backend {ta:.2f} ms ({B / ta * 1e3:.0f} img/s), with MIOpen backward convs "
          f"{tb:.2f} ms, reference op sequence on the same GPU {tc:.2f} ms ({B / tc * 1e3:.0f} img/s)")
    print(f"loss {la:.6f} / {lb:.6f} / {lc:.6f}; worst normalised gradient difference vs the reference ops {worst:.2e} "
          "(with real-valued pixels conv1's fp32 rounding differs between the routes, a few signs flip behind the training-mode "
          "BatchNorm and the nets diverge; PM1_INPUT=1 feeds +-1 pixels, for which the forward passes are identical)")


if __name__ == "__main__":
    main()
