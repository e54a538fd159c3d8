"""Whole-network comparison "up to quantiser ties" (VERDICT r3 item 8).

Two executions of one network that differ in fp32 rounding (device vs host BatchNorm arithmetic, accumulation order of a
real-valued first layer) can only differ where a quantiser's input sits within rounding of one of its decision boundaries; such
a flip then moves every later integer sum by a whole step, which is why whole-network logits used to be compared with percent-level
bounds.  Here the second execution is run with the FIRST one's quantiser outputs forced in (forward hooks on the BinaryConnect /
nnDorefaQuant modules): every disagreement is counted and must be a tie — the quantiser's input within ``tol`` (relative to the
tensor's mean magnitude) of a boundary — and with the codes forced the logits have to agree to the float tail (1e-5 normalised)."""
import torch

from pytorch_quantize_impls_amd import lazy
from pytorch_quantize_impls_amd.functions.common import _FunctionModule
from pytorch_quantize_impls_amd.functions.binary_connect import BinaryConnectDeterministic


def _quantisers(model):
    return [m for m in model.modules() if isinstance(m, _FunctionModule)
            and (m.core is BinaryConnectDeterministic or getattr(m.core, "_qt_quant_bits", None) is not None)]


def forward_forcing_codes(first_model, second_model, x_first, x_second, tol=1e-5, max_flip_frac=1e-4):
    """(logits of the first execution, logits of the second with the first's codes forced, {"flips", "elements"}).  Both models
    have the same module structure (e.g. a deepcopy moved to the device); ``first_model`` runs module by module."""
    rec = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: rec.append(out.detach().float().cpu().contiguous())) for m in _quantisers(first_model)]
    try:
        with torch.no_grad(), lazy.eager():
            y1 = first_model(x_first).detach().float().cpu()
    finally:
        for h in hooks:
            h.remove()
    it = iter(rec)
    stats = {"flips": 0, "elements": 0}

    def force(mod, inp, out):
        forced = next(it).to(out.device)
        v = inp[0].detach().float()
        o = out.detach().float()
        assert forced.shape == o.shape
        diff = forced != o
        stats["elements"] += o.numel()
        if bool(diff.any()):
            stats["flips"] += int(diff.sum())
            bits = getattr(mod.core, "_qt_quant_bits", None)
            if bits is None:          # sign: the boundary is 0
                dist = v[diff].abs()
                scale = v.abs().mean()
            else:                     # k-bit DoReFa quantiser rint(n v) / n: boundaries at the half-integers of n v
                nlev = float((1 << int(bits)) - 1)
                u = v[diff] * nlev
                assert bool(((forced[diff] - o[diff]).abs() * nlev - 1.0).abs().max() <= 1e-3), "codes differ by more than one level"
                dist = ((u - torch.floor(u)) - 0.5).abs()
                scale = (v.abs().mean() * nlev).clamp_min(1.0)
            assert float(dist.max()) <= tol * float(scale), (type(mod.core).__name__, float(dist.max()) / float(scale))
        return forced.to(out.dtype)

    hooks = [m.register_forward_hook(force) for m in _quantisers(second_model)]
    try:
        with torch.no_grad(), lazy.eager():
            y2 = second_model(x_second).detach().float().cpu()
    finally:
        for h in hooks:
            h.remove()
    assert next(it, None) is None, "the two executions called a different number of quantisers"
    assert stats["flips"] <= max_flip_frac * max(1, stats["elements"]), stats
    return y1, y2, stats
