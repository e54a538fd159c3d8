"""Packed checkpoint of a quantised network: 1 bit (binary, DoReFa k=1) or 2 bits (ternary) per weight.

SURVEY.md section 5 / 8(f) n3: upstream's ``state_dict()`` holds whatever ``weight.data`` holds at
the time — the real weight in training mode, the quantised fp32 image in eval mode — and the real
weight of an eval-mode layer lives in the non-persistent ``weight.org``.  A deployment checkpoint
needs only the quantised image, which is 32x (16x) smaller as bit planes.

Format (a plain dict, ``torch.save``-able):
    {"format": "qt-packed-v1",
     "layers": {module_name: {"kind": "binary" | "ternary" | "dorefa1", "shape": [..],
                              "sign": int32 [rows, ld]  (bit j of word w = element 32w+j of the row,
                                                        1 <=> negative; rows = shape[0], K = prod(shape[1:]),
                                                        ld = words per row rounded up to 4, pad bits zero),
                              "mask": int32 [rows, ld]  (ternary only: 1 <=> non-zero),
                              "scale": fp32 scalar      (dorefa1 only: E = mean|W|),
                              "bias": fp32 or None}},
     "rest": ordinary state_dict entries of everything else (BatchNorm, float layers, ...)}
The plane layout is exactly what ``qt_sign_pack_f32`` / ``qt_ternary_pack_f32`` emit (include/qt_hip.h),
so a device model packs with the HIP kernels; CPU tensors are packed with numpy (a serialisation
helper, not the compute path).

``load_packed_state_dict`` writes the de-quantised image (+-1 / 0 [* E]) into ``weight.data`` — the
same values an eval-mode layer holds — and drops any stale ``weight.org``; the model is left in
eval mode (a packed checkpoint cannot resume float training).
"""
from collections import OrderedDict

import numpy as np
import torch

from ..layers.binary_layers import LinearBin, BinConv2d
from ..layers.terner_layers import LinearTer, TerConv2d
from ..layers.dorefa_layers import LinearDorefa, DorefaConv2d

FORMAT = "qt-packed-v1"


def _kind(module):
    if isinstance(module, (LinearBin, BinConv2d)):
        return "binary"
    if isinstance(module, (LinearTer, TerConv2d)):
        return "ternary"
    if isinstance(module, (LinearDorefa, DorefaConv2d)) and module.bit_width == 1:
        return "dorefa1"
    return None


def _ld(K):
    return max(4, ((K + 31) // 32 + 3) // 4 * 4)


def _pack_bits_cpu(bits):
    """bool [rows, K] -> int32 [rows, ld], little-endian bit order inside each word."""
    rows, K = bits.shape
    ld = _ld(K)
    padded = np.zeros((rows, ld * 32), dtype=np.uint8)
    padded[:, :K] = bits
    words = np.packbits(padded, axis=1, bitorder="little").view("<u4")
    return torch.from_numpy(words.astype(np.uint32).view(np.int32).reshape(rows, ld).copy())


def _unpack_bits(plane, K):
    """int32 [rows, ld] (any device) -> bool [rows, K]"""
    shifts = torch.arange(32, device=plane.device, dtype=torch.int32)
    bits = (plane.unsqueeze(-1) >> shifts) & 1
    return bits.reshape(plane.shape[0], -1)[:, :K].bool()


def _quantized_image(module, kind):
    """The fp32 tensor an eval-mode layer holds in weight.data (without touching the module)."""
    w = module.weight.detach()
    if module.training:
        with torch.no_grad():
            return module._quantized_weight_for_eval().detach()
    return w


def _pack_layer(module, kind):
    q = _quantized_image(module, kind)
    rows = int(q.shape[0])
    q2 = q.reshape(rows, -1).contiguous()
    entry = {"kind": kind, "shape": list(q.shape)}
    if kind == "dorefa1":
        entry["scale"] = q2.abs().amax().to(torch.float32).cpu()   # |sign(W)*E| == E everywhere
    if q2.is_cuda:
        from .. import ops
        if kind == "ternary":
            planes = ops.ternary_pack(q2)      # q is already in {-1,0,+1}: ternarising again is the identity
            entry["mask"] = planes.mask.cpu()
        else:
            planes, _ = ops.sign_pack(q2)
        entry["sign"] = planes.sign.cpu()
    else:
        a = q2.numpy()
        entry["sign"] = _pack_bits_cpu(a < 0)
        if kind == "ternary":
            entry["mask"] = _pack_bits_cpu(a != 0)
    entry["bias"] = None if module.bias is None else module.bias.detach().cpu().clone()
    return entry


def packed_state_dict(model):
    layers, skip = OrderedDict(), set()
    for name, module in model.named_modules():
        kind = _kind(module)
        if kind is None:
            continue
        layers[name] = _pack_layer(module, kind)
        prefix = name + "." if name else ""
        skip.update({prefix + "weight", prefix + "bias"})
    rest = OrderedDict((k, v.detach().cpu().clone()) for k, v in model.state_dict().items() if k not in skip)
    return {"format": FORMAT, "layers": layers, "rest": rest}


def packed_state_nbytes(state):
    n = sum(v.numel() * v.element_size() for v in state["rest"].values())
    for e in state["layers"].values():
        for key in ("sign", "mask", "scale", "bias"):
            t = e.get(key)
            if t is not None:
                n += t.numel() * t.element_size()
    return n


def load_packed_state_dict(model, state):
    if state.get("format") != FORMAT:
        raise ValueError(f"not a {FORMAT} checkpoint: format = {state.get('format')!r}")
    modules = dict(model.named_modules())
    missing = [n for n in state["layers"] if n not in modules]
    if missing:
        raise KeyError(f"packed checkpoint has layers the model lacks: {missing}")
    model.eval()     # eval-swap first, so the protocol's own copy cannot overwrite what we load
    for name, e in state["layers"].items():
        module = modules[name]
        kind = _kind(module)
        if kind != e["kind"]:
            raise TypeError(f"{name}: checkpoint holds a {e['kind']} layer, the model a {kind or type(module).__name__}")
        if list(module.weight.shape) != list(e["shape"]):
            raise ValueError(f"{name}: weight shape {list(module.weight.shape)} != checkpoint {e['shape']}")
        dev = module.weight.device
        rows = e["shape"][0]
        K = int(np.prod(e["shape"][1:])) if len(e["shape"]) > 1 else 1
        neg = _unpack_bits(e["sign"].to(dev), K)
        q = torch.where(neg, -1.0, 1.0).to(torch.float32)
        if kind == "ternary":
            q = q * _unpack_bits(e["mask"].to(dev), K).to(torch.float32)
        if kind == "dorefa1":
            q = q * e["scale"].to(dev)
        with torch.no_grad():
            module.weight.data.copy_(q.reshape(e["shape"]))
            if hasattr(module.weight, "org"):
                module.weight.org.data.copy_(module.weight.data)   # no stale float weight to resurrect
            if module.bias is not None and e["bias"] is not None:
                module.bias.data.copy_(e["bias"].to(dev))
        module._qt_eval_planes = None
    res = model.load_state_dict(state["rest"], strict=False)
    unexpected = list(res.unexpected_keys)
    if unexpected:
        raise KeyError(f"unexpected entries in the checkpoint: {unexpected}")
    return model
