"""Deferred activations: the reference's module-by-module inference graph, executed as the fused chain.

The reference builds its binarised CNNs from separate modules (models/Alexnet/Alexnet_Bin.py:12-33,
models/FullNet/terMNIST.py:48-59):

    BinConv2d / TerConv2d -> [MaxPool2d] -> BatchNorm2d -> Hardtanh -> BinaryConnect -> [MaxPool2d] -> next layer

and every arrow is an fp32 tensor.  On the device four of those five modules are not this package's (torch's pooling /
Hardtanh kernels, MIOpen's BatchNorm) and they are 40-60 % of the forward's kernel time.  ``layers.fused`` removes them,
but only for callers that re-build their model with ``fuse_sequential`` / ``FusedFeatureClassifier``.

This module gets the same execution WITHOUT touching the model: in eval mode under ``torch.no_grad()`` /
``inference_mode()`` a binarised conv on a HIP device does not run; it returns a ``LazyActivation`` — a ``torch.Tensor``
wrapper subclass with the right shape / dtype / device and no storage — that remembers (layer, input).  The
``__torch_function__`` protocol then sees ``F.max_pool2d``, ``F.batch_norm`` (eval), ``F.hardtanh``, ``reshape`` /
``flatten`` and BinaryConnect as they are applied by the un-modified ``nn`` modules and only RECORDS them.  Execution
is pulled by the consumer:

  * the next BinConv2d / TerConv2d / LinearBin / LinearTer asks for the activation as a ``PackedActivation`` and the
    recorded chain runs as ``layers.fused.FusedConvPoolBnSign`` — conv with the BatchNorm-threshold epilogue, MaxPool on
    bits, output written directly as the consumer's operand (its padding as a zero border) — i.e. exactly the kernels
    of the opt-in fused form, bit-identical to it;
  * ANY other use (an op that is not in the grammar above, printing, ``.cpu()``, arithmetic, a hook, autograd) makes
    the tensor materialise: the conv runs through the layer's ordinary eval path and the recorded ops are replayed with
    the torch functions that were intercepted, so the caller gets what the module-by-module graph would have produced.

Exactness: the fused sign chain takes its per-channel thresholds from this device's own ``F.batch_norm`` (bisection over
the fp32 bit patterns, ``layers.fused.device_sign_fold``), so the bits — and everything computed from them — equal the
module-by-module graph's exactly (tests assert ``torch.equal`` against ``lazy.eager()``).  The DoReFa code chain cannot be
reduced to thresholds (a residual is added before the quantiser): its epilogue evaluates BatchNorm in this device's own
arithmetic instead, verified on a probe (``layers.fused.device_bn_fold``).

Nothing is deferred in training mode, with autograd enabled, on CPU tensors, for non-fp32 dtypes, grouped convs or
non-zero padding modes.  ``lazy.ENABLED = False`` (or the ``eager()`` context manager) switches the mechanism off;
``lazy.STATS`` counts what happened (tests assert on it).
"""
from __future__ import annotations

import collections
import contextlib
import threading
import weakref
from typing import Optional

import torch
import torch.nn.functional as F
from torch.utils._pytree import tree_map_only

from . import lazy_train, ops, packed

#: master switch (module-by-module execution when False)
ENABLED = True
#: DoReFa chains (DorefaConv2d -> BatchNorm -> [+ shortcut] -> ReLU -> nnDorefaQuant) need BatchNorm's VALUE, not just its
#: sign.  Their fused blocks therefore evaluate BatchNorm exactly as this device's eval-mode F.batch_norm does —
#: fma(fl(fl(x - mean) * rs), weight, bias) with rs read back from the library's own kernel — and verify that emulation against
#: F.batch_norm on a probe when a block is built (layers.fused.device_bn_fold; a mismatch sends the chain to its eager path).
#: The codes then equal the module-by-module graph's bit for bit (tests/test_gpu_lazy.py: torch.equal).  False: DoReFa
#: convs are never deferred.
DEFER_CODES = True
#: the [B, N] result of an eval-mode LinearBin / LinearTer / LinearXNOR is handed out as a deferred activation as well — already
#: COMPUTED (the GEMM ran), but BatchNorm1d -> [Hardtanh] -> BinaryConnect recorded on it run as ONE pass over it (FusedPoolBnSign
#: with this device's own BatchNorm thresholds: the bits of the module chain) instead of three launches; any other use reads the
#: value.  The classifier of an un-modified AlexNet then executes like the explicit fused form.
DEFER_DENSE = True
#: "deferred" convs that returned a LazyActivation, "fused" chains executed as fused blocks, "materialised" lazies that
#: had to produce their fp32 value, "fallback:<func>" the functions that forced it
STATS = collections.Counter()


_tls = threading.local()


def enabled() -> bool:
    """Deferral is on: the master switch ``ENABLED`` and no ``eager()`` block open in this thread."""
    return ENABLED and not getattr(_tls, "eager_depth", 0)


_implicit = None


def note_inference_call(layer, input) -> None:
    """Tell utils/implicit.py that an eval-mode, no-autograd forward of a quantised layer on a device runs (the trigger of the
    implicit hipGraphs: after the second such call the ROOT module of the call stack is wrapped).  Costs two attribute reads in
    training mode."""
    global _implicit
    if layer.training or torch.is_grad_enabled():
        return
    w = layer.weight
    if not w.is_cuda:
        return
    if _implicit is None:
        from .utils import implicit as _mod
        _implicit = _mod
    _implicit.note(layer)


@contextlib.contextmanager
def eager():
    """Run the enclosed forwards module by module (no deferred activations); per thread, re-entrant."""
    _tls.eager_depth = getattr(_tls, "eager_depth", 0) + 1
    try:
        yield
    finally:
        _tls.eager_depth -= 1


@contextlib.contextmanager
def codes_deferred(on: bool = True):
    """Defer DoReFa code chains (``DEFER_CODES``) inside the block; process-wide, restores the previous setting."""
    global DEFER_CODES
    prev, DEFER_CODES = DEFER_CODES, bool(on)
    try:
        yield
    finally:
        DEFER_CODES = prev


class _Node:
    """One recorded step.  ``parent is None``: the deferred conv itself (``layer``, ``kind``, ``input`` = device tensor or
    PackedActivation); otherwise ``op`` applied to ``parent``.  The flags summarise the chain from the root."""
    __slots__ = ("parent", "op", "layer", "kind", "input", "shape", "value", "packed_cache",
                 "pool", "bn", "hardtanh", "flat", "signed", "pool2", "chw", "stamp", "add", "relu", "quant", "device",
                 "captured")

    def __init__(self, parent: Optional["_Node"], op, shape, layer=None, kind=None, input=None):
        self.parent, self.op, self.shape = parent, op, shape          # shape: tuple of ints
        self.value = None
        self.packed_cache = None
        if parent is None:
            self.layer, self.kind, self.input = layer, kind, input
            self.device = input.device
            self.pool = self.bn = self.hardtanh = self.pool2 = self.add = self.quant = None
            self.flat = self.signed = self.relu = False
            self.chw = None
            self.stamp = _stamp(input, layer.weight, layer.bias) if hasattr(layer, "weight") else ()
            self.captured = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
            return
        p = parent
        self.layer, self.kind, self.input, self.device = p.layer, p.kind, p.input, p.device
        self.pool, self.bn, self.hardtanh, self.flat, self.signed = p.pool, p.bn, p.hardtanh, p.flat, p.signed
        self.pool2, self.chw, self.add, self.relu, self.quant = p.pool2, p.chw, p.add, p.relu, p.quant
        self.stamp = ()
        tag = op[0]
        if tag == "pool":
            self.pool = op[1:]
        elif tag == "bn":
            self.bn = op[1:]
            self.stamp = _stamp(*op[1:5])
        elif tag == "hardtanh":
            self.hardtanh = op[1:]
        elif tag == "flat":
            self.flat, self.chw = True, parent.shape[1:]
        elif tag == "sign":
            self.signed = True
        elif tag == "pool2":
            self.pool2 = op[1:]
        elif tag == "add":
            self.add = op[1]
            self.stamp = _stamp(op[1])
        elif tag == "relu":
            self.relu = True
        elif tag == "quant":
            self.quant = op[1]

    def check_unmodified(self):
        """A deferred activation reads its producers when it is USED; like autograd's saved tensors they must not have
        been written in place in between (version counters: load_state_dict / copy_ / optimizer steps / in-place ops)."""
        node = self
        while node is not None:
            if node.parent is None and node.captured and not torch.cuda.is_current_stream_capturing():
                raise RuntimeError(
                    "a deferred activation created while a stream was being captured (hipGraph) was not used inside the "
                    "captured region, so its kernels are not part of the graph; end the region with an op on it (or "
                    ".value()), use pytorch_quantize_impls_amd.utils.graphed(), or capture under lazy.eager()")
            for t, ptr, version in node.stamp:
                if t.data_ptr() != ptr or t._version != version:
                    raise RuntimeError(
                        "a tensor a deferred activation depends on (the input, weight / bias of a binarised conv, or the "
                        "statistics of the BatchNorm that followed it) was modified in place before the activation was "
                        "used; use the activation first (any tensor op, or .value()), or run the forward under "
                        "pytorch_quantize_impls_amd.lazy.eager()")
            node = node.parent

    # ---- the un-fused value ------------------------------------------------------------------------------------
    def materialise(self) -> torch.Tensor:
        """The fp32 tensor.  Always evaluated without autograd — the mode the chain was deferred in — whatever mode the
        first use happens in (a deferred activation never requires grad)."""
        if self.value is None:
            with torch.no_grad():
                self._materialise()
        return self.value

    def _materialise(self) -> None:
        if self.value is None:
            self.check_unmodified()
            STATS["materialised"] += 1
            codes = self.force_any() if (self.kind == "dorefa" and self.quant is not None) else None
            if codes is not None:
                # a quantised DoReFa chain always evaluates through the code epilogue (one arithmetic for every consumer):
                # the int8 codes, expanded; the chain's int8-range flag is applied on the device (NaN, no host sync)
                y = codes.float()
            elif self.parent is None:
                y = self.layer._forward_impl(self.input)
            else:
                x = self.parent.materialise()
                tag = self.op[0]
                if tag in ("pool", "pool2"):
                    y = F.max_pool2d(x, self.op[1], self.op[2])
                elif tag == "bn":
                    rm, rv, w, b, eps = self.op[1:]
                    y = F.batch_norm(x, rm, rv, w, b, False, 0.0, eps)
                elif tag == "hardtanh":
                    y = F.hardtanh(x, self.op[1], self.op[2])
                elif tag == "flat":
                    y = x.reshape(self.shape)
                elif tag == "add":
                    y = x + resolve(self.op[1])
                elif tag == "relu":
                    y = torch.relu(x)
                elif tag == "quant":
                    from .functions.dorefa_connect import _quantize
                    y = _quantize(x, bit_width=self.op[1])
                else:   # sign
                    from .functions.binary_connect import _binarize_and_tag
                    y = _binarize_and_tag(x)
            self.value = y

    # ---- the fused execution -----------------------------------------------------------------------------------
    def force(self, halo=None):
        """PackedActivation of a signed chain: (N, C, H, W) bit planes, or the consumer's nibble operand with a zero border
        of ``halo`` pixels, or (flat chains) row planes in (h, w, c) order.  DoReFa chains: the CodeActivation of a
        quantised chain (int8 code plane, ``halo`` = zero border).  None if this chain cannot run fused."""
        if self.kind == "dorefa":
            return self._force_codes(halo)
        if self.kind == "dense":
            return self._force_dense()
        if not self.signed or self.bn is None:
            return None
        if self.flat:
            halo = None
        key = halo
        if self.packed_cache is None:
            self.packed_cache = {}
        if key in self.packed_cache:
            return self.packed_cache[key]
        self.check_unmodified()
        try:
            with torch.no_grad():
                block = _fused_block(self.layer, self.bn, self.pool, flatten=self.flat and self.pool2 is None,
                                     halo=halo if self.pool2 is None else None)
                act = block(self.input)
                if self.pool2 is not None:
                    act = _packed_pool(self.pool2, halo)(act)
                    if self.flat:
                        act = act.flatten_hwc()
        except ValueError:
            act = None
        if act is not None:
            STATS["fused"] += 1
        self.packed_cache[key] = act
        return act


def _stamp(*tensors):
    """(tensor, storage pointer, version counter) of every real tensor among ``tensors``.  Inference tensors track no
    version counter; layers whose producers are inference tensors are not deferred at all (``_untracked``), so the ones
    skipped here are internal planes nobody else holds."""
    return tuple((t, t.data_ptr(), t._version) for t in tensors
                 if isinstance(t, torch.Tensor) and not isinstance(t, LazyActivation) and not t.is_inference())


def _force_codes(self, halo=None):
    if self.quant is None or self.bn is None or self.flat:
        return None
    halo = tuple(halo) if halo is not None else (0, 0)
    if self.packed_cache is None:
        self.packed_cache = {}
    if halo in self.packed_cache:
        return self.packed_cache[halo]
    self.check_unmodified()
    from .layers import fused
    try:
        with torch.no_grad():
            blk = _code_block(self.layer, self.bn, self.quant, self.relu, halo if self.pool2 is None else (0, 0))
            res = res_bn = res_conv = None
            if self.add is not None:
                other = self.add
                if isinstance(other, LazyActivation):
                    o = other._qt
                    if o.quant is not None:                       # identity shortcut: the block's own (quantised) input
                        res = o.force_any()
                    else:                                         # conv + BatchNorm shortcut: fp32 conv output, BN folded
                        root = o.parent
                        if (root.parent is None and root.value is None and isinstance(root.input, packed.CodeActivation)
                                and o.bn is not None and o.pool is None and o.relu is False and o.add is None
                                and blk.shortcut_in_one_launch(root.layer, root.input, _bn_view(blk, o.bn))):
                            # ... the un-materialised conv of the branch on a code plane: conv + BatchNorm in one launch
                            o.check_unmodified()                       # the branch's BatchNorm tensors and, through the parent, its conv's
                            res_conv, res_bn = (root.layer, root.input), _bn_view(blk, o.bn)
                        else:
                            res, res_bn = root.materialise(), _bn_view(blk, o.bn)
                else:
                    res = other
                    # an fp32 residual that carries the int8 codes of the quantiser that produced it (the identity shortcut of the
                    # FIRST block of a ResNet: the stem's nnDorefaQuant output, which is also where this chain's codes started —
                    # same range flag): join as codes, 1 byte per element and the straight-line epilogue instead of 4 bytes and
                    # the general one (87 -> ~24 us for the 64 -> 64 conv at 32 x 32, batch 256).  fl(inv_n * q) is what the
                    # quantiser wrote into the fp32 image and what the code path adds: the same bits
                    tag = packed.lookup_codes(other, packed.NHWC) if (isinstance(other, torch.Tensor) and other.dim() == 4
                                                                      and other.dtype == torch.float32) else None
                    inp = self.input
                    if (tag is not None and isinstance(inp, packed.CodeActivation) and tag.overflow is not None
                            and tag.overflow is inp.codes.overflow and tag.K == int(other.shape[1])
                            and tag.rows == int(other.shape[0]) * int(other.shape[2]) * int(other.shape[3])):
                        res = packed.CodeActivation(tag, tuple(int(v) for v in other.shape))
                        STATS["residual_as_codes"] += 1
                if res is None and res_conv is None:
                    raise ValueError("residual cannot join the fused chain")
            act = blk(self.input, residual=res, residual_bn=res_bn, residual_conv=res_conv)
            if self.pool2 is not None:
                act = fused.CodeMaxPool(torch.nn.MaxPool2d(self.pool2[0], self.pool2[1]), out_halo=halo)(act)
    except ValueError:
        act = None
    if act is not None:
        STATS["fused"] += 1
    self.packed_cache[halo] = act
    return act


def _force_any(self):
    """The chain's CodeActivation with whatever halo it was already produced with (a residual may carry any)."""
    for act in (self.packed_cache or {}).values():
        if act is not None:
            return act
    return self.force(None)


_Node._force_codes = _force_codes
_Node.force_any = _force_any


def _bn_view(owner_block, bn):
    """_BnView of the shortcut's BatchNorm, cached on the fused block that folds it."""
    key = tuple(id(t) for t in bn[:4]) + (bn[4],)
    cache = owner_block.__dict__.setdefault("_qt_res_bn", {})
    if key not in cache:
        if len(cache) > 4:
            cache.clear()
        cache[key] = _BnView(*bn)
    return cache[key]


class _BnView(torch.nn.BatchNorm2d):
    """The tensors F.batch_norm was called with, presented as the BatchNorm2d module layers.fused folds (shares them)."""

    def __init__(self, rm, rv, w, b, eps):
        torch.nn.Module.__init__(self)
        self.num_features, self.eps, self.momentum = int(rm.numel()), float(eps), None
        self.affine, self.track_running_stats = w is not None, True
        for name, t in (("running_mean", rm), ("running_var", rv), ("weight", w), ("bias", b)):
            object.__setattr__(self, name, t)
        self.training = False


_BLOCKS = weakref.WeakKeyDictionary()      # conv layer -> OrderedDict(key -> FusedConvPoolBnSign)
_POOLS = {}
_MAX_BLOCKS_PER_LAYER = 8


def _fused_block(layer, bn, pool, flatten: bool, halo):
    from .layers import fused
    rm, rv, w, b, eps = bn
    key = (id(rm), id(rv), id(w), id(b), eps, pool, flatten, halo)
    per = _BLOCKS.get(layer)
    if per is None:
        per = _BLOCKS[layer] = collections.OrderedDict()
    blk = per.get(key)
    if blk is None:
        pm = torch.nn.MaxPool2d(pool[0], pool[1]) if pool is not None else None
        blk = fused.FusedConvPoolBnSign(layer, _BnView(rm, rv, w, b, eps), pm, flatten_hwc=flatten, fold="device")
        blk.out_nib_halo = halo
        per[key] = blk
        while len(per) > _MAX_BLOCKS_PER_LAYER:
            per.popitem(last=False)
    else:
        per.move_to_end(key)
    return blk


def _code_block(layer, bn, bit_width, relu, halo):
    from .layers import fused
    rm, rv, w, b, eps = bn
    key = (id(rm), id(rv), id(w), id(b), eps, "codes", bit_width, relu, halo)
    per = _BLOCKS.get(layer)
    if per is None:
        per = _BLOCKS[layer] = collections.OrderedDict()
    blk = per.get(key)
    if blk is None:
        blk = fused.FusedDorefaConvBnQuant(layer, _BnView(rm, rv, w, b, eps), bit_width, relu=relu, out_halo=halo, fold="device")
        per[key] = blk
        while len(per) > _MAX_BLOCKS_PER_LAYER:
            per.popitem(last=False)
    else:
        per.move_to_end(key)
    return blk


def _packed_pool(pool, halo):
    from .layers import fused
    key = (pool, halo)
    pm = _POOLS.get(key)
    if pm is None:
        pm = fused.PackedMaxPool(torch.nn.MaxPool2d(pool[0], pool[1]))
        pm.out_nib_halo = halo
        _POOLS[key] = pm
    return pm


class LazyActivation(torch.Tensor):
    """fp32 activation of a binarised conv chain that has not been computed (see the module docstring)."""

    @staticmethod
    def __new__(cls, node: _Node, device=None):
        t = torch.Tensor._make_wrapper_subclass(cls, node.shape, dtype=torch.float32, device=node.device,
                                                requires_grad=False)
        t._qt = node
        return t

    def value(self) -> torch.Tensor:
        """The ordinary fp32 tensor this activation stands for (computed module by module on first use)."""
        return self._qt.materialise()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        handler = _HANDLERS.get(func)
        if handler is not None:
            out = handler(*args, **kwargs)
            if out is not NotImplemented:
                return out
        elif func in _METADATA:
            fast = _META_FAST.get(func)
            if fast is not None and not kwargs:
                return fast(*args)
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        name = getattr(func, "__name__", str(func))
        if args and isinstance(args[0], LazyActivation) and args[0]._qt.value is not None:
            # a use of an already computed value (dense / constant nodes) forces nothing; the common shape — the activation first, no
            # other deferred operand, not an in-place / out= form — skips the generic argument walk (log_softmax on the last layer)
            if (not _writes_in_place(name) and "out" not in kwargs
                    and not any(isinstance(a, LazyActivation) for a in args[1:])
                    and not any(isinstance(v, LazyActivation) for v in kwargs.values())):
                return func(args[0]._qt.value, *args[1:], **kwargs)
        else:
            STATS["fallback:" + name] += 1
        out_kw = kwargs.get("out")
        if isinstance(out_kw, LazyActivation):
            # func(..., out=deferred): the result replaces what the wrapper stood for — computed into a fresh tensor (the cached
            # value of the old node may be the parent of chains recorded earlier), the wrapper moves on to a constant node
            rest = {k: v for k, v in kwargs.items() if k != "out"}
            a2, rest = tree_map_only(LazyActivation, lambda t: t._qt.materialise(), (tuple(args), rest))
            own = _own_storage(out_kw)
            if own is not None:                 # a LazyDense IS its memory: the result is written there (C++ readers see it)
                func(*a2, out=own, **rest)
                out_kw._qt = _const_node(own)
            else:
                out_kw._qt = _const_node(func(*a2, **rest))
            return out_kw
        if _writes_in_place(name) and args and isinstance(args[0], LazyActivation):
            # x.op_(...) on a deferred activation outside the grammar: the module-by-module value, mutated — but in a
            # private copy (the cached value of this node may be the parent of chains recorded earlier), and the wrapper
            # moves on to a constant node holding the result, as the tensor an in-place op returns would
            self_ = args[0]
            own = _own_storage(self_)
            v = own if own is not None else self_._qt.materialise().clone()
            rest, kwargs = tree_map_only(LazyActivation, lambda t: t._qt.materialise(), (tuple(args[1:]), kwargs))
            func(v, *rest, **kwargs)
            self_._qt = _const_node(v)
            return self_
        args, kwargs = tree_map_only(LazyActivation, lambda t: t._qt.materialise(), (args, kwargs))
        return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # safety net: an ATen op reached the dispatcher with a deferred activation (C++ callers, autograd internals)
        args, kwargs = tree_map_only(LazyActivation, lambda t: t._qt.materialise(), (args, kwargs or {}))
        return func(*args, **kwargs)


def _own_storage(t):
    """For a LazyDense about to be overwritten (in-place op / out=): its own memory as a plain tensor, after the node that
    chains recorded earlier hang on has been given a private copy of the old value; None for storage-less activations."""
    if type(t) is not LazyDense:
        return None
    with torch._C.DisableTorchFunctionSubclass():
        own = t.as_subclass(torch.Tensor)
    node = t._qt
    if node.value is not None and node.value.data_ptr() == own.data_ptr():
        node.value = own.clone()
        if node.kind == "dense":
            node.stamp = _stamp(node.value)          # chains recorded on the node now read (and version-check) the copy
    return own


class LazyDense(LazyActivation):
    """The COMPUTED [B, N] result of an eval-mode quantised Linear layer (DEFER_DENSE): the real tensor — same storage, made
    with ``as_subclass`` — that can still record BatchNorm1d -> [Hardtanh] -> BinaryConnect.  Unlike a storage-less
    LazyActivation it is an ordinary tensor to everything that bypasses ``__torch_function__`` (torch.distributed collectives,
    C++ extensions, pickling / torch.save): a model's final logits travel as plain memory."""

    @staticmethod
    def __new__(cls, node: _Node, device=None):
        with torch._C.DisableTorchFunctionSubclass():
            t = node.value.as_subclass(cls)
        t._qt = node
        return t

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def plain(t):
            v = t._qt.value
            if v is None:                                  # the wrapper moved on (in-place op outside the grammar)
                v = t._qt.materialise()
            return v
        args, kwargs = tree_map_only(LazyActivation, plain, (args, kwargs or {}))
        return func(*args, **kwargs)


def _writes_in_place(name: str) -> bool:
    return (name.endswith("_") and not name.endswith("__")) or name == "__setitem__" or (
        name.startswith("__i") and name.endswith("__") and name not in ("__int__", "__index__", "__invert__"))


def _const_node(value: torch.Tensor) -> _Node:
    """A node that only holds a computed value (nothing can be recorded on it: every handler declines)."""
    n = _Node.__new__(_Node)
    n.parent = n.op = n.layer = n.input = None
    n.kind, n.shape, n.value, n.packed_cache = "const", tuple(int(v) for v in value.shape), value, None
    n.pool = n.pool2 = n.add = n.quant = n.chw = None
    n.bn = n.hardtanh = n.flat = True          # "already there": pool / BatchNorm / Hardtanh / flatten handlers decline
    n.signed = n.relu = n.captured = False
    n.stamp, n.device = (), value.device
    return n


def _dense_node(value: torch.Tensor) -> _Node:
    """Root that holds the computed [B, N] result of a quantised Linear layer: BatchNorm1d, Hardtanh and BinaryConnect can be
    recorded on it (kind "dense"); everything else reads the value."""
    n = _const_node(value)
    n.kind = "dense"
    n.bn = n.hardtanh = None                  # still to come; flat stays True: pooling / flatten handlers decline
    n.stamp = _stamp(value)
    n.captured = value.is_cuda and torch.cuda.is_current_stream_capturing()
    return n


_DENSE_BLOCKS = collections.OrderedDict()     # BatchNorm tensors' ids -> FusedPoolBnSign (its fold is re-validated per call)


def _force_dense(self):
    """PackedActivation (row planes) of a recorded Linear -> BatchNorm1d -> [Hardtanh] -> BinaryConnect chain: one pass over the
    layer's fp32 result with this device's BatchNorm thresholds."""
    if not self.signed or self.bn is None:
        return None
    if self.packed_cache is None:
        self.packed_cache = {}
    if None in self.packed_cache:
        return self.packed_cache[None]
    self.check_unmodified()
    from .layers import fused
    rm, rv, w, b, eps = self.bn
    key = (id(rm), id(rv), id(w), id(b), eps)
    blk = _DENSE_BLOCKS.get(key)
    if blk is None:
        blk = _DENSE_BLOCKS[key] = fused.FusedPoolBnSign(_BnView(rm, rv, w, b, eps), fold="device")
        while len(_DENSE_BLOCKS) > 64:
            _DENSE_BLOCKS.popitem(last=False)
    else:
        _DENSE_BLOCKS.move_to_end(key)
    root = self
    while root.parent is not None:
        root = root.parent
    try:
        with torch.no_grad():
            act = blk(root.value)
    except ValueError:
        act = None
    if act is not None:
        STATS["dense_fused"] += 1
    self.packed_cache[None] = act
    return act


_Node._force_dense = _force_dense


_T = torch.Tensor
_METADATA = {_T.dim, _T.size, _T.numel, _T.ndimension, _T.nelement, _T.is_floating_point, _T.is_complex, _T.element_size,
             _T.shape.__get__, _T.dtype.__get__, _T.device.__get__, _T.is_cuda.__get__, _T.ndim.__get__,
             _T.requires_grad.__get__, _T.layout.__get__, _T.grad_fn.__get__, _T.is_leaf.__get__, _T.is_inference,
             _T.is_sparse.__get__, _T.is_quantized.__get__, _T.is_meta.__get__, _T.get_device, _T.__len__,
             _T._version.__get__}


def _size(t, dim=None):
    shape = t._qt.shape
    return torch.Size(shape) if dim is None else shape[dim]


def _numel(t):
    n = 1
    for v in t._qt.shape:
        n *= v
    return n


#: the metadata queries nn modules make on every call, answered from the node without entering the dispatcher
_META_FAST = {_T.dim: lambda t: len(t._qt.shape), _T.ndimension: lambda t: len(t._qt.shape),
              _T.ndim.__get__: lambda t: len(t._qt.shape), _T.size: _size, _T.shape.__get__: lambda t: torch.Size(t._qt.shape),
              _T.dtype.__get__: lambda t: torch.float32, _T.device.__get__: lambda t: t._qt.device,
              _T.is_cuda.__get__: lambda t: t._qt.device.type == "cuda", _T.numel: _numel, _T.nelement: _numel,
              _T.requires_grad.__get__: lambda t: False, _T.is_floating_point: lambda t: True,
              _T.__len__: lambda t: t._qt.shape[0]}


def _wrap(node: _Node, device=None) -> LazyActivation:
    return LazyActivation(node)


def resolve(x):
    """``x`` itself, or the fp32 tensor a deferred activation stands for (entry guard of every consumer that does not
    take part in the fused chain)."""
    if isinstance(x, LazyActivation):
        return x._qt.materialise()
    if type(x) in lazy_train._DEFERRED:
        return lazy_train.resolve(x)
    return x


# ---- recording handlers ----------------------------------------------------------------------------------------------

def _square(v):
    if isinstance(v, int):
        return int(v)
    v = tuple(v)
    if len(v) == 1:
        return int(v[0])
    if len(v) == 2 and v[0] == v[1]:
        return int(v[0])
    return None


def _h_max_pool2d(input, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False, return_indices=False):
    if not isinstance(input, LazyActivation) or return_indices or ceil_mode:
        return NotImplemented
    n = input._qt
    k = _square(kernel_size)
    s = _square(stride) if stride not in (None, [], ()) else k
    if k is None or s is None or _square(padding) != 0 or _square(dilation) != 1 or len(n.shape) != 4:
        return NotImplemented
    N, C, H, W = n.shape
    if H < k or W < k:
        return NotImplemented
    shape = (N, C, (H - k) // s + 1, (W - k) // s + 1)
    if n.kind == "dorefa":
        if n.quant is not None and n.pool2 is None:          # DoReFa CNNs pool AFTER the quantiser: max of the codes
            return _wrap(_Node(n, ("pool2", k, s), shape))
        return NotImplemented
    if n.bn is None and n.pool is None:
        return _wrap(_Node(n, ("pool", k, s), shape))
    if n.signed and not n.flat and n.pool2 is None:
        return _wrap(_Node(n, ("pool2", k, s), shape))
    return NotImplemented


def _h_batch_norm(input, running_mean, running_var, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
    if not isinstance(input, LazyActivation) or training or running_mean is None or running_var is None:
        return NotImplemented
    n = input._qt
    if n.kind == "dense":
        if n.bn is not None or len(n.shape) != 2:
            return NotImplemented
    elif n.bn is not None or n.flat or len(n.shape) != 4 or n.add is not None:
        return NotImplemented
    C = n.shape[1]
    for t in (running_mean, running_var, weight, bias):
        if t is None:
            continue
        if (isinstance(t, LazyActivation) or t.device != n.device or t.dtype != torch.float32 or t.dim() != 1
                or t.numel() != C):
            return NotImplemented
    if (weight is None) != (bias is None) or _untracked(running_mean, running_var, weight, bias):
        return NotImplemented
    return _wrap(_Node(n, ("bn", running_mean, running_var, weight, bias, float(eps)), n.shape))


def _h_hardtanh(input, min_val=-1.0, max_val=1.0, inplace=False):
    if not isinstance(input, LazyActivation):
        return NotImplemented
    n = input._qt
    if n.kind == "dorefa" or n.bn is None or n.hardtanh is not None or n.signed or not (min_val < 0 < max_val):
        return NotImplemented
    child = _Node(n, ("hardtanh", float(min_val), float(max_val)), n.shape)
    if inplace:                      # same shape: the wrapper object itself moves on, as an in-place op's result would
        input._qt = child
        return input
    return _wrap(child)


def _h_dropout(input, p=0.5, training=True, inplace=False):
    if not isinstance(input, LazyActivation) or training:
        return NotImplemented
    return input


def _h_relu(input, inplace=False):
    if not isinstance(input, LazyActivation):
        return NotImplemented
    n = input._qt
    if n.kind != "dorefa" or n.bn is None or n.relu or n.quant is not None:
        return NotImplemented
    child = _Node(n, ("relu",), n.shape)
    if inplace:
        input._qt = child
        return input
    return _wrap(child)


def _h_relu_(input):
    return _h_relu(input, inplace=True)


def _conv_macs(n: _Node) -> int:
    w = n.layer.weight
    return int(w.shape[1]) * int(w.shape[2]) * int(w.shape[3])


def _residual_ok(main: _Node, other) -> bool:
    if isinstance(other, LazyActivation):
        o = other._qt
        if o.kind != "dorefa" or o.shape != main.shape or o.flat:
            return False
        if o.quant is not None:
            return True
        return o.bn is not None and o.parent is not None and o.parent.parent is None       # exactly conv -> BatchNorm
    return (isinstance(other, torch.Tensor) and other.device == main.input.device and other.dtype == torch.float32
            and tuple(other.shape) == main.shape and not other.requires_grad and not other.is_inference())


def _mainline(n: _Node) -> bool:
    return (n.kind == "dorefa" and n.bn is not None and n.add is None and not n.relu and n.quant is None and not n.flat
            and len(n.shape) == 4)


def _h_add(a, b, *, alpha=1, out=None, _inplace=False):
    """bn(conv(x)) + shortcut: the residual add of a DoReFa ResNet block, folded into the conv's code epilogue."""
    if alpha != 1 or out is not None:
        return NotImplemented
    cands = []
    if isinstance(a, LazyActivation) and _mainline(a._qt) and _residual_ok(a._qt, b):
        cands.append((a, b))
    if not _inplace and isinstance(b, LazyActivation) and _mainline(b._qt) and _residual_ok(b._qt, a):
        cands.append((b, a))
    if not cands:
        return NotImplemented
    # both operands conv -> BatchNorm (3x3 main path + 1x1 shortcut): the bigger conv keeps its epilogue
    main, other = max(cands, key=lambda c: _conv_macs(c[0]._qt))
    child = _Node(main._qt, ("add", other), main._qt.shape)
    if _inplace:
        a._qt = child
        return a
    return _wrap(child)


def _h_iadd(a, b, *, alpha=1):
    return _h_add(a, b, alpha=alpha, _inplace=True)


def _flat_target(n: _Node, shape):
    """True iff ``shape`` (ints, at most one -1) flattens the (N, C, H, W) chain to (N, C*H*W)."""
    if len(n.shape) != 4 or n.flat or n.bn is None or n.kind == "dorefa":
        return False
    N, C, H, W = n.shape
    shape = tuple(int(v) for v in shape)
    if len(shape) != 2:
        return False
    a, b = shape
    if a == -1 and b == -1:
        return False
    if a == -1:
        a = N if b == C * H * W else -2
    if b == -1:
        b = C * H * W if a == N else -2
    return (a, b) == (N, C * H * W)


def _as_flat(input, ok):
    if not ok:
        return NotImplemented
    n = input._qt
    N, C, H, W = n.shape
    return _wrap(_Node(n, ("flat",), (N, C * H * W)))


def _h_reshape(input, *shape):
    if not isinstance(input, LazyActivation):
        return NotImplemented
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    if not all(isinstance(v, int) for v in shape):
        return NotImplemented
    return _as_flat(input, _flat_target(input._qt, shape))


def _h_flatten(input, start_dim=0, end_dim=-1):
    if not isinstance(input, LazyActivation):
        return NotImplemented
    n = input._qt
    ok = (len(n.shape) == 4 and start_dim == 1 and end_dim in (-1, 3) and not n.flat and n.bn is not None
          and n.kind != "dorefa")
    return _as_flat(input, ok)


def _h_avg_pool2d(input, kernel_size, stride=None, padding=0, ceil_mode=False, count_include_pad=True, divisor_override=None):
    """F.avg_pool2d(k) (kernel = stride, no padding) of a QUANTISED DoReFa chain — the head of a DoReFa ResNet — reads the chain's
    int8 codes and pools their fp32 image in the same pass (packed.CodeActivation.avg_pool2d: ATen's order, bit-identical to
    pooling the materialised image); the result is a computed value (constant node)."""
    if not isinstance(input, LazyActivation) or ceil_mode or divisor_override is not None:
        return NotImplemented
    n = input._qt
    k = _square(kernel_size)
    s = _square(stride) if stride not in (None, [], ()) else k
    if (k is None or s != k or _square(padding) != 0 or len(n.shape) != 4 or n.kind != "dorefa" or n.quant is None
            or n.pool2 is not None or n.value is not None or n.shape[2] % k or n.shape[3] % k):
        return NotImplemented
    n.check_unmodified()
    with torch.no_grad():
        codes = n.force_any()
    if codes is None:
        return NotImplemented
    STATS["avg_pool_on_codes"] += 1
    return codes.avg_pool2d(k)


_HANDLERS = {
    F.max_pool2d: _h_max_pool2d,
    F.avg_pool2d: _h_avg_pool2d,
    F.batch_norm: _h_batch_norm,
    F.hardtanh: _h_hardtanh,
    F.dropout: _h_dropout,
    _T.reshape: _h_reshape,
    _T.view: _h_reshape,
    torch.reshape: _h_reshape,
    _T.flatten: _h_flatten,
    torch.flatten: _h_flatten,
    torch.relu: _h_relu, _T.relu: _h_relu, F.relu: _h_relu, torch.relu_: _h_relu_, _T.relu_: _h_relu_,
    torch.add: _h_add, _T.add: _h_add, _T.__add__: _h_add, _T.__radd__: _h_add,
    _T.add_: _h_iadd, _T.__iadd__: _h_iadd,
}


def sign(x: LazyActivation):
    """BinaryConnect (deterministic) of a deferred activation: recorded if the chain allows it, else None (the caller
    then binarises the materialised value)."""
    n = x._qt
    if n.kind in ("dorefa", "const"):
        return None
    if n.signed:
        return x                     # sign(+-1) == itself
    if n.bn is None:
        return None
    return _wrap(_Node(n, ("sign",), n.shape))


def quant(x: LazyActivation, bit_width: int):
    """nnDorefaQuant(k) of a deferred DorefaConv2d chain (conv -> BatchNorm [-> + shortcut] [-> ReLU]): recorded when
    the code epilogue can produce it (2 <= k <= 8), else None."""
    n = x._qt
    if n.kind != "dorefa" or n.bn is None or n.quant is not None or n.flat or not 2 <= int(bit_width) <= 8:
        return None
    return _wrap(_Node(n, ("quant", int(bit_width)), n.shape))


# ---- layer entry points ----------------------------------------------------------------------------------------------

def _no_autograd(layer) -> bool:
    return not torch.is_grad_enabled()


def _untracked(*tensors) -> bool:
    """An inference tensor among the producers: it has no version counter, so a write in place between the deferral and
    the use (possible inside ``torch.inference_mode()``) could not be detected — such a layer runs eagerly instead."""
    return any(isinstance(t, torch.Tensor) and not isinstance(t, LazyActivation) and t.is_inference() for t in tensors)


def _conv_can_defer(layer, input) -> bool:
    if layer.training or not _no_autograd(layer) or layer.groups != 1 or layer.padding_mode != "zeros" \
            or isinstance(layer.padding, str) or not getattr(layer, "_qt_can_defer", True):
        return False
    w = layer.weight
    if not w.is_cuda or w.dtype != torch.float32:
        return False
    if isinstance(input, packed.PackedActivation):
        ok = len(input.shape) == 4
    else:
        ok = (isinstance(input, torch.Tensor) and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4
              and input.numel() > 0)
    if _untracked(input, w, layer.bias):
        return False
    return ok and int(input.shape[1]) == layer.in_channels and layer._eval_on_grid()


def conv_forward(layer, input, kind: str):
    """forward() of BinConv2d / TerConv2d: consumes a deferred activation as packed planes and, when it may, defers
    itself; everything else goes to the layer's ordinary path (``_forward_impl``)."""
    note_inference_call(layer, input)
    if isinstance(input, LazyActivation):
        act = None
        n = input._qt
        if n.signed and not n.flat and _conv_can_defer(layer, _ShapeOnly(n.shape)):
            from .functions import _fused
            act = n.force(tuple(int(v) for v in ops._pairs(layer.padding)) if _fused._cfg("PAD_PLANES") else None)
        input = act if act is not None else n.materialise()
    if enabled() and _conv_can_defer(layer, input):
        N, C, H, W = (int(v) for v in input.shape)
        kh, kw = layer.kernel_size
        Ho, Wo = ops.conv_out_hw(H, W, kh, kw, layer.stride, layer.padding, layer.dilation)
        if Ho > 0 and Wo > 0:
            STATS["deferred"] += 1
            return _wrap(_Node(None, None, (N, int(layer.out_channels), int(Ho), int(Wo)), layer=layer, kind=kind,
                               input=input))
    if type(input) in lazy_train._DEFERRED:
        input = lazy_train.resolve(input)
    return lazy_train.wrap(layer, layer._forward_impl(input))


def _dorefa_can_defer(layer, input) -> bool:
    if (layer.training or not _no_autograd(layer) or layer.groups != 1 or layer.padding_mode != "zeros"
            or isinstance(layer.padding, str) or layer.bit_width != 1):
        return False
    w = layer.weight
    if not w.is_cuda or w.dtype != torch.float32 or not isinstance(input, packed.CodeActivation) or len(input.shape) != 4:
        return False
    if _untracked(w, layer.bias):
        return False
    return int(input.shape[1]) == layer.in_channels and layer._eval_on_grid()


def dorefa_conv_forward(layer, input):
    """forward() of DorefaConv2d: a 1-bit-weight conv in eval mode whose input carries int8 codes (the tag nnDorefaQuant
    leaves on its result, a CodeActivation, or a deferred quantised chain) defers itself."""
    note_inference_call(layer, input)
    if isinstance(input, LazyActivation):
        n = input._qt
        act = None
        if n.kind == "dorefa" and n.quant is not None and _dorefa_can_defer(layer, _CodeShapeOnly(n.shape)):
            pad = tuple(int(v) for v in ops._pairs(layer.padding))
            # a plane already produced for another consumer serves this one too if its zero border covers the padding
            # (the kernels read a halo plane at an offset of halo - padding pixels): e.g. the 1x1 shortcut conv of a
            # ResNet block after the 3x3 conv that shares its input
            for halo, done in (n.packed_cache or {}).items():
                if done is not None and halo[0] >= pad[0] and halo[1] >= pad[1]:
                    act = done
                    break
            if act is None:
                act = n.force(pad)
        input = act if act is not None else n.materialise()
    tagged = None
    defer = enabled() and DEFER_CODES
    if defer and not isinstance(input, packed.CodeActivation) and isinstance(input, torch.Tensor) and input.is_cuda \
            and input.dtype == torch.float32 and input.dim() == 4 and not torch.is_grad_enabled() and not layer.training \
            and layer.bit_width == 1:
        codes = packed.lookup_codes(input, packed.NHWC)
        N, C, H, W = (int(v) for v in input.shape)
        if codes is not None and codes.K == C and codes.rows == N * H * W:
            cand = packed.CodeActivation(codes, (N, C, H, W))
            if _dorefa_can_defer(layer, cand):
                tagged, input = input, cand
    if defer and _dorefa_can_defer(layer, input):
        N, C, H, W = (int(v) for v in input.shape)
        kh, kw = layer.kernel_size
        Ho, Wo = ops.conv_out_hw(H, W, kh, kw, layer.stride, layer.padding, layer.dilation)
        if Ho > 0 and Wo > 0 and 127 * kh * kw * int(input.codes.codes.shape[1]) < (1 << 24):
            STATS["deferred"] += 1
            node = _Node(None, None, (N, int(layer.out_channels), int(Ho), int(Wo)), layer=layer, kind="dorefa", input=input)
            node.stamp += _stamp(tagged)          # the fp32 tensor whose code tag is being used
            return _wrap(node)
    if type(input) in lazy_train._DEFERRED:
        input = lazy_train.resolve(input)
    return lazy_train.wrap(layer, layer._forward_impl(tagged if tagged is not None else input))


class _CodeShapeOnly(packed.CodeActivation):
    def __init__(self, shape):
        self.shape = tuple(shape)


class _ShapeOnly(packed.PackedActivation):
    """Stand-in used to ask _conv_can_defer about an activation that has not been forced yet."""

    def __init__(self, shape):
        self.shape = tuple(shape)


def _dense_result(layer, y):
    """``y`` as a deferred activation that BatchNorm1d -> [Hardtanh] -> BinaryConnect can be recorded on (DEFER_DENSE)."""
    if (DEFER_DENSE and enabled() and type(y) is torch.Tensor and y.dim() == 2 and y.is_cuda and y.dtype == torch.float32
            and not y.requires_grad and not layer.training and _no_autograd(layer) and not _untracked(y)):
        STATS["dense_deferred"] += 1
        return LazyDense(_dense_node(y))
    return y


def linear_forward(layer, input, kind: str):
    """forward() of LinearBin / LinearTer / LinearXNOR: a flattened, binarised deferred activation arrives as row planes in
    (h, w, c) order and meets the weight with its columns permuted to that order (cached per weight version).  In eval mode without
    autograd the result goes out as a "dense" deferred activation (DEFER_DENSE)."""
    note_inference_call(layer, input)
    if isinstance(input, LazyActivation):
        n = input._qt
        if (n.signed and n.flat and not layer.training and _no_autograd(layer) and layer.weight.is_cuda
                and layer.weight.dtype == torch.float32 and n.shape[1] == layer.in_features and layer._eval_on_grid()):
            act = n.force()
            if act is not None:
                from .functions import _fused
                if kind == "xnor":
                    return _dense_result(layer, _fused.packed_xnor_linear(layer, act, hwc=n.chw))
                return _dense_result(layer, _fused.packed_linear(layer, act, kind, hwc=n.chw))
        input = n.materialise()
    if type(input) in lazy_train._DEFERRED:
        input = lazy_train.resolve(input)
    return _dense_result(layer, lazy_train.wrap(layer, layer._forward_impl(input)))
