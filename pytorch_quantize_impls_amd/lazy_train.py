"""TRAINING-mode counterpart of lazy.py: the modules BETWEEN two quantised layers of an un-modified graph run as the fused
training nodes of layers/fused.py, without the model being rewritten.

In training mode (autograd on, HIP device, fp32) a quantised conv / linear layer hands its output back as a ``TrainOut`` — the
real tensor (same storage, same autograd node) under a Tensor subclass.  The functional calls the reference's graphs make
on it are RECORDED instead of executed, as long as they follow one of the two grammars the fused nodes implement:

    [max_pool2d(k, s)] -> batch_norm(training) -> [hardtanh] -> [reshape (N, C*H*W)] -> BinaryConnect(deterministic)
            == layers.fused._TrainPoolBnSignFn        (models/Alexnet/Alexnet_Bin.py:13-17, benchmark/BinaryNet/MLPBin.py:42-44)
    batch_norm(training) -> [+ shortcut] -> [relu] -> nnDorefaQuant(k)
            == layers.fused._TrainBnActQuantFn        (models/Resnet/Resnet_bin.py:63-97)

and the chain executes as ONE autograd node when BinaryConnect / nnDorefaQuant is reached (functions/common.py's module proxy
and the Functions' ``apply`` ask ``sign`` / ``quant`` here).  The result, the gradients and the running-statistics update are
those of the explicit ``FusedTrainPoolBnSign`` / ``FusedTrainBnActQuant`` modules bit for bit: the same autograd.Function runs
with the same arguments.  A BatchNorm whose chain ends anywhere else (the shortcut branch's, a network's last BatchNorm) is
normalised by the same node without a quantiser; every other use of a recorded chain replays the recorded calls with torch's
own functions on the real tensor (``materialise``), so an op outside the grammar costs nothing but the deferral.

A ``TrainOut`` with nothing recorded IS the layer's output: C++ callers that bypass ``__torch_function__``
(``torch.autograd.grad`` on the network's last layer) see the ordinary tensor.  Only the short-lived stand-ins between the
BatchNorm call and its consumer (``TrainChain``) have no storage of their own.

``ENABLED = False`` (or ``with lazy_train.eager():``) switches the recording off: every module runs by itself.
"""
from __future__ import annotations

import collections
import contextlib
import threading

import torch
import torch.nn.functional as F
from torch.utils._pytree import tree_map_only

ENABLED = True
STATS = collections.Counter()


_tls = threading.local()


def enabled() -> bool:
    """Recording is on: the master switch ``ENABLED`` and no ``eager()`` block open in THIS thread."""
    return ENABLED and not getattr(_tls, "eager_depth", 0)


@contextlib.contextmanager
def eager():
    """Module-by-module training inside the block (torch / MIOpen pooling, BatchNorm, Hardtanh, add, ReLU kernels).  Thread-local,
    like lazy.eager(): two training threads of one process do not switch each other's recording off."""
    _tls.eager_depth = getattr(_tls, "eager_depth", 0) + 1
    try:
        yield
    finally:
        _tls.eager_depth -= 1


class _Step:
    """One recorded call.  Immutable; ``value`` caches the replayed tensor (a chain used twice replays once)."""
    __slots__ = ("parent", "op", "shape", "value", "pool", "bn", "ht", "add", "relu", "flat", "fused", "__weakref__")

    def __init__(self, parent, op, shape):
        self.parent, self.op, self.shape, self.value, self.fused = parent, op, tuple(int(v) for v in shape), None, False
        if parent is None:
            self.pool = self.bn = self.ht = self.add = self.flat = None
            self.relu = False
        else:
            self.pool, self.bn, self.ht, self.add, self.relu, self.flat = (parent.pool, parent.bn, parent.ht, parent.add,
                                                                           parent.relu, parent.flat)
            kind = op[0]
            if kind == "pool":
                self.pool = op
            elif kind == "bn":
                self.bn = op
            elif kind == "hardtanh":
                self.ht = op
            elif kind == "add":
                self.add = op
            elif kind == "relu":
                self.relu = True
            elif kind == "flat":
                self.flat = op

    @property
    def base(self):
        n = self
        while n.parent is not None:
            n = n.parent
        return n.value

    def _untouched(self) -> bool:
        """No step of this chain has been replayed or run as a fused node yet (the BatchNorm's running statistics have not been
        updated for this batch)."""
        n = self
        while n.parent is not None:
            if n.value is not None or n.fused:
                return False
            n = n.parent
        return True

    def _mark_fused(self):
        n = self
        while n.parent is not None:
            n.fused = True
            n = n.parent

    def materialise(self) -> torch.Tensor:
        if self.value is not None:
            return self.value
        h = self.parent.materialise()
        kind = self.op[0]
        if kind == "pool":
            h = F.max_pool2d(h, self.op[1], self.op[2])
        elif kind == "bn":
            h = _batch_norm(h, self)
        elif kind == "hardtanh":
            h = F.hardtanh(h, self.op[1], self.op[2])
        elif kind == "add":
            h = h + _resolve_operand(self.op[1])
        elif kind == "relu":
            h = torch.relu(h)
        elif kind == "flat":
            h = h.reshape(self.shape)
        STATS["replayed:" + kind] += 1
        self.value = h
        return h


def _resolve_operand(o):
    """A recorded residual operand: an immutable _Step (deferred when recorded) or an ordinary tensor."""
    return o.materialise() if type(o) is _Step else resolve(o)


def _batch_norm(h, step):
    """A recorded training-mode BatchNorm outside a fused chain: this backend's statistics / normalise / backward kernels
    (the quantiser-less form of _TrainBnActQuantFn — what TrainFused* graphs use for a shortcut branch's BatchNorm)."""
    _, rm, rv, w, b, momentum, eps = step.op
    if step.fused:
        # the fused node of another consumer has already updated the running statistics for this batch
        return F.batch_norm(h, None, None, w, b, True, momentum, eps)
    from .layers.fused import _TrainBnActQuantFn
    return _TrainBnActQuantFn.apply(h, None, w, b, rm, rv, eps, momentum, False, 0)


class TrainOut(torch.Tensor):
    """Output of a quantised layer in training mode: the real tensor, able to record the calls that follow (module docstring)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return _torch_function(func, types, args, kwargs)


class TrainChain(torch.Tensor):
    """Stand-in for the result of recorded calls that have not run (module docstring)."""

    @staticmethod
    def __new__(cls, step: _Step):
        base = step.base
        # requires_grad=True + the hook below: an autograd.Function this package cannot see (Function.apply bypasses
        # __torch_function__) takes the stand-in as a graph LEAF; its backward then reaches _foreign_function_grad and fails
        # loudly instead of silently cutting the graph to the BatchNorm and every layer upstream.
        t = torch.Tensor._make_wrapper_subclass(cls, step.shape, dtype=torch.float32, device=base.device, requires_grad=True)
        t._qt = step
        with torch._C.DisableTorchFunctionSubclass():
            t.register_hook(_foreign_function_grad)
        return t

    def value(self) -> torch.Tensor:
        return self._qt.materialise()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return _torch_function(func, types, args, kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        args, kwargs = tree_map_only(TrainChain, lambda t: t._qt.materialise(), (args, kwargs or {}))
        return func(*args, **kwargs)


_DEFERRED = (TrainOut, TrainChain)


def _foreign_function_grad(grad):
    raise RuntimeError(
        "pytorch_quantize_impls_amd.lazy_train: a gradient arrived at the stand-in of a deferred training-mode "
        "[pool] -> BatchNorm -> ... chain.  A torch.autograd.Function outside this package consumed the stand-in through "
        "Function.apply (which bypasses __torch_function__), so its backward cannot reach the BatchNorm.  Pass "
        "lazy_train.resolve(x) to that Function, derive it from functions.common.QtFunction, or run the step under "
        "`with lazy_train.eager():`.")


def resolve_args(args):
    """Function.apply arguments with every deferred training activation replaced by its ordinary tensor."""
    for a in args:
        if type(a) in _DEFERRED:
            return tuple(resolve(b) for b in args)
    return args


def _step_of(t) -> _Step:
    if type(t) is TrainChain:
        return t._qt
    with torch._C.DisableTorchFunctionSubclass():
        s = t.__dict__.get("_qt")
        if s is None:
            s = _Step(None, None, t.shape)
            s.value = t.as_subclass(torch.Tensor)
            t._qt = s
    return s


def resolve(x):
    """The ordinary tensor behind a deferred training activation (anything else unchanged)."""
    if type(x) is TrainChain:
        return x._qt.materialise()
    if type(x) is TrainOut:
        with torch._C.DisableTorchFunctionSubclass():
            return x.as_subclass(torch.Tensor)
    return x


def wrap(layer, out):
    """What a quantised layer returns in training mode: ``out`` itself unless the chain behind it can be fused."""
    if (enabled() and layer.training and torch.is_grad_enabled() and type(out) is torch.Tensor and out.is_cuda
            and out.dtype == torch.float32 and out.dim() in (2, 4) and out.requires_grad and out.numel() > 0):
        STATS["wrapped"] += 1
        return out.as_subclass(TrainOut)
    return out


_T = torch.Tensor
_META_FAST = {_T.dim: lambda t: len(t._qt.shape), _T.ndimension: lambda t: len(t._qt.shape),
              _T.ndim.__get__: lambda t: len(t._qt.shape),
              _T.size: lambda t, dim=None: torch.Size(t._qt.shape) if dim is None else t._qt.shape[dim],
              _T.shape.__get__: lambda t: torch.Size(t._qt.shape),
              _T.dtype.__get__: lambda t: torch.float32, _T.device.__get__: lambda t: t._qt.base.device,
              _T.is_cuda.__get__: lambda t: True, _T.is_floating_point: lambda t: True,
              _T.__len__: lambda t: t._qt.shape[0]}


def _torch_function(func, types, args, kwargs):
    handler = _HANDLERS.get(func)
    if handler is not None:
        out = handler(*args, **(kwargs or {}))
        if out is not NotImplemented:
            return out
    if TrainChain not in types:
        # only TrainOut operands: they ARE ordinary tensors — the op itself, with subclass dispatch off (plain results)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))
    kwargs = kwargs or {}
    if args and type(args[0]) is TrainChain:
        if not kwargs:
            fast = _META_FAST.get(func)
            if fast is not None:
                return fast(*args)
        name = getattr(func, "__name__", str(func))
        if _writes_in_place(name):
            # x.op_(...) outside the grammar on a stand-in: the replayed value, mutated; the stand-in moves on to the result
            self_ = args[0]
            # a private copy: the step's cached value may be the parent of chains recorded earlier (a = y + r; y.mul_(2)),
            # which must replay on the un-mutated tensor, as eager computed them before the mutation (lazy.py does the same)
            v = self_._qt.materialise().clone()
            rest, kwargs = tree_map_only(TrainChain, resolve, (tuple(args[1:]), kwargs))
            with torch._C.DisableTorchFunctionSubclass():
                res = func(v, *rest, **kwargs)
            done = _Step(None, None, res.shape)
            done.value = res
            self_._qt = done
            return self_
    args, kwargs = tree_map_only(TrainChain, resolve, (args, kwargs))
    with torch._C.DisableTorchFunctionSubclass():
        return func(*args, **kwargs)


def _writes_in_place(name: str) -> bool:
    return (name.endswith("_") and not name.endswith("__")) or name == "__setitem__" or (
        name.startswith("__i") and name.endswith("__") and name not in ("__int__", "__index__", "__invert__"))


def _square(v):
    if isinstance(v, int):
        return v
    if isinstance(v, (tuple, list)) and len(v) in (1, 2) and all(isinstance(a, int) for a in v) and len(set(v)) == 1:
        return v[0]
    return None


def _deferred(x) -> bool:
    return enabled() and type(x) in _DEFERRED


def _h_max_pool2d(input, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False, return_indices=False):
    if not _deferred(input) or return_indices or ceil_mode:
        return NotImplemented
    n = _step_of(input)
    k = _square(kernel_size)
    s = _square(stride) if stride not in (None, [], ()) else k
    if (n.parent is not None or k is None or s is None or _square(padding) != 0 or _square(dilation) != 1
            or len(n.shape) != 4):
        return NotImplemented
    N, C, H, W = n.shape
    if H < k or W < k:
        return NotImplemented
    return TrainChain(_Step(n, ("pool", k, s), (N, C, (H - k) // s + 1, (W - k) // s + 1)))


def _h_batch_norm(input, running_mean, running_var, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
    if not _deferred(input) or not training or running_mean is None or running_var is None or momentum is None:
        return NotImplemented
    n = _step_of(input)
    if n.bn is not None or n.flat is not None or len(n.shape) not in (2, 4) or (weight is None) != (bias is None):
        return NotImplemented
    C = n.shape[1]
    count = n.shape[0] * (n.shape[2] * n.shape[3] if len(n.shape) == 4 else 1)
    if C % 4 != 0 or count <= 1:
        return NotImplemented
    dev = n.base.device
    for t in (running_mean, running_var, weight, bias):
        if t is not None and (type(t) in _DEFERRED or t.device != dev or t.dtype != torch.float32 or t.dim() != 1
                              or t.numel() != C):
            return NotImplemented
    return TrainChain(_Step(n, ("bn", running_mean, running_var, weight, bias, float(momentum), float(eps)), n.shape))


def _h_hardtanh(input, min_val=-1.0, max_val=1.0, inplace=False):
    if type(input) is not TrainChain or not enabled():
        return NotImplemented
    n = input._qt
    if n.bn is None or n.ht is not None or n.add is not None or n.relu or n.flat is not None or not (min_val < 0 < max_val):
        return NotImplemented
    child = _Step(n, ("hardtanh", float(min_val), float(max_val)), n.shape)
    if inplace:
        input._qt = child
        return input
    return TrainChain(child)


def _h_hardtanh_(input, min_val=-1.0, max_val=1.0):
    return _h_hardtanh(input, min_val, max_val, inplace=True)


def _h_relu(input, inplace=False):
    if type(input) is not TrainChain or not enabled():
        return NotImplemented
    n = input._qt
    if n.bn is None or n.pool is not None or n.ht is not None or n.relu or n.flat is not None:
        return NotImplemented
    child = _Step(n, ("relu",), n.shape)
    if inplace:
        input._qt = child
        return input
    return TrainChain(child)


def _h_relu_(input):
    return _h_relu(input, inplace=True)


def _mainline(x) -> bool:
    if type(x) is not TrainChain:
        return False
    n = x._qt
    return (n.bn is not None and n.pool is None and n.ht is None and n.add is None and not n.relu and n.flat is None
            and n.value is None)


def _residual_ok(main: _Step, other) -> bool:
    if type(other) in _DEFERRED:
        return _step_of(other).shape == main.shape
    return (isinstance(other, torch.Tensor) and type(other) is torch.Tensor and other.device == main.base.device
            and other.dtype == torch.float32 and tuple(other.shape) == main.shape)


def _h_add(a, b, *, alpha=1, out=None, _inplace=False):
    if alpha != 1 or out is not None or not enabled():
        return NotImplemented
    if _mainline(a) and _residual_ok(a._qt, b):
        main, other = a, b
    elif not _inplace and _mainline(b) and _residual_ok(b._qt, a):
        main, other = b, a
    else:
        return NotImplemented
    # a deferred residual is captured as its CURRENT (immutable) step: a later shortcut.relu_() / shortcut += ... rebinds the
    # stand-in's _qt and must not change what this already-recorded add resolves to
    child = _Step(main._qt, ("add", _step_of(other) if type(other) in _DEFERRED else other), main._qt.shape)
    if _inplace:
        a._qt = child
        return a
    return TrainChain(child)


def _h_iadd(a, b, *, alpha=1):
    return _h_add(a, b, alpha=alpha, _inplace=True)


def _flat_ok(n: _Step, shape) -> bool:
    if len(n.shape) != 4 or n.flat is not None or n.bn is None or n.add is not None or n.relu:
        return False
    N, C, H, W = n.shape
    shape = tuple(shape)
    if len(shape) != 2 or not all(isinstance(v, int) for v in shape):
        return False
    a, b = shape
    if a == -1 and b == -1:
        return False
    if a == -1:
        a = N if b == C * H * W else -2
    if b == -1:
        b = C * H * W if a == N else -2
    return (a, b) == (N, C * H * W)


def _as_flat(input):
    n = input._qt
    N, C, H, W = n.shape
    return TrainChain(_Step(n, ("flat",), (N, C * H * W)))


def _h_reshape(input, *shape):
    if type(input) is not TrainChain or not enabled():
        return NotImplemented
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    return _as_flat(input) if _flat_ok(input._qt, shape) else NotImplemented


def _h_flatten(input, start_dim=0, end_dim=-1):
    if type(input) is not TrainChain or not enabled():
        return NotImplemented
    n = input._qt
    if len(n.shape) == 4 and start_dim == 1 and end_dim in (-1, 3) and _flat_ok(n, (n.shape[0], -1)):
        return _as_flat(input)
    return NotImplemented


_HANDLERS = {
    F.max_pool2d: _h_max_pool2d,
    F.batch_norm: _h_batch_norm,
    F.hardtanh: _h_hardtanh, F.hardtanh_: _h_hardtanh_,
    _T.reshape: _h_reshape, _T.view: _h_reshape, torch.reshape: _h_reshape,
    _T.flatten: _h_flatten, torch.flatten: _h_flatten,
    torch.relu: _h_relu, _T.relu: _h_relu, F.relu: _h_relu, torch.relu_: _h_relu_, _T.relu_: _h_relu_,
    torch.add: _h_add, _T.add: _h_add, _T.__add__: _h_add, _T.__radd__: _h_add,
    _T.add_: _h_iadd, _T.__iadd__: _h_iadd,
}


# ---- the consumers ---------------------------------------------------------------------------------------------------

def sign(x):
    """BinaryConnect(deterministic) of a deferred training activation: the fused node when the recorded chain is
    [pool] -> BatchNorm -> [Hardtanh] -> [flatten], else None (the caller binarises ``resolve(x)``)."""
    if type(x) is not TrainChain or not enabled():
        return None
    n = x._qt
    if n.bn is None or n.add is not None or n.relu or not n._untouched():
        return None
    from .layers.fused import _TrainPoolBnSignFn
    _, rm, rv, w, b, momentum, eps = n.bn
    k, s = (n.pool[1], n.pool[2]) if n.pool is not None else (1, 1)
    lo, hi = (n.ht[1], n.ht[2]) if n.ht is not None else (-float("inf"), float("inf"))
    out = _TrainPoolBnSignFn.apply(n.base, w, b, rm, rv, eps, momentum, k, s, lo, hi)
    n._mark_fused()
    STATS["fused:sign"] += 1
    return out.reshape(n.shape) if n.flat is not None else out


def quant(x, bit_width: int):
    """nnDorefaQuant(k) of a deferred training activation: the fused node when the recorded chain is
    BatchNorm -> [+ shortcut] -> [ReLU] and 2 <= k <= 8, else None."""
    if type(x) is not TrainChain or not enabled():
        return None
    n = x._qt
    if (n.bn is None or n.pool is not None or n.ht is not None or n.flat is not None or not 2 <= int(bit_width) <= 8
            or not n._untouched()):
        return None
    from .layers.fused import _TrainBnActQuantFn
    _, rm, rv, w, b, momentum, eps = n.bn
    res = _resolve_operand(n.add[1]) if n.add is not None else None
    out = _TrainBnActQuantFn.apply(n.base, res, w, b, rm, rv, eps, momentum, n.relu, int(bit_width))
    n._mark_fused()
    STATS["fused:quant"] += 1
    return out
