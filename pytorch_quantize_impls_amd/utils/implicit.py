"""Implicit hipGraphs: an eval-mode model with quantised layers on a HIP device replays its fixed-shape inference forward as a
hipGraph BY ITSELF from the third identical call on — no wrapper, opt-OUT.

Why: the module graphs of small-map nets are host-bound when run eagerly (DoReFa ResNet-18 W1A4 at 32 x 32, batch 256: 1.8 ms of
Python dispatch around 0.55 ms of device work — 140 k img/s eager against 450 k as a replay), and ``utils.auto_graphed(model)``
only helps callers who know about it.

How: every quantised layer of the package counts its own eval-mode / no-autograd forwards on a HIP device (one dictionary
look-up per call; nothing at all in training mode).  At a layer's ``CAPTURE_AFTER``-th such call the call stack is walked for the
frames of ``torch.nn.Module._call_impl`` (identified by code object): the OUTERMOST one's ``self`` is the root the user called.  (A
process-wide module hook would say the same, but costs 3.6 us on EVERY module call of the process — measured — because it takes
``_call_impl`` off its fast path.)  When that root

  * has every module of its tree in eval mode, autograd is off,
  * was called with ONE positional fp32 device tensor and no keyword argument, outside any stream capture,
  * carries no forward / backward hooks anywhere in its tree (user instrumentation must keep seeing every call), and is not a
    ``torch.nn.parallel`` / graph wrapper,

the root gets an INSTANCE attribute ``forward`` — an
``_ImplicitForward`` around its original forward — which from then on is ``utils.AutoGraphed`` in all but name: captured per input
signature (shape, dtype, strides, device), dropped when a parameter / buffer of the tree is written (version counters), eager for
anything that cannot be replayed (training mode, autograd, CPU tensors, a forward that synchronises), outputs returned as copies.
Two safety nets on top of the opt-in wrapper: the captured replay must reproduce the eager result of the same input BIT FOR BIT
(a forward whose value depends on anything but its input and the tree's tensors is left eager), and the replay must be faster
than the eager forward (timed once at capture, which synchronises anyway: a device-bound forward keeps its eager path and the
graph's private memory pool is released).

What a replay cannot preserve: Python side effects of a user-defined ``forward`` (counters, prints, appended lists) happen in
the eager calls only.  Opt out: ``QT_AUTO_GRAPH=0`` in the environment, ``utils.implicit_graphs(False)`` (process-wide; also a
context manager), ``utils.implicit_graphs_off(model)`` for one model (sticky: the model is never wrapped again).
``utils.implicit_graph_stats(model)`` reports what happened.

Deep copies and pickles of a wrapped model carry a fresh, empty wrapper bound to the copy (no graph is ever copied)."""
import contextlib
import os
import threading
import weakref

import torch

from .. import lazy as _lazy
from .graphs import AutoGraphed

CAPTURE_AFTER = 2            # eager calls with one signature before the capture ("after the second identical call")
KEEP_IF_FASTER = 0.9         # a replay has to need < 0.9 x the eager wall time to be kept

_enabled = os.environ.get("QT_AUTO_GRAPH", "1") not in ("0", "off", "false", "no")
_tls = threading.local()
_PENDING = weakref.WeakKeyDictionary()     # root -> why it is not wrapped (yet)
_OPTED_OUT = weakref.WeakSet()


def enabled() -> bool:
    return _enabled and getattr(_tls, "off", 0) == 0


class implicit_graphs:
    """``implicit_graphs(False)`` switches the default off process-wide (``True``: on again); used as a context manager it
    restores the previous setting on exit."""

    def __init__(self, on: bool = True):
        global _enabled
        self._prev = _enabled
        _enabled = bool(on)
        if on:
            install()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        global _enabled
        _enabled = self._prev
        return False


@contextlib.contextmanager
def _thread_off():
    _tls.off = getattr(_tls, "off", 0) + 1
    try:
        yield
    finally:
        _tls.off -= 1


def implicit_graphs_off(model: torch.nn.Module) -> torch.nn.Module:
    """Never replay this model implicitly (removes an installed wrapper); returns the model."""
    fw = model.__dict__.get("forward")
    if isinstance(fw, _ImplicitForward):
        del model.__dict__["forward"]
    _OPTED_OUT.add(model)
    _PENDING.pop(model, None)
    return model


def implicit_graph_stats(model: torch.nn.Module) -> dict:
    fw = model.__dict__.get("forward")
    if isinstance(fw, _ImplicitForward):
        e = fw.engine
        return {"wrapped": True, "replays": e.replays, "eager_calls": e.eager_calls, "graphs": sum(g is not None for g in e._graphs.values()),
                "capture_failures": dict(e.capture_failures), "not_faster": e.not_faster, "value_mismatch": fw.value_mismatch,
                "eager_signatures": len(fw.eager_keys)}
    return {"wrapped": False, "opted_out": model in _OPTED_OUT, "why": _PENDING.get(model)}


def _has_quantised_layer(module) -> bool:
    from ..layers.common import QLayer
    return any(isinstance(m, QLayer) for m in module.modules())


def _hooked(module) -> bool:
    for m in module.modules():
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
            return True
    return False


def _signature(args, kwargs):
    if kwargs or len(args) != 1:
        return None
    x = args[0]
    if not (type(x) is torch.Tensor and x.is_cuda and x.dtype == torch.float32 and x.numel() > 0):
        return None
    return (tuple(x.shape), x.dtype, tuple(x.stride()), x.device)


class _ImplicitForward:
    """The instance-level ``forward`` of a wrapped root: AutoGraphed over the root's ORIGINAL forward."""

    def __init__(self, module: torch.nn.Module):
        self.module_ref = weakref.ref(module)
        self.original = type(module).forward.__get__(module, type(module))          # the class's forward, bound
        self.engine = AutoGraphed(module, capture_after=0, max_graphs=4, clone_output=True, call=self._eager,
                                  keep_if_faster=KEEP_IF_FASTER)
        self.engine.refuse_lazy_outputs = True
        # the engine must not register `module` as a child of itself as an attribute cycle that state_dict() walks: it is an
        # nn.Module that is never called as one and never appears in the root's tree (held here, not on the root's _modules)
        self.value_mismatch = 0
        self.busy = threading.Lock()
        self._checked = set()
        self.eager_keys = set()

    def _eager(self, x):
        with _thread_off():
            return self.original(x)

    def __call__(self, *args, **kwargs):
        # (inside `with lazy.eager():` the caller asks for the module-by-module execution: it gets it)
        if kwargs or len(args) != 1 or not enabled() or not _lazy.enabled() or not self.busy.acquire(blocking=False):
            with _thread_off():
                return self.original(*args, **kwargs)
        try:
            x = args[0]
            eng = self.engine
            key = (tuple(x.shape), x.dtype, tuple(x.stride()), x.device) if type(x) is torch.Tensor else None
            if key in self.eager_keys:
                # this signature was measured (replay not faster: a device-bound forward) or could not be captured: the original
                # forward, without the per-call state signature / training-mode walk of the engine (a 16-launch AlexNet forward
                # at batch 256 is close enough to host-bound for those 60 us to show: 313 k -> 294 k img/s)
                return self.original(x)
            before, state_before = eng.replays, eng._state
            out = eng(x)
            if eng._state is not state_before:
                self._checked.clear()                               # graphs were dropped and / or a new one captured: check again
            if key is not None and eng._graphs.get(key, False) is None:
                self.eager_keys.add(key)
            if eng.replays != before:
                if key not in self._checked:
                    # first replay of this capture: it must reproduce the eager value of the same input bit for bit
                    self._checked.add(key)
                    ref = self._eager(x)
                    if not _same(out, ref):
                        self.value_mismatch += 1
                        eng._graphs[key] = None                     # this signature stays eager
                        self.eager_keys.add(key)
                        return ref
            return out
        finally:
            self.busy.release()

    # a copy / pickle of the model gets an empty wrapper bound to the copy
    def __deepcopy__(self, memo):
        m = self.module_ref()
        new = memo.get(id(m)) if m is not None else None
        return _ImplicitForward(new) if isinstance(new, torch.nn.Module) else _Unbound()

    def __reduce__(self):
        return (_rebuild, (self.module_ref(),))


class _Unbound:
    """What a wrapper turns into when it is copied without its module: the class's forward takes over at the first call."""

    def __call__(self, *a, **k):
        raise RuntimeError("an implicit-graph wrapper was copied without its module; delete the instance attribute 'forward'")


def _rebuild(module):
    return _ImplicitForward(module) if isinstance(module, torch.nn.Module) and hasattr(type(module), "forward") else _Unbound()


def _same(a, b) -> bool:
    from .. import lazy
    if isinstance(a, lazy.LazyActivation):
        a = a.value()
    if isinstance(b, lazy.LazyActivation):
        b = b.value()
    if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
        return a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a, b))
    if isinstance(a, (tuple, list)) and isinstance(b, (tuple, list)) and len(a) == len(b):
        return all(_same(p, q) for p, q in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict) and a.keys() == b.keys():
        return all(_same(a[k], b[k]) for k in a)
    return type(a) is type(b) and not isinstance(a, torch.Tensor) and a == b


# ---- the trigger: called by the quantised layers' forwards (lazy.py) -----------------------------------------------------------------

_CALL_IMPL_CODE = torch.nn.Module._call_impl.__code__


def note(layer) -> None:
    """An eval-mode, no-autograd forward of a quantised layer on a device tensor is starting (the caller checked that much)."""
    if not _enabled or getattr(_tls, "off", 0):
        return
    d = layer.__dict__
    n = d.get("_qt_ig_n", 0)
    if n < 0:                                       # this layer's root is resolved (wrapped, or refused for good)
        return
    n += 1
    if n < CAPTURE_AFTER:
        d["_qt_ig_n"] = n
        return
    d["_qt_ig_n"] = _resolve(layer)


def _find_root():
    """(root module, its args, its kwargs) of the outermost torch.nn.Module._call_impl frame on this thread's stack, or None."""
    import sys
    f = sys._getframe(0)
    found = None
    while f is not None:
        if f.f_code is _CALL_IMPL_CODE:
            found = f
        f = f.f_back
    if found is None:
        return None
    loc = found.f_locals
    return loc.get("self"), loc.get("args", ()), loc.get("kwargs", {})


def _resolve(layer) -> int:
    """Wrap the root of the forward this layer is running in, if it qualifies.  Returns the layer's new counter: -1 = resolved (do
    not ask again), 0 = not now (a transient reason: ask again after CAPTURE_AFTER more calls)."""
    from .graphs import _any_training
    hit = _find_root()
    if hit is None:
        return 0
    root, args, kwargs = hit
    if not isinstance(root, torch.nn.Module):
        return 0
    if root is layer:
        return -1          # the layer itself was called: nothing host-bound to replay, and its deferred result keeps its type
    if "forward" in root.__dict__ or root in _OPTED_OUT:
        return -1
    if isinstance(root, (torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel, AutoGraphed)) \
            or type(root).__module__.startswith("torch.nn.parallel") or type(root).__name__ in ("GraphedModule",):
        _PENDING[root] = "refused: a parallel / graph wrapper"
        return -1
    if torch.cuda.is_current_stream_capturing():
        return 0
    if _signature(args, kwargs) is None:
        _PENDING[root] = "not now: the root was not called with one positional fp32 device tensor"
        return 0
    if _any_training(root):
        _PENDING[root] = "not now: a module of the tree is in training mode"
        return 0
    if _hooked(root):
        _PENDING[root] = "not now: hooks registered in the tree"
        return 0
    _PENDING.pop(root, None)
    object.__setattr__(root, "forward", _ImplicitForward(root))
    return -1


def install():
    """Kept for callers of the first design (process-wide hooks): nothing to install — the layers call ``note`` themselves."""
    return None
