"""Shared pieces of the STE function family (reference: QuantTorch/functions/common.py)."""
import torch

from .. import ops, lazy, lazy_train


def safeSign(tensor: torch.Tensor) -> torch.Tensor:
    """-1 where tensor < 0, +1 everywhere else (+0.0, -0.0 and NaN map to +1).

    Reference: QuantTorch/functions/common.py:4-7 (torch.sign followed by a masked write of the
    zeros).  HIP device tensors go through qt_binarize_f32; CPU tensors use the same predicate in
    torch.
    """
    if tensor.is_cuda and tensor.dtype == torch.float32:
        return ops.binarize(tensor)
    one = torch.ones((), dtype=tensor.dtype, device=tensor.device)
    return torch.where(tensor < 0, -one, one)


class QtFunction(torch.autograd.Function):
    """Base of every autograd.Function of this package.  ``Function.apply`` bypasses ``__torch_function__``, so a deferred
    activation (the storage-less stand-in of a recorded training-mode [pool] -> BatchNorm -> ... chain, lazy_train.py; a deferred
    inference chain, lazy.py) handed to ``apply`` would enter ``forward`` as a no-grad leaf and cut the graph to everything
    upstream.  Every ``apply`` therefore swaps deferred arguments for their ordinary tensors first; subclasses that can FUSE
    the chain instead (BinaryConnectDeterministic, nnDorefaQuant's Function) override ``apply`` and end here."""

    def __init_subclass__(cls, **kwargs):
        """A graph is differentiated under the thread-local switches it was built under: ``forward`` records the thread's
        ``ops.float_split`` / ``_fused.detect_scope`` / ``ops.scope`` / ``_fused.scope`` overrides on ctx, ``backward`` (run by the
        autograd engine's own thread, which sees no thread-local state of the caller) re-opens them.  Consequence: a
        scope opened ONLY around ``loss.backward()`` does not reach the backward kernels of a graph built outside it — open it around
        the forward, or set the module-level default (``ops.FLOAT_SPLIT``, ``_fused.DETECT_MODE``, ...), which every thread reads."""
        super().__init_subclass__(**kwargs)
        fwd, bwd = cls.__dict__.get("forward"), cls.__dict__.get("backward")
        if isinstance(fwd, staticmethod) and isinstance(bwd, staticmethod) and not getattr(fwd.__func__, "_qt_wrapped", False):
            f0, b0 = fwd.__func__, bwd.__func__

            def forward(ctx, *a, **k):
                from . import _fused
                ctx._qt_split, ctx._qt_detect = ops.float_split_override(), _fused.detect_mode_override()
                ctx._qt_scopes = (ops.scope_overrides(), _fused.scope_overrides())
                return f0(ctx, *a, **k)

            def backward(ctx, *g):
                split, detect = getattr(ctx, "_qt_split", None), getattr(ctx, "_qt_detect", None)
                so, sf = getattr(ctx, "_qt_scopes", (None, None))
                if split is None and detect is None and not so and not sf:
                    return b0(ctx, *g)
                from . import _fused
                with ops.float_split(split), _fused.detect_scope(detect), ops.scope(so), _fused.scope(sf):
                    return b0(ctx, *g)
            forward._qt_wrapped = backward._qt_wrapped = True
            forward.__doc__, backward.__doc__ = f0.__doc__, b0.__doc__
            cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)

    @classmethod
    def apply(cls, *args, **kwargs):
        args = lazy_train.resolve_args(args)
        for a in args:
            if isinstance(a, lazy.LazyActivation):
                args = tuple(b.value() if isinstance(b, lazy.LazyActivation) else b for b in args)
                break
        return super().apply(*args, **kwargs)


class _FunctionModule(torch.nn.Module):
    """nn.Module that applies an autograd.Function class (what front()/front2() hand out)."""

    def __init__(self, fn_class):
        super().__init__()
        self.core = fn_class

    def forward(self, x):
        if isinstance(x, lazy.LazyActivation):
            # a deferred conv chain (lazy.py): BinaryConnect(deterministic) is recorded, anything else gets the value
            out = lazy.sign(x) if getattr(self.core, "_qt_records_sign", False) else None
            bits = getattr(self.core, "_qt_quant_bits", None)
            if bits is not None:                   # nnDorefaQuant(k) on a deferred DorefaConv2d chain
                out = lazy.quant(x, bits)
            if out is not None:
                return out
            x = x.value()
        elif type(x) in lazy_train._DEFERRED:
            # training mode: the recorded [pool] -> BatchNorm -> ... chain runs as one autograd node (lazy_train.py)
            out = None
            if getattr(self.core, "_qt_records_sign", False):
                out = lazy_train.sign(x)
            elif getattr(self.core, "_qt_quant_bits", None) is not None:
                out = lazy_train.quant(x, self.core._qt_quant_bits)
            if out is not None:
                return out
            x = lazy_train.resolve(x)
        return self.core.apply(x)


def front(claaz):
    """Module proxy of an autograd.Function class (reference: functions/common.py:10-20)."""
    return _FunctionModule(claaz)


def front2(claaz):
    """Same proxy, keeping the Function on ``.core`` (reference: functions/common.py:23-31)."""
    return _FunctionModule(claaz)


def ste_mask(grad_output: torch.Tensor, saved_input: torch.Tensor, thr: float = ops.STE_THRESHOLD):
    """grad * 1[|x| <= thr]: the straight-through mask every Binary/Ternary op shares
    (functions/binary_connect.py:31-38, terner_connect.py:29-34)."""
    if grad_output.is_cuda and grad_output.dtype == torch.float32 and saved_input.dtype == torch.float32:
        return ops.ste_mask(grad_output, saved_input, thr)
    return torch.where(saved_input.abs() > thr, torch.zeros_like(grad_output), grad_output)
