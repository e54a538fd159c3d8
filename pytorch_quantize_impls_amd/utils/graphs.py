"""hipGraph replay of a fixed-shape inference forward (launch-bound small batches).

At batch 256 the fused AlexNet-Bin forward is GPU-bound (1.19 ms, ~25 launches); at batch 1-8 the same 25 launches
cost more host + dispatch time than device time: eager 0.37 ms vs 0.27 ms replayed (tools/bench_graph_small_batch.py).
The C-ABI kernels are launched on torch's current stream, so they are captured like any torch op; the only
requirement is a forward without host synchronisation (no un-tagged +-1 detection: give first layers
``binary_input = False``; no DoReFa code-overflow check inside the captured region).
"""
import torch
from torch.utils._pytree import tree_map_only

from .. import lazy


class GraphedModule(torch.nn.Module):
    """``GraphedModule(module, example_input)(x)``: copies x into the captured input buffer, replays the graph and
    returns the captured output tensor (overwritten by the next call: clone it to keep it).

    ``static_input`` is that buffer: a caller that produces its batches there (the target of its host-to-device copy, or of the
    previous pipeline stage) passes it back — ``g(g.static_input)`` — and the replay starts without the device-to-device copy
    (154 MB / 53 us for a 256 x 3 x 224 x 224 batch)."""

    def __init__(self, module: torch.nn.Module, example_input: torch.Tensor, warmup: int = 3, call=None):
        super().__init__()
        if not example_input.is_cuda:
            raise TypeError("graph capture needs a device input")
        self.module = module
        if call is not None:                        # (implicit graphs, utils/implicit.py: the root's ORIGINAL forward — calling the
            object.__setattr__(self, "_call", call)  #  module itself would re-enter the wrapper installed on it)
            module = call
        self._static_in = example_input.clone()
        self._stream = torch.cuda.Stream(device=example_input.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.stream(self._stream):
            for _ in range(warmup):                 # allocations, weight packing and caches settle before capture
                module(self._static_in)
            torch.cuda.synchronize(example_input.device)
            with torch.cuda.graph(self._graph, stream=self._stream):
                out = module(self._static_in)
                # a module that ends in a quantised conv chain returns a deferred activation (lazy.py): its kernels
                # have to be part of the graph, so it is turned into its value inside the captured region
                #: the forward's own result was (or contained) a deferred activation: a root that hands such a result to a consumer
                #: outside itself is not wrapped implicitly (utils/implicit.py) — the consumer would lose the fused hand-over
                self.lazy_output = _contains_lazy(out)
                self._static_out = tree_map_only(lazy.LazyActivation, lambda t: t.value(), out)
        torch.cuda.synchronize(example_input.device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self._static_in.shape or x.dtype != self._static_in.dtype:
            raise ValueError(f"captured for input {tuple(self._static_in.shape)} {self._static_in.dtype}, "
                             f"got {tuple(x.shape)} {x.dtype}")
        if x is not self._static_in:
            self._static_in.copy_(x)
        self._graph.replay()
        return self._static_out

    @property
    def static_input(self) -> torch.Tensor:
        """The captured input buffer (write the next batch here and call ``self(self.static_input)``: no copy)."""
        return self._static_in


def _contains_lazy(out) -> bool:
    if isinstance(out, lazy.LazyActivation):
        return True
    if isinstance(out, (tuple, list)):
        return any(_contains_lazy(o) for o in out)
    if isinstance(out, dict):
        return any(_contains_lazy(o) for o in out.values())
    return False


def graphed(module: torch.nn.Module, example_input: torch.Tensor, warmup: int = 3) -> GraphedModule:
    return GraphedModule(module, example_input, warmup)


class AutoGraphed(torch.nn.Module):
    """``AutoGraphed(model)``: the un-modified model, with its fixed-shape inference forwards replayed as hipGraphs.

    The module graphs of small-map nets are host-bound when run eagerly (DoReFa ResNet-18 at 32 x 32, batch 256: ~2.0 ms of
    Python and launch latency around 0.7 ms of device work; BinaryNet-AlexNet at batch 1: 0.69 ms against 0.13 ms — module
    machinery spread thin over ~150 module calls).  This wrapper keeps the eager path for everything that
    cannot be replayed — training mode, autograd enabled, CPU tensors, the first ``capture_after`` calls with a new input
    signature — and from then on copies the input into a captured buffer and replays the graph captured for that signature
    (shape, dtype, strides, device).  The captured graphs are dropped when a parameter or buffer of the model is written
    (version counters), e.g. after ``load_state_dict`` or a training epoch.  A forward that cannot be captured (it asks the device
    a question) is remembered as such and stays eager.  The returned tensor is a copy of the captured output unless
    ``clone_output=False`` (then it is overwritten by the next call with the same signature)."""

    def __init__(self, module: torch.nn.Module, capture_after: int = 1, max_graphs: int = 8, clone_output: bool = True, call=None,
                 keep_if_faster: float = 0.0):
        super().__init__()
        self.module = module
        # ``call``: what runs the eager forward (default: the module itself); ``keep_if_faster`` = f > 0: at capture time the replay
        # and the eager forward are both timed (the capture synchronises anyway) and the graph is only kept when it needs less than
        # f x the eager time — a device-bound forward gains nothing from a replay and its private memory pool is given back
        object.__setattr__(self, "_call", call if call is not None else module)
        self.keep_if_faster = float(keep_if_faster)
        self.not_faster = 0
        self.refuse_lazy_outputs = False
        self.capture_after, self.max_graphs, self.clone_output = int(capture_after), int(max_graphs), bool(clone_output)
        self._graphs, self._seen, self._state = {}, {}, None
        self.replays = self.eager_calls = 0
        #: captures that failed, by reason ("ExceptionType: message" -> count): such a signature stays eager — VISIBLY (a genuine
        #: bug inside a captured forward must not turn into a silent performance regression); ``strict=True`` re-raises instead
        self.capture_failures = {}
        self.strict = False

    def _state_sig(self):
        # version counter + storage of every parameter and buffer (optimizer steps, load_state_dict, copy_, .to()); writes through
        # ``.data`` are invisible to it, as they are to the layers' own plane caches: call reset() after one
        return tuple((t._version, t.data_ptr()) for t in self.module.parameters()) + \
            tuple((t._version, t.data_ptr()) for t in self.module.buffers())

    def reset(self):
        """Drop every captured graph (after a manual ``weight.data`` edit)."""
        self._graphs.clear()
        self._seen.clear()
        self._state = None

    def forward(self, x):
        # (a submodule left in train() — BatchNorm fine-tuning, stochastic quantisers — would be captured once and replayed with
        # frozen behaviour: every module has to be in eval mode, not just the root)
        ok = (isinstance(x, torch.Tensor) and x.is_cuda and not torch.is_grad_enabled()
              and not _any_training(self.module) and not torch.cuda.is_current_stream_capturing())
        if not ok:
            self.eager_calls += 1
            return self._call(x)
        sig = self._state_sig()
        if sig != self._state:                          # weights / running statistics were written: the captured kernels
            self._graphs.clear()                        # baked the old packed planes in
            self._seen.clear()
            self._state = sig
        key = (tuple(x.shape), x.dtype, tuple(x.stride()), x.device)
        g = self._graphs.get(key, False)
        if g is False:
            n = self._seen.get(key, 0) + 1
            self._seen[key] = n
            if n <= self.capture_after or len(self._graphs) >= self.max_graphs:
                self.eager_calls += 1
                return self._call(x)
            try:
                g = GraphedModule(self.module, x, warmup=1, call=None if self._call is self.module else self._call)
                if self.refuse_lazy_outputs and g.lazy_output:
                    g = None
                    self.capture_failures["deferred output"] = self.capture_failures.get("deferred output", 0) + 1
                elif self.keep_if_faster > 0.0 and not self._replay_pays(g, x):
                    g = None
                    self.not_faster += 1
            except Exception as exc:                    # e.g. a host synchronisation inside the forward: stay eager for this signature,
                torch.cuda.synchronize(x.device)        # and say so (capture_failures; bench.py prints it)
                if self.strict:
                    raise
                reason = f"{type(exc).__name__}: {str(exc).splitlines()[0][:160] if str(exc) else ''}"
                self.capture_failures[reason] = self.capture_failures.get(reason, 0) + 1
                g = None
            self._graphs[key] = g
            self._state = self._state_sig()             # the warm-up forwards may have built caches, never written parameters
        if g is None:
            self.eager_calls += 1
            return self._call(x)
        self.replays += 1
        out = g(x)
        return tree_map_only(torch.Tensor, torch.clone, out) if self.clone_output else out


def _replay_pays(self, g, x, n: int = 3) -> bool:
    """Wall time of n replays against n eager forwards, each bracketed by a device synchronise (capture time only)."""
    import time
    dev = x.device

    def timed(fn):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    with torch.no_grad():
        t_replay = timed(lambda: g(x))
        t_eager = timed(lambda: tree_map_only(lazy.LazyActivation, lambda t: t.value(), self._call(x)))
    return t_replay < self.keep_if_faster * t_eager


AutoGraphed._replay_pays = _replay_pays


def _any_training(module: torch.nn.Module) -> bool:
    """A module of the tree is in training mode.  The quantised layers' own children (``weight_op``: the weight quantiser as a
    module) are not looked at: the reference's train() / eval() of those layers sets the flag of the layer alone
    (layers/binary_layers.py:30-40 — kept, layers/common.py), so they read ``training`` for ever and never act on it."""
    from ..layers.common import EvalSwapMixin
    if module.training:
        return True
    if isinstance(module, EvalSwapMixin):
        return False
    return any(_any_training(c) for c in module.children())


def auto_graphed(module: torch.nn.Module, **kw) -> AutoGraphed:
    return AutoGraphed(module, **kw)


class GraphedTrainStep:
    """One training step — forward, loss, backward — of a fixed-shape batch captured once as a hipGraph and replayed.

    The training steps of small-map nets are launch-bound (DoReFa ResNet-18 at 32 x 32, batch 256: ~900 kernels of 5-40 us per
    step, 12.5 ms eager, 10.3 ms replayed; DESIGN.md section 1 "Training path").  This backend's autograd Functions launch on
    torch's current stream and allocate through torch's caching allocator, so a whole step is capturable — provided nothing in it
    asks the device a question: the +-1 / int8-range verdicts of un-tagged activations must come from memory
    (``functions._fused.DETECT_MODE = "remember"``: a wrong remembered verdict turns the output into NaN, never into a plausible
    number), which this class switches on for the warm-up, the capture and nothing else.

        step = GraphedTrainStep(model, lambda out, target: F.nll_loss(F.log_softmax(out, 1), target), x0, t0)
        for x, t in loader:            # same shapes / dtypes as x0, t0
            loss = step(x, t)          # gradients are in p.grad (overwritten by the next call)
            opt.step()                 # the optimizer stays outside the graph

    Gradients are zeroed inside the graph (``zero_grad(set_to_none=False)``), so every replay leaves exactly this batch's
    gradients behind.  BatchNorm running statistics and ``num_batches_tracked`` advance on every replay, like eager steps."""

    def __init__(self, model: torch.nn.Module, loss_fn, example_input: torch.Tensor, example_target: torch.Tensor,
                 warmup: int = 3):
        from ..functions import _fused
        if not example_input.is_cuda:
            raise TypeError("graph capture needs device tensors")
        self.model, self.loss_fn = model, loss_fn
        self._x, self._t = example_input.clone(), example_target.clone()
        self._stream = torch.cuda.Stream(device=example_input.device)
        self._graph = torch.cuda.CUDAGraph()
        for p in model.parameters():
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)          # static gradient buffers: the captured backward accumulates into them
        with _fused.detect_scope("remember"):          # thread-local; the Functions carry it into their backward
            with torch.cuda.stream(self._stream):
                for _ in range(max(1, warmup)):       # verdicts asked (and remembered), allocations and weight caches settle
                    self._one()
                torch.cuda.synchronize(example_input.device)
                with torch.cuda.graph(self._graph, stream=self._stream):
                    self._loss = self._one()
            torch.cuda.synchronize(example_input.device)

    def _one(self):
        self.model.zero_grad(set_to_none=False)
        loss = self.loss_fn(self.model(self._x), self._t)
        loss.backward()
        return loss.detach()

    def __call__(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if x.shape != self._x.shape or x.dtype != self._x.dtype or target.shape != self._t.shape or target.dtype != self._t.dtype:
            raise ValueError(f"captured for input {tuple(self._x.shape)} {self._x.dtype} / target {tuple(self._t.shape)} "
                             f"{self._t.dtype}, got {tuple(x.shape)} {x.dtype} / {tuple(target.shape)} {target.dtype}")
        self._x.copy_(x)
        self._t.copy_(target)
        self._graph.replay()
        return self._loss
