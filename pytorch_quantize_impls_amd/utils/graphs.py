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
    returns the captured output tensor (overwritten by the next call: clone it to keep it)."""

    def __init__(self, module: torch.nn.Module, example_input: torch.Tensor, warmup: int = 3):
        super().__init__()
        if not example_input.is_cuda:
            raise TypeError("graph capture needs a device input")
        self.module = module
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
                self._static_out = tree_map_only(lazy.LazyActivation, lambda t: t.value(), out)
        torch.cuda.synchronize(example_input.device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self._static_in.shape or x.dtype != self._static_in.dtype:
            raise ValueError(f"captured for input {tuple(self._static_in.shape)} {self._static_in.dtype}, "
                             f"got {tuple(x.shape)} {x.dtype}")
        self._static_in.copy_(x)
        self._graph.replay()
        return self._static_out


def graphed(module: torch.nn.Module, example_input: torch.Tensor, warmup: int = 3) -> GraphedModule:
    return GraphedModule(module, example_input, warmup)
