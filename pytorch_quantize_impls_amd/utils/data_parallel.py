"""Data-parallel training of QuantTorch layers, one process per GPU.

The reference trains on one GPU per process (its multi-GPU use is independent optuna workers, utils/jobs/compress.py:88-115);
what data-parallel training of its layers needs on top is (a) the mean of the STE-masked gradients over the ranks before
``optimizer.step()`` and (b) the clamp of the fp32 master weights after it (benchmark/BinaryNet/mnist.py:42-44,
``model.clamp()``).  Both are here, sized for MI355X: gradients travel as flat fp32 buckets over ``torch.distributed``
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring
all-reduce is per-link bound and latency matters more than on a switch: few, large buckets (default 64 MiB — AlexNet-Bin's
292 MB of gradients leave as 5 launches) each started as soon as its last gradient exists, i.e. while backward is still
producing the earlier layers' gradients.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

__all__ = ["GradientSynchronizer", "clamp_weights_", "broadcast_parameters"]


class _Bucket:
    __slots__ = ("params", "offsets", "numel", "flat", "pending", "work")

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        self.flat: Optional[torch.Tensor] = None
        self.pending = len(params)
        self.work = None


class GradientSynchronizer:
    """Averages ``.grad`` of ``params`` over the process group, overlapped with backward.

        sync = GradientSynchronizer(model.parameters())
        loss.backward()          # buckets leave as their last gradient is accumulated
        sync.wait()              # every .grad is now the mean over the ranks
        optimizer.step(); clamp_weights_(model)

    Parameters are bucketed in REVERSE registration order (the order backward produces gradients in), ``bucket_bytes`` per
    bucket.  A parameter that received no gradient in a step is sent as zeros (every rank must issue the same collectives).
    With a single process (or no initialised process group) ``wait`` is a no-op.  ``overlap=False``: no hooks — every
    bucket is launched by ``wait()`` itself, from the calling thread, after backward (same result, nothing hidden).

    Contract: every rank constructs it over the SAME parameter list, and every ``backward()`` is followed by one
    ``wait()`` before the next backward (no gradient accumulation across backwards: accumulate locally with
    ``overlap=False`` and call ``wait()`` once).  Collectives are issued strictly in bucket order on every rank — a
    bucket whose gradients are complete waits for the buckets before it, and whatever has not started when ``wait()`` is
    called (parameters without a gradient this step) is launched there, in order — so ranks that differ in WHICH
    parameters received gradients still issue the same sequence of all-reduces.  A second backward before ``wait()``
    raises instead of silently averaging a stale gradient."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, group=None, overlap: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        plist = [p for p in params if p.requires_grad]
        self.buckets: List[_Bucket] = []
        self._bucket_of = {}
        cur, cur_bytes, key = [], 0, None
        for p in reversed(plist):
            k = (p.device, p.dtype)
            nbytes = p.numel() * p.element_size()
            if cur and (k != key or cur_bytes + nbytes > bucket_bytes):
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            key = k
        if cur:
            self.buckets.append(_Bucket(cur))
        self._hooks = []
        self._next = 0                      # index of the first bucket whose all-reduce has not been issued
        if self.world > 1 and overlap:
            for b in self.buckets:
                for p in b.params:
                    self._bucket_of[p] = b
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _launch(self, b: _Bucket):
        p0 = b.params[0]
        if b.flat is None:
            b.flat = torch.empty((b.numel,), dtype=p0.dtype, device=p0.device)
        for p, off in zip(b.params, b.offsets):
            dst = b.flat[off:off + p.numel()]
            if p.grad is None:
                dst.zero_()
            else:
                dst.copy_(p.grad.reshape(-1))
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        b = self._bucket_of[p]
        if b.work is not None or b.pending <= 0:
            raise RuntimeError("GradientSynchronizer: a gradient arrived for a bucket that is already reduced / complete — "
                               "a second backward() ran before wait() (for gradient accumulation use overlap=False)")
        b.pending -= 1
        # strictly in bucket order: start every complete bucket at the head of the line, stop at the first incomplete one
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def wait(self):
        """Blocks until every bucket is reduced and writes the means back into ``.grad``; re-arms for the next backward."""
        if self.world == 1:
            return
        for b in self.buckets[self._next:]:      # not started by the hooks (a parameter without a gradient, or overlap=False)
            self._launch(b)
        self._next = 0
        for b in self.buckets:
            b.work.wait()
            b.flat.div_(self.world)
            for p, off in zip(b.params, b.offsets):
                g = b.flat[off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            b.work, b.pending = None, len(b.params)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Identical replicas at start: parameters and buffers of ``module`` from rank ``src``."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)


def clamp_weights_(module: torch.nn.Module):
    """``model.clamp()`` of the reference's nets (models/.../clamp -> every quantised layer's ``clamp()``,
    layers/binary_layers.py:24-28, terner_layers.py, log_lin_layers.py): after ``optimizer.step()`` the fp32 master weights
    go back into the quantiser's range.  Calls ``clamp()`` of every layer (a module with a ``weight`` parameter) that defines one."""
    for m in module.modules():
        if isinstance(getattr(m, "weight", None), torch.nn.Parameter) and callable(getattr(m, "clamp", None)):
            m.clamp()
