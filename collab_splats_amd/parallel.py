"""Multi-GPU: one process per GPU, camera views sharded across ranks.

The reference has no distributed code at all; north_star adds exactly one
pattern: every rank holds the same N Gaussians, renders different views, and the six parameter
gradients (59 floats = 236 B per Gaussian at SH degree 3) are summed with ONE all-reduce over
RCCL/xGMI.  Independent views (BASELINE config 4) need no collective.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist

GRAD_KEYS = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; 'nccl' IS RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("MISPLAT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_views(n_views: int, rank: int, world: int) -> List[int]:
    """Views owned by ``rank``: contiguous blocks, remainder to the low ranks."""
    base, rem = divmod(n_views, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


class GradientBuckets:
    """The six parameter gradients of the shared Gaussians in ONE preallocated flat fp32 buffer (59 floats = 236 B per
    Gaussian at SH degree 3), reduced over the ranks as TWO buckets:

      colour bucket    features_dc / features_rest (or ``sh``): 192 of the 236 B.  The SH backward kernel writes it
                       STRAIGHT into the buffer (``ops.GRAD_SINK``), and its ``all_reduce(async_op=True)`` is launched as
                       soon as that kernel has been enqueued -- it travels over xGMI while the projection backward and
                       the activation backward still run;
      geometry bucket  means, quats, scales, opacities: launched from ``allreduce()`` after ``backward()``.  means / quats
                       come out of the projection backward in place; gradients that autograd produced elsewhere (the
                       log-scale / logit activations of the caller) are copied in (16 B per Gaussian).

    After ``allreduce()`` every ``p.grad`` IS its slice of the buffer: no flatten / unflatten copies (1.18 GB each way
    at 5 M Gaussians in the first version).  One large collective per bucket suits the point-to-point xGMI links (7 x
    ~153 GB/s per GPU): RCCL's direct reduce-scatter + all-gather moves 2 x 7/8 of the bucket per rank over 7 links in
    parallel.  ``NCCL_DEBUG=INFO`` (stderr) shows the algorithm / protocol RCCL picked."""

    def __init__(self, params: Sequence[torch.Tensor], geometry: Sequence[int] | None = None,
                 colour: Sequence[int] | None = None):
        self.params = list(params)
        if colour is None:                              # by convention: the SH tensors are the big trailing ones
            colour = [i for i, p in enumerate(self.params) if p.dim() == 3 or (p.dim() == 2 and p.shape[-1] == 3 and i >= 4)]
        if geometry is None:
            geometry = [i for i in range(len(self.params)) if i not in colour]
        self.colour, self.geometry = list(colour), list(geometry)
        order = self.colour + self.geometry
        total = sum(self.params[i].numel() for i in order)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views: List[torch.Tensor] = [None] * len(self.params)
        o = 0
        for i in order:
            n = self.params[i].numel()
            self.views[i] = self.flat[o:o + n].view_as(self.params[i])
            o += n
        self.n_colour = sum(self.params[i].numel() for i in self.colour)
        self._colour_work = None
        self._colour_launched = False
        self._by_ptr = {}

    # -- the step
    def attach(self) -> None:
        """Call before ``backward()``: gradients start from None and the backward kernels are pointed at the buffer."""
        from . import ops
        for p in self.params:
            p.grad = None
        self._by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params)}
        self._colour_work, self._colour_launched = None, False
        ops.GRAD_SINK = self

    def sink(self, inp: torch.Tensor):
        """Output buffer for the gradient of ``inp`` if it is one of the parameters themselves (else None).  A FRESH
        view object every time: autograd adopts an incoming gradient without a copy only if nobody else holds it."""
        i = self._by_ptr.get(inp.data_ptr())
        if i is None or self.views[i].shape != inp.shape:
            return None
        return self.views[i].view(inp.shape)

    def colour_ready(self) -> None:
        """Called by the backward right after the colour kernel has been enqueued."""
        if self._colour_launched or not self.colour:
            return
        self._colour_launched = True
        if all(self._by_ptr.get(self.params[i].data_ptr()) == i for i in self.colour) and _world() > 1:
            self._colour_work = self._launch(self.flat[:self.n_colour])

    def allreduce(self, average: bool = False):
        """After ``backward()``: reduce what is still pending, make every ``p.grad`` its slice of the buffer.  Returns a
        (start, end) pair of device events around the collectives on a GPU, else None."""
        from . import ops
        ops.GRAD_SINK = None
        ev = None
        if self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        pending_colour = self._colour_work is None
        for i, p in enumerate(self.params):
            if pending_colour or i in self.geometry:
                g = p.grad
                if g is None:
                    self.views[i].zero_()
                elif g.data_ptr() != self.views[i].data_ptr():
                    self.views[i].copy_(g)               # produced by autograd outside the rasterizer (activations)
        works = []
        if _world() > 1:
            if pending_colour:
                works.append(self._launch(self.flat))      # nothing was overlapped: one collective for everything
            else:
                works.append(self._launch(self.flat[self.n_colour:]))
                works.append(self._colour_work)
        for w in works:
            self._finish(w)
        if average and _world() > 1:
            self.flat /= _world()
        for i, p in enumerate(self.params):
            p.grad = self.views[i].view(p.shape)
        self._colour_work = None
        if ev is not None:
            ev[1].record()
        return ev

    # -- collectives
    def _launch(self, t: torch.Tensor):
        if dist.get_backend() == "gloo" and t.is_cuda:       # CPU rehearsal of the multi-rank path on a GPU box
            host = t.cpu()
            return (dist.all_reduce(host, op=dist.ReduceOp.SUM, async_op=True), host, t)
        return (dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), None, t)

    @staticmethod
    def _finish(w) -> None:
        work, host, t = w
        work.wait()
        if host is not None:
            t.copy_(host)


def _world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def flatten_grads(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """One contiguous fp32 bucket (kept for callers that hold ordinary ``.grad`` tensors; GradientBuckets avoids the copies)."""
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])


def unflatten_into_grads(flat: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    o = 0
    for p in params:
        n = p.numel()
        p.grad = flat[o:o + n].view_as(p).clone() if p.grad is None else p.grad.copy_(flat[o:o + n].view_as(p))
        o += n


def _reduce_device(t: torch.Tensor) -> torch.Tensor:
    """gloo (CPU rehearsal of the multi-rank path) cannot reduce device tensors: stage through host."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu()
    return t


def allreduce_gradients(params: Sequence[torch.Tensor], average: bool = False) -> torch.Tensor:
    """Sum (or mean) the gradients of the shared Gaussians over all ranks, in place (simple form: flatten, one
    all-reduce, copy back; the training path uses GradientBuckets)."""
    flat = flatten_grads(params)
    if dist.is_initialized() and dist.get_world_size() > 1:
        buf = _reduce_device(flat)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if buf is not flat:
            flat.copy_(buf)
        if average:
            flat /= dist.get_world_size()
    unflatten_into_grads(flat, params)
    return flat


def max_over_ranks(value: float, device: torch.device) -> float:
    """bench.py's max-over-ranks of the timed region."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
