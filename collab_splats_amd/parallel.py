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
    Gaussian at SH degree 3), reduced over the ranks with RCCL.

    The slices of the buffer ARE the gradient tensors: ``ops.GRAD_SINK`` hands them to the rasterizer's backward as its
    output pointers, so the ONE-CALL backward (``misplat_raster_bwd``: graph replay, both per-Gaussian stages in one
    launch, the zeros written in the background of the compositing backward) writes all 236 B/Gaussian in place -- the
    data-parallel backward is the same backward as on one GPU.  Two collectives follow it, one per bucket:

      * the colour bucket (features_dc / features_rest or ``sh``: 192 of the 236 B) is complete the moment that call has
        been enqueued -- nothing but the rasterizer ever writes a colour gradient -- and its
        ``all_reduce(async_op=True)`` starts right there (``rasterizer_done``), while autograd finishes the step;
      * the geometry bucket (means, quats, scales, opacities) goes from ``allreduce()`` after ``backward()``: gradients
        that autograd adds elsewhere (the caller's own ``exp`` / ``sigmoid``, a scale regulariser) are then included --
        accumulated in place where the slice already is ``p.grad``, copied in (16 B/Gaussian) where it is not.

    One large collective per bucket suits the point-to-point xGMI links (7 x ~153 GB/s per GPU): RCCL's direct
    reduce-scatter + all-gather moves 2 x 7/8 of the bucket per rank over 7 links in parallel.  ``NCCL_DEBUG=INFO``
    (stderr) shows the algorithm / protocol RCCL picked.

    A slice is handed out ONCE per parameter and ``attach()``: a second rasterization node on the same parameters in one
    ``backward()`` (several views per step) gets an ordinary fresh tensor, which autograd then adds into the slice; with
    more than one view per backward nothing may be reduced before ``allreduce()`` -- say so with
    ``attach(views_per_backward=k)`` (a second node after an early launch raises instead of corrupting the sum)."""

    ALIGN = 4                                           # floats: every slice starts on a 16-byte boundary (the backward's
                                                        # background fill stores 16 bytes at a time)

    def __init__(self, params: Sequence[torch.Tensor], geometry: Sequence[int] | None = None,
                 colour: Sequence[int] | None = None):
        self.params = list(params)
        if colour is None:                              # by convention: the SH tensors are the big trailing ones
            colour = [i for i, p in enumerate(self.params) if p.dim() == 3 or (p.dim() == 2 and p.shape[-1] == 3 and i >= 4)]
        if geometry is None:
            geometry = [i for i in range(len(self.params)) if i not in colour]
        self.colour, self.geometry = list(colour), list(geometry)
        order = self.colour + self.geometry

        def padded(n):
            return (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN

        total = sum(padded(self.params[i].numel()) for i in order)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views: List[torch.Tensor] = [None] * len(self.params)
        o = 0
        self.n_colour = 0
        for k, i in enumerate(order):
            n = self.params[i].numel()
            self.views[i] = self.flat[o:o + n].view_as(self.params[i])
            o += padded(n)
            if k + 1 == len(self.colour):
                self.n_colour = o                        # the colour bucket is the prefix flat[:n_colour]
        self.views_per_backward = 1
        self._work: list = []                            # collectives in flight: (work, host copy or None, slice)
        self._reduced = 0                                # floats of the buffer (a prefix) whose collective has been launched
        self._by_ptr = {}
        self._handed: set = set()
        self._nodes_done = 0

    # -- the step
    def attach(self, views_per_backward: int = 1) -> None:
        """Call before ``backward()``: gradients start from None and the backward kernels are pointed at the buffer."""
        from . import ops
        for p in self.params:
            p.grad = None
        self._by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params)}
        self._work, self._reduced, self._nodes_done = [], 0, 0
        self._handed = set()
        self.views_per_backward = max(1, int(views_per_backward))
        ops.GRAD_SINK = self

    def sink(self, inp: torch.Tensor):
        """Output buffer for the gradient of ``inp`` if it is one of the parameters themselves and its slice has not been
        handed out since ``attach()`` (else None: the caller allocates, autograd accumulates).  A FRESH view object
        every time: autograd adopts an incoming gradient without a copy only if nobody else holds it."""
        i = self._by_ptr.get(inp.data_ptr())
        if i is None or self.views[i].shape != inp.shape:
            return None
        if i in self._handed:
            if self._reduced:
                from ._lib import MisplatError
                raise MisplatError("GradientBuckets: a second rasterization node reached the parameters after their all-reduce "
                                   "had been launched; call attach(views_per_backward=k) for k views per backward()")
            return None
        self._handed.add(i)
        return self.views[i].view(inp.shape)

    def rasterizer_done(self) -> None:
        """Called by the rasterizer's backward right after its one C call has been enqueued: everything it wrote into the
        buffer is final unless another view follows in the same backward."""
        self._nodes_done += 1
        if self.views_per_backward != 1 or self._reduced or _world() <= 1:
            return
        if self.colour and all(i in self._handed for i in self.colour):
            self._launch(self.n_colour)

    def colour_ready(self) -> None:
        """Two-node / stage-by-stage form: called right after the colour backward kernel has been enqueued."""
        if self.views_per_backward == 1 and not self._reduced and self.colour and _world() > 1 \
                and all(i in self._handed for i in self.colour):
            self._launch(self.n_colour)

    def allreduce(self, average: bool = False):
        """After ``backward()``: reduce what is still pending, make every ``p.grad`` its slice of the buffer.  Returns a
        (start, end) pair of device events around the collectives on a GPU, else None."""
        from . import ops
        ops.GRAD_SINK = None
        ev = None
        if self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        n_prefix = self._reduced
        for i, p in enumerate(self.params):
            start = self.views[i].storage_offset()
            if start < n_prefix:
                continue                                 # already travelling (written in place by the kernels)
            g = p.grad
            if g is None:
                self.views[i].zero_()
            elif g.data_ptr() != self.views[i].data_ptr():
                self.views[i].copy_(g)                   # produced by autograd outside the rasterizer (activations)
        if _world() > 1 and self._reduced < self.flat.numel():
            self._launch(self.flat.numel())
        for w in self._work:
            self._finish(w)
        self._work = []
        if average and _world() > 1:
            self.flat /= _world()
        for i, p in enumerate(self.params):
            p.grad = self.views[i].view(p.shape)
        if ev is not None:
            ev[1].record()
        return ev

    # -- collectives
    def _launch(self, upto: int) -> None:
        """all_reduce of flat[self._reduced : upto], asynchronously."""
        t = self.flat[self._reduced:upto]
        self._reduced = upto
        if t.numel() == 0:
            return
        if dist.get_backend() == "gloo" and t.is_cuda:       # CPU rehearsal of the multi-rank path on a GPU box
            host = t.cpu()
            self._work.append((dist.all_reduce(host, op=dist.ReduceOp.SUM, async_op=True), host, t))
        else:
            self._work.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), None, t))

    @staticmethod
    def _finish(w) -> None:
        work, host, t = w
        work.wait()
        if host is not None:
            t.copy_(host)


def _world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def max_over_ranks(value: float, device: torch.device) -> float:
    """bench.py's max-over-ranks of the timed region."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
