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
    if (world > 1 or FORCE_COLLECTIVES) and not dist.is_initialized():    # (FORCE_COLLECTIVES: a group of one that still issues its collectives)
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


# How the shared-Gaussian gradients travel.
#   MISPLAT_SPARSE_REDUCE  "auto" (default): when the rasterizer's row flags are there and the union of the ranks' touched
#                          rows is at most SPARSE_MAX_FRACTION of the Gaussians, only those rows are reduced (236 B x |union|
#                          instead of 236 B x N: a view's backward reaches 2 % of the rows at 5 M, 11 % at 1 M); "1": whenever
#                          the flags are there; "0": always the dense buffer.
#   MISPLAT_ALLREDUCE      "auto": one ``all_reduce`` per bucket, algorithm left to RCCL; "rs_ag": an explicit
#                          ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` -- every rank exchanges 1/world of the
#                          buffer with every peer at once, the pattern that uses all 7 xGMI links of a GPU (~1.9 ms for
#                          1.18 GB where a per-link-bound ring takes ~13.5 ms) -- for the case RCCL picks a ring.
SPARSE = os.environ.get("MISPLAT_SPARSE_REDUCE", "auto")
SPARSE_MAX_FRACTION = 0.35
ALLREDUCE = os.environ.get("MISPLAT_ALLREDUCE", "auto")
# One-GPU rehearsal (bench.py --buckets --rehearse-sparse): run the flags -> bitmap -> union -> pack -> [no collective] -> unpack path with a
# world of one, to measure what the sparse reduce costs on the device besides the bytes it saves on the links.
REHEARSE = False                                  # bench.py --rehearse-sparse sets it
# A process group of ONE rank normally short-cuts every collective (there is nothing to sum).  With this switch the calls are
# issued all the same -- uint8 all_gather_into_tensor, the packed all_reduce, reduce_scatter_tensor + all_gather_into_tensor,
# the dense all_reduce, the early colour launch -- so that one GPU can put the whole sequence through RCCL (dtype, size,
# stream and ordering errors show up there, not on the first 8-GPU run): tests/test_parity_gpu.py::test_rccl_world_of_one_*.
FORCE_COLLECTIVES = os.environ.get("MISPLAT_FORCE_COLLECTIVES", "0") == "1"
STATS: dict = {"dense": 0, "sparse": 0, "rows_reduced": 0, "rows_total": 0, "union_overflow": 0, "host_reads_in_step": 0,
               "collectives": 0, "geometry_touched_late": 0}


class _Done:
    def wait(self):
        return True


def _collective() -> bool:
    """True when collectives are to be issued: more than one rank, or a group of one with FORCE_COLLECTIVES."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def _gather_bits(gathered: torch.Tensor, bits: torch.Tensor) -> None:
    if _collective():
        dist.all_gather_into_tensor(gathered, bits)
    else:
        gathered.copy_(bits)


def _backend() -> str:
    return dist.get_backend() if dist.is_initialized() else "none"


def _reduce(t: torch.Tensor, async_op: bool):
    """Sum ``t`` (flat fp32, a multiple of the world size long in rs_ag mode) over the ranks, in place."""
    if not _collective():
        return _Done()
    STATS["collectives"] += 1
    if ALLREDUCE == "rs_ag" and t.numel() % _world() == 0 and t.numel() > 0:
        chunk = t.numel() // _world()
        mine = torch.empty(chunk, device=t.device, dtype=t.dtype)
        dist.reduce_scatter_tensor(mine, t, op=dist.ReduceOp.SUM)
        return dist.all_gather_into_tensor(t, mine, async_op=async_op)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


class GradientBuckets:
    """The six parameter gradients of the shared Gaussians in ONE preallocated flat fp32 buffer (59 floats = 236 B per
    Gaussian at SH degree 3), reduced over the ranks with RCCL.

    The slices of the buffer ARE the gradient tensors: ``ops.GRAD_SINK`` hands them to the rasterizer's backward as its
    output pointers, so the ONE-CALL backward (``misplat_raster_bwd``: graph replay, both per-Gaussian stages in one
    launch, the zeros written in the background of the compositing backward) writes all 236 B/Gaussian in place -- the
    data-parallel backward is the same backward as on one GPU.  Two collectives follow it, one per bucket:

      * the colour bucket (features_dc / features_rest or ``sh``: 192 of the 236 B) is complete the moment that call has
        been enqueued and its collective starts right there (``rasterizer_done``), while autograd finishes the step.  This
        assumes that nothing but the rasterizer writes a colour gradient in this backward: the colour slices have a version
        counter of their own, and ``allreduce()`` raises if anything was added to them after the launch (an SH regulariser,
        a second loss path: use ``attach(early_colour=False)`` then);
      * the geometry bucket (means, quats, scales, opacities) goes from ``allreduce()`` after ``backward()``: gradients
        that autograd adds elsewhere (the caller's own ``exp`` / ``sigmoid``, a scale regulariser) are then included --
        accumulated in place where the slice already is ``p.grad``, copied in (16 B/Gaussian) where it is not.

    **Only the rows that have a gradient travel** when the rasterizer's row flags are available (``rasterizer_done(touched)``:
    one byte per Gaussian, set by the compositing backward): the ranks all-gather their bitmaps (N / 8 bytes), OR them, list
    the union's rows (the same list everywhere), pack 236 B x |union| from the buffer, reduce that and scatter it back
    (``misplat_touched_bits`` / ``union_*`` / ``rows_pack`` / ``rows_unpack``; torch indexing for CPU rehearsals).  The
    geometry bucket goes sparse only when the rasterizer wrote every one of its slices in place and nothing else touched
    them (the activations inside the kernels, no regulariser), or when the caller vouches for it (``allreduce(sparse=True)``:
    rows outside the flags carry no gradient); above ``SPARSE_MAX_FRACTION`` the dense buffer is cheaper.

    One large collective per bucket suits the point-to-point xGMI links (7 x ~153 GB/s per GPU): RCCL's direct
    reduce-scatter + all-gather moves 2 x 7/8 of the bucket per rank over 7 links in parallel; ``MISPLAT_ALLREDUCE=rs_ag``
    issues exactly that pair.  ``NCCL_DEBUG=INFO`` (stderr) shows the algorithm / protocol RCCL picked.

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
        starts = {}
        o = 0
        self.n_colour = 0
        for k, i in enumerate(order):
            starts[i] = o
            o += padded(self.params[i].numel())
            if k + 1 == len(self.colour):
                self.n_colour = o                        # the colour bucket is the prefix flat[:n_colour]
        # The colour prefix as a tensor of its OWN over the same memory (not a view of ``flat``): its version counter then
        # counts what autograd does to the colour slices alone -- the guard of the early launch.
        self._colour_alias = torch.empty(0, device=dev, dtype=torch.float32).set_(self.flat.untyped_storage(), 0, (self.n_colour,))
        # ... and the same for the rest of the buffer: autograd accumulating IN PLACE into a geometry slice that the
        # rasterizer's gradient had become (a regulariser node behind the rasterizer's) shows in THIS counter and nowhere else
        # (the kernels and the collectives go through raw pointers / through ``flat``).
        self._geom_alias = torch.empty(0, device=dev, dtype=torch.float32).set_(self.flat.untyped_storage(), 0, (total,))
        self._geom_version = None
        for i in order:
            n = self.params[i].numel()
            base = self._colour_alias if i in self.colour else self._geom_alias
            self.views[i] = base[starts[i]:starts[i] + n].view_as(self.params[i])
        self._starts = starts
        self.n_rows = int(self.params[0].shape[0])
        self.row_params = all(p.dim() >= 1 and p.shape[0] == self.n_rows for p in self.params)   # (else: never sparse)
        self.views_per_backward = 1
        self.early_colour = True
        self._work: list = []                            # collectives in flight
        self._reduced = 0                                # floats of the buffer (a prefix) whose collective has been launched
        self._by_ptr = {}
        self._handed: set = set()
        self._nodes_done = 0
        self._touched = None
        self._union = None
        self._colour_version = None
        # Row capacity of the packed buffers, from the unions of earlier steps (None: unknown -- the first sparse step reads the
        # union's size on the host once).  With it a step enqueues bitmap -> gather -> OR -> ids -> pack -> reduce -> unpack
        # without waiting for anything: the size stays on the device (misplat_rows_pack's count_dev) and is read back after
        # the collectives have been waited for -- a union that outgrew the capacity leaves the dense buffer untouched and is
        # reduced densely then.  (The host read in the middle of the backward cost the 5 M step 0.6 ms on one GPU.)
        self._row_cap = None
        self._ws = None                                  # device work arrays of the sparse path (bitmaps, counts, offsets, ids)
        self._packed: dict = {}                          # packed row buffers per bucket span, sized by the row capacity
        self._pin = None
        self._pending = None                             # (event, the capacity the step ran with) of the count copy in flight

    # -- the step
    def attach(self, views_per_backward: int = 1, early_colour: bool = True) -> None:
        """Call before ``backward()``: gradients start from None and the backward kernels are pointed at the buffer.
        ``early_colour=False``: the colour bucket waits for ``allreduce()`` like the geometry bucket (needed when anything
        besides the rasterizer contributes a colour gradient)."""
        from . import ops
        for p in self.params:
            p.grad = None
        self._by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params)}
        self._work, self._reduced, self._nodes_done = [], 0, 0
        self._handed = set()
        self._touched, self._union, self._colour_version, self._geom_version = None, None, None, None
        self._union_counted, self._pending = False, None
        self.views_per_backward = max(1, int(views_per_backward))
        self.early_colour = bool(early_colour)
        ops.GRAD_SINK = self

    def sink(self, inp: torch.Tensor):
        """Output buffer for the gradient of ``inp`` if it is one of the parameters themselves and its slice has not been
        handed out since ``attach()`` (else None: the caller allocates, autograd accumulates).  A FRESH view object
        every time: autograd adopts an incoming gradient without a copy only if nobody else holds it."""
        i = self._by_ptr.get(inp.data_ptr())
        # (the parameter itself, or a reshaped view of ALL of it -- the model passes ``opacities.squeeze(-1)``,
        # rade_gs_model.py:444: the gradient of the view, written into the slice, reaches ``p.grad`` as a view of the same memory)
        if i is None or self.views[i].numel() != inp.numel() or not inp.is_contiguous():
            return None
        if i in self._handed:
            if self._reduced:
                from ._lib import MisplatError
                raise MisplatError("GradientBuckets: a second rasterization node reached the parameters after their all-reduce "
                                   "had been launched; call attach(views_per_backward=k) for k views per backward()")
            return None
        self._handed.add(i)
        return self.views[i].view(inp.shape)

    def rasterizer_done(self, touched: "torch.Tensor | None" = None) -> None:
        """Called by the rasterizer's backward right after its one C call has been enqueued: everything it wrote into the
        buffer is final unless another view follows in the same backward.  ``touched`` (or None): uint8 [N], non-zero for
        the rows that received a gradient in this backward (``misplat_params.touched``)."""
        self._nodes_done += 1
        self._touched = touched if (self._nodes_done == 1 and touched is not None and touched.numel() == self.n_rows) else None
        if self._nodes_done == 1:
            self._geom_version = self._geom_alias._version
        if self.views_per_backward != 1 or self._reduced or not (_collective() or REHEARSE) or not self.early_colour:
            return
        if self.colour and all(i in self._handed for i in self.colour):
            self._colour_version = self._colour_alias._version
            self._launch(self.n_colour)

    def colour_ready(self) -> None:
        """Two-node / stage-by-stage form: called right after the colour backward kernel has been enqueued."""
        if self.views_per_backward == 1 and not self._reduced and self.colour and _collective() and self.early_colour \
                and all(i in self._handed for i in self.colour):
            self._colour_version = self._colour_alias._version
            self._launch(self.n_colour)

    def allreduce(self, average: bool = False, sparse: "bool | None" = None):
        """After ``backward()``: reduce what is still pending, make every ``p.grad`` its slice of the buffer.  Returns a
        (start, end) pair of device events around the collectives on a GPU, else None.  ``sparse``: True -- the caller
        vouches that rows outside the rasterizer's flags carry no gradient (no regulariser on the geometry parameters);
        False -- dense; None -- decided here (class docstring)."""
        from . import ops
        from ._lib import MisplatError
        ops.GRAD_SINK = None
        ev = None
        if self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        n_prefix = self._reduced
        if n_prefix and self._colour_version is not None:
            # the colour bucket left early: nothing may have been added to it since (its slices have their own version counter)
            changed = self._colour_alias._version != self._colour_version
            foreign = [i for i in self.colour if self.params[i].grad is not None
                       and self.params[i].grad.data_ptr() != self.views[i].data_ptr()]
            if changed or foreign:
                raise MisplatError("GradientBuckets: a gradient reached the colour parameters AFTER their all-reduce had been "
                                   "launched from the rasterizer's backward (an SH regulariser, a second loss path?) -- the sum "
                                   "that travelled does not contain it.  Use attach(early_colour=False)")
        in_place = True                                   # every pending slice was written by the rasterizer, nothing else
        for i, p in enumerate(self.params):
            start = self._starts[i]
            if start < n_prefix:
                continue                                 # already travelling (written in place by the kernels)
            g = p.grad
            if g is None:
                self.views[i].zero_()
                in_place = False
            elif g.data_ptr() != self.views[i].data_ptr():
                self.views[i].copy_(g)                   # produced by autograd outside the rasterizer (activations)
                in_place = False
        if in_place and self._geom_version is not None and self._geom_alias._version != self._geom_version:
            # A slice the rasterizer wrote in place became p.grad, and something later in the same backward was ADDED to it in
            # place (a scale / opacity regulariser whose node ran after the rasterizer's): rows outside the rasterizer's flags
            # then carry a gradient too, and the sparse row reduce would leave them unreduced.  The geometry slices have a
            # version counter of their own for exactly this (advisor, round 4): such a step goes dense.
            in_place = False
            STATS["geometry_touched_late"] += 1
        if (_collective() or REHEARSE) and self._reduced < self.flat.numel():
            self._launch(self.flat.numel(), sparse=(in_place if sparse is None else bool(sparse)))
        for w in self._work:
            self._finish(w)
        self._work = []
        if self._pending is not None:                    # (also when this step went dense: the union may have shrunk)
            self._settle_count()
            self._pending = None
        if average and _world() > 1:
            self.flat /= _world()
        for i, p in enumerate(self.params):
            p.grad = self.views[i].view(p.shape)
        if ev is not None:
            ev[1].record()
        return ev

    # -- collectives
    def _union_rows(self):
        """Row ids of the union of the ranks' touched rows (ascending int32, the same on every rank), or None when the
        flags are missing or the union is too large a part of the scene for packing to pay.  One small collective per step;
        on the device the list is a CAPACITY of rows sized from earlier steps (``_union_counted``) and nothing is read back
        before ``allreduce()`` has waited for its collectives -- only a sink's first sparse step reads the size at once."""
        if self._union is not None:
            return self._union if self._union is not False else None
        self._union = False
        self._union_counted = False
        if SPARSE == "0" or self._touched is None or not self.row_params:
            return None
        n, world, t = self.n_rows, _world(), self._touched
        nbytes = (n + 7) // 8
        if t.is_cuda and _backend() != "gloo":
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            n_blocks = (nbytes + 255) // 256
            # (the small work arrays of this path live with the sink: no allocator call and no library scan per step -- the
            # first torch op behind the backward's launch used to block the autograd thread for 0.9 ms a step at 5 M)
            ws = self._ws
            if ws is None or ws["key"] != (n, world, t.device):
                ws = self._ws = dict(key=(n, world, t.device), bits=torch.empty(nbytes, device=t.device, dtype=torch.uint8),
                                     gathered=torch.empty(world * nbytes, device=t.device, dtype=torch.uint8),
                                     counts=torch.empty(n_blocks, device=t.device, dtype=torch.int32),
                                     offs=torch.empty(n_blocks, device=t.device, dtype=torch.int64),
                                     total=torch.empty(1, device=t.device, dtype=torch.int64), ids=None)
            bits, gathered, counts, offs, total_dev = ws["bits"], ws["gathered"], ws["counts"], ws["offs"], ws["total"]
            _lib.check(lib.misplat_touched_bits(_lib.ptr(t), C.c_int64(n), _lib.ptr(bits), _lib.stream_ptr()), "misplat_touched_bits")
            _gather_bits(gathered, bits)
            _lib.check(lib.misplat_union_count(_lib.ptr(gathered), C.c_int32(world), C.c_int64(nbytes), _lib.ptr(counts),
                                               _lib.stream_ptr()), "misplat_union_count")
            _lib.check(lib.misplat_union_scan(_lib.ptr(counts), C.c_int64(n_blocks), _lib.ptr(offs), _lib.ptr(total_dev),
                                              _lib.stream_ptr()), "misplat_union_scan")
            cap = self._row_cap
            if cap is None:                                                # first sparse step of this sink: one host read
                total = int(total_dev.item())
                STATS["host_reads_in_step"] += 1
                self._row_cap = _row_capacity(total)
                if SPARSE != "1" and total > SPARSE_MAX_FRACTION * n:
                    return None
                ids = torch.empty(max(total, 1), device=t.device, dtype=torch.int32)
                _lib.check(lib.misplat_union_ids(_lib.ptr(gathered), C.c_int32(world), C.c_int64(nbytes), _lib.ptr(offs), _lib.ptr(ids),
                                                 C.c_int64(ids.numel()), _lib.stream_ptr()), "misplat_union_ids")
                ids = ids[:total]
            else:
                # the size goes to a pinned slot behind everything else of the step; allreduce() reads it at the end
                if self._pin is None:
                    self._pin = torch.empty(1, dtype=torch.int64).pin_memory()
                self._count_dev = total_dev
                self._pin.copy_(self._count_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending = (ev, cap)
                if SPARSE != "1" and cap > SPARSE_MAX_FRACTION * n:
                    return None                                            # (dense; the count is still tracked: the union may shrink)
                if ws["ids"] is None or ws["ids"].numel() != cap:
                    ws["ids"] = torch.empty(cap, device=t.device, dtype=torch.int32)
                ids = ws["ids"]
                _lib.check(lib.misplat_union_ids(_lib.ptr(gathered), C.c_int32(world), C.c_int64(nbytes), _lib.ptr(offs), _lib.ptr(ids),
                                                 C.c_int64(cap), _lib.stream_ptr()), "misplat_union_ids")
                self._union = ids
                self._union_counted = True
                return ids
        else:                                             # CPU rehearsal (gloo): the same steps with numpy / torch
            import numpy as np
            host = (t.detach().cpu().numpy() != 0)
            bits = torch.from_numpy(np.packbits(host, bitorder="little"))
            gathered = torch.empty(world * nbytes, dtype=torch.uint8)
            _gather_bits(gathered, bits)
            union = np.bitwise_or.reduce(gathered.numpy().reshape(world, nbytes), axis=0)
            rows = np.flatnonzero(np.unpackbits(union, bitorder="little")[:n])
            if SPARSE != "1" and rows.size > SPARSE_MAX_FRACTION * n:
                return None
            ids = torch.from_numpy(rows.astype(np.int32)).to(t.device)
        self._union = ids
        return ids

    def _launch(self, upto: int, sparse: bool = True) -> None:
        """Sum of flat[self._reduced : upto] over the ranks, asynchronously (the rows of the union only when ``sparse``)."""
        a, b = self._reduced, upto
        self._reduced = upto
        if b <= a:
            return
        ids = self._union_rows() if sparse else None
        STATS["rows_total"] += self.n_rows
        if ids is None:
            STATS["dense"] += 1
            STATS["rows_reduced"] += self.n_rows
            t = self.flat[a:b]
            if _backend() == "gloo" and t.is_cuda:       # CPU rehearsal of the multi-rank path on a GPU box
                host = t.cpu()
                self._work.append(dict(work=_reduce(host, True), host=host, dst=t))
            else:
                self._work.append(dict(work=_reduce(t, True), host=None, dst=t))
            return
        STATS["sparse"] += 1
        STATS["rows_reduced"] += int(ids.numel())
        members = [i for i in range(len(self.params)) if a <= self._starts[i] < b]
        widths = [self.params[i].numel() // self.n_rows for i in members]
        W, world = sum(widths), _world()
        n_pack = int(ids.numel()) * W
        n_pad = (n_pack + world - 1) // world * world                      # (rs_ag wants a multiple of the world size)
        dev = self.flat.device
        counted = bool(getattr(self, "_union_counted", False))
        packed = self._packed.get((a, b)) if (counted and dev.type == "cuda") else None
        if packed is None or packed.numel() != n_pad:
            packed = torch.zeros(n_pad, device=dev, dtype=torch.float32) if n_pad != n_pack else torch.empty(n_pad, device=dev, dtype=torch.float32)
            if counted and dev.type == "cuda":
                self._packed[(a, b)] = packed            # (the padding floats stay zero: pack only writes n_pack of them)
        self._rows_move(True, members, widths, ids, packed, counted)
        if _backend() == "gloo" and packed.is_cuda:
            host = packed.cpu()
            self._work.append(dict(work=_reduce(host, True), host=host, dst=packed, unpack=(members, widths, ids), span=(a, b), counted=counted))
        else:
            self._work.append(dict(work=_reduce(packed, True), host=None, dst=packed, unpack=(members, widths, ids), span=(a, b), counted=counted))

    def _rows_move(self, pack: bool, members, widths, ids, packed, counted: bool = False) -> None:
        """packed[u, :] <-> the rows ids[u] of the member slices, side by side (``counted``: ids is a capacity, the number of
        rows it holds is on the device)."""
        if ids.numel() == 0:
            return
        tensors = [self.views[i].reshape(self.n_rows, w) for i, w in zip(members, widths)]
        if packed.is_cuda:
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            ptrs = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
            wid = (C.c_int32 * len(widths))(*widths)
            fn = lib.misplat_rows_pack if pack else lib.misplat_rows_unpack
            _lib.check(fn(C.c_int32(len(tensors)), ptrs, wid, _lib.ptr(ids), C.c_int64(ids.numel()),
                          _lib.ptr(self._count_dev) if counted else None, _lib.ptr(packed),
                          _lib.stream_ptr()), "misplat_rows_pack" if pack else "misplat_rows_unpack")
            return
        rows = ids.long()
        view = packed[:ids.numel() * sum(widths)].view(ids.numel(), sum(widths))
        c = 0
        for t, w in zip(tensors, widths):
            if pack:
                view[:, c:c + w] = t[rows]
            else:
                t[rows] = view[:, c:c + w]
            c += w

    def _finish(self, w) -> None:
        w["work"].wait()
        if w["host"] is not None:
            w["dst"].copy_(w["host"])
        if "unpack" in w:
            members, widths, ids = w["unpack"]
            if w.get("counted") and self._settle_count() > ids.numel():
                # the union outgrew the capacity: the scatter kernel would have done nothing, the dense slice is as the backward
                # left it -- reduce that (every rank sees the same count, so every rank takes this branch)
                STATS["union_overflow"] += 1
                a, b = w["span"]
                t = self.flat[a:b]
                if _backend() == "gloo" and t.is_cuda:
                    host = t.cpu()
                    _reduce(host, False)
                    t.copy_(host)
                else:
                    _reduce(t, False)
                return
            self._rows_move(False, members, widths, ids, w["dst"], bool(w.get("counted")))

    def _settle_count(self) -> int:
        """This step's union size (waits for the copy that was enqueued behind the union kernels; by the time the collectives
        have been waited for it has long landed) and the capacity of the next step."""
        if self._pending is None:
            return -1
        ev, _cap = self._pending
        ev.synchronize()
        total = int(self._pin[0])
        cap = self._row_cap
        if cap is None or int(total * 1.1) > cap or 4 * _row_capacity(total) < cap:
            self._row_cap = _row_capacity(total)
        return total


def _row_capacity(total: int) -> int:
    """Row capacity for a union of ``total`` rows: half as much again (the unions of consecutive steps of a training run differ by
    the views' own variation), at least 1 024."""
    return max(1024, int(total * 1.5))


def _world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def max_over_ranks(value: float, device: torch.device) -> float:
    """bench.py's max-over-ranks of the timed region."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
