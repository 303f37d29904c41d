"""Autograd-wrapped calls into libmisplat.so (HIP, gfx950).

Python here is plumbing only: it allocates tensors (so PyTorch's caching allocator and current
stream are respected), passes raw pointers through the C ABI of include/misplat.h, and raises on
any non-zero status.  Names and tuple layouts mirror gsplat-rade's ``gsplat.cuda._wrapper`` as the
reference calls it (/root/reference/collab_splats/models/rade_gs_model.py:373-394,
rade_features_model.py:430-434).
"""
from __future__ import annotations

import collections
import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib, arena
from ._lib import MISPLAT_REC, Params, RasterArgs, RasterBwdArgs, check, ptr, require_gpu, stream_ptr


# False (default): every (band, Gaussian) gradient row is added into the per-Gaussian gradient with
# one 64-byte no-return fp32 atomic -- measured fully hidden under the VALU-bound backward kernel
# (needs ~0.5 of the ~1.3 TB/s atomic rate), no second pass; sums depend on arrival order, as in
# gsplat.  True: rows go through a per-intersection slab and are summed in a fixed order -- bitwise
# reproducible gradients, ~0.5 ms slower per step at 1 M Gaussians / 1080p.
DETERMINISTIC_BACKWARD = os.environ.get("MISPLAT_DETERMINISTIC", "0") == "1"


def set_deterministic(flag: bool) -> None:
    global DETERMINISTIC_BACKWARD
    DETERMINISTIC_BACKWARD = bool(flag)

# Launch order of the compositing kernels (speed only; results never depend on it -- DESIGN.md section 6).  The
# forward records what every (tile, band) unit cost; misplat_unit_order turns that into a longest-first order per XCD
# strip, which the BACKWARD of the same step uses (always valid: same lists, same early termination) and which is
# remembered per image shape for the NEXT FORWARD (pays when consecutive calls look at similar views -- evaluation
# sweeps, repeated benchmark steps; harmless otherwise: a stale order is just another arbitrary order).
UNIT_ORDER = True
# keyed by (device, STREAM, shape): the buffer is written by misplat_unit_order on the stream of the call that produced it,
# so only a later call on the same stream is ordered behind that write (a forward on another stream -- an evaluation
# render, a second thread -- starts from the default order instead of reading a half-written permutation)
_LAST_ORDER: Dict[tuple, Tensor] = {}

# Which variants of the path the calls of this process took (tests and bench.py read it; never reset by the library):
#   forward / forward_lazy_colour / forward_merged_phases / forward_view_order (or forward_prev_order) / capacity_redo
#   backward_one_call / backward_background_fill / backward_staged / backward_sink / backward_rows_refilled
PATH_STATS: "collections.Counter" = collections.Counter()


def _stream_id() -> int:
    return int(stream_ptr().value or 0)


BANDS = 2                              # MISPLAT_BANDS: wavefronts (16 x 8 pixel bands) per tile


# View-keyed launch orders (the one-entry forward): per (device, stream, image shape) a persistent device table of
# ORDER_SLOTS records {tag, valid, permutation} and four selector words.  The projection kernel hashes the call's cameras
# and picks the record, the compositing forward runs in that record's order and leaves the order it measured there
# (include/misplat.h: misplat_params.unit_sel) -- a training loop revisits its cameras every epoch, so every view finds
# the order of its own last visit, where a single "previous call" order belongs to some other view (measured on the
# bench's 8 cycling views: as good as no order).  16 336 words per record at 1080p: 16.7 MB for 256 slots.
ORDER_SLOTS = 256
ORDER_HEADER = 16                     # MISPLAT_ORDER_HEADER
ORDER_TABLES_MAX = 8     # (device, stream, shape) keys kept, least recently used beyond
_ORDER_TABLES: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()


def _order_slots(table: Tensor, stride: int) -> int:
    return table.numel() // stride


def _order_table(P: Params, dev: torch.device):
    """(table, sel, stride) of this device / stream / shape, or None when view-keyed orders are off."""
    if not (UNIT_ORDER and ORDER_SLOTS > 0):
        return None
    units = P.tile_w * P.tile_h * P.n_cams * BANDS
    key = (dev.index, _stream_id(), P.n_cams, P.tile_w, P.tile_h, ORDER_SLOTS)
    got = _ORDER_TABLES.get(key)
    if got is not None:
        _ORDER_TABLES.move_to_end(key)
    elif torch.cuda.is_current_stream_capturing():
        # A whole-step capture (graphs.GraphedStep) runs on a stream of its own, which has no table yet -- and creating one
        # now would put its zero-fill INTO the captured graph: every replay would wipe the table in front of the projection
        # kernel's lookup, and the launch-order feedback of the warm-up would be lost.  The warm-up of the same shape ran
        # on another stream of this device and has finished: its table serves the capture (kept alive with the graph).
        for k2, v in _ORDER_TABLES.items():
            if k2[0] == key[0] and k2[2:] == key[2:]:
                got = v
                break
        if got is None:
            return None
        if _CAPTURE_KEEP is not None:
            _CAPTURE_KEEP.extend([got[0], got[1]])
        PATH_STATS["capture_reused_order_table"] += 1
    if got is None:
        # a record: header, the launch order of the view's units, the per-tile depth pivots of front-only ordering
        n_tiles = P.tile_w * P.tile_h * P.n_cams
        stride = ORDER_HEADER + 8 * ((units + 7) // 8) + 8 * ((n_tiles + 7) // 8)
        got = _ORDER_TABLES[key] = (torch.zeros(ORDER_SLOTS * stride, device=dev, dtype=torch.int32),
                                    torch.zeros(4, device=dev, dtype=torch.int32), stride)
        while len(_ORDER_TABLES) > ORDER_TABLES_MAX:            # (a dropped table lives on while a call still refers to it)
            _ORDER_TABLES.popitem(last=False)
    return got


class _UnitSchedule:
    """Per-call launch-order state of one compositing forward/backward pair."""

    def __init__(self, P: Params, dev: torch.device, by_view=None, cv=None):
        self.on = UNIT_ORDER
        self.perm_bwd = None
        self.by_view = by_view if self.on else None        # (table, sel, stride): the order lives in a view-keyed record
        if not self.on:
            return
        self.units = P.tile_w * P.tile_h * P.n_cams * BANDS
        self.key = (dev.index, _stream_id(), P.n_cams, P.tile_w, P.tile_h)
        # (the host-keyed previous-call order outlives the call in _LAST_ORDER: not from the call's arena slot)
        self.work, self.perm = _carve(dev, (self.units, 8 * ((self.units + 7) // 8) if self.by_view is None else 0),
                                      cv if self.by_view is not None else None)

    def before_forward(self, P: Params) -> None:
        if not self.on:
            return
        last = _LAST_ORDER.get(self.key)
        P.unit_perm = last.data_ptr() if last is not None else None
        P.unit_work = self.work.data_ptr()
        self._keep = last                                      # stays alive until the launch has been enqueued

    def after_forward(self, P: Params) -> None:
        P.unit_perm, P.unit_work = None, None
        if not self.on:
            return
        check(_lib.load().misplat_unit_order(C.byref(P), ptr(self.work), ptr(self.perm), stream_ptr()), "misplat_unit_order")
        _LAST_ORDER[self.key] = self.perm
        self.perm_bwd = self.perm

    def before_backward(self, P: Params) -> None:
        if self.by_view is not None:
            table, sel, stride = self.by_view
            P.unit_perm, P.unit_sel, P.unit_stride, P.unit_slots = table.data_ptr(), sel.data_ptr(), stride, _order_slots(table, stride)
        else:
            P.unit_perm = self.perm_bwd.data_ptr() if self.perm_bwd is not None else None
        P.unit_work = None

    @staticmethod
    def done(P: Params) -> None:
        P.unit_perm, P.unit_work, P.unit_sel, P.unit_stride, P.unit_slots = None, None, None, 0, 0


# Data-parallel training: a parallel.GradientBuckets object (or None).  While set, the backward of the per-Gaussian
# stages writes parameter gradients straight into its flat buffer and tells it when the colour bucket is complete.
GRAD_SINK = None


def _grad_out(inp: Tensor, cv: "Optional[arena.Carver]" = None) -> Tensor:
    if GRAD_SINK is not None:
        v = GRAD_SINK.sink(inp)
        if v is not None:
            return v
    if cv is not None:
        return cv.take(inp.numel(), inp.dtype).view(inp.shape)
    return torch.empty_like(inp)


# Optional per-kernel timing (bench.py): name -> list of (start_event, end_event) recorded on the
# current stream, i.e. the stream the kernels are launched on.  None = off (no events recorded).
KERNEL_EVENTS: Optional[Dict[str, list]] = None


class _timed:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if KERNEL_EVENTS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if KERNEL_EVENTS is not None:
            self.e1.record()
            KERNEL_EVENTS.setdefault(self.name, []).append((self.e0, self.e1))
        return False


def _kernel_event_pair():
    """Two timing events whose raw handles the one-entry C calls record around a compositing kernel (measurement passes:
    KERNEL_EVENTS is set); a torch event only gets its handle when it is first recorded."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    e1.record()
    return e0, e1


# SH colours: the forward keeps the Jacobian d rgb / d dir for the backward (see misplat_color_fwd)
def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else t.contiguous()


def _f32(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference trains in fp32, "
                        f"rade_gs_method.py:31); got {t.dtype}")
    return t.contiguous()


_PLAN_CACHE: Dict[tuple, Tuple[int, int]] = {}


def _carve(dev: torch.device, sizes, cv: "Optional[arena.Carver]" = None, dtype=torch.int32) -> list:
    """One allocation cut into views of the given element counts (each 256-byte aligned): a single trip through the
    caching allocator -- or one contiguous piece of the call's arena slot (``cv``) -- instead of one per scratch array."""
    # (ONE split call makes all the views: a Python-level slice per array was the largest single item of a small scene's
    # forward on the host -- ~25 arrays per call, 55 of its ~270 us under the profiler)
    parts, tot = [], 0
    for n in sizes:
        n = int(n)
        padded = (n + 63) // 64 * 64
        parts.append(n)
        parts.append(padded - n)
        tot += padded
    if tot < 64:
        parts.append(64 - tot)
        tot = 64
    buf = cv.take(tot, dtype) if cv is not None else torch.empty(tot, device=dev, dtype=dtype)
    return list(torch.split_with_sizes(buf, parts)[0:2 * len(sizes):2])


def bucket_plan(P: Params) -> Tuple[int, int]:
    """(n_cells, n_blocks) of the cell-ordered bucketing for this configuration (host-side query, cached)."""
    key = (P.n_gauss, P.n_cams, P.tile_w, P.tile_h)
    if key not in _PLAN_CACHE:
        nc, nb = C.c_int32(0), C.c_int32(0)
        check(_lib.load().misplat_bucket_plan(C.byref(P), C.byref(nc), C.byref(nb)), "misplat_bucket_plan")
        _PLAN_CACHE[key] = (int(nc.value), int(nb.value))
    return _PLAN_CACHE[key]


# ----------------------------------------------------------------------------- one host entry per phase

# The reference's path (RGB / RGB+ED with SH or pass-through colours, atomic gradient mode, "cells" ordering) goes
# through misplat_raster_fwd: phase A inside _ProjectPack.forward, phase B inside _BlendPacked.forward -- two C calls
# and three allocations per forward instead of ~15 calls and ~30 allocations.  SPECULATE: phase B is enqueued with a
# capacity guessed from the previous call of the same shape BEFORE the host has seen the intersection count; the
# count is checked afterwards (exact results always: an overflow re-runs phase B with the exact size).
FUSED_ENTRY = True
SPECULATE = True
CAP_MARGIN = 1.1        # re-chosen when the count comes within this of it
# A capacity, once chosen, STAYS: it is part of every phase-B argument block (a change re-captures every graph of the shape)
# and it sizes the per-tile lists inside the arena slot.  The first choice (and every later increase) is generous -- the views
# of one scene differ by tens of percent in their counts (the bench's eight: 3.8 - 6.4 M) --, address space is what it costs.
CAP_FIRST_MARGIN = 2.0
_CAP_CHOSEN: Dict[tuple, int] = {}
# (measurement aid: a list here collects (entry, byte image of the argument blocks) per call -- the graph cache's keys;
# key_trace_report() names the fields that differ between a call and the one `period` calls earlier)
KEY_TRACE: Optional[list] = None          # bench.py --key-trace sets it to a list
# (the first call of a shape learns its intersection count from a counting pass and then runs in the steady form)


def key_trace_report(period: int, start: int = 0) -> list:
    out = []
    tr = KEY_TRACE or []
    for i in range(max(start, period), len(tr)):
        (na, a), (nb, b) = tr[i - period], tr[i]
        if na != nb or len(a) != len(b):
            out.append((i, "entry", na, nb))
            continue
        if a == b:
            continue
        np_ = C.sizeof(Params)
        T = _lib.RasterArgs if nb.startswith("fwd") else _lib.RasterBwdArgs
        names = []
        for blk, cls, off in ((b[:np_], Params, 0), (b[np_:], T, np_)):
            for f in cls._fields_:
                d = getattr(cls, f[0])
                if a[off + d.offset: off + d.offset + d.size] != b[off + d.offset: off + d.offset + d.size]:
                    names.append(f[0])
        out.append((i, nb, names))
    return out
_CAP_HINT: Dict[tuple, int] = {}
_READBACK: Dict[tuple, Tensor] = {}


def _carve_f(dev: torch.device, sizes, cv: "Optional[arena.Carver]" = None) -> list:
    return _carve(dev, sizes, cv, torch.float32)


# hipGraph replay of the launch sequences (csrc/raster.hip): one graph per distinct argument block, LRU per device.
GRAPHS = os.environ.get("MISPLAT_GRAPH", "1") == "1"
# one graph per distinct argument block: a trainer that cycles through a few resident camera tensors needs forward +
# backward graphs for each of them (8 views -> 16 graphs and the first-call variants)
GRAPH_CACHE_ENTRIES = int(os.environ.get("MISPLAT_GRAPH_ENTRIES", "256"))   # (2 per resident camera tensor: forward + backward)
_GRAPH_CACHE: Dict[int, int] = {}


def _readback_slot(dev: torch.device, static: bool = False) -> Tensor:
    """Pinned int64[1] of this device that receives the intersection count.  An eager forward waits for its own count
    before it returns, so one slot per device serves all of them; fixed-capacity calls (``static_capacity``: nobody
    waits, a captured graph replays the copy) have a slot of their own, so that a replay queued in front of an eager
    forward can never be mistaken for that forward's count."""
    if static and _STATIC_SLOT is not None:
        return _STATIC_SLOT
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, bool(static))
    if key not in _READBACK:
        _READBACK[key] = torch.zeros(1, dtype=torch.int64, pin_memory=True)
    return _READBACK[key]


def _graph_cache(dev: torch.device):
    if not GRAPHS or _STATIC_CAP is not None and torch.cuda.is_current_stream_capturing():
        return None                                     # (the caller is capturing the whole step: plain launches)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _GRAPH_CACHE:
        h = _lib.load().misplat_graph_cache_create(C.c_int32(GRAPH_CACHE_ENTRIES))
        if not h:
            raise _lib.MisplatError("misplat_graph_cache_create failed")
        _GRAPH_CACHE[idx] = h
    return C.c_void_p(_GRAPH_CACHE[idx])


def reset_graph_cache(dev: Optional[torch.device] = None) -> None:
    """Drop the device's graph cache (waits for the device): the next call starts a fresh one -- captured graphs, hit
    statistics and a pause of captures after a run of misses (csrc/raster.hip, run_cached) all go."""
    idx = torch.cuda.current_device() if dev is None or dev.index is None else dev.index
    h = _GRAPH_CACHE.pop(idx, None)
    if h:
        _lib.load().misplat_graph_cache_destroy(C.c_void_p(h))


def graph_cache_stats(dev: Optional[torch.device] = None) -> Dict[str, int]:
    """{"hits", "captures"} of the device's graph cache (zeros when graphs are off or unused)."""
    idx = torch.cuda.current_device() if dev is None or dev.index is None else dev.index
    if idx not in _GRAPH_CACHE:
        return dict(hits=0, captures=0)
    h, c = C.c_int64(0), C.c_int64(0)
    check(_lib.load().misplat_graph_cache_stats(C.c_void_p(_GRAPH_CACHE[idx]), C.byref(h), C.byref(c)), "misplat_graph_cache_stats")
    return dict(hits=int(h.value), captures=int(c.value))


def _wait_count(host: Tensor) -> int:
    n = int(_lib.load().misplat_wait_count(C.c_void_p(host.data_ptr()), C.c_int64(20_000_000)))
    if n < 0:
        raise _lib.MisplatError("timed out waiting for the intersection count (GPU hung?)")
    return n


def _choose_cap(key: tuple, hint: int) -> int:
    """The intersection capacity of a speculative forward of this shape: CAP_FIRST_MARGIN x the running count when it is
    chosen, then KEPT -- until the count comes within CAP_MARGIN of it (chosen again, as generously) or has fallen to an
    eighth of it (given back)."""
    cap = _CAP_CHOSEN.get(key)
    if cap is None or int(hint * CAP_MARGIN) > cap or 4 * _quantise_cap(int(hint * CAP_FIRST_MARGIN)) < cap:
        cap = _CAP_CHOSEN[key] = min(_quantise_cap(int(hint * CAP_FIRST_MARGIN)), 2 ** 31 - 1)
    return cap


def _quantise_cap(x: int) -> int:
    """Round a capacity up to 8 steps per octave: the speculative buffers (and with them the pointers the allocator
    hands out and the graph-cache key) then stay the same while the count moves by a few percent from step to step."""
    x = max(int(x), 4096)
    step = 1 << max(x.bit_length() - 4, 0)
    return (x + step - 1) // step * step


def fused_entry_ok() -> bool:
    return FUSED_ENTRY and not DETERMINISTIC_BACKWARD


def _dp(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


# One call for BOTH phases when the capacity of the intersection buffers is known before phase A runs (a hint from the
# previous call of this shape, or a capacity fixed by the caller): one hipGraphLaunch and no Python between the phases.
MERGE_PHASES = True
# Fixed capacity (``static_capacity``): the intersection buffers get exactly this size and NOTHING on the host waits for
# the count -- the call contains no host synchronisation at all, so a whole training step can be captured into one
# hipGraph (graphs.GraphedStep).  ``meta["n_isects"]`` is then a device tensor, the lists beyond it are unspecified, and
# an overflow (count > capacity) is reported by ``check_static_capacity()`` after the fact.
_STATIC_CAP: Optional[int] = None
_STATIC_SLOT: Optional[Tensor] = None      # pinned int64[1] of the caller's own that receives the count (else: one per device)


class static_capacity:
    """``with ops.static_capacity(n): rasterization(...)`` -- see _STATIC_CAP.  ``slot``: a pinned int64[1] that receives the
    intersection count of the calls inside (a captured graph keeps writing it on every replay): several graphs that are
    replayed in turn each have their own, so that an overflow of any of them is seen (graphs.GraphedViews)."""

    def __init__(self, n: Optional[int], slot: Optional[Tensor] = None):
        self.n = None if n is None else int(n)
        self.slot = slot

    def __enter__(self):
        global _STATIC_CAP, _STATIC_SLOT
        self.old, _STATIC_CAP = _STATIC_CAP, self.n
        self.old_slot, _STATIC_SLOT = _STATIC_SLOT, self.slot
        return self

    def __exit__(self, *exc):
        global _STATIC_CAP, _STATIC_SLOT
        _STATIC_CAP = self.old
        _STATIC_SLOT = self.old_slot
        return False


_STATIC_SEEN: Dict[int, int] = {}
_CAPTURE_KEEP: Optional[list] = None      # set by graphs.GraphedStep while it captures: buffers its graph reads later


def check_static_capacity(dev: Optional[torch.device] = None) -> int:
    """After a synchronisation: the intersection count of the last fixed-capacity forward on this device; raises if it
    did not fit (its images are then incomplete)."""
    idx = torch.cuda.current_device() if dev is None or dev.index is None else dev.index
    if (idx, True) not in _READBACK or idx not in _STATIC_SEEN:
        return 0
    n = int(_READBACK[(idx, True)][0])
    if n > _STATIC_SEEN[idx]:
        raise _lib.MisplatError(f"{n} tile intersections exceed the fixed capacity {_STATIC_SEEN[idx]}: rerun with a larger one")
    return n


def _cap_key(P: Params, dev: torch.device):
    return (dev.index, P.n_gauss, P.n_cams, P.width, P.height)


# On-demand SH colours (csrc/blend.hip, LazyColour): no colour kernel -- the compositing forward evaluates the colour of
# a record when it first stages it.  Pays when most visible Gaussians are never composited (dense scenes: 1 M random
# Gaussians at 1080p stage a third of the visible ones); "auto" switches it on from the typical bucket length.
SPARSE_BWD_MIN_ROWS = 262144        # raster.hip: background_fill_ok
ROWS_ON_TOUCH = True
LAZY_SH = os.environ.get("MISPLAT_LAZY_SH", "auto")
LAZY_ND = "1"                       # on-demand N-D records (features model) where LAZY_SH allows
LAZY_SH_MIN_BUCKET = 450   # measured crossover at 1080p: 391 even, 549 ahead
# Front-only ordering (csrc/binning.hip, tile_sort_front_kernel): in a dense scene the compositing stops long before the end of
# a tile's list, so only the part of every bucket in front of the depth the view's LAST visit reached (x a margin; the
# pivots live in the view-keyed launch-order records) is sorted; a tile whose pixels outlive its sorted part is sorted in
# full and composited again (exact images either way), and meta["flatten_ids"] / ["isect_ids"] are completed on access.
# "auto": from a typical bucket of FRONT_MIN_AVG entries (the hint of the previous call of the shape); "1": always; "0": off.
FRONT_ONLY = os.environ.get("MISPLAT_FRONT_ONLY", "auto")
FRONT_MIN_AVG = 1024
FRONT_MIN_BUCKET = 256
FRONT_MARGIN = 1.05


def _lazy_colour_ok(P: Params, dev, deg: int, kd: int, n_color: int, want_grad: bool, cd: int, nxq: int = 0,
                    features=None) -> bool:
    if LAZY_SH == "0" or deg < 0 or kd != 16 or n_color != 3 or not want_grad:
        return False
    if nxq == 0 and cd not in (3, 4):
        return False
    # N-D records on demand (the features model's call: SH colours + a feature tensor, D' = 16 / 17, one camera)
    if nxq > 0 and (LAZY_ND == "0" or features is None or nxq not in (3, 4) or P.n_cams != 1):
        return False
    if LAZY_SH == "1":
        return True
    hint = _CAP_HINT.get(_cap_key(P, dev))
    return hint is not None and hint >= LAZY_SH_MIN_BUCKET * P.tile_w * P.tile_h * P.n_cams


def _front_only_wanted(P: Params, dev) -> bool:
    """Front-only ordering for this call?  Needs the view-keyed records (pivots), a capacity hint (the typical bucket) and
    an eager, non-static call; decided before phase A, which then also lays the buckets out for it (entries = positions in the cell-ordered row list)."""
    if FRONT_ONLY == "0" or _STATIC_CAP is not None or not (SPECULATE and UNIT_ORDER):
        return False
    hint = _CAP_HINT.get(_cap_key(P, dev))
    return hint is not None and (FRONT_ONLY == "1" or hint >= FRONT_MIN_AVG * P.tile_w * P.tile_h * P.n_cams)


def _fwd_ring_key(P: Params, dev, kd: int, want_grad, want_aux, absgrad, depth_channel, indexed, nxq) -> tuple:
    return ("fwd", dev.index, _stream_id(), P.n_gauss, P.n_cams, P.width, P.height, kd, int(bool(want_grad)), int(bool(want_aux)),
            int(bool(absgrad)), int(bool(depth_channel)), int(bool(indexed)), int(nxq))


def _raster_phase_a(P: Params, means, quats, scales, opacities, colors, colors_rest, viewmats, Ks, deg: int, kd: int,
                    n_color: int, per_cam: int, depth_channel: bool, want_aux: bool, want_grad: bool, defer: bool = False,
                    lazy: bool = False, flags: bool = False, absgrad: bool = False, nxq: int = 0, features=None,
                    probe: bool = False, cd_hint: int = 20, cv: "Optional[arena.Carver]" = None):
    """Allocations + phase A of misplat_raster_fwd (``defer``: phase A is launched together with B, by _raster_phase_b).
    Returns (radii, means2d, depths, comps, grec, sh_aux, state)."""
    lib = _lib.load()
    dev = means.device
    N, Cn = P.n_gauss, P.n_cams
    rows = Cn * N
    n_tiles = P.tile_w * P.tile_h * Cn
    n_cells, n_blocks = bucket_plan(P)
    # Every array of the call from ONE slot of a persistent ring (arena.py): the same call finds the same addresses every
    # step -- an argument block recurs whenever its camera tensor does, so its hipGraph is replayed --, and the gather targets
    # (records, lists) start on 2 MiB boundaries.  A slot is handed out again only when nothing refers to its storage.
    # (bucket entries as positions in the cell-ordered row list: pays where the per-tile sort's depth gather misses the L2 --
    # dense scenes, i.e. together with front-only ordering; at 1 M Gaussians it costs the sort a second gather: 55 -> 86 us)
    indexed = _front_only_wanted(P, dev) and rows < (1 << 23)          # (23 index bits + 9 bits of depth code)
    if cv is None:
        cv = arena.Carver(None if probe else _fwd_ring_key(P, dev, kd, want_grad, want_aux, absgrad, depth_channel, indexed, nxq), dev)
    if defer and not probe and _STATIC_CAP is None and _CAP_HINT.get(_cap_key(P, dev)) is not None:
        # a ring's first call: its demand, summed the way the takes below (and phase B's) go -- 64-element rounding inside a
        # carve, 256 bytes between takes --, so that the slot exists from call one (arena.Carver.reserve)
        g_, nd_ = int(bool(want_grad)), int(nxq > 0)
        n_pix_, units_ = Cn * P.height * P.width, n_tiles * BANDS
        cap_ = _choose_cap(_cap_key(P, dev), _CAP_HINT[_cap_key(P, dev)])
        carves = ((2 * rows, rows, rows, 12 * rows * int(bool(want_aux))), (MISPLAT_REC * rows,), (MISPLAT_REC * rows * g_,),
                  (2 * rows * g_ * int(bool(absgrad)),), (4 * nxq * rows * nd_ * int(not lazy),), (4 * nxq * rows * nd_ * g_,),
                  (2 * rows, rows, 2 * rows, n_blocks * n_cells, n_cells, n_cells, 4, n_tiles + 1, n_cells + 1, rows, 2 * rows,
                   (rows + 3) // 4 * g_, rows * int(bool(indexed))),
                  (cd_hint * n_pix_, n_pix_, n_pix_, n_pix_, 3 * n_pix_), (units_, 8 * ((units_ + 7) // 8)),
                  (n_pix_, n_pix_, n_tiles + 2, units_, n_tiles, n_tiles), (cap_,), (cap_,), (2 * cap_,))
        cv.reserve(sum(4 * sum((int(n) + 63) // 64 * 64 for n in c) + 512 for c in carves))
    means2d, depths, comps, sh_aux = _carve_f(dev, (2 * rows, rows, rows, 12 * rows if want_aux else 0), cv)
    grec = cv.take(MISPLAT_REC * rows, torch.float32)
    v_grec_zero = cv.take(MISPLAT_REC * rows, torch.float32).view(rows, MISPLAT_REC) if want_grad else None
    v_abs_zero = cv.take(2 * rows, torch.float32).view(rows, 2) if (want_grad and absgrad) else None
    # N-D colours (the features model, nxq > 0): channels 4.. of every row, and their gradient rows
    # (on-demand N-D records: no featx -- the compositing kernels read the feature rows themselves)
    featx = cv.take(4 * nxq * rows, torch.float32).view(rows, 4 * nxq) if (nxq > 0 and not lazy) else None
    v_featx_zero = cv.take(4 * nxq * rows, torch.float32).view(rows, 4 * nxq) if (nxq > 0 and want_grad) else None
    # (cell_count, cell_cursor, counters, tile_count back to back: the projection kernel clears that contiguous range -- no
    # memset, no clearing launch)
    (radii, tiles_per_gauss, rect2, cellhist, cell_count, cell_cursor, counters, tile_count, cell_offs, order, rect_sorted,
     touched, depth_sorted) = _carve(
        dev, (2 * rows, rows, 2 * rows, n_blocks * n_cells, n_cells, n_cells, 4, n_tiles + 1, n_cells + 1, rows, 2 * rows,
              (rows + 3) // 4 if want_grad else 0, rows if indexed else 0), cv)
    # one byte per row: cleared by the projection kernel, set by the compositing backward, read by the per-Gaussian
    # backward kernels (misplat_params.touched).  Only where the compositing backward is the ONLY source of the packed
    # gradient rows (the single autograd node): with the two-node form a loss on the projection's own outputs reaches the
    # projection backward without ever passing the compositing kernels.
    P.touched = touched.data_ptr() if (want_grad and flags) else None
    host = _readback_slot(dev, static=_STATIC_CAP is not None and defer)
    host[0] = -1                                                      # overwritten by the asynchronous copy of phase A
    a = RasterArgs()
    a.means, a.quats, a.scales, a.opacities = _dp(means), _dp(quats), _dp(scales), _dp(opacities)
    a.colors, a.colors_rest, a.viewmats, a.Ks = _dp(colors), _dp(colors_rest), _dp(viewmats), _dp(Ks)
    a.sh_degree, a.K_or_D, a.n_color, a.per_cam, a.depth_channel = deg, kd, n_color, per_cam, int(depth_channel)
    a.radii, a.means2d, a.depths, a.compensations, a.grec = _dp(radii), _dp(means2d), _dp(depths), _dp(comps), _dp(grec)
    a.sh_aux = _dp(sh_aux) if want_aux else None
    a.v_grec_zero = _dp(v_grec_zero)
    a.v_abs_zero = _dp(v_abs_zero)
    a.nxq, a.featx, a.v_featx_zero = int(nxq), _dp(featx), _dp(v_featx_zero)
    a.features, a.n_feat = _dp(features), (int(features.shape[-1]) if features is not None else 0)
    a.tiles_per_gauss, a.rect2, a.cellhist, a.cell_count = _dp(tiles_per_gauss), _dp(rect2), _dp(cellhist), _dp(cell_count)
    a.cell_offs, a.order, a.counters, a.tile_count = _dp(cell_offs), _dp(order), _dp(counters), _dp(tile_count)
    a.cell_cursor = _dp(cell_cursor)
    a.rect_sorted = _dp(rect_sorted)
    # bucket entries = positions in the cell-ordered row list (the per-tile sort then gathers its depth keys locally)
    a.depth_sorted = depth_sorted.view(torch.float32).data_ptr() if indexed else None
    a.n_isects_host = host.data_ptr()
    # Gradient rows cleared on first touch (lazy_colour = 2) where the backward is going to read flagged rows only -- the
    # static part of raster.hip's background_fill_ok; the backward checks the actual plan and clears v_grec itself otherwise.
    rows_on_touch = bool(lazy and want_grad and flags and Cn == 1 and N >= SPARSE_BWD_MIN_ROWS and not want_aux and kd == 16
                         and ROWS_ON_TOUCH)
    a.lazy_colour = 2 if rows_on_touch else int(lazy)
    row_order = order                                                 # (the cell-ordered row list; `order` below: the launch-order table)
    order = _order_table(P, dev)
    if order is not None:
        a.order_table, a.order_sel, a.order_slots, a.order_stride = (_dp(order[0]), _dp(order[1]),
                                                                     _order_slots(order[0], order[2]), order[2])
    PATH_STATS["forward_rows_on_touch"] += int(rows_on_touch)
    if not defer:
        check(lib.misplat_raster_fwd(C.byref(P), C.byref(a), C.c_int32(1), stream_ptr(), _graph_cache(dev)),
              "misplat_raster_fwd(A)")
    state = dict(args=a, host=host, tiles_per_gauss=tiles_per_gauss, depths=depths, v_grec_zero=v_grec_zero, v_abs_zero=v_abs_zero,
                 deferred=defer, rows_on_touch=rows_on_touch, order=order, carver=cv, featx=featx, v_featx_zero=v_featx_zero,
                 counters=counters, touched=touched,
                 keep=(rect2, cellhist, cell_count, cell_offs, row_order, counters, tile_count, radii, cell_cursor, depth_sorted),
                 row_map=row_order, depth_sorted=depth_sorted.view(torch.float32) if indexed else None)
    return (radii.view(Cn, N, 2), means2d.view(Cn, N, 2), depths.view(Cn, N), comps.view(Cn, N), grec.view(rows, MISPLAT_REC),
            sh_aux.view(rows, 12) if want_aux else None, state)


def _raster_phase_b(P: Params, state: dict, cd: int):
    """Phase B of misplat_raster_fwd with a speculative capacity; returns (images..., bins, sched)."""
    prep = _phase_b_prepare(P, state, cd)
    return _phase_b_launch(P, state, prep, cd, state.get("carver"))


def _phase_b_prepare(P: Params, state: dict, cd: int) -> dict:
    """Everything of phase B that does not depend on this call's intersection count: the capacity, the images and lists carved
    from the call's arena slot, the argument block filled in.  A function of (P, the state phase A left, the capacity hint) --
    which is what makes it cacheable per arena slot (_FwdPlan)."""
    a = state["args"]
    dev = state["depths"].device
    Cn, H, W = P.n_cams, P.height, P.width
    n_pix = Cn * H * W
    n_tiles = P.tile_w * P.tile_h * Cn
    key = _cap_key(P, dev)
    hint = _CAP_HINT.get(key) if SPECULATE else None
    n_known = None
    static = _STATIC_CAP is not None and state.get("deferred")
    if static:
        cap = max(int(_STATIC_CAP), 1)
    elif hint is None:                                                # first call of this shape: exact, as before
        assert not state.get("deferred")
        n_known = _wait_count(state["host"])
        cap = max(n_known, 1)
    else:
        cap = _choose_cap(key, hint)
        # (the estimate that picks the sort's size classes is tied to the capacity, not to the running count: the count moves
        # every step, and the argument block is the graph key)
        a.est_isects = int(cap / CAP_FIRST_MARGIN)
    if cap >= 2 ** 31:
        raise _lib.MisplatError(f"{cap} tile intersections exceed int32 indexing")
    cv = state.get("carver")
    render, alpha, exp_depth, med_depth, normal = _carve_f(dev, (cd * n_pix, n_pix, n_pix, n_pix, 3 * n_pix), cv)
    sched = _UnitSchedule(P, dev, by_view=state.get("order"), cv=cv)
    by_view = sched.on and sched.by_view is not None
    front = bool(by_view and not static and hint is not None and state.get("depth_sorted") is not None)   # (_front_only_wanted)
    last_ids, median_ids, offsets, reach, front_n, tile_flag = _carve(
        dev, (n_pix, n_pix, n_tiles + 2, n_tiles * BANDS if by_view else 0, n_tiles if front else 0, n_tiles if front else 0), cv)
    a.unit_reach = _dp(reach) if by_view else None
    a.front_n, a.tile_flag = (_dp(front_n), _dp(tile_flag)) if front else (None, None)
    a.front_margin, a.front_min_bucket = FRONT_MARGIN, FRONT_MIN_BUCKET
    PATH_STATS["forward_front_only"] += int(front)
    payload, flatten_ids, scratch = _isect_buffers(dev, cv, cap)
    a.color_dim = cd
    a.offsets, a.render, a.alpha, a.exp_depth, a.med_depth, a.normal = (_dp(offsets), _dp(render), _dp(alpha), _dp(exp_depth),
                                                                        _dp(med_depth), _dp(normal))
    a.last_ids, a.median_ids = _dp(last_ids), _dp(median_ids)
    if sched.on and sched.by_view is not None:
        PATH_STATS["forward_view_order"] += 1
        a.unit_perm_in, a.unit_work, a.unit_perm_out = None, _dp(sched.work), None
    elif sched.on:
        a.order_table, a.order_sel = None, None
        last = _LAST_ORDER.get(sched.key)
        if last is not None and _CAPTURE_KEEP is not None:
            _CAPTURE_KEEP.append(last)                                # a captured graph keeps reading this buffer
        if last is not None:
            PATH_STATS["forward_prev_order"] += 1
        a.unit_perm_in, a.unit_work, a.unit_perm_out = _dp(last), _dp(sched.work), _dp(sched.perm)
    else:
        a.unit_perm_in, a.unit_work, a.unit_perm_out = None, None, None
    a.payload, a.flatten_ids, a.scratch, a.cap_isects = _dp(payload), _dp(flatten_ids), _dp(scratch), cap
    return dict(cap=cap, n_known=n_known, static=static, front=front, sched=sched, render=render, alpha=alpha, exp_depth=exp_depth,
                med_depth=med_depth, normal=normal, last_ids=last_ids, median_ids=median_ids, offsets=offsets, reach=reach,
                front_n=front_n, tile_flag=tile_flag, payload=payload, flatten_ids=flatten_ids, scratch=scratch)


def _isect_buffers(dev, cv, c: int):
    if cv is None:
        return _carve(dev, (c, c, 2 * c))
    return cv.take(c, torch.int32), cv.take(c, torch.int32), cv.take(2 * c, torch.int32)


def _phase_b_launch(P: Params, state: dict, prep: dict, cd: int, cv):
    """The launch (both phases in one call when phase A was deferred), the wait for this call's intersection count, the exact redo
    when the capacity fell short, and the per-call results: (images, bins, sched).  Returns prep["overflow"] = True after a redo
    (the buffers of ``prep`` are then no longer the ones the call used)."""
    lib = _lib.load()
    a = state["args"]
    dev = state["depths"].device
    Cn, H, W = P.n_cams, P.height, P.width
    n_tiles = P.tile_w * P.tile_h * Cn
    key = _cap_key(P, dev)
    cap, n_known, static, front, sched = prep["cap"], prep["n_known"], prep["static"], prep["front"], prep["sched"]
    payload, flatten_ids, scratch = prep["payload"], prep["flatten_ids"], prep["scratch"]
    offsets, reach, front_n, tile_flag = prep["offsets"], prep["reach"], prep["front_n"], prep["tile_flag"]
    phases = 3 if state.get("deferred") else 2
    while True:
        a.payload, a.flatten_ids, a.scratch, a.cap_isects = _dp(payload), _dp(flatten_ids), _dp(scratch), cap
        if KERNEL_EVENTS is not None:                                 # a measurement pass: events around the compositing forward
            ev = _kernel_event_pair()
            a.ev_blend_begin, a.ev_blend_end = ev[0].cuda_event, ev[1].cuda_event
            KERNEL_EVENTS.setdefault("blend_fwd", []).append(ev)
        with _timed("raster_fwd_B"):
            if KEY_TRACE is not None:
                KEY_TRACE.append(("fwd%d" % phases, bytes(P) + bytes(a)))
            check(lib.misplat_raster_fwd(C.byref(P), C.byref(a), C.c_int32(phases), stream_ptr(), _graph_cache(dev)),
                  "misplat_raster_fwd(B)")
        a.ev_blend_begin, a.ev_blend_end = None, None
        phases = 2
        if static:                                                    # nobody waits: the count stays on the device
            _STATIC_SEEN[dev.index if dev.index is not None else torch.cuda.current_device()] = cap
            break
        if n_known is None:
            n_known = _wait_count(state["host"])                      # usually long there: phase B was enqueued meanwhile
        if n_known <= cap:
            break
        cap = n_known                                                 # the guess was too small: exact size, once more
        prep["overflow"] = True
        PATH_STATS["capacity_redo"] += 1
        tc = state["keep"][6]                                         # tile_count: phase B expects it cleared
        check(lib.misplat_zero_bytes(ptr(tc), C.c_size_t(4 * tc.numel()), stream_ptr()), "misplat_zero_bytes")
        if cap >= 2 ** 31:
            raise _lib.MisplatError(f"{cap} tile intersections exceed int32 indexing")
        payload, flatten_ids, scratch = _isect_buffers(dev, cv, cap)
    if not static:
        # a very slowly decaying maximum: consecutive training views differ in their counts by tens of percent (1 M random
        # Gaussians, the eight views of configs[3]: 3.8 - 6.4 M), a guess that falls short costs a second phase B, a
        # generous one costs nothing but address space (288 GB of HBM) -- and a capacity that stays put keeps the buffers'
        # addresses and with them the graph keys (a change of 1/8 octave needs ~100 calls at this decay)
        _CAP_HINT[key] = max(n_known, int(0.999 * _CAP_HINT.get(key, 0)))
    if sched.on and sched.by_view is None:
        _LAST_ORDER[sched.key] = sched.perm
        sched.perm_bwd = sched.perm
    if cv is not None:
        cv.done()
        PATH_STATS["forward_arena_slot"] += int(cv.slot is not None)
    bins = dict(tiles_per_gauss=state["tiles_per_gauss"], n_isects=cap if static else n_known, depths=state["depths"],
                tile_ids=None, v_grec_zero=state.get("v_grec_zero"), v_abs_zero=state.get("v_abs_zero"),
                featx=state.get("featx"),
                # (on-demand N-D records without rows-on-touch: nobody has cleared the featx gradient rows -- the backward does)
                v_featx_zero=(state.get("v_featx_zero") if (a.lazy_colour != 1 or a.nxq == 0) else None),
                rows_on_touch=bool(state.get("rows_on_touch")), n_isects_dev=state["counters"].view(torch.int64)[0] if static else None,
                n_tiles=n_tiles, slots=None, flatten_ids=flatten_ids[:cap if static else n_known], cap_isects=int(cap),
                isect_offsets=offsets[:n_tiles + 1], _keep=(scratch, payload, reach),
                # one byte per row, set by the atomic compositing backward for the rows it adds a gradient to
                touched=(state["touched"].view(torch.uint8)[:P.n_gauss * Cn] if P.touched else None),
                # front-only ordering: flatten_ids holds the sorted head of every list (all the compositing and the
                # backward read); complete_bins() sorts the rest when someone wants the whole lists
                partial=(dict(cap=cap, offsets=offsets, payload=payload, scratch=scratch, flatten_ids=flatten_ids,
                              front_n=front_n, tile_flag=tile_flag, row_map=state["row_map"],
                              depth_sorted=state["depth_sorted"]) if front else None))
    imgs = (prep["render"].view(Cn, H, W, cd), prep["alpha"].view(Cn, H, W, 1), prep["exp_depth"].view(Cn, H, W, 1),
            prep["med_depth"].view(Cn, H, W, 1), prep["normal"].view(Cn, H, W, 3), prep["last_ids"].view(Cn, H, W),
            prep["median_ids"].view(Cn, H, W))
    return imgs, bins, sched


# ----------------------------------------------------------------------------- one autograd node for the whole call

# The reference's path as ONE autograd node: forward = phases A + B (two C calls, or one graph each), backward = one C
# call (misplat_raster_bwd: compositing, colour, projection backward; replayed as a graph when memset-free).  Halves
# the autograd bookkeeping of the two-node form, which is what a small scene spends its time on.  ``means2d`` is an
# OUTPUT of the node; gsplat's contract -- ``meta["means2d"].grad`` / ``.absgrad`` hold the screen-space gradient
# after ``backward()`` (rade_gs_model.py:191-198) -- is kept by assigning both from inside the backward.  The other
# per-Gaussian intermediates in ``meta`` (conics, ray planes, ...) are not differentiable in this form; set
# MISPLAT_FUSED_NODE=0 (the two-node form) to differentiate through them.
FUSED_NODE = True


def fused_node_ok() -> bool:
    return FUSED_NODE and fused_entry_ok()


# A steady-state call is a function of its inputs' addresses: the arena hands the same slot to the same call, so the ~30 views
# carved from it and the ~90 fields of the argument block come out identical every time.  They are kept ON THE SLOT
# (arena.Slot.plan) keyed by everything that went into them; a call that finds its plan resets the pinned count, makes the one
# C call and hands out fresh views.  What stays per call: the count, the length of flatten_ids, the bins dict, the tensor
# objects autograd gets.  A small scene's eager step is host-bound: this is most of what the host did.
PLAN_CACHE = True


class _FwdPlan:
    __slots__ = ("key", "P_image", "state", "prep", "outs", "stats", "off", "demand", "ident", "refs")


class _BwdPlan:
    __slots__ = ("key", "b", "grads", "v_grec", "v_abs", "v_featx", "v_m2d", "sparse", "off", "demand", "on_touch", "flags", "refs")


_PLAN_IDS = iter(range(1, 1 << 62))


class _RasterFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, colors_rest, features, viewmats, Ks, P: Params, sh_degree,
                depth_channel: bool, cd: int, absgrad: bool, extra: dict):
        import weakref
        require_gpu(means, quats, scales, opacities, colors, viewmats, Ks, features)
        if sh_degree is not None:
            kd = colors.shape[1] if colors_rest is None else 1 + colors_rest.shape[1]
            deg, n_color, per_cam = int(sh_degree), 3, 0
        else:
            deg, kd, per_cam = -1, colors.shape[-1], int(colors.dim() == 3)
            n_color = kd
        want_grad = any(ctx.needs_input_grad[:7])
        # N-D colours in one pass (a8, rade_features_model.py:427-476): cd = D' composited channels, 4 in the record + 4 nxq
        nxq = (cd - 4 + 3) // 4 if cd > 4 else 0
        lazy = _lazy_colour_ok(P, means.device, deg, kd, n_color, want_grad, cd, nxq, features)
        want_aux = deg >= 0 and want_grad and not lazy
        if (MERGE_PHASES and SPECULATE and _STATIC_CAP is None and _cap_key(P, means.device) not in _CAP_HINT
                and not torch.cuda.is_current_stream_capturing()):
            # First call of a shape: a counting pass (projection + bucketing, results dropped) learns the intersection count,
            # so that THIS call already runs in the steady form -- one merged entry with a speculative capacity, the argument
            # block the later visits of the view will present (graph key), the arena demand of the steady call.
            Pp = Params.from_buffer_copy(bytes(P))
            st = _raster_phase_a(Pp, means, quats, scales, opacities, colors, colors_rest, viewmats, Ks, deg, kd, n_color, per_cam,
                                 depth_channel, False, False, nxq=nxq, features=features, probe=True)[-1]
            _CAP_HINT[_cap_key(P, means.device)] = max(_wait_count(st["host"]), 1)
            del st
            PATH_STATS["forward_probe"] += 1
            lazy = _lazy_colour_ok(P, means.device, deg, kd, n_color, want_grad, cd, nxq, features)
            want_aux = deg >= 0 and want_grad and not lazy
        defer = _STATIC_CAP is not None or (MERGE_PHASES and SPECULATE and _cap_key(P, means.device) in _CAP_HINT)
        PATH_STATS["forward"] += 1
        PATH_STATS["forward_lazy_colour"] += int(lazy)
        PATH_STATS["forward_merged_phases"] += int(bool(defer))
        PATH_STATS["forward_nd"] += int(nxq > 0)
        dev = means.device
        N, Cn = P.n_gauss, P.n_cams
        rows = N * Cn
        # the steady form: both phases in one call with a capacity that is known, nothing being measured or captured
        plannable = (PLAN_CACHE and defer and _STATIC_CAP is None and KERNEL_EVENTS is None and KEY_TRACE is None and UNIT_ORDER
                     and arena.ENABLED and not torch.cuda.is_current_stream_capturing())
        cv, plan, pkey = None, None, None
        if plannable:
            indexed = _front_only_wanted(P, dev) and rows < (1 << 23)
            cv = arena.Carver(_fwd_ring_key(P, dev, kd, want_grad, want_aux, absgrad, depth_channel, indexed, nxq), dev)
            if cv.slot is not None:
                ck = _cap_key(P, dev)
                pkey = (bytes(P), means.data_ptr(), quats.data_ptr(), scales.data_ptr(), opacities.data_ptr(), colors.data_ptr(),
                        _dp(colors_rest), _dp(features), viewmats.data_ptr(), Ks.data_ptr(), deg, kd, n_color, per_cam,
                        bool(depth_channel), want_aux, want_grad, lazy, bool(absgrad), nxq, cd, _choose_cap(ck, _CAP_HINT[ck]),
                        FRONT_MARGIN, FRONT_MIN_BUCKET, ROWS_ON_TOUCH, id(_ORDER_TABLES.get((dev.index, _stream_id(), Cn, P.tile_w,
                                                                                              P.tile_h, ORDER_SLOTS))))
                plan = cv.slot.get_plan(pkey)
        if plan is not None:
            # ---- the call's plan is on its slot: restore what the build left (P with its pointers, the carve position), launch
            C.memmove(C.addressof(P), plan.P_image, C.sizeof(P))
            state, prep = plan.state, plan.prep
            cv.slot.off, cv.slot.demand = plan.off, plan.demand
            state["host"][0] = -1
            for k, v in plan.stats.items():
                PATH_STATS[k] += v
            PATH_STATS["forward_plan_hit"] += 1
            r0, m0, d0, c0, g0, s0 = plan.outs
        else:
            c_before = cv.slot._count() if (pkey is not None) else 0
            stats0 = dict(PATH_STATS) if pkey is not None else None
            r0, m0, d0, c0, g0, s0, state = _raster_phase_a(
                P, means, quats, scales, opacities, colors, colors_rest, viewmats, Ks, deg, kd, n_color, per_cam, depth_channel,
                want_aux, want_grad, defer=defer, lazy=lazy, flags=True, absgrad=bool(absgrad), nxq=nxq, features=features, cd_hint=cd,
                cv=cv)
            cv = state["carver"]
            prep = _phase_b_prepare(P, state, cd)
            state["carver"] = None            # (a plan that kept its carver would close a cycle slot -> plan -> carver -> ring -> slot:
                                              #  a dropped ring's memory would then wait for the cyclic collector)
            if pkey is not None and prep["sched"].by_view is not None and cv.slot is not None:
                plan = _FwdPlan()
                plan.key, plan.state, plan.prep, plan.outs = pkey, state, prep, (r0, m0, d0, c0, g0, s0)
                plan.P_image = C.create_string_buffer(bytes(P), C.sizeof(P))
                plan.off, plan.demand = cv.slot.off, cv.slot.demand
                plan.stats = {k: v - stats0.get(k, 0) for k, v in PATH_STATS.items() if v != stats0.get(k, 0)}
                plan.ident = next(_PLAN_IDS)
                # (every tensor that refers to the slot at this point -- nothing per-call exists yet -- is the plan's own)
                new_refs = cv.slot._count() - c_before
        imgs, bins, sched = _phase_b_launch(P, state, prep, cd, cv)
        if plan is not None:
            if prep.get("overflow"):                                  # the capacity fell short: this plan's lists were replaced
                cv.slot.drop_plan(pkey)
                prep.pop("overflow", None)
            elif cv.slot.plans.get(pkey) is not plan:
                cv.slot.put_plan(pkey, plan, new_refs)
            bins["plan"] = plan.ident
        # the node's outputs are fresh tensor objects (autograd writes its history into what forward returns)
        radii, means2d, depths, comps, grec = r0.view(Cn, N, 2), m0.view(Cn, N, 2), d0.view(Cn, N), c0.view(Cn, N), g0.view(rows, MISPLAT_REC)
        sh_aux = s0.view(rows, 12) if s0 is not None else None
        render, alpha, exp_depth, med_depth, normal, last_ids, median_ids = imgs
        extra["bins"] = bins
        bins["grec"] = grec                                              # (the packed records: bench.py counts the colours that were set)
        ctx.P, ctx.bins, ctx.sched, ctx.cd, ctx.absgrad = P, bins, sched, cd, absgrad
        ctx.color_args = (deg, kd, n_color, per_cam)
        ctx.depth_slot = 12 + n_color if (depth_channel and nxq == 0) else -1
        ctx.nd = (nxq, bool(depth_channel), features is not None)
        ctx.has_rest, ctx.has_aux = colors_rest is not None, sh_aux is not None
        ctx.means2d_ref = weakref.ref(means2d)
        ctx.save_for_backward(means, quats, scales, opacities, colors, viewmats, Ks, radii, comps,
                              colors_rest if colors_rest is not None else colors,
                              sh_aux if sh_aux is not None else comps, grec, alpha, last_ids, median_ids, render,
                              features if features is not None else comps)
        ctx.mark_non_differentiable(radii, depths, comps, grec, last_ids, median_ids)
        ctx.set_materialize_grads(False)
        return render, alpha, exp_depth, med_depth, normal, means2d, radii, depths, comps, grec, last_ids, median_ids

    @staticmethod
    def backward(ctx, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal, v_means2d_in, *_unused):
        lib = _lib.load()
        (means, quats, scales, opacities, colors, viewmats, Ks, radii, comps, colors_rest, sh_aux, grec, alpha, last_ids,
         median_ids, render, features) = ctx.saved_tensors
        nxq, nd_depth, has_feat = ctx.nd
        if not has_feat:
            features = None
        if not ctx.has_rest:
            colors_rest = None
        if not ctx.has_aux:
            sh_aux = None
        P, bins, cd = ctx.P, ctx.bins, ctx.cd
        deg, kd, n_color, per_cam = ctx.color_args
        dev = grec.device
        rows = P.n_cams * P.n_gauss
        ups = _upstream(P, cd, dev, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal)
        # the gradients this call hands to autograd, from the backward's own arena ring (they may live on as `.grad` of the
        # caller's parameters: the slot is reused when nothing refers to it any more -- arena.py)
        cvb = arena.Carver(("bwd", dev.index, _stream_id(), P.n_gauss, P.n_cams, kd, int(colors_rest is not None), deg), dev)
        # (the ring's first call: an upper bound of what the takes below ask for -- every gradient tensor, the packed rows, the
        # direction gradient, means2d's own tensor)
        cvb.reserve(sum(4 * ((int(n) + 63) // 64 * 64) + 512 for n in (
            rows * MISPLAT_REC, colors.numel(), colors_rest.numel() if colors_rest is not None else 0,
            features.numel() if features is not None else 0, rows * 4 * nxq, means.numel(), means.numel(), quats.numel(),
            scales.numel(), opacities.numel(), rows * 2)))
        simple = v_means2d_in is None
        m2d_alive = ctx.means2d_ref() is not None
        # A steady-state backward finds its argument block and its gradient views on its arena slot (see _FwdPlan): keyed by
        # the forward's plan, the upstream gradients' addresses and what the forward left cleared.
        bp, bkey = None, None
        if (PLAN_CACHE and simple and KERNEL_EVENTS is None and KEY_TRACE is None and GRAD_SINK is None and cvb.slot is not None
                and bins.get("plan") is not None and (not ctx.absgrad or bins.get("v_abs_zero") is not None)):
            bkey = (bins["plan"], tuple(_dp(t) for t in ups), m2d_alive, bool(ctx.absgrad), bins.get("v_grec_zero") is not None,
                    bins.get("v_abs_zero") is not None, bins.get("v_featx_zero") is not None, bins.get("cap_isects"))
            bp = cvb.slot.get_plan(bkey)
        if bp is not None:
            # (the rows the forward left cleared are the FORWARD slot's: taken from this call's bins, never kept in this plan -- a
            # tensor of another slot held here would keep that slot busy for ever)
            v_grec, v_abs, v_featx = bins.pop("v_grec_zero", None), bins.pop("v_abs_zero", None), bins.pop("v_featx_zero", None)
            v_grec = bp.v_grec if v_grec is None else v_grec
            v_featx = bp.v_featx if v_featx is None else v_featx
            b, sparse, v_m2d, on_touch, flags = bp.b, bp.sparse, bp.v_m2d, bp.on_touch, bp.flags
            v_means, v_quats, v_scales, v_opac, v_colors, v_colors_rest, v_features = bp.grads
            cvb.slot.off, cvb.slot.demand = bp.off, bp.demand
            PATH_STATS["backward_plan_hit"] += 1
        else:
            c_before = cvb.slot._count() if bkey is not None else 0
            v_grec = bins.pop("v_grec_zero", None)
            flags = 1 if v_grec is not None else 0
            # (rows cleared on first touch by the forward: as good as cleared for a backward that reads flagged rows only)
            on_touch = bool(bins.get("rows_on_touch")) and v_grec is not None
            if v_grec is None:
                v_grec = cvb.take(rows * MISPLAT_REC, torch.float32).view(rows, MISPLAT_REC)
            v_abs = None
            if ctx.absgrad:
                v_abs = bins.pop("v_abs_zero", None)                # cleared by the forward's projection kernel
                if v_abs is None:
                    v_abs = torch.zeros(rows, 2, device=dev, dtype=torch.float32)
                flags |= 2
            v_colors = _grad_out(colors, cvb)
            v_colors_rest = _grad_out(colors_rest, cvb) if colors_rest is not None else None
            v_features = _grad_out(features, cvb) if features is not None else None
            v_featx = None
            if nxq > 0:
                v_featx = bins.pop("v_featx_zero", None)                 # cleared by the forward's colour stage
                if v_featx is not None:
                    flags |= 4
                else:
                    v_featx = cvb.take(rows * 4 * nxq, torch.float32).view(rows, 4 * nxq)
                if v_means2d_in is not None:
                    raise _lib.MisplatError("a gradient reached meta['means2d'] from outside the rasterizer in a call with more than four "
                                            "colour channels: that combination goes stage by stage only (MISPLAT_FUSED_NODE=0)")
            v_means_dir = cvb.take(means.numel(), torch.float32).view(means.shape) if deg >= 0 else None
            v_means, v_quats = _grad_out(means, cvb), _grad_out(quats, cvb)
            v_scales, v_opac = _grad_out(scales, cvb), _grad_out(opacities, cvb)
            perm = ctx.sched.perm_bwd if ctx.sched is not None else None
            by_view = ctx.sched.by_view if ctx.sched is not None else None
            # (a data-parallel gradient sink only changes where the six outputs are written: the slices of its flat buffer are
            # plain pointers like any other, so the one-call backward -- graph replay, both per-Gaussian stages in one launch,
            # zeros written in the background of the compositing backward -- serves it too)
            v_m2d = None
            if simple:
                b = RasterBwdArgs()
                b.Ks, b.grec, b.flatten_ids, b.offsets = _dp(Ks), _dp(grec), _dp(bins["flatten_ids"]), _dp(bins["isect_offsets"])
                # (the CAPACITY of the lists, not this call's count: the kernels take every range from `offsets`, and the count of
                # a scene that is being trained changes with every step -- it must not be part of the graph key)
                b.n_isects = bins.get("cap_isects") or bins["n_isects"]
                b.alpha, b.last_ids, b.median_ids, b.render = _dp(alpha), _dp(last_ids), _dp(median_ids), _dp(render)
                b.v_render, b.v_alpha, b.v_exp_depth, b.v_med_depth, b.v_normal = [_dp(t) for t in ups]
                b.v_grec, b.v_abs, b.unit_perm = _dp(v_grec), _dp(v_abs), _dp(perm)
                if by_view is not None:
                    b.unit_perm, b.unit_sel, b.unit_stride = _dp(by_view[0]), _dp(by_view[1]), by_view[2]
                    b.unit_slots = _order_slots(by_view[0], by_view[2])
                b.color_dim, b.zero_flags = cd, flags
                b.sh_degree, b.K_or_D, b.n_color, b.per_cam, b.depth_slot = deg, kd, n_color, per_cam, ctx.depth_slot
                b.means, b.quats, b.scales, b.opacities = _dp(means), _dp(quats), _dp(scales), _dp(opacities)
                b.colors, b.colors_rest, b.viewmats, b.radii = _dp(colors), _dp(colors_rest), _dp(viewmats), _dp(radii)
                b.compensations, b.sh_aux, b.v_means2d = _dp(comps), _dp(sh_aux), None
                b.v_colors, b.v_colors_rest, b.v_means_dir = _dp(v_colors), _dp(v_colors_rest), _dp(v_means_dir)
                b.v_means, b.v_quats, b.v_scales, b.v_opacities = _dp(v_means), _dp(v_quats), _dp(v_scales), _dp(v_opac)
                if nxq > 0:
                    b.nxq, b.featx, b.v_featx, b.depth_channel = nxq, _dp(bins["featx"]), _dp(v_featx), int(nd_depth)
                    b.depths = _dp(bins["depths"]) if bins["featx"] is None else None
                    b.features, b.v_features, b.n_feat = _dp(features), _dp(v_features), (features.shape[-1] if features is not None else 0)
                    b.zero_flags = flags
                if KERNEL_EVENTS is not None:                             # a measurement pass: events around the compositing backward
                    ev = _kernel_event_pair()
                    b.ev_blend_begin, b.ev_blend_end = ev[0].cuda_event, ev[1].cuda_event
                    KERNEL_EVENTS.setdefault("blend_bwd", []).append(ev)
                # meta["means2d"].grad, when someone still holds meta["means2d"]: a tensor of its own in the flagged-rows
                # backward (v_grec may then be defined in flagged rows only), a slice of v_grec otherwise
                m2d_alive = ctx.means2d_ref() is not None
                v_m2d = cvb.take(rows * 2, torch.float32).view(rows, 2) if m2d_alive else None
                b.v_means2d_out = _dp(v_m2d)
                sparse = int(lib.misplat_raster_bwd_plan(C.byref(P), C.byref(b)) & 1)
                if not sparse:
                    v_m2d, b.v_means2d_out = None, None
                if on_touch and not sparse:                               # every row is going to be read: clear them all first
                    b.zero_flags = flags & ~1 & (~4 if nxq > 0 else ~0)
                    PATH_STATS["backward_rows_refilled"] += 1
                if bkey is not None:
                    bp = _BwdPlan()
                    bp.key, bp.b, bp.sparse, bp.v_abs, bp.v_m2d = bkey, b, sparse, None, v_m2d
                    # (only what was carved from THIS slot: the rows the forward cleared come with every call's bins)
                    bp.v_grec = v_grec if bkey[4] is False else None
                    bp.v_featx = v_featx if (nxq > 0 and bkey[6] is False) else None
                    bp.on_touch, bp.flags = on_touch, flags
                    bp.grads = (v_means, v_quats, v_scales, v_opac, v_colors, v_colors_rest, v_features)
                    bp.off, bp.demand = cvb.slot.off, cvb.slot.demand
                    del v_means_dir
                    # (what refers to the backward's slot now is the plan's: the gradient views are handed out afresh below)
                    cvb.slot.put_plan(bkey, bp, cvb.slot._count() - c_before)
        if simple:
            with _timed("raster_bwd"):
                if KEY_TRACE is not None:
                    KEY_TRACE.append(("bwd", bytes(P) + bytes(b)))
                check(lib.misplat_raster_bwd(C.byref(P), C.byref(b), stream_ptr(), _graph_cache(dev)), "misplat_raster_bwd")
            PATH_STATS["backward_one_call"] += 1
            PATH_STATS["backward_background_fill"] += sparse
            if GRAD_SINK is not None:
                PATH_STATS["backward_sink"] += 1
                # (the row flags let the data-parallel reduce move only the rows that have a gradient: parallel.py)
                GRAD_SINK.rasterizer_done(bins.get("touched") if P.n_cams == 1 else None)
        else:
            # a gradient that reached means2d from another consumer: stage by stage
            PATH_STATS["backward_staged"] += 1
            if on_touch:
                flags &= ~1
                PATH_STATS["backward_rows_refilled"] += 1
            _with_perm = C.c_void_p(perm.data_ptr()) if perm is not None else None
            P.unit_perm = _with_perm
            if by_view is not None:
                P.unit_perm, P.unit_sel, P.unit_stride = by_view[0].data_ptr(), by_view[1].data_ptr(), by_view[2]
                P.unit_slots = _order_slots(by_view[0], by_view[2])
            check(lib.misplat_blend_bwd_atomic(C.byref(P), C.c_int32(cd), ptr(Ks), ptr(grec), ptr(bins["flatten_ids"]),
                                               ptr(bins["isect_offsets"]), C.c_int64(bins["n_isects"]), ptr(alpha),
                                               ptr(last_ids), ptr(median_ids), ptr(render), *[ptr(t) for t in ups],
                                               ptr(v_grec), ptr(v_abs), C.c_int32(flags), stream_ptr()),
                  "misplat_blend_bwd_atomic")
            P.unit_perm, P.unit_sel, P.unit_stride, P.unit_slots = None, None, 0, 0
            check(lib.misplat_color_bwd(C.byref(P), C.c_int32(deg), C.c_int32(kd), C.c_int32(n_color),
                                        C.c_int32(per_cam), ptr(means), ptr(viewmats), ptr(colors), ptr(colors_rest),
                                        ptr(radii), ptr(v_grec), ptr(v_colors), ptr(v_colors_rest), ptr(v_means_dir),
                                        ptr(sh_aux), stream_ptr()), "misplat_color_bwd")
            if GRAD_SINK is not None:
                GRAD_SINK.colour_ready()
            vm2d = None
            if v_means2d_in is not None:
                vm2d = (v_grec.view(P.n_cams, P.n_gauss, MISPLAT_REC)[..., 0:2] + v_means2d_in).contiguous()
            check(lib.misplat_project_pack_bwd(C.byref(P), C.c_int32(ctx.depth_slot), ptr(means), ptr(quats),
                                               ptr(scales), ptr(opacities), ptr(viewmats), ptr(Ks), ptr(radii),
                                               ptr(comps), ptr(vm2d), ptr(v_grec), ptr(v_means_dir), ptr(v_means),
                                               ptr(v_quats), ptr(v_scales), ptr(v_opac), None, C.c_int32(0), stream_ptr()),
                  "misplat_project_pack_bwd")
        cvb.done()
        # gsplat's contract: the screen-space gradient rides on meta["means2d"]
        m2d = ctx.means2d_ref()
        if m2d is not None:
            if simple and v_m2d is not None:
                g2d = v_m2d.view(P.n_cams, P.n_gauss, 2)
            else:
                g2d = v_grec.view(P.n_cams, P.n_gauss, MISPLAT_REC)[..., 0:2]
            m2d.grad = g2d if v_means2d_in is None else g2d + v_means2d_in
            if ctx.absgrad:
                m2d.absgrad = v_abs.view(P.n_cams, P.n_gauss, 2)
        if simple and bp is not None:
            # (the plan keeps its gradient tensors: autograd adopts an incoming gradient without a copy only if nobody else holds
            # it -- so what is returned are fresh views of them)
            v_means, v_quats, v_scales, v_opac, v_colors, v_colors_rest, v_features = [
                None if t is None else t.view(t.shape) for t in (v_means, v_quats, v_scales, v_opac, v_colors, v_colors_rest, v_features)]
        return (v_means, v_quats, v_scales, v_opac, v_colors, v_colors_rest, v_features, None, None, None, None, None, None, None,
                None)


def raster_fused(means, quats, scales, opacities, colors, viewmats, Ks, P: Params, sh_degree, depth_channel: bool,
                 cd: int, absgrad: bool, features: Optional[Tensor] = None):
    """(render, alpha, exp_depth, med_depth, normal, means2d, radii, depths, comps, grec, last_ids, median_ids), bins.
    ``features`` [N,F] (with SH ``colors``): the features model's call as one entry -- channels 0..2 = max(SH + 0.5, 0),
    channels 3.. = the features (rade_features_model.py:427-441), ``cd`` = 3 + F (+ 1 with a depth channel)."""
    rest = None
    if isinstance(colors, (tuple, list)):
        colors, rest = colors
        rest = _f32(rest, "features_rest")
    args = [_f32(t, n) for t, n in ((means, "means"), (quats, "quats"), (scales, "scales"),
                                    (opacities, "opacities"), (colors, "colors"))]
    extra: dict = {}
    feat = None if features is None else _f32(features, "features")
    out = _RasterFused.apply(*args, rest, feat, _f32(viewmats, "viewmats"), _f32(Ks, "Ks"), P, sh_degree, bool(depth_channel),
                             int(cd), bool(absgrad), extra)
    return out, extra["bins"]


def _upstream(P: Params, cd: int, dev, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal):
    """Contiguous upstream gradients of the five images; an output that took no part in the loss (None: the
    nodes do not materialise gradients) contributes zeros."""
    widths = (cd, 1, 1, 1, 3)
    return [_c(t) if t is not None else torch.zeros(P.n_cams, P.height, P.width, w, device=dev, dtype=torch.float32)
            for t, w in zip((v_render, v_alpha, v_exp_depth, v_med_depth, v_normal), widths)]

# ----------------------------------------------------------------------------- the other nodes
# The stage-by-stage nodes live in ops_stages.py, the model-level ones in ops_epilogue.py; ``ops.X`` resolves there on first use
# (lazily: ops_stages reads this module's switches, so neither can import the other's names at import time).
def __getattr__(name: str):
    if name.startswith("__"):
        raise AttributeError(name)
    from . import ops_epilogue, ops_stages
    for m in (ops_stages, ops_epilogue):
        if name in m.__dict__:
            return m.__dict__[name]
    raise AttributeError(f"module 'collab_splats_amd.ops' has no attribute {name!r}")
