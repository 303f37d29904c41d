"""Per-call device buffers from a small ring of persistent arenas (plumbing; no kernels here).

One ``rasterization()`` call needs ~25 scratch / output arrays (packed records, bucketing workspace, images, per-tile lists).
Taken from PyTorch's caching allocator they land wherever it has room: the launch arguments of two identical steps then
differ in a dozen pointers (a hipGraph captured for one step is useless for the next: DESIGN.md section 8), and the 64-byte
record gathers of the compositing kernels run 10 - 15 % slower in some placements than in others (section 13.7).  So the
arrays of a call are carved from ONE persistent slot of a ring kept per (device, stream, shape): the same call gets the same
addresses every time (the placement of the arrays inside a slot: see ALIGN_BIG below -- measured, not assumed).

Ownership: the tensors handed out are views of the slot's storage, and a slot is handed out again only when NOTHING refers to
its storage any more (``torch._C._storage_Use_Count``: outputs the caller still holds, tensors saved for a backward that has not
run, a ``meta`` dict kept for later all count) -- a call that finds every slot of its ring in use gets a new slot (up to
``MAX_SLOTS``) or plain allocator memory.  Reuse needs no event: a ring belongs to one stream, and later work on a stream is
ordered behind earlier work on it, exactly the rule by which the caching allocator itself hands a freed block out again.
"""
from __future__ import annotations

import collections
import os
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

ENABLED = os.environ.get("MISPLAT_ARENA", "1") == "1"
MAX_SLOTS = 4
MAX_RINGS = 12      # distinct (stream, shape, role) keys kept; LRU beyond
MAX_PLANS = 32      # cached call plans per slot (ops._FwdPlan / _BwdPlan: one per resident camera of a training loop)
# A ring nobody has asked for in this many lookups is dropped: densification changes the number of Gaussians every few hundred
# steps and never comes back to the old one (scripts/soak.py: +0.7 GB of reserved memory per change at 1 M without this).
IDLE_LOOKUPS = 64
# Placement inside a slot.  Measured (5 M Gaussians / 1080p, one box): with every array of >= 1 MiB on a 2 MiB boundary
# the kernels that write several arrays at the same element offset (bucket_rows: order / rect_sorted / depth_sorted at
# [pos]) ran 2 x slower (74 -> 182 us; their streams then map to the same memory channels) and the step lost 0.1 ms; the
# 1 M step lost 4 %.  So arrays are packed at 256-byte granularity, as a one-allocation carve always was -- what the arena
# contributes is that the addresses REPEAT (graph replay), not where they are.
BIG = 1 << 20
ALIGN_SLOT = 2 << 20              # the slot itself starts on a 2 MiB boundary
ALIGN_BIG = 256
ALIGN_SMALL = 256
STATS: "collections.Counter" = collections.Counter()             # slots_created / slot_hits / fallbacks / regrown
# The slots (and the allocator memory a ring's FIRST call runs from) come from a memory pool of their own
# (torch.cuda.MemPool): creating a 200 MB slot, or freeing the first call's arrays, then does not reshuffle the free blocks the
# CALLER's tensors come from -- the torch.exp / torch.sigmoid outputs of the reference's call kept moving for two steps after
# our slots appeared, and every address that moves is a graph key that does not recur.
# ONE POOL PER RING (round 5; round 4 had one per device).  A private pool never gives a freed block back to the device while the
# pool object lives, and densification makes every ring obsolete sooner or later (N only grows): with one pool per device every
# slot ever created stayed reserved (advisor, round 4: 4.25 GB reserved against 0.79 GB allocated after two growths at 400 k).
# A ring's pool dies with the ring; the allocator may then hand its blocks to anyone on an out-of-memory retry, and ``trim()``
# -- called from ``ring()`` when the next NEW shape arrives with a dropped ring on record -- returns them to the device
# (torch.cuda.empty_cache(): a synchronisation, paid once per refinement step of a densifying run, at the moment every
# address changes anyway; at most two generations of rings are held).
USE_POOL = True
# ... and only once the dropped rings add up to this much (a trim also moves the blocks the CALLER's temporaries come from --
# one round of graph captures for a steady loop that happens to see it: not worth it for a few megabytes of small scenes)
TRIM_BYTES = int(float(os.environ.get("MISPLAT_ARENA_TRIM_MB", "256")) * (1 << 20))
_TRIM_PENDING = 0                                                 # bytes of the slots of dropped rings not yet returned


def _new_pool(dev: torch.device):
    if not USE_POOL or not hasattr(torch.cuda, "MemPool"):
        return None
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        return torch.cuda.MemPool()


def pool_ctx(dev: torch.device, pool=None):
    """Context in which allocations of this thread on ``dev`` come from ``pool`` (a ring's private pool; a no-op context when
    there is none or a graph is being captured: a capture has a pool of its own)."""
    import contextlib
    if pool is None or torch.cuda.is_current_stream_capturing():
        return contextlib.nullcontext()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return torch.cuda.use_mem_pool(pool, device=idx)


def trim() -> None:
    """Give the memory of dropped rings back to the device (their pools are gone: their cached blocks are free to go)."""
    global _TRIM_PENDING
    if _TRIM_PENDING >= max(TRIM_BYTES, 1) and not torch.cuda.is_current_stream_capturing():
        _TRIM_PENDING = 0
        torch.cuda.empty_cache()
        STATS["trims"] += 1


_use_count = getattr(torch._C, "_storage_Use_Count", None)
_ESIZE = {torch.float32: 4, torch.int32: 4, torch.uint8: 1, torch.int64: 8, torch.int16: 2, torch.float64: 8}


class Slot:
    """One arena: a 2 MiB-aligned byte range carved front to back by ``take``; ``demand`` records what a call wanted."""

    def __init__(self, dev: torch.device, nbytes: int, pool=None):
        self.size = int(nbytes)
        with pool_ctx(dev, pool):
            self.raw = torch.empty(self.size + ALIGN_SLOT, device=dev, dtype=torch.uint8)
        shift = (-self.raw.data_ptr()) % ALIGN_SLOT
        self.base = self.raw[shift:shift + self.size]
        # one typed view of the whole slot per element size: an array is then ONE slice (host time: ~60 arrays per step)
        self.typed = {torch.uint8: self.base, torch.float32: self.base.view(torch.float32), torch.int32: self.base.view(torch.int32),
                      torch.int64: self.base.view(torch.int64), torch.float64: self.base.view(torch.float64),
                      torch.int16: self.base.view(torch.int16)}
        self.storage = self.raw.untyped_storage()                 # (kept: the use count below then has a fixed floor)
        self.floor = self._count()                                # raw + base + the typed views + the storage wrapper
        self.off = 0
        self.demand = 0
        # A steady-state call may leave its carved views and filled argument blocks here for its next visit (ops._FwdPlan /
        # _BwdPlan): `plan_refs` = how many tensors of the plan refer to this storage -- they are the library's own and do not
        # make the slot busy.
        self.plans: "collections.OrderedDict" = collections.OrderedDict()     # key -> plan (a plan has `.refs`); LRU of MAX_PLANS
        self.plan_refs = 0

    def _count(self) -> int:
        return int(_use_count(self.storage._cdata))

    def free(self) -> bool:
        return self._count() <= self.floor + self.plan_refs

    def get_plan(self, key):
        p = self.plans.get(key)
        if p is not None:
            self.plans.move_to_end(key)
        return p

    def put_plan(self, key, plan, refs: int) -> None:
        """``refs`` = how many tensors of ``plan`` refer to this slot's storage (measured by the caller around the build)."""
        self.drop_plan(key)
        plan.refs = int(refs)
        self.plans[key] = plan
        self.plan_refs += plan.refs
        while len(self.plans) > MAX_PLANS:                        # (a trainer's resident cameras: one plan per view)
            _, old = self.plans.popitem(last=False)
            self.plan_refs -= old.refs

    def drop_plan(self, key=None) -> None:
        if key is None:
            self.plans.clear()
            self.plan_refs = 0
        else:
            old = self.plans.pop(key, None)
            if old is not None:
                self.plan_refs -= old.refs

    def begin(self) -> None:
        self.off = 0
        self.demand = 0

    def take(self, count: int, dtype: torch.dtype) -> Optional[Tensor]:
        """``count`` elements of ``dtype`` from the slot, or None when it is full (the caller then asks the allocator)."""
        esize = _ESIZE[dtype]
        nbytes = int(count) * esize
        align = ALIGN_BIG if nbytes >= BIG else ALIGN_SMALL
        start = (self.off + align - 1) // align * align
        self.demand = start + nbytes                                        # (what a slot must hold to serve this call)
        if start + nbytes > self.size:
            self.off = start + nbytes                                       # keep counting: the demand of the whole call
            return None
        self.off = start + nbytes
        first = start // esize                                              # (align is a multiple of every element size)
        return self.typed[dtype][first:first + int(count)]


class Ring:
    def __init__(self, dev: torch.device):
        self.dev = dev
        self.slots: List[Slot] = []
        self.want = 0                                             # bytes the largest call so far asked for
        self.last_lookup = 0
        self.pool = _new_pool(dev)                                # (dies with the ring: see USE_POOL)

    def acquire(self) -> Optional[Slot]:
        """A slot nothing refers to (regrown first if the last call did not fit), a new one, or None."""
        for i, s in enumerate(self.slots):
            if not s.free():
                continue
            if s.size < self.want:                                # the shape's demand grew (a larger capacity class)
                self.slots[i] = s = Slot(self.dev, self._padded(self.want), self.pool)
                STATS["regrown"] += 1
            else:
                STATS["slot_hits"] += 1
            s.begin()
            return s
        if self.want > 0 and len(self.slots) < MAX_SLOTS:
            s = Slot(self.dev, self._padded(self.want), self.pool)
            self.slots.append(s)
            STATS["slots_created"] += 1
            s.begin()
            return s
        return None

    @staticmethod
    def _padded(n: int) -> int:
        """Slot size for a demand of ``n`` bytes: the next step of a geometric ladder (x 1.25 from 2 MiB) above n + 10 % -- a
        slot regrown for a slightly larger capacity class then has the size of a block its pool already caches more often
        than not, and a ring's memory stays within a constant factor of its demand."""
        need = int(n * 1.10)
        size = ALIGN_SLOT
        while size < need:
            size = (int(size * 1.25) + ALIGN_SLOT - 1) // ALIGN_SLOT * ALIGN_SLOT
        return size

    def nbytes(self) -> int:
        return sum(s.size for s in self.slots)

    def release(self, slot: Optional[Slot], demand: int) -> None:
        """What the call asked for in total (also when it had no slot): the next slot of this ring is sized for it."""
        self.want = max(self.want, int(demand))


_RINGS: "collections.OrderedDict[tuple, Ring]" = collections.OrderedDict()
_LOOKUPS = 0


def ring(key: tuple, dev: torch.device) -> Optional[Ring]:
    """The ring of this key (LRU over ``MAX_RINGS`` keys), or None when arenas are off or a graph is being captured (memory
    allocated during a capture belongs to that graph's private pool and must not outlive it in a ring)."""
    if not ENABLED or _use_count is None or torch.cuda.is_current_stream_capturing():
        return None
    global _LOOKUPS
    _LOOKUPS += 1
    global _TRIM_PENDING
    r = _RINGS.get(key)
    if r is None:
        trim()                                                    # (a new shape: the moment abandoned rings' memory is wanted)
        r = _RINGS[key] = Ring(dev)
        while len(_RINGS) > MAX_RINGS:
            _TRIM_PENDING += _RINGS.popitem(last=False)[1].nbytes()   # (its slots live on while views of them do)
    else:
        _RINGS.move_to_end(key)
    r.last_lookup = _LOOKUPS
    if IDLE_LOOKUPS > 0:                                          # (oldest first: the dict is in LRU order)
        while len(_RINGS) > 1:
            k0 = next(iter(_RINGS))
            if _LOOKUPS - _RINGS[k0].last_lookup <= IDLE_LOOKUPS:
                break
            _TRIM_PENDING += _RINGS.pop(k0).nbytes()
            STATS["rings_dropped_idle"] += 1
                                                                  # (returned to the device when the next new shape arrives: a
                                                                  #  trim moves the caller's allocator blocks too, i.e. costs
                                                                  #  the steady state a round of graph captures -- at a shape
                                                                  #  change everything is new anyway)
    return r


class Carver:
    """The allocations of one call: from a ring slot where one is free and large enough, from the allocator otherwise.
    ``done()`` tells the ring how much the call wanted, so that the next call of the shape finds a slot that fits."""

    def __init__(self, key: tuple, dev: torch.device):
        self.dev = dev
        self.ring = ring(key, dev) if key is not None else None    # (key None: a throw-away call -- plain allocator memory)
        self.slot = self.ring.acquire() if self.ring is not None else None
        self.shadow = 0                                           # demand counted when there is no slot

    def reserve(self, nbytes: int) -> None:
        """Before the first ``take`` of a ring's FIRST call: the caller's own estimate of the call's demand (an upper bound
        is fine; an estimate that falls short only means allocator memory for what does not fit, as without it).  The ring
        then owns a slot from call one -- the addresses, and with them the graph keys, of the first call are those of
        every later one."""
        if self.ring is not None and self.slot is None and self.ring.want == 0 and not self.ring.slots and nbytes > 0:
            self.ring.want = int(nbytes)
            self.slot = self.ring.acquire()

    def take(self, count: int, dtype: torch.dtype) -> Tensor:
        count = int(count)
        if self.slot is not None:
            t = self.slot.take(count, dtype)
            if t is not None:
                return t
            STATS["fallbacks"] += 1
        elif self.ring is not None:
            nbytes = count * _ESIZE[dtype]
            align = ALIGN_BIG if nbytes >= BIG else ALIGN_SMALL
            self.shadow = (self.shadow + align - 1) // align * align + nbytes
            with pool_ctx(self.dev, self.ring.pool):              # (a ring's first call: see USE_POOL)
                return torch.empty(max(count, 0), device=self.dev, dtype=dtype)
        return torch.empty(max(count, 0), device=self.dev, dtype=dtype)

    def carve(self, sizes: Sequence[int], dtype: torch.dtype) -> list:
        return [self.take(n, dtype) for n in sizes]

    def done(self) -> None:
        if self.ring is not None:
            self.ring.release(self.slot, self.slot.demand if self.slot is not None else self.shadow)

    @property
    def slot_id(self) -> int:
        return -1 if self.slot is None else self.ring.slots.index(self.slot)


def reset() -> None:
    global _TRIM_PENDING
    _TRIM_PENDING += sum(r.nbytes() for r in _RINGS.values())
    _RINGS.clear()
    if torch.cuda.is_available():
        trim()
