"""Whole-step hipGraph capture for small scenes.

At 10 k Gaussians / 256 x 256 the kernels of one forward + backward take ~0.14 ms, the eager PyTorch step around them
0.25 - 0.45 ms: autograd bookkeeping, the caller's ``exp`` / ``sigmoid``, and the one host wait for the intersection
count.  With a FIXED intersection capacity (``ops.static_capacity``) the rasterizer needs nothing from the host, so the
standard PyTorch recipe applies: run the step a few times, capture it once, replay it.

    step = GraphedStep(fn, capacity=200_000)      # fn(): zero grads, forward, loss, backward -- on static tensors
    for it in range(n):
        step.replay()                              # one hipGraphLaunch
        optimizer.step()
        if it % 100 == 0:
            step.check()                           # (synchronises) raises if the intersection count outgrew `capacity`

``fn`` follows the usual rules of CUDA-graph capture in PyTorch: static input tensors updated in place, and nothing with
autograd history stashed across calls (return / keep detached tensors: an autograd graph kept alive from the previous
iteration makes PyTorch run its AccumulateGrad nodes on the wrong stream and the capture crashes the process).  What
``fn`` RETURNS is checked: a result that carries autograd history raises ``MisplatError`` during the warm-up, before any
capture has begun.

The reference's trainer (nerfstudio) runs eagerly; this is an extension on the caller's side of the boundary, not part
of the drop-in surface.  Densification changes tensor shapes: re-create the GraphedStep after a refinement step.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

import types

from . import ops


# (the text of PyTorch's warning for an AccumulateGrad node that runs on a stream other than the one its graph was built on;
# tests/test_host.py checks that the installed torch still says this -- the secondary net behind _tensors_with_history)
STALE_GRAPH_WARNING = "AccumulateGrad node's stream does not match"


def _tensors_with_history(obj, path="result", seen=None):
    """Paths of the tensors inside ``obj`` (lists / tuples / dicts / objects with __dict__ are walked) that carry autograd
    history."""
    seen = set() if seen is None else seen
    if id(obj) in seen:
        return []
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        return [path] if obj.grad_fn is not None else []
    out = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            out += _tensors_with_history(v, f"{path}[{k!r}]", seen)
    elif isinstance(obj, (list, tuple, set, frozenset)):
        for i, v in enumerate(obj):
            out += _tensors_with_history(v, f"{path}[{i}]", seen)
    elif hasattr(obj, "__dict__") and not isinstance(obj, (type, types.ModuleType, types.FunctionType, types.MethodType)):
        for k, v in vars(obj).items():                       # plain objects, dataclasses, SimpleNamespace: a `meta` holder
            out += _tensors_with_history(v, f"{path}.{k}", seen)
    return out


class GraphedStep:
    def __init__(self, fn: Callable[[], object], capacity: int, warmup: int = 3, device: Optional[torch.device] = None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.capacity = int(capacity)
        if not ops.fused_node_ok():
            raise ops._lib.MisplatError("GraphedStep needs the default single-node path (MISPLAT_FUSED / MISPLAT_FUSED_NODE / "
                                        "atomic gradient mode)")
        # the count of every replay lands in a pinned word of this graph's own (advisor, round 4: with one word per device a
        # view's overflow was overwritten by the next view's replay before anybody looked)
        self._count = torch.zeros(1, dtype=torch.int64, pin_memory=True)
        self._worst = 0
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        # An autograd graph kept alive across calls (returned by fn, or stashed anywhere else: a dict of "last outputs", a
        # meta object) makes PyTorch run the parameters' AccumulateGrad nodes on the stream of the call that created
        # them: inside a capture that is a cross-stream dependency and the process dies in capture_end (observed:
        # SIGSEGV).  Both forms are refused here, during the warm-up, before any capture has begun: what fn returns is
        # inspected, and PyTorch's own stream-mismatch warning (made to fire every time for the duration) is an error.
        import warnings
        stale = []
        warn_always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)
        try:
            with warnings.catch_warnings(record=True) as caught, torch.cuda.stream(side), ops.static_capacity(self.capacity, self._count):
                warnings.simplefilter("always")
                for _ in range(max(warmup, 1)):               # allocator warm-up, launch-order feedback, lazy initialisation
                    res = fn()
                    bad = _tensors_with_history(res)
                    del res
                    if bad:
                        torch.cuda.current_stream(self.device).wait_stream(side)
                        raise ops._lib.MisplatError(
                            "GraphedStep: fn() returned tensor(s) with autograd history (" + ", ".join(bad[:4])
                            + ("..." if len(bad) > 4 else "") + "): return detached tensors (t.detach()) or nothing -- "
                            "a graph of the previous iteration that is still alive cannot be captured")
                stale = [w for w in caught if STALE_GRAPH_WARNING in str(w.message)]
        finally:
            torch.set_warn_always(warn_always)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        if stale:
            raise ops._lib.MisplatError(
                "GraphedStep: an autograd graph of an EARLIER call on these parameters is still alive (something keeps a "
                "tensor with autograd history: outputs, a meta dict, a loss) -- PyTorch reported AccumulateGrad nodes on a "
                "foreign stream during the warm-up, and capturing in that state crashes the process.  Drop or detach those "
                "references before creating the GraphedStep")
        self._check_count()                                   # too small already: fail before capturing
        self.graph = torch.cuda.CUDAGraph()
        self._keep: list = []
        ops._CAPTURE_KEEP = self._keep
        try:
            with ops.static_capacity(self.capacity, self._count), torch.cuda.graph(self.graph):
                self.result = fn()
        finally:
            ops._CAPTURE_KEEP = None

    def replay(self):
        self.graph.replay()
        return self.result

    def _check_count(self) -> int:
        n = int(self._count[0])
        if n > self.capacity:
            raise ops._lib.MisplatError(f"{n} tile intersections exceed the fixed capacity {self.capacity}: rerun with a larger one")
        return max(n, 0)

    def check(self) -> int:
        """Synchronise and return the intersection count of this graph's last replay; raises if it exceeded the capacity
        (the images and gradients of that replay are then incomplete)."""
        torch.cuda.synchronize(self.device)
        return self._check_count()


class GraphedViews:
    """One whole-step graph per RESIDENT camera: a trainer that keeps its views on the device replays ``fn(v)`` -- activations,
    rasterization of view ``v``, loss, backward -- as one hipGraph per view, in whatever order it visits them.  ``fn(v)`` obeys
    GraphedStep's rules (no tensor with autograd history may outlive the call; parameters are updated in place) and one more:
    gradients must live in STATIC tensors that every graph accumulates into -- clear them with ``p.grad.zero_()`` inside
    ``fn``, not with ``p.grad = None`` (a replay does not run Python: a ``.grad`` attribute assigned while one view's graph was
    captured does not follow the replays of another's), or put the optimiser step inside ``fn``.  ``capacity`` covers the
    largest view.  ``bench.py --graphed`` on the cycling views: 0.14 ms per step at 10 k / 256^2, 0.38 ms at
    100 k / 1080p, where the eager loop is host-bound at 0.5 ms."""

    def __init__(self, fn: Callable[[int], object], n_views: int, capacity: int, warmup: int = 3,
                 device: Optional[torch.device] = None):
        self.capacity = int(capacity)
        self.steps = [GraphedStep((lambda v=v: fn(v)), capacity, warmup=warmup, device=device) for v in range(int(n_views))]

    def __len__(self) -> int:
        return len(self.steps)

    def replay(self, v: int):
        return self.steps[v].replay()

    def check(self, v: Optional[int] = None) -> int:
        """Synchronise; the intersection count of view ``v``'s last replay, or (``v`` None) the largest among all views' last
        replays -- raises if ANY of them exceeded the capacity: every graph has its own pinned count."""
        torch.cuda.synchronize(self.steps[0].device)
        worst = max(st._check_count() for st in self.steps)
        return worst if v is None else self.steps[v]._check_count()
