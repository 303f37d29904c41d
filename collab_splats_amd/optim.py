"""Fused multi-tensor Adam (HIP) for the Gaussian parameter groups -- SURVEY.md section 8(f) rank 3.

The reference trains with one ``torch.optim.Adam`` per parameter group (nerfstudio ``AdamOptimizerConfig``,
``/root/reference/collab_splats/configs/rade_gs_method.py:44-71``: lr per group, ``eps=1e-15``).  ``FusedAdam``
keeps that interface (``param_groups``, ``state[p]["exp_avg"|"exp_avg_sq"|"step"]``, so the densification
strategy's optimizer-state surgery works unchanged) and ``step_all`` updates every group of every optimizer in
ONE kernel launch (``misplat_adam_step``) instead of ~6 x 5 elementwise kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Tuple

import torch

from . import _lib
from ._lib import check, stream_ptr

MAX_TENSORS = 8                                  # MISPLAT_ADAM_MAX_TENSORS


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` semantics (no weight decay, no amsgrad, maximize=False), fp32 CUDA parameters."""

    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    def _entries(self) -> List[tuple]:
        out = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda:
                    raise _lib.MisplatError("FusedAdam: parameters must be float32 tensors on the GPU (no CPU fallback)")
                if p.grad.is_sparse:
                    raise _lib.MisplatError("FusedAdam: sparse gradients are not supported")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                out.append((p, st, float(group["lr"]), float(b1), float(b2), float(group["eps"])))
        return out

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _launch(self._entries())
        return loss


@torch.no_grad()
def step_all(optimizers: Iterable[FusedAdam] | Dict[str, FusedAdam]) -> None:
    """One fused launch (per distinct (betas, eps)) over all parameters of several ``FusedAdam`` instances."""
    if isinstance(optimizers, dict):
        optimizers = optimizers.values()
    entries: List[tuple] = []
    for opt in optimizers:
        if not isinstance(opt, FusedAdam):
            raise TypeError("step_all() takes FusedAdam optimizers")
        entries += opt._entries()
    _launch(entries)


def _launch(entries: List[tuple]) -> None:
    if not entries:
        return
    lib = _lib.load()
    by_hyper: Dict[tuple, List[tuple]] = {}
    for e in entries:
        by_hyper.setdefault(e[3:], []).append(e)
    for (b1, b2, eps), es in by_hyper.items():
        for i in range(0, len(es), MAX_TENSORS):
            chunk = es[i:i + MAX_TENSORS]
            n = len(chunk)
            ps, gs, ms, vs, numel, lrs, steps = [], [], [], [], [], [], []
            keep = []                                    # contiguous gradient copies must outlive the launch
            for p, st, lr, _, _, _ in chunk:
                if not p.is_contiguous() or not st["exp_avg"].is_contiguous() or not st["exp_avg_sq"].is_contiguous():
                    raise _lib.MisplatError("FusedAdam: parameters and their moments must be contiguous")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g.dtype != torch.float32 or g.shape != p.shape:
                    raise _lib.MisplatError("FusedAdam: gradient must be float32 with the parameter's shape")
                keep.append(g)
                st["step"] += 1
                ps.append(p.data_ptr()); gs.append(g.data_ptr())
                ms.append(st["exp_avg"].data_ptr()); vs.append(st["exp_avg_sq"].data_ptr())
                numel.append(p.numel()); lrs.append(lr); steps.append(int(st["step"].item()))
            vp = C.c_void_p * n
            check(lib.misplat_adam_step(C.c_int32(n), vp(*ps), vp(*gs), vp(*ms), vp(*vs), (C.c_int64 * n)(*numel),
                                        (C.c_float * n)(*lrs), (C.c_int64 * n)(*steps), C.c_double(b1), C.c_double(b2),
                                        C.c_double(eps), stream_ptr()), "misplat_adam_step")
            del keep
