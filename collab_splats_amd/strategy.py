"""``gsplat.strategy.DefaultStrategy`` surface used by the reference
(/root/reference/collab_splats/models/rade_gs_model.py:19, 191-198, 456-458): an ``isinstance``
check, the ``absgrad`` flag and ``step_pre_backward(params, optimizers, state, step, info)``.

The densification controller itself (split / clone / prune / opacity reset) is the step on the
other side of the rasterizer and is NOT part of this round: only the
statistics it would consume are gathered here, from the rasterizer's ``meta``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict

import torch


@dataclass
class DefaultStrategy:
    absgrad: bool = False
    key_for_gradient: str = "means2d"
    refine_start_iter: int = 500
    refine_stop_iter: int = 15_000
    verbose: bool = False

    def initialize_state(self, scene_scale: float = 1.0) -> Dict[str, Any]:
        return {"grad2d": None, "count": None, "radii": None, "scene_scale": scene_scale}

    def check_sanity(self, params, optimizers) -> None:
        for key in ("means", "scales", "quats", "opacities"):
            assert key in params, f"{key} is required in params but missing."

    def step_pre_backward(self, params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any]) -> None:
        """Keep the 2-D mean gradient: ``info["means2d"]`` is a non-leaf of the autograd graph."""
        assert self.key_for_gradient in info, "The 2D means of the Gaussians is required but missing."
        info[self.key_for_gradient].retain_grad()

    @torch.no_grad()
    def step_post_backward(self, params, optimizers, state: Dict[str, Any], step: int, info: Dict[str, Any],
                           packed: bool = False) -> None:
        """Accumulate the per-Gaussian screen-space gradient norm (the densification statistic)."""
        if step >= self.refine_stop_iter:
            return
        m2d = info[self.key_for_gradient]
        grads = (m2d.absgrad if self.absgrad else m2d.grad).clone()
        grads[..., 0] *= info["width"] / 2.0 * info["n_cameras"]
        grads[..., 1] *= info["height"] / 2.0 * info["n_cameras"]
        sel = (info["radii"] > 0).any(dim=-1)                      # [C, N]
        n = m2d.shape[1]
        if state["grad2d"] is None:
            state["grad2d"] = torch.zeros(n, device=grads.device)
            state["count"] = torch.zeros(n, device=grads.device)
        norm = grads.norm(dim=-1) * sel
        state["grad2d"] += norm.sum(0)
        state["count"] += sel.sum(0).to(state["count"].dtype)
