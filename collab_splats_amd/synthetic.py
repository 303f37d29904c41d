"""Deterministic synthetic scenes (SURVEY.md section 8(d) / BASELINE.md section 4).

Generated on the CPU with ``torch.Generator().manual_seed(seed)`` and copied to the device, so RNG
implementation differences between hosts cannot matter.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def random_scene(n: int, width: int, height: int, seed: int = 42, sh_degree: int = 3,
                 device: str = "cpu") -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    fx = fy = 0.9 * width
    z = torch.rand(n, generator=g) * 10.0 + 2.0
    x = (torch.rand(n, generator=g) * 2 - 1) * 1.15 * z * (width / 2) / fx
    y = (torch.rand(n, generator=g) * 2 - 1) * 1.15 * z * (height / 2) / fy
    means = torch.stack([x, y, z], dim=-1)
    log_s = torch.rand(n, 3, generator=g) * (math.log(0.04) - math.log(0.004)) + math.log(0.004)
    flat = torch.randint(0, 3, (n,), generator=g)
    log_s[torch.arange(n), flat] += math.log(0.1)
    quats = torch.randn(n, 4, generator=g)
    opacity_logits = torch.rand(n, generator=g) * 6.0 - 2.0
    K = (sh_degree + 1) ** 2
    sh = torch.empty(n, K, 3)
    sh[:, 0] = torch.rand(n, 3, generator=g) * 3.0 - 1.5
    if K > 1:
        sh[:, 1:] = torch.randn(n, K - 1, 3, generator=g) * 0.1
    viewmat = torch.eye(4)[None]
    Ks = torch.tensor([[[fx, 0.0, width / 2.0], [0.0, fy, height / 2.0], [0.0, 0.0, 1.0]]])
    out = dict(means=means, log_scales=log_s, quats=quats, opacity_logits=opacity_logits, sh=sh,
               viewmats=viewmat, Ks=Ks)
    return {k: v.to(device).contiguous() for k, v in out.items()}


def view_matrix(view_index: int, n_views: int = 8) -> torch.Tensor:
    """8 cameras rotated +-20 degrees about x/y around the point (0,0,7) (config 4)."""
    ang = math.radians(20.0)
    k = view_index % 8
    ax, ay = [(0, 0), (ang, 0), (-ang, 0), (0, ang), (0, -ang), (ang, ang), (-ang, -ang), (ang, -ang)][k]
    cx, sx, cy, sy = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay)
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float32)
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float32)
    R = Rx @ Ry
    pivot = torch.tensor([0.0, 0.0, 7.0])
    t = pivot - R @ pivot
    V = torch.eye(4)
    V[:3, :3] = R
    V[:3, 3] = t
    return V[None]
