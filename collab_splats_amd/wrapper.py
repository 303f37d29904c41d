"""``gsplat.cuda._wrapper``-compatible entry points, as the reference imports them
(/root/reference/collab_splats/models/rade_gs_model.py:20, rade_features_model.py:20)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import make_params


def fully_fused_projection(
    means: Tensor, covars: Optional[Tensor], quats: Optional[Tensor], scales: Optional[Tensor],
    viewmats: Tensor, Ks: Tensor, width: int, height: int, eps2d: float = 0.3,
    near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, packed: bool = False,
    sparse_grad: bool = False, calc_compensations: bool = False, camera_model: str = "pinhole",
    opacities: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor], Tensor, Tensor, Tensor]:
    """Signature and 8-tuple pinned by rade_gs_model.py:373-394:
    ``radii[C,N,2], means2d[C,N,2], depths[C,N], conics[C,N,3], compensations[C,N] | None,
    ray_ts[C,N], ray_planes[C,N,2], normals[C,N,3]``; only rows with radii > 0 are valid."""
    if covars is not None:
        raise NotImplementedError("covars: the reference passes None (rade_gs_model.py:375)")
    if packed or sparse_grad:
        raise NotImplementedError("packed / sparse_grad: the reference passes False (rade_gs_model.py:383, 387)")
    if camera_model != "pinhole":
        raise NotImplementedError("camera_model must be 'pinhole'")
    assert quats is not None and scales is not None, "quats and scales are required when covars is None"
    N, Cn = means.shape[0], viewmats.shape[0]
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape
    assert viewmats.shape == (Cn, 4, 4), viewmats.shape
    assert Ks.shape == (Cn, 3, 3), Ks.shape
    P = make_params(N, Cn, int(width), int(height), antialiased=bool(calc_compensations), eps2d=eps2d,
                    near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip)
    radii, means2d, depths, conics, comps, ray_ts, ray_planes, normals = ops.project(
        means, quats, scales, opacities, viewmats.contiguous().float(), Ks.contiguous().float(), P)
    return radii, means2d, depths, conics, (comps if calc_compensations else None), ray_ts, ray_planes, normals


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """``spherical_harmonics(degrees_to_use=, dirs=[...,3], coeffs=[...,K,3])`` -> ``[...,3]``
    (rade_features_model.py:430-434).  Raw SH: the caller adds 0.5 and clamps (:438)."""
    if masks is not None:
        raise NotImplementedError("masks: not passed by the reference (rade_features_model.py:430-434)")
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3 and coeffs.shape[-1] == 3
    lead = dirs.shape[:-1]
    out = ops.spherical_harmonics_raw(int(degrees_to_use), dirs.reshape(-1, 3),
                                      coeffs.reshape(-1, coeffs.shape[-2], 3))
    return out.reshape(lead + (3,))
