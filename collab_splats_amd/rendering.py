"""``rasterization(...)`` -- drop-in for ``gsplat.rendering.rasterization`` of gsplat-rade as the
reference calls it (/root/reference/collab_splats/models/rade_gs_model.py:439-465,
rade_features_model.py:450-476), running on hand-written HIP kernels for MI355X (gfx950).

Keyword surface and return arity follow SURVEY.md Appendix A.1: with
``return_depth_normal=True`` the result is the 6-tuple
``(render[C,H,W,D'], alpha[C,H,W,1], expected_depths[C,H,W,1], median_depths[C,H,W,1],
expected_normals[C,H,W,3], meta)``; otherwise gsplat's 3-tuple ``(render, alpha, meta)``.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import MisplatError, make_params

_RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")
ENABLE_ND_ONE_PASS = True        # a8: 5..20 channels in one compositing pass (False: 4-channel passes; for A/B)


class _Meta(dict):
    """``meta`` dict.  Two of gsplat's keys are produced on demand: ``isect_ids`` (the sorted 64-bit keys, which nothing on
    the reference's path reads) is rebuilt from the tile ids and depths, and ``flatten_ids`` is absent after a front-only
    forward (only the heads of the buckets are sorted) and completed by the regular per-tile sort.  EVERY way of asking for
    them completes them -- ``meta[k]``, ``meta.get(k)``, ``k in meta``, ``keys()`` / ``items()`` / ``values()`` / iteration,
    ``dict(meta)`` / ``copy()`` -- so code written against gsplat's contract (the keys are always there) sees them whatever its
    access style (advisor, round 4).  ``_has(k)`` asks without completing.  NOTE: ``meta`` refers to arrays of the call's
    arena slot (``meta["_bins"]``): the slot stays in use for as long as ``meta`` is held."""

    _LAZY = ("isect_ids", "flatten_ids")

    def __missing__(self, key):
        if key == "isect_ids":
            self[key] = ops.isect_ids(dict.__getitem__(self, "_bins"))
            return dict.__getitem__(self, key)
        if key == "flatten_ids":                       # (absent after a front-only forward: only the heads are sorted)
            self[key] = ops.complete_bins(dict.__getitem__(self, "_bins"))
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def _has(self, key) -> bool:
        """Is ``key`` materialised already?  (plain dict membership: does not complete anything)"""
        return dict.__contains__(self, key)

    def _complete(self) -> None:
        if dict.__contains__(self, "_bins"):
            for k in self._LAZY:
                if not dict.__contains__(self, k):
                    self[k]                              # (through __missing__)

    def __contains__(self, key) -> bool:
        return dict.__contains__(self, key) or (key in self._LAZY and dict.__contains__(self, "_bins"))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        self._complete()
        return dict.keys(self)

    def items(self):
        self._complete()
        return dict.items(self)

    def values(self):
        self._complete()
        return dict.values(self)

    def __iter__(self):
        self._complete()
        return dict.__iter__(self)

    def __len__(self) -> int:
        return dict.__len__(self) + (sum(1 for k in self._LAZY if not dict.__contains__(self, k)) if dict.__contains__(self, "_bins") else 0)

    def copy(self):
        self._complete()
        return _Meta(dict.items(self))


def rasterization(
    means: Tensor,                    # [N, 3]
    quats: Tensor,                    # [N, 4] wxyz, un-normalised
    scales: Tensor,                   # [N, 3] (already exp'd, rade_gs_model.py:443)
    opacities: Tensor,                # [N]    (already sigmoid'ed, :444)
    colors: Tensor,                   # [N, K, 3] SH coeffs | [N, D] | [C, N, D]
    viewmats: Tensor,                 # [C, 4, 4] world->camera, OpenCV axes
    Ks: Tensor,                       # [C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = False,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    camera_model: str = "pinhole",
    covars: Optional[Tensor] = None,
    return_depth_normal: bool = False,
    # open parameters of SURVEY.md Appendix B (see DESIGN.md "Numerical choices")
    radius_sigma: float = 3.33,
    opacity_aware_radius: bool = True,
    alpha_max: float = 0.999,
    normalise_expected_depth: bool = False,
    # extension: ``scales`` holds log-scales / ``opacities`` holds logits -- the projection kernels apply exp / sigmoid
    # themselves and the gradients come back for the raw parameters (the caller's torch.exp / torch.sigmoid of
    # rade_gs_model.py:443-444 and their backward: four launches per step less).  Same results up to rounding.
    scales_are_log: bool = False,
    opacities_are_logit: bool = False,
    # extension: the features model's call as ONE entry (rade_features_model.py:427-476).  ``features`` [N,F] with SH
    # ``colors`` and ``sh_degree``: the render has 3 + F (+ 1) channels -- channels 0..2 = max(SH(colors) + 0.5, 0), channels
    # 3.. = the features -- exactly what the reference gets from spherical_harmonics -> clamp_min(c + 0.5, 0) -> cat((colors,
    # features)) -> rasterization(sh_degree=None), without the [N,16] concatenation and with gradients to the coefficients
    # AND the features.
    features: Optional[Tensor] = None,
):
    if render_mode not in _RENDER_MODES:
        raise ValueError(f"Unknown render_mode: {render_mode}")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")   # cf. rade_gs_model.py:150-151
    if packed or sparse_grad:
        raise NotImplementedError("packed=True / sparse_grad=True are not on the reference's path "
                                  "(it passes packed=False, sparse_grad=False: rade_gs_model.py:450, 455)")
    if distributed or camera_model != "pinhole" or covars is not None:
        raise NotImplementedError("only camera_model='pinhole', covars=None, distributed=False")
    N = means.shape[0]
    Cn = viewmats.shape[0]
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    assert scales.shape == (N, 3), scales.shape
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (Cn, 4, 4), viewmats.shape
    assert Ks.shape == (Cn, 3, 3), Ks.shape
    split_sh = isinstance(colors, (tuple, list))
    if split_sh:
        # extension: (features_dc [N,3], features_rest [N,K-1,3]) -- the reference's two colour parameters
        # (rade_gs_model.py:119-120) consumed in place, without its per-step torch.cat (:128-130)
        assert sh_degree is not None and len(colors) == 2, "a (features_dc, features_rest) pair needs sh_degree"
        dc, rest = colors
        assert dc.shape == (N, 3) and rest.dim() == 3 and rest.shape[0] == N and rest.shape[2] == 3, (dc.shape, rest.shape)
        assert (sh_degree + 1) ** 2 <= 1 + rest.shape[1], rest.shape
    elif sh_degree is None:
        assert (colors.dim() == 2 and colors.shape[0] == N) or (colors.dim() == 3 and colors.shape[:2] == (Cn, N)), colors.shape
    else:
        assert colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3, colors.shape
        assert (sh_degree + 1) ** 2 <= colors.shape[1], colors.shape
    if not means.is_cuda:
        raise MisplatError("rasterization() runs on the MI355X only (no CPU fallback)")
    width, height = int(width), int(height)
    aa = rasterize_mode == "antialiased"
    viewmats = viewmats.contiguous().float()
    Ks = Ks.contiguous().float()

    if features is not None:
        if sh_degree is None or render_mode in ("D", "ED"):
            raise ValueError("features=... rides behind SH colours: it needs sh_degree and an RGB render mode")
        assert features.shape[0] == N and features.dim() == 2 and features.shape[1] >= 1, features.shape
    depth_channel = render_mode in ("RGB+D", "RGB+ED", "D", "ED")
    n_user = 0 if render_mode in ("D", "ED") else (3 if sh_degree is not None else colors.shape[-1])
    if features is not None:
        n_user += features.shape[1]
    n_sh = (1 + colors[1].shape[1]) if split_sh else (colors.shape[1] if sh_degree is not None else 0)
    fused = n_user + int(depth_channel) <= 4 and n_sh <= 16
    # N-D colours in one pass (a8): 5..20 channels, pass-through colours, atomic gradient mode
    n_total = n_user + int(depth_channel)
    fused_x = (ENABLE_ND_ONE_PASS and (not fused) and sh_degree is None and 5 <= n_total <= 20
               and not ops.DETERMINISTIC_BACKWARD)
    fused_feat = features is not None and ENABLE_ND_ONE_PASS and 5 <= n_total <= 20 and n_sh <= 16 and Cn == 1
    if features is not None and not (fused_feat and ops.fused_node_ok() and N > 0):
        # outside the one-entry path (deterministic mode, > 20 channels, several cameras): as the reference composes it
        from .wrapper import spherical_harmonics
        coeffs = torch.cat((colors[0][:, None, :], colors[1]), dim=1) if split_sh else colors
        cam_centers = -torch.einsum("cji,cj->ci", viewmats[:, :3, :3], viewmats[:, :3, 3])
        rgb = torch.clamp_min(spherical_harmonics(sh_degree, means[None] - cam_centers[:, None, :], coeffs) + 0.5, 0.0)
        fused_cols = torch.cat((rgb, features[None].expand(Cn, N, features.shape[1])), dim=-1)
        return rasterization(means, quats, scales, opacities, fused_cols if Cn > 1 else fused_cols[0], viewmats, Ks, width, height,
                             near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip, eps2d=eps2d, sh_degree=None,
                             tile_size=tile_size, backgrounds=backgrounds, render_mode=render_mode, absgrad=absgrad,
                             rasterize_mode=rasterize_mode, return_depth_normal=return_depth_normal, radius_sigma=radius_sigma,
                             opacity_aware_radius=opacity_aware_radius, alpha_max=alpha_max,
                             normalise_expected_depth=normalise_expected_depth, scales_are_log=scales_are_log,
                             opacities_are_logit=opacities_are_logit)
    # fused paths: the kernels divide the depth channel by max(alpha, 1e-10) themselves ("ED")
    if split_sh and not fused and not fused_feat:
        colors = torch.cat((colors[0][:, None, :], colors[1]), dim=1)
    ed_fused = (fused or fused_x or fused_feat) and render_mode in ("ED", "RGB+ED")
    P = make_params(N, Cn, width, height, tile_size=tile_size, antialiased=aa,
                    opacity_aware_radius=opacity_aware_radius, eps2d=eps2d, near_plane=near_plane,
                    far_plane=far_plane, radius_clip=radius_clip, radius_sigma=radius_sigma,
                    alpha_max=alpha_max, ed_slot=n_user if ed_fused else -1)
    act = (1 if scales_are_log else 0) | (2 if opacities_are_logit else 0)
    nd_one_node = (fused_x or fused_feat) and ops.fused_node_ok() and N > 0
    if act and not ((fused or nd_one_node) and ops.fused_node_ok() and N > 0 and Cn == 1):
        # only the single-node path of one camera carries the activations inside its kernels: elsewhere, as the caller would
        scales = torch.exp(scales) if scales_are_log else scales
        opacities = torch.sigmoid(opacities) if opacities_are_logit else opacities
        act = 0
    P.activations = act
    if (fused and ops.fused_node_ok() and N > 0) or nd_one_node:
        # ---- the reference's path as ONE autograd node (ops._RasterFused): two C calls forward, one backward -- RGB(+ED),
        # and the features model's 16 / 17 channels (pass-through colours [N,D], or SH coefficients + features side by side)
        cin = colors if n_user > 0 else means.new_zeros(N, 1)[:, :0].contiguous()
        D = n_user + int(depth_channel)
        out12, bins = ops.raster_fused(means, quats, scales, opacities, cin, viewmats, Ks, P,
                                       sh_degree if n_user > 0 else None, depth_channel, D, absgrad,
                                       features=features if fused_feat else None)
        render, alpha_, ed_, md_, nrm_, means2d, radii, depths, comps, grec, last_ids_, median_ids_ = out12
        first = (render, alpha_, ed_, md_, nrm_, last_ids_, median_ids_)
        gv = grec.view(Cn, N, 16)
        conics, opac, ray_ts, ray_planes, normals = gv[..., 2:5], gv[..., 5], gv[..., 6], gv[..., 7:9], gv[..., 9:12]
    elif fused:
        # ---- the same path as two autograd nodes (differentiable per-Gaussian intermediates in meta), no glue kernels
        cin = colors if n_user > 0 else means.new_zeros(N, 1)
        prebin = {}                                   # tile counting starts inside the node, before the colour kernel
        radii, means2d, depths, comps, grec = ops.project_pack(
            means, quats, scales, opacities, cin if n_user > 0 else cin[:, :0].contiguous(), viewmats, Ks, P,
            sh_degree if n_user > 0 else None, depth_channel, prebin)
        D = n_user + int(depth_channel)
        if "fused" in prebin:                         # one host entry per phase (ops._raster_phase_a / _b)
            bins = prebin["fused"]
        else:
            bins = ops.bin_tiles(P, means2d, radii, depths, pending=prebin.get("pending"))
        first = ops.blend_packed(means2d, grec, Ks, P, bins, absgrad, D)
        render = first[0]
        gv = grec.view(Cn, N, 16)
        conics, opac, ray_ts, ray_planes, normals = gv[..., 2:5], gv[..., 5], gv[..., 6], gv[..., 7:9], gv[..., 9:12]
    elif fused_x:
        # ---- rade_features_model.py:441-476 (16 fused channels, 17 with ED): one compositing pass
        nxq = (n_total - 4 + 3) // 4
        radii, means2d, depths, comps, grec, featx = ops.project_pack_x(
            means, quats, scales, opacities, colors, viewmats, Ks, P, depth_channel, nxq)
        bins = ops.bin_tiles(P, means2d, radii, depths)
        first = ops.blend_packed_x(means2d, grec, featx, Ks, P, bins, absgrad, n_total, nxq)
        render = first[0]
        gv = grec.view(Cn, N, 16)
        conics, opac, ray_ts, ray_planes, normals = gv[..., 2:5], gv[..., 5], gv[..., 6], gv[..., 7:9], gv[..., 9:12]
    else:
        # ---- generic path: any number of colour channels (rade_features_model.py:441-476, D = 16 / 17),
        # composited 4 channels per pass; the geometry outputs come from pass 0
        radii, means2d, depths, conics, comps, ray_ts, ray_planes, normals = ops.project(
            means, quats, scales, opacities, viewmats, Ks, P)
        opac = opacities[None, :].expand(Cn, N)
        opac = opac * comps if aa else opac.contiguous()
        if sh_degree is not None:
            cam_centers = -torch.einsum("cji,cj->ci", viewmats[:, :3, :3], viewmats[:, :3, 3])   # -R^T t
            dirs = means[None, :, :] - cam_centers[:, None, :]
            cols = ops.spherical_harmonics_raw(sh_degree, dirs, colors, radii)
            cols = torch.clamp_min(cols + 0.5, 0.0)                  # rade_features_model.py:438
        else:
            cols = colors if colors.dim() == 3 else colors[None].expand(Cn, N, colors.shape[-1])
        if render_mode in ("RGB+D", "RGB+ED"):
            cols = torch.cat([cols, depths[..., None]], dim=-1)
        elif render_mode in ("D", "ED"):
            cols = depths[..., None]
        cols = cols.contiguous()
        D = cols.shape[-1]
        bins = ops.bin_tiles(P, means2d, radii, depths)
        renders = []
        first = None
        for s in range(0, D, 4):
            out = ops.blend(means2d, conics, opac, cols[..., s:s + 4], ray_ts, ray_planes, normals, Ks, P, bins,
                            absgrad=absgrad, pass_index=s // 4)
            renders.append(out[0])
            if first is None:
                first = out
        render = renders[0] if len(renders) == 1 else torch.cat(renders, dim=-1)
    alpha, exp_depth, med_depth, exp_normal = first[1], first[2], first[3], first[4]

    if render_mode in ("ED", "RGB+ED") and not ed_fused:
        render = torch.cat([render[..., :-1], render[..., -1:] / alpha.clamp(min=1e-10)], dim=-1)
    if normalise_expected_depth:
        exp_depth = exp_depth / alpha.clamp(min=1e-10)
    if backgrounds is not None:
        render = render + (1.0 - alpha) * backgrounds[:, None, None, :]

    meta = _Meta({
        "_bins": bins, "camera_ids": None, "gaussian_ids": None,
        "radii": radii, "means2d": means2d, "depths": depths, "conics": conics, "opacities": opac,
        "compensations": comps, "ray_ts": ray_ts, "ray_planes": ray_planes, "normals": normals,
        "tile_width": P.tile_w, "tile_height": P.tile_h, "tiles_per_gauss": bins["tiles_per_gauss"].view(Cn, N),
        "isect_offsets": bins["isect_offsets"][:-1].view(Cn, P.tile_h, P.tile_w),
        "n_isects": bins["n_isects"] if bins.get("n_isects_dev") is None else bins["n_isects_dev"], "last_ids": first[5], "median_ids": first[6],
        "width": width, "height": height, "tile_size": tile_size, "n_cameras": Cn,
    })
    if bins.get("partial") is None:
        meta["flatten_ids"] = bins["flatten_ids"]      # (else completed on first access: _Meta.__missing__)
    if return_depth_normal:
        return render, alpha, exp_depth, med_depth, exp_normal, meta
    return render, alpha, meta
