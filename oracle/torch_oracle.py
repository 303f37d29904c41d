"""fp64-capable PyTorch/autograd restatement of the RaDe-GS rasterizer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``collab_splats_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

PARITY UNPINNED: the arithmetic this restates lives in the
third-party dependency ``gsplat @ git+https://github.com/brian-xu/gsplat-rade.git``
(unpinned, un-vendored: /root/reference/pyproject.toml:38) which is absent from the
reference tree, and the reference's own tests never render
(/root/reference/tests/test_models.py:45-63).  What this file follows instead:

* the call-site contract of ``rasterization(...)`` / ``fully_fused_projection(...)`` /
  ``spherical_harmonics(...)`` pinned at /root/reference/collab_splats/models/
  rade_gs_model.py:373-394, 439-465 and rade_features_model.py:427-476;
* the conventions pinned by the reference's consumers of the outputs: quaternions are
  wxyz and normalised inside (camera_utils.py:138-168), depth maps are z-depth at
  pixel centres +0.5 (camera_utils.py:228-245), rendered normals face the camera in
  OpenCV camera space (camera_utils.py:269-273 + rade_gs_model.py:212-214), SH colour is
  ``clamp_min(sh + 0.5, 0)`` (rade_features_model.py:438), ``alpha = 1 - T``
  (rade_gs_model.py:228), the ``RGB+ED`` depth channel sits at ``render[..., 3:4]``
  (rade_gs_model.py:237);
* SURVEY.md Appendix B for the published RaDe-GS / gsplat math (every constant is a
  named parameter of :class:`RasterSpec` so it can be flipped).

The backward is autograd's: branch decisions (alpha skip, transmittance stop, median
selection, culling) are boolean masks, i.e. frozen, which is exactly the adjoint the
HIP kernels implement by hand.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

# Real SH constants (3DGS / gsplat convention).
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


@dataclass
class RasterSpec:
    """Open parameters of SURVEY.md Appendix B (defaults = the shipped HIP path)."""
    tile_size: int = 16
    eps2d: float = 0.3                  # rade_gs_model.py:382
    near_plane: float = 0.01            # rade_gs_model.py:451
    far_plane: float = 1e10             # rade_gs_model.py:452
    radius_clip: float = 0.0            # rade_gs_model.py:386
    radius_sigma: float = 3.33          # exp(-3.33^2/2) ~= 1/255
    opacity_aware_radius: bool = True   # shrink extent to the 1/255 level set of o*exp(-s)
    alpha_max: float = 0.999
    alpha_min: float = 1.0 / 255.0
    t_stop: float = 1e-4
    median_t: float = 0.5
    jacobian_margin: float = 0.3        # lim = (W-cx)/fx + margin * tan_fov  (== 1.3 tan at centre)
    normalise_expected_depth: bool = False
    plane_eps: float = 1e-6


# --------------------------------------------------------------------------- helpers

def quat_to_rotmat(quats: torch.Tensor) -> torch.Tensor:
    """wxyz -> 3x3, normalised inside (follows camera_utils.py:138-168)."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    r, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y),
    ], dim=-1)
    return R.reshape(quats.shape[:-1] + (3, 3))


def eval_sh(degree: int, dirs: torch.Tensor, coeffs: torch.Tensor) -> torch.Tensor:
    """Real SH up to ``degree``; dirs [...,3] (normalised here), coeffs [...,K,3] -> [...,3].

    Signature follows ``spherical_harmonics(degrees_to_use, dirs, coeffs)``
    (rade_features_model.py:430-434).  No +0.5 / clamp here (the caller does it, :438).
    """
    d = dirs / dirs.norm(dim=-1, keepdim=True).clamp_min(1e-30)
    x, y, z = d[..., 0:1], d[..., 1:2], d[..., 2:3]
    res = SH_C0 * coeffs[..., 0, :]
    if degree > 0:
        res = res - SH_C1 * y * coeffs[..., 1, :] + SH_C1 * z * coeffs[..., 2, :] - SH_C1 * x * coeffs[..., 3, :]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * coeffs[..., 4, :] + SH_C2[1] * yz * coeffs[..., 5, :]
               + SH_C2[2] * (2 * zz - xx - yy) * coeffs[..., 6, :]
               + SH_C2[3] * xz * coeffs[..., 7, :] + SH_C2[4] * (xx - yy) * coeffs[..., 8, :])
    if degree > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * coeffs[..., 9, :]
               + SH_C3[1] * xy * z * coeffs[..., 10, :]
               + SH_C3[2] * y * (4 * zz - xx - yy) * coeffs[..., 11, :]
               + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * coeffs[..., 12, :]
               + SH_C3[4] * x * (4 * zz - xx - yy) * coeffs[..., 13, :]
               + SH_C3[5] * z * (xx - yy) * coeffs[..., 14, :]
               + SH_C3[6] * x * (xx - 3 * yy) * coeffs[..., 15, :])
    return res


# --------------------------------------------------------------------------- projection

def project(means, quats, scales, viewmat, K, width: int, height: int,
            opacities: Optional[torch.Tensor] = None, spec: RasterSpec = RasterSpec(),
            calc_compensations: bool = True) -> Dict[str, torch.Tensor]:
    """Per-Gaussian 3D->2D projection for ONE camera.

    Returns a dict with radii [N,2] int32, means2d [N,2], depths [N], conics [N,3],
    compensations [N], ray_ts [N], ray_planes [N,2], normals [N,3], valid [N] bool.
    Names of the 8 outputs follow rade_gs_model.py:392-394.
    """
    dt = means.dtype
    Rwc, twc = viewmat[:3, :3], viewmat[:3, 3]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    Rg = quat_to_rotmat(quats)                                   # [N,3,3]
    mu = means @ Rwc.T + twc                                     # [N,3]
    x, y, z = mu.unbind(-1)
    Rc = Rwc @ Rg                                                # [N,3,3]
    M = Rc * scales[:, None, :]                                  # Rc diag(s)
    cov = M @ M.transpose(1, 2)                                  # Sigma_c

    valid = (z >= spec.near_plane) & (z <= spec.far_plane)
    zs = torch.where(valid, z, torch.ones_like(z))               # keep culled rows finite
    u, v = x / zs, y / zs
    tan_fx, tan_fy = 0.5 * width / fx, 0.5 * height / fy
    lim_xp = (width - cx) / fx + spec.jacobian_margin * tan_fx
    lim_xn = cx / fx + spec.jacobian_margin * tan_fx
    lim_yp = (height - cy) / fy + spec.jacobian_margin * tan_fy
    lim_yn = cy / fy + spec.jacobian_margin * tan_fy
    tx = zs * torch.minimum(lim_xp, torch.maximum(-lim_xn, u))
    ty = zs * torch.minimum(lim_yp, torch.maximum(-lim_yn, v))
    zero = torch.zeros_like(zs)
    J = torch.stack([fx / zs, zero, -fx * tx / (zs * zs),
                     zero, fy / zs, -fy * ty / (zs * zs)], dim=-1).reshape(-1, 2, 3)
    cov2d = J @ cov @ J.transpose(1, 2)
    a0, b0, c0 = cov2d[:, 0, 0], cov2d[:, 0, 1], cov2d[:, 1, 1]
    det0 = a0 * c0 - b0 * b0
    a, c, b = a0 + spec.eps2d, c0 + spec.eps2d, b0
    det = a * c - b * b
    valid = valid & (det > 0)
    dets = torch.where(det > 0, det, torch.ones_like(det))
    comp = torch.sqrt(torch.clamp(det0 / dets, min=0.0))
    conics = torch.stack([c / dets, -b / dets, a / dets], dim=-1)
    means2d = torch.stack([fx * u + cx, fy * v + cy], dim=-1)

    # extent: radius_sigma, optionally tightened to the alpha_min level set of the opacity
    extend = torch.full_like(zs, spec.radius_sigma)
    if opacities is not None and spec.opacity_aware_radius:
        o = opacities * comp if calc_compensations else opacities
        valid = valid & (o >= spec.alpha_min)
        os_ = torch.where(o >= spec.alpha_min, o, torch.full_like(o, spec.alpha_min))
        extend = torch.minimum(extend, torch.sqrt(2.0 * torch.log(os_ / spec.alpha_min)))
    mid = 0.5 * (a + c)
    v1 = mid + torch.sqrt(torch.clamp(mid * mid - dets, min=0.01))
    rx = torch.ceil(torch.minimum(extend * torch.sqrt(a), extend * torch.sqrt(v1)))
    ry = torch.ceil(torch.minimum(extend * torch.sqrt(c), extend * torch.sqrt(v1)))
    valid = valid & ~((rx <= spec.radius_clip) & (ry <= spec.radius_clip))
    valid = valid & ~((means2d[:, 0] + rx <= 0) | (means2d[:, 0] - rx >= width)
                      | (means2d[:, 1] + ry <= 0) | (means2d[:, 1] - ry >= height))
    radii = torch.stack([rx, ry], dim=-1).to(torch.int32) * valid[:, None].to(torch.int32)

    # RaDe-GS extras: m ~ Sigma_c^-1 mu_c computed stably.
    smin = scales.min(dim=-1, keepdim=True).values
    w = (smin / scales) ** 2                                     # s_min^2 / s^2   [N,3]
    Rt_mu = (Rc.transpose(1, 2) @ mu[:, :, None])[:, :, 0]       # Rc^T mu
    m = (Rc @ (w * Rt_mu)[:, :, None])[:, :, 0]
    mn = m.norm(dim=-1, keepdim=True)
    nhat = m / mn.clamp_min(1e-300 if dt == torch.float64 else 1e-30)
    h = torch.stack([u, v, torch.ones_like(u)], dim=-1)
    ell = h.norm(dim=-1)
    nh = (nhat * h).sum(-1)
    plane_ok = (nh.abs() >= spec.plane_eps) & torch.isfinite(nh) & (mn[:, 0] > 0)
    nhs = torch.where(plane_ok, nh, torch.ones_like(nh))
    ray_t = zs * ell
    dtdu = -zs * ell * nhat[:, 0] / nhs + zs * u / ell
    dtdv = -zs * ell * nhat[:, 1] / nhs + zs * v / ell
    ray_planes = torch.stack([dtdu / fx, dtdv / fy], dim=-1) * plane_ok[:, None]
    normals = -nhat * plane_ok[:, None]

    return dict(radii=radii, means2d=means2d, depths=z, conics=conics, compensations=comp,
                ray_ts=ray_t, ray_planes=ray_planes, normals=normals, valid=valid)


# --------------------------------------------------------------------------- binning

def tile_rects(means2d: np.ndarray, radii: np.ndarray, tile_size: int, tile_w: int, tile_h: int):
    """Integer tile rectangle of each Gaussian, computed in the dtype of ``means2d``."""
    ft = means2d.dtype.type
    ts = ft(tile_size)
    rx, ry = radii[:, 0].astype(means2d.dtype), radii[:, 1].astype(means2d.dtype)
    x0 = np.clip(np.floor((means2d[:, 0] - rx) / ts), 0, tile_w).astype(np.int64)
    x1 = np.clip(np.ceil((means2d[:, 0] + rx) / ts), 0, tile_w).astype(np.int64)
    y0 = np.clip(np.floor((means2d[:, 1] - ry) / ts), 0, tile_h).astype(np.int64)
    y1 = np.clip(np.ceil((means2d[:, 1] + ry) / ts), 0, tile_h).astype(np.int64)
    vis = (radii[:, 0] > 0) | (radii[:, 1] > 0)
    x0, x1, y0, y1 = [np.where(vis, t, 0) for t in (x0, x1, y0, y1)]
    return x0, x1, y0, y1


def bin_and_sort(means2d: np.ndarray, radii: np.ndarray, depths: np.ndarray, width: int,
                 height: int, tile_size: int = 16, cam: int = 0, n_cams: int = 1):
    """Tile intersection + stable sort + offsets, numpy, exact.

    key = ((cam * n_tiles + tile) << 32) | bits(float32(depth));  value = Gaussian id.
    Emission order is ascending Gaussian id, then row-major tiles; the sort is stable, so
    ties in depth resolve by Gaussian id.
    """
    tile_w = (width + tile_size - 1) // tile_size
    tile_h = (height + tile_size - 1) // tile_size
    x0, x1, y0, y1 = tile_rects(means2d, radii, tile_size, tile_w, tile_h)
    nt = (x1 - x0) * (y1 - y0)
    cum = np.concatenate([[0], np.cumsum(nt)])
    I = int(cum[-1])
    keys = np.zeros(I, dtype=np.uint64)
    gids = np.zeros(I, dtype=np.int32)
    dbits = depths.astype(np.float32).view(np.uint32).astype(np.uint64)
    for g in np.nonzero(nt)[0]:
        ys, xs = np.meshgrid(np.arange(y0[g], y1[g]), np.arange(x0[g], x1[g]), indexing="ij")
        tid = (cam * tile_w * tile_h + ys * tile_w + xs).reshape(-1).astype(np.uint64)
        keys[cum[g]:cum[g + 1]] = (tid << np.uint64(32)) | dbits[g]
        gids[cum[g]:cum[g + 1]] = g
    order = np.argsort(keys, kind="stable")
    keys_s, gids_s = keys[order], gids[order]
    tiles_sorted = (keys_s >> np.uint64(32)).astype(np.int64) - cam * tile_w * tile_h
    offsets = np.searchsorted(tiles_sorted, np.arange(tile_w * tile_h), side="left").astype(np.int32)
    return dict(tiles_per_gauss=nt.astype(np.int32), isect_ids=keys_s, flatten_ids=gids_s,
                isect_offsets=offsets.reshape(tile_h, tile_w), tile_width=tile_w, tile_height=tile_h,
                n_isects=I, rects=(x0, x1, y0, y1))


# --------------------------------------------------------------------------- blending

def blend(means2d, conics, opac, colors, ray_ts, ray_planes, normals, depths, order,
          rects, K, width: int, height: int, spec: RasterSpec = RasterSpec(),
          pixel_chunk: int = 4096):
    """Front-to-back compositing for ONE camera, dense.

    ``order``: LongTensor of visible Gaussian ids sorted by (depth, id).  ``rects``: tile
    rectangle per Gaussian (x0,x1,y0,y1 LongTensors) -- a Gaussian only reaches the pixels of
    its tiles, as in the tiled kernels.  colors [N,D].
    Returns render [H,W,D], alpha [H,W,1], exp_depth [H,W,1], med_depth [H,W,1],
    normal [H,W,3], plus last_ids/median_ids [H,W] (position in ``order``, -1 if none).
    """
    dt, dev = means2d.dtype, means2d.device
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    D = colors.shape[-1]
    G = order.numel()
    P = width * height
    ts = spec.tile_size
    m2, cn, op = means2d[order], conics[order], opac[order]
    col, rt, rp, nr = colors[order], ray_ts[order], ray_planes[order], normals[order]
    x0, x1, y0, y1 = [r[order] for r in rects]
    out_c = torch.zeros(P, D, dtype=dt, device=dev)
    out_a = torch.zeros(P, dtype=dt, device=dev)
    out_d = torch.zeros(P, dtype=dt, device=dev)
    out_m = torch.zeros(P, dtype=dt, device=dev)
    out_n = torch.zeros(P, 3, dtype=dt, device=dev)
    last_ids = torch.full((P,), -1, dtype=torch.long, device=dev)
    med_ids = torch.full((P,), -1, dtype=torch.long, device=dev)
    if G == 0:
        shp = (height, width)
        return (out_c.reshape(*shp, D), out_a.reshape(*shp, 1), out_d.reshape(*shp, 1),
                out_m.reshape(*shp, 1), out_n.reshape(*shp, 3), last_ids.reshape(shp), med_ids.reshape(shp))
    cc, cd, cm, cn_, ca, cl, cmi = [], [], [], [], [], [], []
    for s in range(0, P, pixel_chunk):
        idx = torch.arange(s, min(P, s + pixel_chunk), device=dev)
        ix, iy = idx % width, idx // width
        px, py = ix.to(dt) + 0.5, iy.to(dt) + 0.5
        tx_, ty_ = ix // ts, iy // ts
        in_rect = ((tx_[:, None] >= x0[None]) & (tx_[:, None] < x1[None])
                   & (ty_[:, None] >= y0[None]) & (ty_[:, None] < y1[None]))
        dx = m2[None, :, 0] - px[:, None]
        dy = m2[None, :, 1] - py[:, None]
        sigma = 0.5 * (cn[None, :, 0] * dx * dx + cn[None, :, 2] * dy * dy) + cn[None, :, 1] * dx * dy
        alpha = torch.clamp(op[None] * torch.exp(-sigma), max=spec.alpha_max)
        contrib = in_rect & (sigma >= 0) & (alpha >= spec.alpha_min)
        a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
        one_m = 1.0 - a_eff
        t_incl = torch.cumprod(one_m, dim=1)
        t_before = torch.cat([torch.ones_like(t_incl[:, :1]), t_incl[:, :-1]], dim=1)
        live = t_incl > spec.t_stop                       # monotone: first failure stops the pixel
        use = contrib & live
        wgt = torch.where(use, a_eff * t_before, torch.zeros_like(a_eff))
        t_final = torch.prod(torch.where(live, one_m, torch.ones_like(one_m)), dim=1)
        # per-(pixel, Gaussian) z-depth from the RaDe ray-distance plane
        tpix = rt[None] - (rp[None, :, 0] * dx + rp[None, :, 1] * dy)
        ell = torch.sqrt(((px - cx) / fx) ** 2 + ((py - cy) / fy) ** 2 + 1.0)
        zpix = tpix / ell[:, None]
        cc.append(wgt @ col)
        cd.append((wgt * zpix).sum(1))
        cn_.append(wgt @ nr)
        ca.append(1.0 - t_final)
        pos = torch.arange(G, device=dev)[None].expand_as(use)
        last = torch.where(use, pos, torch.full_like(pos, -1)).max(dim=1).values
        medsel = use & (t_before > spec.median_t)
        med = torch.where(medsel, pos, torch.full_like(pos, -1)).max(dim=1).values
        cm.append(torch.where(med >= 0, zpix.gather(1, med.clamp_min(0)[:, None])[:, 0],
                              torch.zeros_like(ell)))
        cl.append(last)
        cmi.append(med)
    out_c, out_d, out_n = torch.cat(cc), torch.cat(cd), torch.cat(cn_)
    out_a, out_m = torch.cat(ca), torch.cat(cm)
    if spec.normalise_expected_depth:
        out_d = out_d / out_a.clamp_min(1e-10)
    shp = (height, width)
    return (out_c.reshape(*shp, D), out_a.reshape(*shp, 1), out_d.reshape(*shp, 1),
            out_m.reshape(*shp, 1), out_n.reshape(*shp, 3),
            torch.cat(cl).reshape(shp), torch.cat(cmi).reshape(shp))


# --------------------------------------------------------------------------- top level

def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width: int, height: int,
                  near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
                  eps2d: float = 0.3, sh_degree: Optional[int] = None, packed: bool = False,
                  tile_size: int = 16, backgrounds=None, render_mode: str = "RGB",
                  sparse_grad: bool = False, absgrad: bool = False,
                  rasterize_mode: str = "classic", return_depth_normal: bool = True,
                  spec: Optional[RasterSpec] = None, pixel_chunk: int = 4096):
    """Oracle with the keyword surface the reference passes (rade_gs_model.py:440-464).

    Returns ``(render[C,H,W,D'], alpha[C,H,W,1], expected_depths[C,H,W,1],
    median_depths[C,H,W,1], expected_normals[C,H,W,3], meta)``.
    """
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED")
    assert rasterize_mode in ("classic", "antialiased")
    spec = spec or RasterSpec()
    spec = RasterSpec(**{**spec.__dict__, "tile_size": tile_size, "eps2d": eps2d,
                         "near_plane": near_plane, "far_plane": far_plane,
                         "radius_clip": radius_clip})
    C = viewmats.shape[0]
    outs = [[] for _ in range(5)]
    metas = []
    for ci in range(C):
        V, K = viewmats[ci], Ks[ci]
        pr = project(means, quats, scales, V, K, width, height, opacities, spec,
                     calc_compensations=(rasterize_mode == "antialiased"))
        if sh_degree is not None:
            cam_center = -(V[:3, :3].T @ V[:3, 3])
            cols = torch.clamp_min(eval_sh(sh_degree, means - cam_center, colors) + 0.5, 0.0)
        else:
            cols = colors
        opac = opacities * pr["compensations"] if rasterize_mode == "antialiased" else opacities
        if render_mode in ("RGB+D", "RGB+ED"):
            cols = torch.cat([cols, pr["depths"][:, None]], dim=-1)
        elif render_mode in ("D", "ED"):
            cols = pr["depths"][:, None]
        radii_np = pr["radii"].detach().cpu().numpy()
        m2_np = pr["means2d"].detach().cpu().numpy()
        bs = bin_and_sort(m2_np, radii_np, pr["depths"].detach().cpu().numpy(), width, height,
                          tile_size, cam=ci, n_cams=C)
        vis = torch.from_numpy((radii_np[:, 0] > 0) | (radii_np[:, 1] > 0))
        dkey = pr["depths"].detach().to(torch.float32).cpu().numpy().view(np.uint32).astype(np.int64)
        ids = np.nonzero(vis.numpy())[0]
        order = torch.from_numpy(ids[np.lexsort((ids, dkey[ids]))]).long()
        rects = tuple(torch.from_numpy(r) for r in bs["rects"])
        r = blend(pr["means2d"], pr["conics"], opac, cols, pr["ray_ts"], pr["ray_planes"],
                  pr["normals"], pr["depths"], order, rects, K, width, height, spec, pixel_chunk)
        render, alpha, ed, md, nrm = r[:5]
        if render_mode in ("ED", "RGB+ED"):
            render = torch.cat([render[..., :-1], render[..., -1:] / alpha.clamp_min(1e-10)], dim=-1)
        if backgrounds is not None:
            render = render + (1.0 - alpha) * backgrounds[ci]
        for lst, t in zip(outs, (render, alpha, ed, md, nrm)):
            lst.append(t)
        metas.append(dict(pr, **{k: v for k, v in bs.items() if k != "rects"}, order=order,
                          last_ids=r[5], median_ids=r[6], colors=cols, opacities_eff=opac))
    render, alpha, ed, md, nrm = [torch.stack(o) for o in outs]
    meta = dict(
        radii=torch.stack([m["radii"] for m in metas]),
        means2d=torch.stack([m["means2d"] for m in metas]),
        depths=torch.stack([m["depths"] for m in metas]),
        conics=torch.stack([m["conics"] for m in metas]),
        opacities=torch.stack([m["opacities_eff"] for m in metas]),
        ray_ts=torch.stack([m["ray_ts"] for m in metas]),
        ray_planes=torch.stack([m["ray_planes"] for m in metas]),
        normals=torch.stack([m["normals"] for m in metas]),
        compensations=torch.stack([m["compensations"] for m in metas]),
        colors=torch.stack([m["colors"] for m in metas]),
        tiles_per_gauss=np.stack([m["tiles_per_gauss"] for m in metas]),
        isect_ids=np.concatenate([m["isect_ids"] for m in metas]),
        flatten_ids=np.concatenate([m["flatten_ids"] + ci * means.shape[0]
                                    for ci, m in enumerate(metas)]),
        isect_offsets=np.stack([m["isect_offsets"] + base for m, base in zip(
            metas, np.concatenate([[0], np.cumsum([m["n_isects"] for m in metas])[:-1]]).astype(np.int32))]),
        order_ids=[m["order"].numpy() for m in metas],
        last_ids=torch.stack([m["last_ids"] for m in metas]),
        median_ids=torch.stack([m["median_ids"] for m in metas]),
        width=width, height=height, tile_size=tile_size, n_cameras=C,
        tile_width=metas[0]["tile_width"], tile_height=metas[0]["tile_height"],
    )
    return render, alpha, ed, md, nrm, meta
