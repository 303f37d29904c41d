"""The stage-by-stage entries of the path, each an autograd node of its own (plumbing around the C ABI; no kernels here).

What runs through them: the deterministic gradient mode, N-D colours beyond 20 channels, a loss on the projection's own outputs
with N-D colours, the two-node form (``ops.FUSED_NODE = False``: ``_ProjectPack`` -> ``_BlendPacked``, in which every per-Gaussian
intermediate of ``meta`` is differentiable) and the public stage functions of ``wrapper.py`` (``fully_fused_projection``,
``spherical_harmonics``).  The default call is ONE node: ``ops._RasterFused`` (ops.py).  Switches and shared state live in
``ops`` and are read from there at call time (``_o.X``); every public name here is reachable as ``ops.X`` too.
SURVEY.md section 8 rows a2.1 - a2.5, a8; reference call sites /root/reference/collab_splats/models/rade_gs_model.py:373-394,
439-465 and rade_features_model.py:430-476."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib, arena
from . import ops as _o
from ._lib import MISPLAT_REC, Params, check, ptr, require_gpu, stream_ptr

# ----------------------------------------------------------------------------- projection

class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, opacities, viewmats, Ks, P: Params):
        lib = _lib.load()
        require_gpu(means, quats, scales, viewmats, Ks)
        N, Cn = P.n_gauss, P.n_cams
        dev = means.device
        f = dict(device=dev, dtype=torch.float32)
        radii = torch.empty(Cn, N, 2, device=dev, dtype=torch.int32)
        means2d = torch.empty(Cn, N, 2, **f)
        depths = torch.empty(Cn, N, **f)
        conics = torch.empty(Cn, N, 3, **f)
        comps = torch.empty(Cn, N, **f)
        ray_ts = torch.empty(Cn, N, **f)
        ray_planes = torch.empty(Cn, N, 2, **f)
        normals = torch.empty(Cn, N, 3, **f)
        check(lib.misplat_project_fwd(C.byref(P), ptr(means), ptr(quats), ptr(scales), ptr(opacities),
                                      ptr(viewmats), ptr(Ks), ptr(radii), ptr(means2d), ptr(depths),
                                      ptr(conics), ptr(comps), ptr(ray_ts), ptr(ray_planes), ptr(normals),
                                      stream_ptr()), "misplat_project_fwd")
        ctx.P = P
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii)
        ctx.mark_non_differentiable(radii)
        return radii, means2d, depths, conics, comps, ray_ts, ray_planes, normals

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps, v_ray_ts, v_ray_planes, v_normals):
        lib = _lib.load()
        means, quats, scales, viewmats, Ks, radii = ctx.saved_tensors
        P = ctx.P
        v_means = torch.empty_like(means)
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        g = [_o._c(t) for t in (v_means2d, v_depths, v_conics, v_comps, v_ray_ts, v_ray_planes, v_normals)]
        check(lib.misplat_project_bwd(C.byref(P), ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks),
                                      ptr(radii), *[ptr(t) for t in g], ptr(v_means), ptr(v_quats),
                                      ptr(v_scales), stream_ptr()), "misplat_project_bwd")
        return v_means, v_quats, v_scales, None, None, None, None


def project(means: Tensor, quats: Tensor, scales: Tensor, opacities: Optional[Tensor], viewmats: Tensor,
            Ks: Tensor, P: Params):
    """8-tuple (radii, means2d, depths, conics, compensations, ray_ts, ray_planes, normals)."""
    means, quats, scales = _o._f32(means, "means"), _o._f32(quats, "quats"), _o._f32(scales, "scales")
    viewmats, Ks = _o._f32(viewmats, "viewmats"), _o._f32(Ks, "Ks")
    op = None if opacities is None else _o._f32(opacities.detach(), "opacities")
    return _Project.apply(means, quats, scales, op, viewmats, Ks, P)


# ----------------------------------------------------------------------------- SH

class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dirs, coeffs, degree: int, radii):
        lib = _lib.load()
        require_gpu(dirs, coeffs)
        N, K = coeffs.shape[0], coeffs.shape[1]
        Cn = dirs.numel() // (3 * N) if N > 0 else 1
        colors = torch.empty(dirs.shape[:-1] + (3,), device=dirs.device, dtype=torch.float32)
        check(lib.misplat_sh_fwd(C.c_int32(N), C.c_int32(Cn), C.c_int32(K), C.c_int32(degree), ptr(dirs),
                                 ptr(coeffs), ptr(radii), ptr(colors), stream_ptr()), "misplat_sh_fwd")
        ctx.save_for_backward(dirs, coeffs, radii)
        ctx.degree, ctx.Cn = degree, Cn
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        lib = _lib.load()
        dirs, coeffs, radii = ctx.saved_tensors
        N, K = coeffs.shape[0], coeffs.shape[1]
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs)
        check(lib.misplat_sh_bwd(C.c_int32(N), C.c_int32(ctx.Cn), C.c_int32(K), C.c_int32(ctx.degree),
                                 ptr(dirs), ptr(coeffs), ptr(radii), ptr(_o._c(v_colors)), ptr(v_coeffs),
                                 ptr(v_dirs), stream_ptr()), "misplat_sh_bwd")
        return v_dirs, v_coeffs, None, None


def spherical_harmonics_raw(degree: int, dirs: Tensor, coeffs: Tensor, radii: Optional[Tensor] = None) -> Tensor:
    """dirs [..., N, 3] (C leading cameras allowed), coeffs [N, K, 3] -> [..., N, 3]; raw SH."""
    if coeffs.dim() != 3 or coeffs.shape[-1] != 3:
        raise ValueError(f"coeffs must be [N, K, 3], got {tuple(coeffs.shape)}")
    if dirs.shape[-2:] != (coeffs.shape[0], 3):
        raise ValueError(f"dirs {tuple(dirs.shape)} does not match coeffs {tuple(coeffs.shape)}")
    if not 0 <= degree <= 3 or (degree + 1) ** 2 > coeffs.shape[1]:
        raise ValueError(f"degree {degree} needs {(degree + 1) ** 2} <= K={coeffs.shape[1]} and degree <= 3")
    r = None if radii is None else radii.contiguous()
    return _SphericalHarmonics.apply(_o._f32(dirs, "dirs"), _o._f32(coeffs, "coeffs"), int(degree), r)


# ----------------------------------------------------------------------------- binning

# Ordering (csrc/bucket.hip + csrc/binning.hip; gives exactly the (tile, depth, Gaussian id) order of a one-shot 64-bit key
# sort): rows are put into coarse screen-cell order, a workgroup of 1024 neighbouring rows counts / fills its intersections
# per tile through an LDS window with one global atomic per (workgroup, tile): every intersection is written once (its row,
# 4 bytes) and no tile-id array exists; one workgroup per tile then sorts its bucket by (depth, row).  All sizes live on
# the device.
def _read_back(n_dev: Tensor) -> Dict:
    """Asynchronous read-back of a device int64 (pinned buffer + event)."""
    host = torch.empty(1, dtype=torch.int64, pin_memory=True)
    host.copy_(n_dev, non_blocking=True)
    event = torch.cuda.Event()
    event.record()
    return dict(host=host, event=event)


@torch.no_grad()
def start_binning(P: Params, means2d: Tensor, radii: Tensor) -> Dict[str, Tensor]:
    """First half of ``bin_tiles``: everything that does not need the number of intersections on the host (tile
    counts, the cell ordering of the rows), and an ASYNCHRONOUS read-back of that number.  Called right
    after the projection kernel, before the colour kernel is launched, so that the host's wait for n_isects -- the one
    sync of the step -- and the launches that follow it are hidden behind the colour kernel instead of idling the GPU."""
    lib = _lib.load()
    dev = means2d.device
    total = P.n_gauss * P.n_cams
    n_tiles = P.tile_w * P.tile_h * P.n_cams
    n_cells, n_blocks = _o.bucket_plan(P)
    tiles_per_gauss, rect2, cellhist, cell_count, cell_cursor, cell_offs, order, rect_sorted, counters, tile_count = _o._carve(
        dev, (total, 2 * total, n_blocks * n_cells, n_cells, n_cells, n_cells + 1, total, 2 * total, 4, n_tiles + 1))
    counters = counters.view(torch.int64)
    check(lib.misplat_bucket_count(C.byref(P), ptr(means2d), ptr(radii), ptr(tiles_per_gauss), ptr(rect2),
                                   ptr(cellhist), ptr(cell_count), ptr(counters), C.c_int32(0), stream_ptr()),
          "misplat_bucket_count")
    pend = _read_back(counters[0:1])
    check(lib.misplat_bucket_rows(C.byref(P), ptr(tiles_per_gauss), ptr(rect2), ptr(cellhist), ptr(cell_count),
                                  ptr(cell_cursor), ptr(cell_offs), ptr(order), ptr(rect_sorted), ptr(counters),
                                  ptr(tile_count), ptr(None), C.c_int32(0), stream_ptr()), "misplat_bucket_rows")
    pend.update(tiles_per_gauss=tiles_per_gauss, rect2=rect_sorted, order=order, counters=counters, tile_count=tile_count)
    return pend


@torch.no_grad()
def bin_tiles(P: Params, means2d: Tensor, radii: Tensor, depths: Tensor,
              pending: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    """Tile intersection + ordering + offsets.  One host read-back: n_isects.

    Result: ``flatten_ids[n_isects]`` (Gaussian rows in (tile, depth, id) order), ``isect_offsets[n_tiles + 1]`` (the
    last entry is n_isects), ``tiles_per_gauss``; in deterministic mode also ``slots`` (the emission slot of every
    sorted intersection: the row of the gradient slab)."""
    lib = _lib.load()
    dev = means2d.device
    total = P.n_gauss * P.n_cams
    n_tiles = P.tile_w * P.tile_h * P.n_cams
    deterministic = _o.DETERMINISTIC_BACKWARD
    pend = pending if pending is not None else start_binning(P, means2d, radii)
    tiles_per_gauss = pend["tiles_per_gauss"]
    pend["event"].synchronize()                                   # the one sync of the step (usually long past)
    n_isects = int(pend["host"][0])
    if n_isects >= 2 ** 31:
        raise _lib.MisplatError(f"{n_isects} tile intersections exceed int32 indexing")
    depths = depths.contiguous()
    out = dict(tiles_per_gauss=tiles_per_gauss, n_isects=n_isects, depths=depths, tile_ids=None, n_tiles=n_tiles)
    offsets, payload, flatten_ids, scratch, isect_gid = _o._carve(
        dev, (n_tiles + 2, n_isects, n_isects, 2 * n_isects, n_isects if deterministic else 0))
    cum = None
    if deterministic:                                         # emission slots index the gradient slab
        cum = (torch.cumsum(tiles_per_gauss, dim=0, dtype=torch.int64) - tiles_per_gauss).contiguous()
        out["cum"] = cum
    else:
        isect_gid = None
    check(lib.misplat_bucket_tiles(C.byref(P), ptr(pend["order"]), ptr(pend["rect2"]), ptr(pend["counters"]),
                                   ptr(pend["tile_count"]), ptr(offsets), ptr(cum), C.c_int64(n_isects), ptr(payload),
                                   ptr(isect_gid), stream_ptr()), "misplat_bucket_tiles")
    if n_isects > 0:
        check(lib.misplat_tile_sort(ptr(offsets), C.c_int32(n_tiles), C.c_int64(n_isects), ptr(depths), ptr(isect_gid),
                                    ptr(payload), ptr(flatten_ids), ptr(scratch), C.c_int32(3), stream_ptr()),
              "misplat_tile_sort")
    out.update(slots=payload if deterministic else None, flatten_ids=flatten_ids, isect_offsets=offsets[:n_tiles + 1])
    return out


@torch.no_grad()
def complete_bins(bins: Dict[str, Tensor]) -> Tensor:
    """``flatten_ids`` with every tile's list sorted to its end.  After a front-only forward only the head of each list
    is there (the part the compositing and the backward read); this sorts every bucket in full from ``payload``, which is
    still a permutation of it -- the heads come out as they were (they are the first entries of the sorted lists), so it may
    run before or after the backward."""
    part = bins.get("partial")
    if part is not None:
        n_tiles = bins["n_tiles"]
        if part["cap"] > 0:
            # (the bucket entries are positions in the cell-ordered row list: flags bit 2)
            check(_lib.load().misplat_tile_sort(ptr(part["offsets"]), C.c_int32(n_tiles), C.c_int64(part["cap"]),
                                                ptr(part["depth_sorted"]), ptr(part["row_map"]), ptr(part["payload"]),
                                                ptr(part["flatten_ids"]), ptr(part["scratch"]), C.c_int32(7), stream_ptr()),
                  "misplat_tile_sort")
        bins["partial"] = None
        _o.PATH_STATS["bins_completed"] += 1
    return bins["flatten_ids"]


@torch.no_grad()
def isect_ids(bins: Dict[str, Tensor]) -> Tensor:
    """gsplat's ``meta["isect_ids"]``: the sorted 64-bit keys (tile << 32 | depth bits), on demand."""
    lib = _lib.load()
    complete_bins(bins)
    n = bins["n_isects"]
    out = torch.empty(n, device=bins["flatten_ids"].device, dtype=torch.int64)
    if bins["tile_ids"] is None:                       # no sorted tile-id array exists: rebuild it from the offsets
        cnt = torch.diff(bins["isect_offsets"].long())
        bins["tile_ids"] = torch.repeat_interleave(torch.arange(cnt.numel(), device=cnt.device, dtype=torch.int32), cnt)
    check(lib.misplat_isect_ids(ptr(bins["tile_ids"]), ptr(bins["flatten_ids"]), ptr(bins["depths"]),
                                C.c_int64(n), ptr(out), stream_ptr()), "misplat_isect_ids")
    return out


def _cum_by_row(bins: Dict[str, Tensor]) -> Tensor:
    """First emission slot of every Gaussian row (deterministic backward only; built lazily)."""
    if "cum" not in bins:
        tpg = bins["tiles_per_gauss"]
        bins["cum"] = (torch.cumsum(tpg, dim=0, dtype=torch.int64) - tpg).contiguous()
    return bins["cum"]


# ----------------------------------------------------------------------------- blending

class _Blend(torch.autograd.Function):
    """Compositing of <= 4 colour channels + alpha + expected/median depth + normal."""

    @staticmethod
    def forward(ctx, means2d, conics, opac, colors, ray_ts, ray_planes, normals, Ks, P: Params,
                bins: Dict[str, Tensor], absgrad: bool, pass_index: int = 0):
        lib = _lib.load()
        ctx.pass_index = pass_index
        require_gpu(means2d)
        Cn, N, H, W = P.n_cams, P.n_gauss, P.height, P.width
        cd = colors.shape[-1]
        dev = means2d.device
        rows = Cn * N
        grec = torch.empty(rows, MISPLAT_REC, device=dev, dtype=torch.float32)
        check(lib.misplat_pack(C.c_int64(rows), C.c_int32(cd), ptr(means2d), ptr(conics), ptr(opac),
                               ptr(ray_ts), ptr(ray_planes), ptr(normals), ptr(colors), ptr(grec),
                               stream_ptr()), "misplat_pack")
        f = dict(device=dev, dtype=torch.float32)
        render = torch.empty(Cn, H, W, cd, **f)
        alpha = torch.empty(Cn, H, W, 1, **f)
        exp_depth = torch.empty(Cn, H, W, 1, **f)
        med_depth = torch.empty(Cn, H, W, 1, **f)
        normal = torch.empty(Cn, H, W, 3, **f)
        last_ids = torch.empty(Cn, H, W, device=dev, dtype=torch.int32)
        median_ids = torch.empty(Cn, H, W, device=dev, dtype=torch.int32)
        sched = _o._UnitSchedule(P, dev)
        with _o._timed("blend_fwd"):
            sched.before_forward(P)
            check(lib.misplat_blend_fwd(C.byref(P), C.c_int32(cd), ptr(Ks), ptr(grec), ptr(bins["flatten_ids"]),
                                        ptr(bins["isect_offsets"]), C.c_int64(bins["n_isects"]), ptr(render),
                                        ptr(alpha), ptr(exp_depth), ptr(med_depth), ptr(normal), ptr(last_ids),
                                        ptr(median_ids), stream_ptr()), "misplat_blend_fwd")
        sched.after_forward(P)
        ctx.sched = sched
        ctx.P, ctx.bins, ctx.absgrad, ctx.cd = P, bins, absgrad, cd
        ctx.means2d_ref = means2d if absgrad else None
        ctx.save_for_backward(grec, Ks, alpha, last_ids, median_ids, render)
        ctx.mark_non_differentiable(last_ids, median_ids)
        ctx.set_materialize_grads(False)               # absent upstream gradients arrive as None
        return render, alpha, exp_depth, med_depth, normal, last_ids, median_ids

    @staticmethod
    def backward(ctx, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal, _l, _m):
        ctx.sched.before_backward(ctx.P)
        v_grec, v_abs = _blend_backward(ctx, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal)
        _o._UnitSchedule.done(ctx.P)
        P, cd = ctx.P, ctx.cd
        Cn, N = P.n_cams, P.n_gauss
        g = v_grec.view(Cn, N, MISPLAT_REC)
        if ctx.absgrad:
            # gsplat convention: the 2-D |gradient| rides on the means2d tensor for the strategy; the 4-channel
            # passes of one render (rendering.py generic path) each add their channels' share
            # (|.| is taken per pass, so with more than one pass the sum is an upper bound of the one-pass value)
            parts = ctx.means2d_ref.__dict__.setdefault("_absgrad_parts", {})
            parts[ctx.pass_index] = v_abs.view(Cn, N, 2)
            ctx.means2d_ref.absgrad = parts[0] if len(parts) == 1 and 0 in parts else sum(parts.values())
        return (g[..., 0:2], g[..., 2:5], g[..., 5], g[..., 12:12 + cd], g[..., 6], g[..., 7:9],
                g[..., 9:12], None, None, None, None, None)


def blend(means2d, conics, opac, colors, ray_ts, ray_planes, normals, Ks, P: Params, bins, absgrad=False,
          pass_index: int = 0):
    """``pass_index``: which 4-channel pass of one render this is (rendering.py generic path); every pass adds its
    channels' share to ``means2d.absgrad``."""
    if colors.shape[-1] < 1 or colors.shape[-1] > 4:
        raise ValueError("blend() takes 1..4 colour channels per pass")
    if pass_index == 0:
        means2d.__dict__.pop("_absgrad_parts", None)
    args = [_o._f32(t, n) for t, n in ((means2d, "means2d"), (conics, "conics"), (opac, "opacities"),
                                    (colors, "colors"), (ray_ts, "ray_ts"), (ray_planes, "ray_planes"),
                                    (normals, "normals"), (Ks, "Ks"))]
    if args[0] is not means2d:
        raise ValueError("means2d must be contiguous float32 so that its .grad/.absgrad can be retained")
    return _Blend.apply(*args, P, bins, bool(absgrad), int(pass_index))


# ----------------------------------------------------------------------------- fused path

def _grads_of_pack(P: Params, v_means2d, v_grec):
    """Incoming gradients of (means2d, grec).  When the mean gradient is exactly the view of columns 0:2 of
    the packed rows that _BlendPacked returned (nothing else was accumulated into means2d), the kernel
    reads it from the rows and the strided copy is skipped (v_means2d -> None)."""
    rows = P.n_cams * P.n_gauss
    if v_grec is None:
        ref = v_means2d
        v_grec = torch.zeros(rows, MISPLAT_REC, device=ref.device, dtype=torch.float32)
    v_grec = _o._c(v_grec)
    if v_means2d is None:
        v_means2d = torch.zeros(rows, 2, device=v_grec.device, dtype=torch.float32)
    elif (v_means2d.data_ptr() == v_grec.data_ptr() and v_means2d.dtype == torch.float32
          and v_means2d.stride() == (P.n_gauss * MISPLAT_REC, MISPLAT_REC, 1)):
        v_means2d = None
    else:
        v_means2d = _o._c(v_means2d)
    return v_means2d, v_grec


class _ProjectPack(torch.autograd.Function):
    """projection + colour (SH or pass-through) -> packed blend records, one autograd node.

    Outputs (radii, means2d, depths, compensations, grec).  ``means2d`` is a separate
    differentiable output so that ``meta["means2d"].retain_grad()`` works (rade_gs_model.py:191-198);
    every other gradient travels in the packed rows ``v_grec`` (columns 0:2 of which are ignored
    here -- the mean2d gradient arrives through ``v_means2d``)."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, colors_rest, viewmats, Ks, P: Params, sh_degree,
                depth_channel: bool, prebin: Optional[dict] = None):
        lib = _lib.load()
        require_gpu(means, quats, scales, opacities, colors, viewmats, Ks)
        N, Cn = P.n_gauss, P.n_cams
        dev = means.device
        if sh_degree is not None:
            kd = colors.shape[1] if colors_rest is None else 1 + colors_rest.shape[1]
            deg, n_color, per_cam = int(sh_degree), 3, 0
        else:
            deg, kd, per_cam = -1, colors.shape[-1], int(colors.dim() == 3)
            n_color = kd
        if prebin is not None and _o.fused_entry_ok() and N > 0:
            # one host entry: projection, row bucketing, asynchronous n_isects read-back, colours
            want_grad = any(ctx.needs_input_grad[:6])
            want_aux = deg >= 0 and want_grad
            radii, means2d, depths, comps, grec, sh_aux, state = _o._raster_phase_a(
                P, means, quats, scales, opacities, colors, colors_rest, viewmats, Ks, deg, kd, n_color, per_cam,
                depth_channel, want_aux, want_grad)
            prebin["fused"] = state
            ctx.P, ctx.color_args = P, (deg, kd, n_color, per_cam)
            ctx.depth_slot = 12 + n_color if depth_channel else -1
            ctx.has_rest = colors_rest is not None
            ctx.has_aux = sh_aux is not None
            ctx.save_for_backward(means, quats, scales, opacities, colors, viewmats, Ks, radii, comps,
                                  colors_rest if colors_rest is not None else colors,
                                  sh_aux if sh_aux is not None else comps)
            ctx.mark_non_differentiable(radii, depths, comps)
            ctx.set_materialize_grads(False)
            return radii, means2d, depths, comps, grec
        radii = torch.empty(Cn, N, 2, device=dev, dtype=torch.int32)
        means2d = torch.empty(Cn, N, 2, device=dev, dtype=torch.float32)
        depths = torch.empty(Cn, N, device=dev, dtype=torch.float32)
        comps = torch.empty(Cn, N, device=dev, dtype=torch.float32)
        grec = torch.empty(Cn * N, MISPLAT_REC, device=dev, dtype=torch.float32)
        check(lib.misplat_project_pack_fwd(C.byref(P), ptr(means), ptr(quats), ptr(scales), ptr(opacities),
                                           ptr(viewmats), ptr(Ks), ptr(radii), ptr(means2d), ptr(depths),
                                           ptr(comps), ptr(grec), ptr(None), C.c_int32(0), ptr(None), ptr(None), stream_ptr()),
              "misplat_project_pack_fwd")
        if prebin is not None:                         # count tiles + start the n_isects read-back before the colours
            prebin["pending"] = start_binning(P, means2d, radii)
        # SH + a backward to come: keep d rgb / d dir (48 B per (camera, Gaussian)) so that the backward does not
        # read the coefficients (192 B at degree 3) again
        sh_aux = None
        if deg >= 0 and any(ctx.needs_input_grad[:6]):
            sh_aux = torch.empty(Cn * N, 12, device=dev, dtype=torch.float32)
        check(lib.misplat_color_fwd(C.byref(P), C.c_int32(deg), C.c_int32(kd), C.c_int32(n_color),
                                    C.c_int32(per_cam), C.c_int32(int(depth_channel)), ptr(means), ptr(viewmats),
                                    ptr(colors), ptr(colors_rest), ptr(radii), ptr(depths), ptr(grec), ptr(sh_aux),
                                    ptr(None), stream_ptr()), "misplat_color_fwd")
        ctx.P, ctx.color_args = P, (deg, kd, n_color, per_cam)
        ctx.depth_slot = 12 + n_color if depth_channel else -1
        ctx.has_rest = colors_rest is not None
        ctx.has_aux = sh_aux is not None
        ctx.save_for_backward(means, quats, scales, opacities, colors, viewmats, Ks, radii, comps,
                              colors_rest if colors_rest is not None else colors,
                              sh_aux if sh_aux is not None else comps)
        ctx.mark_non_differentiable(radii, depths, comps)
        ctx.set_materialize_grads(False)               # no zero tensors for the non-differentiable outputs
        return radii, means2d, depths, comps, grec

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, _v_depths, _v_comps, v_grec):
        lib = _lib.load()
        means, quats, scales, opacities, colors, viewmats, Ks, radii, comps, colors_rest, sh_aux = ctx.saved_tensors
        if not ctx.has_rest:
            colors_rest = None
        if not ctx.has_aux:
            sh_aux = None
        P = ctx.P
        deg, kd, n_color, per_cam = ctx.color_args
        v_means2d, v_grec = _grads_of_pack(P, v_means2d, v_grec)
        v_colors = _o._grad_out(colors)
        v_colors_rest = _o._grad_out(colors_rest) if colors_rest is not None else None
        v_means_dir = torch.empty_like(means) if deg >= 0 else None
        check(lib.misplat_color_bwd(C.byref(P), C.c_int32(deg), C.c_int32(kd), C.c_int32(n_color),
                                    C.c_int32(per_cam), ptr(means), ptr(viewmats), ptr(colors), ptr(colors_rest),
                                    ptr(radii), ptr(v_grec), ptr(v_colors), ptr(v_colors_rest), ptr(v_means_dir),
                                    ptr(sh_aux), stream_ptr()), "misplat_color_bwd")
        if _o.GRAD_SINK is not None:
            _o.GRAD_SINK.colour_ready()                   # the colour bucket's all-reduce starts now, overlapped with the rest
        v_means, v_quats = _o._grad_out(means), _o._grad_out(quats)
        v_scales, v_opac = _o._grad_out(scales), _o._grad_out(opacities)
        check(lib.misplat_project_pack_bwd(C.byref(P), C.c_int32(ctx.depth_slot), ptr(means), ptr(quats),
                                           ptr(scales), ptr(opacities), ptr(viewmats), ptr(Ks), ptr(radii),
                                           ptr(comps), ptr(v_means2d), ptr(v_grec), ptr(v_means_dir), ptr(v_means),
                                           ptr(v_quats), ptr(v_scales), ptr(v_opac), None, C.c_int32(0), stream_ptr()),
              "misplat_project_pack_bwd")
        return v_means, v_quats, v_scales, v_opac, v_colors, v_colors_rest, None, None, None, None, None, None


def project_pack(means, quats, scales, opacities, colors, viewmats, Ks, P: Params, sh_degree, depth_channel,
                 prebin: Optional[dict] = None):
    """``colors`` may be a pair (features_dc [N,3], features_rest [N,K-1,3]) when ``sh_degree`` is given.
    ``prebin``: a dict that receives ``["pending"]`` = ``start_binning(...)`` for ``bin_tiles(pending=...)``."""
    rest = None
    if isinstance(colors, (tuple, list)):
        colors, rest = colors
        rest = _o._f32(rest, "features_rest")
    args = [_o._f32(t, n) for t, n in ((means, "means"), (quats, "quats"), (scales, "scales"),
                                    (opacities, "opacities"), (colors, "colors"))]
    return _ProjectPack.apply(*args, rest, _o._f32(viewmats, "viewmats"), _o._f32(Ks, "Ks"), P, sh_degree,
                              bool(depth_channel), prebin)


class _BlendPacked(torch.autograd.Function):
    """Compositing straight from the packed records (<= 4 colour slots)."""

    @staticmethod
    def forward(ctx, means2d, grec, Ks, P: Params, bins: Dict[str, Tensor], absgrad: bool, cd: int):
        lib = _lib.load()
        Cn, H, W = P.n_cams, P.height, P.width
        dev = grec.device
        if "args" in bins:                             # phase-A state of the one-entry path: buckets, sort, compositing
            imgs, done, sched = _o._raster_phase_b(P, bins, cd)
            bins.clear()
            bins.update(done)                          # the caller's dict becomes the finished bins (meta reads it)
            render, alpha, exp_depth, med_depth, normal, last_ids, median_ids = imgs
            ctx.sched = sched
            ctx.P, ctx.bins, ctx.absgrad, ctx.cd = P, bins, absgrad, cd
            ctx.means2d_ref = means2d if absgrad else None
            ctx.save_for_backward(grec, Ks, alpha, last_ids, median_ids, render)
            ctx.mark_non_differentiable(last_ids, median_ids)
            ctx.set_materialize_grads(False)
            return render, alpha, exp_depth, med_depth, normal, last_ids, median_ids
        f = dict(device=dev, dtype=torch.float32)
        render = torch.empty(Cn, H, W, cd, **f)
        alpha = torch.empty(Cn, H, W, 1, **f)
        exp_depth = torch.empty(Cn, H, W, 1, **f)
        med_depth = torch.empty(Cn, H, W, 1, **f)
        normal = torch.empty(Cn, H, W, 3, **f)
        last_ids = torch.empty(Cn, H, W, device=dev, dtype=torch.int32)
        median_ids = torch.empty(Cn, H, W, device=dev, dtype=torch.int32)
        sched = _o._UnitSchedule(P, dev)
        with _o._timed("blend_fwd"):
            sched.before_forward(P)
            check(lib.misplat_blend_fwd(C.byref(P), C.c_int32(cd), ptr(Ks), ptr(grec), ptr(bins["flatten_ids"]),
                                        ptr(bins["isect_offsets"]), C.c_int64(bins["n_isects"]), ptr(render),
                                        ptr(alpha), ptr(exp_depth), ptr(med_depth), ptr(normal), ptr(last_ids),
                                        ptr(median_ids), stream_ptr()), "misplat_blend_fwd")
        sched.after_forward(P)
        ctx.sched = sched
        ctx.P, ctx.bins, ctx.absgrad, ctx.cd = P, bins, absgrad, cd
        ctx.means2d_ref = means2d if absgrad else None
        ctx.save_for_backward(grec, Ks, alpha, last_ids, median_ids, render)
        ctx.mark_non_differentiable(last_ids, median_ids)
        ctx.set_materialize_grads(False)               # absent upstream gradients arrive as None
        return render, alpha, exp_depth, med_depth, normal, last_ids, median_ids

    @staticmethod
    def backward(ctx, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal, _l, _m):
        ctx.sched.before_backward(ctx.P)
        v_grec, v_abs = _blend_backward(ctx, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal)
        _o._UnitSchedule.done(ctx.P)
        P = ctx.P
        if ctx.absgrad:
            ctx.means2d_ref.absgrad = v_abs.view(P.n_cams, P.n_gauss, 2)
        return v_grec.view(P.n_cams, P.n_gauss, MISPLAT_REC)[..., 0:2], v_grec, None, None, None, None, None


def _blend_backward(ctx, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal):
    """blend_bwd -> per-intersection rows -> fixed-order per-Gaussian sum.  Returns (v_grec, v_abs)."""
    lib = _lib.load()
    grec, Ks, alpha, last_ids, median_ids, render = ctx.saved_tensors
    P, bins, cd = ctx.P, ctx.bins, ctx.cd
    n_isects = bins["n_isects"]
    rows = P.n_cams * P.n_gauss
    dev = grec.device
    ups = _o._upstream(P, cd, dev, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal)
    if bins["slots"] is None:                      # binned in atomic mode (ops.DETERMINISTIC_BACKWARD was False)
        # the one-entry forward left a cleared gradient buffer behind (written by the colour kernel): first backward only
        v_grec = bins.pop("v_grec_zero", None)
        prezeroed = v_grec is not None and not bins.get("rows_on_touch")
        if v_grec is None:
            v_grec = torch.empty(rows, MISPLAT_REC, device=dev, dtype=torch.float32)
        v_abs = torch.empty(rows, 2, device=dev, dtype=torch.float32) if ctx.absgrad else None
        with _o._timed("blend_bwd"):
            check(lib.misplat_blend_bwd_atomic(C.byref(P), C.c_int32(cd), ptr(Ks), ptr(grec), ptr(bins["flatten_ids"]),
                                               ptr(bins["isect_offsets"]), C.c_int64(n_isects), ptr(alpha),
                                               ptr(last_ids), ptr(median_ids), ptr(render), *[ptr(t) for t in ups],
                                               ptr(v_grec), ptr(v_abs), C.c_int32(int(prezeroed)), stream_ptr()),
                  "misplat_blend_bwd_atomic")
        return v_grec, v_abs
    planes = int(lib.misplat_blend_planes(C.byref(P)))
    rows_s = max(n_isects, 1) * planes
    slab = torch.empty(rows_s, MISPLAT_REC, device=dev, dtype=torch.float32)
    slab_abs = torch.empty(rows_s, 2, device=dev, dtype=torch.float32) if ctx.absgrad else None
    slab_valid = torch.empty(rows_s, device=dev, dtype=torch.uint8)
    with _o._timed("blend_bwd"):
        check(lib.misplat_blend_bwd(C.byref(P), C.c_int32(cd), ptr(Ks), ptr(grec), ptr(bins["flatten_ids"]),
                                    ptr(bins["slots"]), ptr(bins["isect_offsets"]), C.c_int64(n_isects),
                                    ptr(alpha), ptr(last_ids), ptr(median_ids), ptr(render), *[ptr(t) for t in ups],
                                    ptr(slab), ptr(slab_abs), ptr(slab_valid), stream_ptr()), "misplat_blend_bwd")
    v_grec = torch.empty(rows, MISPLAT_REC, device=dev, dtype=torch.float32)
    v_abs = torch.empty(rows, 2, device=dev, dtype=torch.float32) if ctx.absgrad else None
    with _o._timed("slab_reduce"):
        check(lib.misplat_slab_reduce(C.byref(P), C.c_int64(rows), C.c_int64(n_isects), ptr(_cum_by_row(bins)),
                                      ptr(bins["tiles_per_gauss"]), ptr(slab), ptr(slab_abs), ptr(slab_valid),
                                      ptr(v_grec), ptr(v_abs), stream_ptr()), "misplat_slab_reduce")
    return v_grec, v_abs


def blend_packed(means2d, grec, Ks, P: Params, bins, absgrad: bool, cd: int):
    if not 1 <= cd <= 4:
        raise ValueError("blend_packed() takes 1..4 colour channels")
    return _BlendPacked.apply(means2d, grec, _o._f32(Ks, "Ks"), P, bins, bool(absgrad), int(cd))


# ----------------------------------------------------------------------------- N-D colours (a8)

class _ProjectPackX(torch.autograd.Function):
    """projection + pass-through colours with D' = D (+ depth) in 5..20 channels: channels 0..3 go to the record,
    the rest to ``featx[C*N, 4*nxq]`` (rade_features_model.py:441-476 renders 16 fused channels, 17 with ED)."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmats, Ks, P: Params, depth_channel: bool, nxq: int):
        lib = _lib.load()
        require_gpu(means, quats, scales, opacities, colors, viewmats, Ks)
        N, Cn = P.n_gauss, P.n_cams
        dev = means.device
        radii = torch.empty(Cn, N, 2, device=dev, dtype=torch.int32)
        means2d = torch.empty(Cn, N, 2, device=dev, dtype=torch.float32)
        depths = torch.empty(Cn, N, device=dev, dtype=torch.float32)
        comps = torch.empty(Cn, N, device=dev, dtype=torch.float32)
        grec = torch.empty(Cn * N, MISPLAT_REC, device=dev, dtype=torch.float32)
        featx = torch.empty(Cn * N, 4 * nxq, device=dev, dtype=torch.float32)
        check(lib.misplat_project_pack_fwd(C.byref(P), ptr(means), ptr(quats), ptr(scales), ptr(opacities),
                                           ptr(viewmats), ptr(Ks), ptr(radii), ptr(means2d), ptr(depths),
                                           ptr(comps), ptr(grec), ptr(None), C.c_int32(0), ptr(None), ptr(None), stream_ptr()),
              "misplat_project_pack_fwd")
        D, per_cam = colors.shape[-1], int(colors.dim() == 3)
        check(lib.misplat_color_fwd_x(C.byref(P), C.c_int32(D), C.c_int32(per_cam), C.c_int32(int(depth_channel)),
                                      C.c_int32(nxq), ptr(colors), ptr(radii), ptr(depths), ptr(grec), ptr(featx),
                                      stream_ptr()), "misplat_color_fwd_x")
        ctx.P, ctx.D, ctx.per_cam, ctx.nxq, ctx.depth_channel = P, D, per_cam, nxq, depth_channel
        ctx.save_for_backward(means, quats, scales, opacities, colors, viewmats, Ks, radii, comps)
        ctx.mark_non_differentiable(radii, depths, comps)
        ctx.set_materialize_grads(False)
        return radii, means2d, depths, comps, grec, featx

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, _v_depths, _v_comps, v_grec, v_featx):
        lib = _lib.load()
        means, quats, scales, opacities, colors, viewmats, Ks, radii, comps = ctx.saved_tensors
        P, D, nxq = ctx.P, ctx.D, ctx.nxq
        v_means2d, v_grec = _grads_of_pack(P, v_means2d, v_grec)
        v_featx = _o._c(v_featx) if v_featx is not None else torch.zeros(v_grec.shape[0], 4 * nxq, device=v_grec.device)
        v_colors = torch.empty_like(colors)
        check(lib.misplat_color_bwd_x(C.byref(P), C.c_int32(D), C.c_int32(ctx.per_cam), C.c_int32(nxq), ptr(radii),
                                      ptr(v_grec), ptr(v_featx), ptr(v_colors), stream_ptr()), "misplat_color_bwd_x")
        depth_slot, v_depth_rows, v_depth_stride = -1, None, 0
        if ctx.depth_channel:                       # channel D carries the depth
            if D < 4:
                depth_slot = 12 + D
            else:                                   # it lives in featx: the projection backward reads it from there (float D - 4
                #                                     of every v_featx row) -- no copy of the 64-byte gradient rows to park it in
                v_depth_rows = C.c_void_p(v_featx.data_ptr() + 4 * (D - 4))
                v_depth_stride = 4 * nxq
        v_means, v_quats = torch.empty_like(means), torch.empty_like(quats)
        v_scales, v_opac = torch.empty_like(scales), torch.empty_like(opacities)
        check(lib.misplat_project_pack_bwd(C.byref(P), C.c_int32(depth_slot), ptr(means), ptr(quats), ptr(scales),
                                           ptr(opacities), ptr(viewmats), ptr(Ks), ptr(radii), ptr(comps),
                                           ptr(v_means2d), ptr(v_grec), ptr(None), ptr(v_means), ptr(v_quats),
                                           ptr(v_scales), ptr(v_opac), v_depth_rows, C.c_int32(v_depth_stride), stream_ptr()),
              "misplat_project_pack_bwd")
        return v_means, v_quats, v_scales, v_opac, v_colors, None, None, None, None, None


def project_pack_x(means, quats, scales, opacities, colors, viewmats, Ks, P: Params, depth_channel: bool, nxq: int):
    args = [_o._f32(t, n) for t, n in ((means, "means"), (quats, "quats"), (scales, "scales"),
                                    (opacities, "opacities"), (colors, "colors"), (viewmats, "viewmats"), (Ks, "Ks"))]
    return _ProjectPackX.apply(*args, P, bool(depth_channel), int(nxq))


class _BlendPackedX(torch.autograd.Function):
    """One-pass compositing of 5..20 colour channels (atomic gradient mode)."""

    @staticmethod
    def forward(ctx, means2d, grec, featx, Ks, P: Params, bins: Dict[str, Tensor], absgrad: bool, n_channels: int,
                nxq: int):
        lib = _lib.load()
        Cn, H, W = P.n_cams, P.height, P.width
        dev = grec.device
        f = dict(device=dev, dtype=torch.float32)
        render = torch.empty(Cn, H, W, n_channels, **f)
        alpha = torch.empty(Cn, H, W, 1, **f)
        exp_depth = torch.empty(Cn, H, W, 1, **f)
        med_depth = torch.empty(Cn, H, W, 1, **f)
        normal = torch.empty(Cn, H, W, 3, **f)
        last_ids = torch.empty(Cn, H, W, device=dev, dtype=torch.int32)
        median_ids = torch.empty(Cn, H, W, device=dev, dtype=torch.int32)
        sched = _o._UnitSchedule(P, dev)
        ctx.sched = sched
        with _o._timed("blend_fwd"):
            sched.before_forward(P)
            check(lib.misplat_blend_fwd_x(C.byref(P), C.c_int32(n_channels), C.c_int32(nxq), ptr(Ks), ptr(grec),
                                          ptr(featx), ptr(bins["flatten_ids"]), ptr(bins["isect_offsets"]),
                                          C.c_int64(bins["n_isects"]), ptr(render), ptr(alpha), ptr(exp_depth),
                                          ptr(med_depth), ptr(normal), ptr(last_ids), ptr(median_ids), stream_ptr()),
                  "misplat_blend_fwd_x")
        sched.after_forward(P)
        ctx.P, ctx.bins, ctx.absgrad, ctx.n_channels, ctx.nxq = P, bins, absgrad, n_channels, nxq
        ctx.means2d_ref = means2d if absgrad else None
        ctx.save_for_backward(grec, featx, Ks, alpha, last_ids, median_ids, render)
        ctx.mark_non_differentiable(last_ids, median_ids)
        ctx.set_materialize_grads(False)               # absent upstream gradients arrive as None
        return render, alpha, exp_depth, med_depth, normal, last_ids, median_ids

    @staticmethod
    def backward(ctx, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal, _l, _m):
        lib = _lib.load()
        grec, featx, Ks, alpha, last_ids, median_ids, render = ctx.saved_tensors
        P, bins = ctx.P, ctx.bins
        rows = P.n_cams * P.n_gauss
        dev = grec.device
        v_grec = torch.empty(rows, MISPLAT_REC, device=dev, dtype=torch.float32)
        v_featx = torch.empty(rows, 4 * ctx.nxq, device=dev, dtype=torch.float32)
        v_abs = torch.empty(rows, 2, device=dev, dtype=torch.float32) if ctx.absgrad else None
        ups = _o._upstream(P, ctx.n_channels, dev, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal)
        ctx.sched.before_backward(P)
        with _o._timed("blend_bwd"):
            check(lib.misplat_blend_bwd_x_atomic(C.byref(P), C.c_int32(ctx.n_channels), C.c_int32(ctx.nxq), ptr(Ks),
                                                 ptr(grec), ptr(featx), ptr(bins["flatten_ids"]),
                                                 ptr(bins["isect_offsets"]), C.c_int64(bins["n_isects"]), ptr(alpha),
                                                 ptr(last_ids), ptr(median_ids), ptr(render), *[ptr(t) for t in ups],
                                                 ptr(v_grec), ptr(v_featx), ptr(v_abs), stream_ptr()),
                  "misplat_blend_bwd_x_atomic")
        _o._UnitSchedule.done(P)
        if ctx.absgrad:
            ctx.means2d_ref.absgrad = v_abs.view(P.n_cams, P.n_gauss, 2)
        return (v_grec.view(P.n_cams, P.n_gauss, MISPLAT_REC)[..., 0:2], v_grec, v_featx, None, None, None, None, None,
                None)


def blend_packed_x(means2d, grec, featx, Ks, P: Params, bins, absgrad: bool, n_channels: int, nxq: int):
    return _BlendPackedX.apply(means2d, grec, featx, _o._f32(Ks, "Ks"), P, bins, bool(absgrad), int(n_channels), int(nxq))
