"""The model-level stages around the rasterizer as autograd nodes (plumbing around the C ABI; no kernels here): the
depth -> normal stencil (a4), the ``get_outputs`` epilogue (a3, and a3 + a4 as one node) and the loss means with SSIM (a5, a5').
SURVEY.md section 8 rows a3 - a5; reference: /root/reference/collab_splats/models/rade_gs_model.py:200-309,
/root/reference/collab_splats/utils/camera_utils.py:176-279.  Every public name here is reachable as ``ops.X`` too."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import check, ptr, require_gpu, stream_ptr


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else t.contiguous()


def _f32(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference trains in fp32, "
                        f"rade_gs_method.py:31); got {t.dtype}")
    return t.contiguous()


# ----------------------------------------------------------------------------- depth -> normal

class _DepthNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, exp_depth, med_depth, n_render, fx: float, fy: float):
        lib = _lib.load()
        require_gpu(exp_depth, med_depth, n_render)
        H, W = exp_depth.shape[-2], exp_depth.shape[-1]
        dev = exp_depth.device
        normals2 = torch.empty(2, H, W, 3, device=dev, dtype=torch.float32)
        err = torch.empty(2, H, W, device=dev, dtype=torch.float32)
        check(lib.misplat_depth_normal_fwd(C.c_int32(W), C.c_int32(H), C.c_float(fx), C.c_float(fy),
                                           ptr(exp_depth), ptr(med_depth), ptr(n_render), ptr(normals2),
                                           ptr(err), stream_ptr()), "misplat_depth_normal_fwd")
        ctx.save_for_backward(exp_depth, med_depth, n_render)
        ctx.fx, ctx.fy = fx, fy
        return normals2, err

    @staticmethod
    def backward(ctx, v_normals2, v_err):
        lib = _lib.load()
        ed, md, nr = ctx.saved_tensors
        H, W = ed.shape[-2], ed.shape[-1]
        v_ed, v_md, v_nr = torch.empty_like(ed), torch.empty_like(md), torch.empty_like(nr)
        check(lib.misplat_depth_normal_bwd(C.c_int32(W), C.c_int32(H), C.c_float(ctx.fx), C.c_float(ctx.fy),
                                           ptr(ed), ptr(md), ptr(nr), ptr(_c(v_normals2)), ptr(_c(v_err)),
                                           ptr(v_ed), ptr(v_md), ptr(v_nr), C.c_int32(0), stream_ptr()),
              "misplat_depth_normal_bwd")
        return v_ed, v_md, v_nr, None, None


def depth_normal(exp_depth: Tensor, med_depth: Tensor, n_render: Tensor, fx: float, fy: float):
    """Fused camera_utils.depth_double_to_normal + error map.  exp/med_depth [H,W], n_render [H,W,3]
    -> (normals2 [2,H,W,3], err [2,H,W])."""
    return _DepthNormal.apply(_f32(exp_depth, "exp_depth"), _f32(med_depth, "med_depth"),
                              _f32(n_render, "n_render"), float(fx), float(fy))


# ----------------------------------------------------------------------------- get_outputs epilogue (a3)

class _Outputs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render, alpha, exp_depth, med_depth, exp_normal, bg, want_depth_im: bool):
        lib = _lib.load()
        require_gpu(render, alpha, exp_depth, med_depth, exp_normal)
        cd = render.shape[-1]
        n_pix = alpha.numel()
        dev = render.device
        f = dict(device=dev, dtype=torch.float32)
        rgb = torch.empty(alpha.shape[:-1] + (3,), **f)
        depth, median = torch.empty_like(alpha), torch.empty_like(alpha)
        normals = torch.empty_like(exp_normal)
        depth_im = torch.empty_like(alpha) if want_depth_im else None
        maxes = torch.empty(4, **f)
        bg_c = (C.c_float * 3)(*[float(b) for b in bg])
        check(lib.misplat_outputs_fwd(C.c_int64(n_pix), C.c_int32(cd), bg_c, ptr(render), ptr(alpha), ptr(exp_depth),
                                      ptr(med_depth), ptr(exp_normal), ptr(maxes), ptr(rgb), ptr(depth), ptr(median),
                                      ptr(normals), ptr(depth_im), stream_ptr()), "misplat_outputs_fwd")
        ctx.save_for_backward(render, alpha)
        ctx.bg, ctx.cd, ctx.want_depth_im = bg_c, cd, want_depth_im
        if want_depth_im:
            return rgb, depth, median, normals, depth_im
        return rgb, depth, median, normals

    @staticmethod
    def backward(ctx, v_rgb, v_depth, v_median, v_normals, v_depth_im=None):
        lib = _lib.load()
        render, alpha = ctx.saved_tensors
        v_render = torch.empty_like(render)
        v_alpha, v_ed, v_md = torch.empty_like(alpha), torch.empty_like(alpha), torch.empty_like(alpha)
        v_nr = torch.empty_like(v_normals)
        ups = [_c(t) for t in (v_rgb, v_depth, v_median, v_normals)]
        vdi = _c(v_depth_im) if ctx.want_depth_im else None
        check(lib.misplat_outputs_bwd(C.c_int64(alpha.numel()), C.c_int32(ctx.cd), ctx.bg, ptr(render), ptr(alpha),
                                      *[ptr(t) for t in ups], ptr(vdi), ptr(v_render), ptr(v_alpha), ptr(v_ed),
                                      ptr(v_md), ptr(v_nr), stream_ptr()), "misplat_outputs_bwd")
        return v_render, v_alpha, v_ed, v_md, v_nr, None, None


def outputs_epilogue(render, alpha, exp_depth, med_depth, exp_normal, background, want_depth_im: bool):
    """rade_gs_model.py:221-254 in three kernels.  ``background``: 3 Python floats."""
    if render.shape[-1] not in (3, 4) or (want_depth_im and render.shape[-1] != 4):
        raise ValueError("outputs_epilogue needs render[..., 3] (RGB) or [..., 4] (RGB+ED)")
    args = [_f32(t, n) for t, n in ((render, "render"), (alpha, "alpha"), (exp_depth, "expected_depths"),
                                    (med_depth, "median_depths"), (exp_normal, "expected_normals"))]
    return _Outputs.apply(*args, tuple(float(b) for b in background), bool(want_depth_im))


class _GetOutputs(torch.autograd.Function):
    """a3 + a4 in ONE autograd node: the depth->normal error maps
    (rade_gs_model.py:206-214) and the output post-processing (:221-254) share their inputs, so the backward is
    outputs_bwd followed by depth_normal_bwd ACCUMULATING into the same five gradient buffers -- no autograd
    add kernels, and upstream gradients of outputs that took no part in the loss stay NULL instead of being
    materialised as zero tensors."""

    @staticmethod
    def forward(ctx, render, alpha, exp_depth, med_depth, exp_normal, bg, want_depth_im: bool, fx: float, fy: float):
        lib = _lib.load()
        require_gpu(render, alpha, exp_depth, med_depth, exp_normal)
        if alpha.shape[0] != 1:
            raise ValueError("get_outputs epilogue: one camera per call (rade_gs_model.py:94-95)")
        cd = render.shape[-1]
        H, W = alpha.shape[1], alpha.shape[2]
        f = dict(device=render.device, dtype=torch.float32)
        rgb = torch.empty(1, H, W, 3, **f)
        depth, median = torch.empty_like(alpha), torch.empty_like(alpha)
        normals = torch.empty_like(exp_normal)
        depth_im = torch.empty_like(alpha) if want_depth_im else None
        maxes = torch.empty(4, **f)
        err = torch.empty(2, H, W, **f)
        normals2 = torch.empty(2, H, W, 3, **f)
        bg_c = (C.c_float * 3)(*[float(b) for b in bg])
        check(lib.misplat_depth_normal_fwd(C.c_int32(W), C.c_int32(H), C.c_float(fx), C.c_float(fy), ptr(exp_depth),
                                           ptr(med_depth), ptr(exp_normal), ptr(normals2), ptr(err), stream_ptr()),
              "misplat_depth_normal_fwd")
        check(lib.misplat_outputs_fwd(C.c_int64(H * W), C.c_int32(cd), bg_c, ptr(render), ptr(alpha), ptr(exp_depth),
                                      ptr(med_depth), ptr(exp_normal), ptr(maxes), ptr(rgb), ptr(depth), ptr(median),
                                      ptr(normals), ptr(depth_im), stream_ptr()), "misplat_outputs_fwd")
        ctx.save_for_backward(render, alpha, exp_depth, med_depth, exp_normal)
        ctx.bg, ctx.cd, ctx.want_depth_im, ctx.fx, ctx.fy = bg_c, cd, want_depth_im, fx, fy
        ctx.set_materialize_grads(False)
        if want_depth_im:
            return rgb, depth, median, normals, err, depth_im
        return rgb, depth, median, normals, err

    @staticmethod
    def backward(ctx, v_rgb, v_depth, v_median, v_normals, v_err, v_depth_im=None):
        lib = _lib.load()
        render, alpha, ed, md, nr = ctx.saved_tensors
        H, W = alpha.shape[1], alpha.shape[2]
        v_render = torch.empty_like(render)
        v_alpha, v_ed, v_md = torch.empty_like(alpha), torch.empty_like(alpha), torch.empty_like(alpha)
        v_nr = torch.empty_like(nr)
        check(lib.misplat_outputs_bwd(C.c_int64(H * W), C.c_int32(ctx.cd), ctx.bg, ptr(render), ptr(alpha),
                                      ptr(_c(v_rgb)), ptr(_c(v_depth)), ptr(_c(v_median)), ptr(_c(v_normals)),
                                      ptr(_c(v_depth_im) if ctx.want_depth_im else None), ptr(v_render), ptr(v_alpha),
                                      ptr(v_ed), ptr(v_md), ptr(v_nr), stream_ptr()), "misplat_outputs_bwd")
        if v_err is not None:
            check(lib.misplat_depth_normal_bwd(C.c_int32(W), C.c_int32(H), C.c_float(ctx.fx), C.c_float(ctx.fy),
                                               ptr(ed), ptr(md), ptr(nr), ptr(None), ptr(_c(v_err)), ptr(v_ed),
                                               ptr(v_md), ptr(v_nr), C.c_int32(1), stream_ptr()),
                  "misplat_depth_normal_bwd")
        return v_render, v_alpha, v_ed, v_md, v_nr, None, None, None, None


def get_outputs_epilogue(render, alpha, exp_depth, med_depth, exp_normal, background, want_depth_im: bool, fx: float,
                         fy: float):
    """(rgb, depth, median_depth, normals, err[2,H,W] (, depth_im)) -- rade_gs_model.py:206-254 as one node."""
    if render.shape[-1] not in (3, 4) or (want_depth_im and render.shape[-1] != 4):
        raise ValueError("get_outputs_epilogue needs render[..., 3] (RGB) or [..., 4] (RGB+ED)")
    args = [_f32(t, n) for t, n in ((render, "render"), (alpha, "alpha"), (exp_depth, "expected_depths"),
                                    (med_depth, "median_depths"), (exp_normal, "expected_normals"))]
    return _GetOutputs.apply(*args, tuple(float(b) for b in background), bool(want_depth_im), float(fx), float(fy))


LOSS_PARTIALS = 3 * 512      # MISPLAT_LOSS_PARTIALS


class _MeanLosses(torch.autograd.Function):
    """a5 (rade_gs_model.py:289-307 + the base model's L1 term) as ONE autograd node: forward = two launches
    (misplat_loss_fwd), backward = one (misplat_loss_bwd) that writes the gradient images the a3 + a4 node consumes --
    instead of ~30 elementwise / reduction launches of a few microseconds each.  ``err``: the [2,H,W] tensor whose halves
    are the two error maps (then e1 / e2 are ignored), or None with e1 / e2 given separately."""

    @staticmethod
    def forward(ctx, rgb, gt, err, e1, e2, depth_ratio: float, lam: float, ssim_lambda: float = 0.0):
        lib = _lib.load()
        with_rgb = rgb is not None and gt is not None
        if err is not None:
            e1, e2 = err[0], err[1]
        with_dn = e1 is not None and e2 is not None
        require_gpu(*[t for t in (rgb, gt, e1, e2) if t is not None])
        n_pix = rgb.numel() // 3 if with_rgb else e1.numel()
        if with_rgb and with_dn and e1.numel() != n_pix:
            raise ValueError("get_loss_dict: the error maps and the image differ in size")
        dev = rgb.device if with_rgb else e1.device
        with_ssim = with_rgb and ssim_lambda > 0.0               # (the SSIM forward then sums the L1 term too: its tiles hold both images)
        l1_here = with_rgb and not with_ssim
        rgb_loss = torch.empty((), device=dev, dtype=torch.float32) if l1_here else None
        dn_loss = torch.empty((), device=dev, dtype=torch.float32) if with_dn else None
        if l1_here or with_dn:
            partials = torch.empty(LOSS_PARTIALS, device=dev, dtype=torch.float32)
            check(lib.misplat_loss_fwd(C.c_int64(n_pix), ptr(rgb if l1_here else None), ptr(gt if l1_here else None),
                                       ptr(e1 if with_dn else None), ptr(e2 if with_dn else None), C.c_float(depth_ratio),
                                       C.c_float(lam), ptr(partials), ptr(rgb_loss), ptr(dn_loss), stream_ptr()), "misplat_loss_fwd")
        # the base model's image loss (Splatfacto: (1 - l) L1 + l (1 - SSIM)): two launches that sum both terms tile by tile
        # and leave the derivative maps of the SSIM for the backward
        ctx.ssim = None
        if with_ssim:
            H, W = int(rgb.shape[-3]), int(rgb.shape[-2])
            if rgb.shape[-1] != 3 or rgb.numel() != 3 * H * W:
                raise ValueError("mean_losses: the SSIM term takes one [H,W,3] image")
            n_scratch = int(lib.misplat_ssim_scratch_floats(C.c_int32(H), C.c_int32(W)))
            if n_scratch < 0:
                raise ValueError(f"mean_losses: the SSIM window needs an image of at least 11 x 11 pixels (got {H} x {W})")
            scratch = torch.empty(n_scratch, device=dev, dtype=torch.float32)
            main = torch.empty((), device=dev, dtype=torch.float32)
            check(lib.misplat_ssim_fwd(C.c_int32(H), C.c_int32(W), ptr(rgb), ptr(gt), ptr(scratch), None,
                                       C.c_float(ssim_lambda), None, ptr(main), stream_ptr()), "misplat_ssim_fwd")
            ctx.ssim = (H, W, scratch, float(ssim_lambda))
            rgb_loss = main
        ctx.save_for_backward(*(t for t in (rgb, gt) if with_rgb))
        ctx.with_rgb, ctx.with_dn, ctx.packed, ctx.n_pix = with_rgb, with_dn, err is not None, n_pix
        ctx.err_shape = tuple(err.shape) if err is not None else (tuple(e1.shape) if with_dn else None)
        ctx.k = (float(depth_ratio), float(lam))
        ctx.dev = dev
        ctx.set_materialize_grads(False)
        return rgb_loss, dn_loss

    @staticmethod
    def backward(ctx, g_rgb, g_dn):
        lib = _lib.load()
        rgb, gt = ctx.saved_tensors if ctx.with_rgb else (None, None)
        want_rgb = ctx.with_rgb and ctx.needs_input_grad[0] and g_rgb is not None
        want_dn = ctx.with_dn and g_dn is not None and (ctx.needs_input_grad[2] if ctx.packed
                                                       else (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]))
        v_rgb = torch.empty_like(rgb) if want_rgb else None
        v_err = v_e1 = v_e2 = None
        if want_dn:
            if ctx.packed:
                v_err = torch.empty(ctx.err_shape, device=ctx.dev, dtype=torch.float32)
                v_e1, v_e2 = v_err[0], v_err[1]
            else:
                v_e1 = torch.empty(ctx.err_shape, device=ctx.dev, dtype=torch.float32)
                v_e2 = torch.empty(ctx.err_shape, device=ctx.dev, dtype=torch.float32)
        g1 = g_rgb.to(torch.float32).contiguous() if want_rgb else None
        g2 = g_dn.to(torch.float32).contiguous() if want_dn else None
        l1_here = want_rgb and ctx.ssim is None
        if want_rgb and ctx.ssim is not None:                         # both halves of the image term in one launch
            H, W, scratch, ssim_lambda = ctx.ssim
            check(lib.misplat_ssim_bwd(C.c_int32(H), C.c_int32(W), ptr(rgb), ptr(gt), ptr(scratch), ptr(g1),
                                       C.c_float(ssim_lambda), ptr(v_rgb), stream_ptr()), "misplat_ssim_bwd")
        if l1_here or want_dn:
            check(lib.misplat_loss_bwd(C.c_int64(ctx.n_pix), ptr(rgb if l1_here else None), ptr(gt if l1_here else None),
                                       ptr(g1 if l1_here else None), ptr(g2), C.c_float(ctx.k[0]), C.c_float(ctx.k[1]),
                                       ptr(v_rgb if l1_here else None), ptr(v_e1), ptr(v_e2), stream_ptr()), "misplat_loss_bwd")
        if ctx.packed:
            return v_rgb, None, v_err, None, None, None, None, None
        return v_rgb, None, None, v_e1, v_e2, None, None, None


def mean_losses(rgb, gt, err=None, e1=None, e2=None, depth_ratio: float = 0.0, depth_normal_lambda: float = 0.0,
                ssim_lambda: float = 0.0):
    """(rgb_loss or None, depth_normal_loss or None): mean |gt - rgb| and lambda * ((1 - r) * mean(e1) + r * mean(e2)).
    Contiguous float32 GPU tensors; ``err`` [2,...] packs e1 / e2 (its gradient then arrives as one tensor).
    ``ssim_lambda`` > 0: the first value is the base model's ``main_loss`` = (1 - l) mean |gt - rgb| + l (1 - SSIM(gt, rgb))
    of one [H,W,3] image (misplat_ssim_fwd / misplat_ssim_bwd)."""
    def ok(t):
        return t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous())
    if not all(ok(t) for t in (rgb, gt, err, e1, e2)):
        raise ValueError("mean_losses: contiguous float32 GPU tensors only")
    if rgb is not None and gt is not None and rgb.shape != gt.shape:
        raise ValueError("mean_losses: image and ground truth differ in shape")
    return _MeanLosses.apply(rgb, gt, err, e1, e2, float(depth_ratio), float(depth_normal_lambda), float(ssim_lambda))
