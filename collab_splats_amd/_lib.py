"""ctypes binding of libmisplat.so (the C ABI declared in include/misplat.h).

There is NO fallback: if the HIP library is missing or a call fails this module raises.  The
CPU restatements under ``oracle/`` are test infrastructure and are never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmisplat.so")

MISPLAT_TILE = 16
MISPLAT_REC = 16

_ERR = {-1: "MISPLAT_EINVAL (bad size / unsupported option)",
        -2: "MISPLAT_ELAUNCH (kernel launch failed)",
        -3: "MISPLAT_EWORKSPACE (scratch buffer too small)"}


class MisplatError(RuntimeError):
    pass


class Params(C.Structure):
    """Mirror of ``misplat_params`` (include/misplat.h)."""
    _fields_ = [("n_gauss", C.c_int32), ("n_cams", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("tile_size", C.c_int32), ("tile_w", C.c_int32),
                ("tile_h", C.c_int32), ("antialiased", C.c_int32),
                ("opacity_aware_radius", C.c_int32), ("eps2d", C.c_float), ("near_plane", C.c_float),
                ("far_plane", C.c_float), ("radius_clip", C.c_float), ("radius_sigma", C.c_float),
                ("alpha_max", C.c_float), ("alpha_min", C.c_float), ("t_stop", C.c_float),
                ("median_t", C.c_float), ("jacobian_margin", C.c_float), ("plane_eps", C.c_float),
                ("ed_slot", C.c_int32), ("reserved_q", C.c_int32),
                ("unit_perm", C.c_void_p), ("unit_work", C.c_void_p), ("touched", C.c_void_p),
                ("activations", C.c_int32), ("unit_stride", C.c_int32), ("unit_sel", C.c_void_p),
                ("unit_slots", C.c_int32), ("front_pass", C.c_int32), ("front_n", C.c_void_p), ("tile_flag", C.c_void_p),
                ("unit_reach", C.c_void_p), ("front_depths", C.c_void_p)]


class RasterArgs(C.Structure):
    """Mirror of ``misplat_raster_args`` (include/misplat.h)."""
    _P = C.c_void_p
    _fields_ = ([(n, C.c_void_p) for n in ("means", "quats", "scales", "opacities", "colors", "colors_rest", "viewmats", "Ks")]
                + [(n, C.c_int32) for n in ("sh_degree", "K_or_D", "n_color", "per_cam", "depth_channel", "color_dim",
                                            "reserved_c", "lazy_colour")]
                + [(n, C.c_void_p) for n in ("radii", "means2d", "depths", "compensations", "grec", "sh_aux", "v_grec_zero",
                                             "tiles_per_gauss", "rect2", "cellhist", "cell_count", "cell_offs", "cell_cursor", "order",
                                             "rect_sorted", "counters", "tile_count", "offsets", "payload", "flatten_ids", "scratch")]
                + [("cap_isects", C.c_int64)]
                + [(n, C.c_void_p) for n in ("n_isects_host", "v_abs_zero", "render", "alpha", "exp_depth", "med_depth",
                                             "normal", "last_ids", "median_ids", "unit_perm_in", "unit_work",
                                             "unit_perm_out", "ev_blend_begin", "ev_blend_end", "order_table", "order_sel")]
                + [("order_slots", C.c_int32), ("order_stride", C.c_int32)]
                + [(n, C.c_void_p) for n in ("unit_reach", "front_n", "tile_flag")]
                + [("front_margin", C.c_float), ("front_min_bucket", C.c_int32), ("depth_sorted", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("features", "featx", "v_featx_zero")] + [("n_feat", C.c_int32), ("nxq", C.c_int32)]
                + [("est_isects", C.c_int64)])


class RasterBwdArgs(C.Structure):
    """Mirror of ``misplat_raster_bwd_args`` (include/misplat.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("Ks", "grec", "flatten_ids", "offsets")] + [("n_isects", C.c_int64)]
                + [(n, C.c_void_p) for n in ("alpha", "last_ids", "median_ids", "render", "v_render", "v_alpha", "v_exp_depth",
                                             "v_med_depth", "v_normal", "v_grec", "v_abs", "unit_perm")]
                + [(n, C.c_int32) for n in ("color_dim", "zero_flags", "sh_degree", "K_or_D", "n_color", "per_cam",
                                            "depth_slot", "reserved")]
                + [(n, C.c_void_p) for n in ("means", "quats", "scales", "opacities", "colors", "colors_rest", "viewmats",
                                             "radii", "compensations", "sh_aux", "v_means2d", "v_colors", "v_colors_rest",
                                             "v_means_dir", "v_means", "v_quats", "v_scales", "v_opacities", "ev_blend_begin", "ev_blend_end",
                                             "v_means2d_out", "unit_sel")]
                + [("unit_stride", C.c_int32), ("unit_slots", C.c_int32)]
                + [(n, C.c_void_p) for n in ("featx", "features", "v_featx", "v_features")]
                + [(n, C.c_int32) for n in ("n_feat", "nxq", "depth_channel", "reserved_x")] + [("depths", C.c_void_p)])


def make_params(n_gauss: int, n_cams: int, width: int, height: int, tile_size: int = 16,
                antialiased: bool = False, opacity_aware_radius: bool = True, eps2d: float = 0.3,
                near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
                radius_sigma: float = 3.33, alpha_max: float = 0.999, alpha_min: float = 1.0 / 255.0,
                t_stop: float = 1e-4, median_t: float = 0.5, jacobian_margin: float = 0.3,
                plane_eps: float = 1e-6, ed_slot: int = -1) -> Params:
    if tile_size != MISPLAT_TILE:
        raise ValueError(f"tile_size must be {MISPLAT_TILE} (got {tile_size})")
    tw = (width + tile_size - 1) // tile_size
    th = (height + tile_size - 1) // tile_size
    return Params(n_gauss, n_cams, width, height, tile_size, tw, th, int(antialiased),
                  int(opacity_aware_radius), eps2d, near_plane, far_plane, radius_clip, radius_sigma,
                  alpha_max, alpha_min, t_stop, median_t, jacobian_margin, plane_eps,
                  int(ed_slot), 0, None, None, None)


# name -> (restype, n_args); every symbol include/misplat.h declares
SYMBOLS = {
    "misplat_project_fwd": (C.c_int, 16), "misplat_project_bwd": (C.c_int, 18),
    "misplat_project_pack_fwd": (C.c_int, 17), "misplat_color_fwd": (C.c_int, 16),
    "misplat_color_bwd": (C.c_int, 16), "misplat_project_pack_bwd": (C.c_int, 20),
    "misplat_sh_fwd": (C.c_int, 9), "misplat_sh_bwd": (C.c_int, 11),
    "misplat_isect_ids": (C.c_int, 6), "misplat_tile_sort": (C.c_int, 10),
    "misplat_adam_step": (C.c_int, 12), "misplat_pack": (C.c_int, 11),
    "misplat_blend_fwd": (C.c_int, 15), "misplat_blend_fwd_lazy": (C.c_int, 23), "misplat_blend_bwd": (C.c_int, 21),
    "misplat_color_fwd_x": (C.c_int, 11), "misplat_color_bwd_x": (C.c_int, 9),
    "misplat_blend_fwd_x": (C.c_int, 17), "misplat_blend_bwd_x_atomic": (C.c_int, 22),
    "misplat_blend_planes": (C.c_int, 1), "misplat_blend_bwd_atomic": (C.c_int, 20), "misplat_slab_reduce": (C.c_int, 11), "misplat_depth_normal_fwd": (C.c_int, 10),
    "misplat_depth_normal_bwd": (C.c_int, 14), "misplat_outputs_fwd": (C.c_int, 15), "misplat_outputs_bwd": (C.c_int, 16),
    "misplat_loss_fwd": (C.c_int, 11), "misplat_loss_bwd": (C.c_int, 11),
    "misplat_ssim_scratch_floats": (C.c_int64, 2), "misplat_ssim_fwd": (C.c_int, 10), "misplat_ssim_bwd": (C.c_int, 9),
    "misplat_bucket_plan": (C.c_int, 3), "misplat_bucket_count": (C.c_int, 10), "misplat_bucket_rows": (C.c_int, 14),
    "misplat_bucket_tiles": (C.c_int, 11),
    "misplat_unit_order": (C.c_int, 4), "misplat_raster_fwd": (C.c_int, 5), "misplat_raster_bwd": (C.c_int, 4), "misplat_raster_bwd_plan": (C.c_int, 2), "misplat_graph_cache_create": (C.c_void_p, 1),
    "misplat_graph_cache_destroy": (None, 1), "misplat_graph_cache_stats": (C.c_int, 3), "misplat_wait_count": (C.c_int64, 2), "misplat_zero_bytes": (C.c_int, 3), "misplat_stream_copy": (C.c_int, 5),
    "misplat_touched_bits": (C.c_int, 4), "misplat_union_count": (C.c_int, 5), "misplat_union_scan": (C.c_int, 5), "misplat_union_ids": (C.c_int, 7),
    "misplat_rows_pack": (C.c_int, 8), "misplat_rows_unpack": (C.c_int, 8),
    "misplat_debug_memset_replay": (C.c_int, 5),
    "misplat_version": (C.c_char_p, 0),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libmisplat.so; raises MisplatError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MISPLAT_LIB", LIB_PATH)      # developer knob: an alternative build of the same ABI
    if not os.path.exists(path):
        raise MisplatError(
            f"{path} not found: build it with `python -m collab_splats_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, n_args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        setattr(lib, name, _arity_checked(name, fn, n_args))
    _lib = lib
    return lib


def _arity_checked(name: str, fn, n_args: int):
    """ctypes without argtypes cannot tell a call with a missing argument from a correct one (the stream pointer
    would silently land in the wrong slot): every entry point is called through this check of the argument count
    that tests/test_abi.py keeps equal to include/misplat.h."""
    def call(*args):
        if len(args) != n_args:
            raise MisplatError(f"{name}: {len(args)} arguments passed, the C ABI takes {n_args}")
        return fn(*args)
    call.__name__ = name
    return call


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> C.c_void_p:
    """The current HIP stream of the current device (what every entry point launches on).  Uses torch's raw
    accessors when present: the public ``torch.cuda.current_stream()`` costs ~10 us per call, 11 calls a step."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "misplat: tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def check(code: int, what: str) -> None:
    if code != 0:
        raise MisplatError(f"{what} failed: {_ERR.get(code, code)}")


def require_gpu(*tensors: torch.Tensor) -> None:
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MisplatError("misplat runs on the MI355X only: got a CPU tensor "
                               "(there is no CPU fallback; oracle/ is test infrastructure)")
        if cur is None:
            cur = _raw_device() if _raw_device is not None else torch.cuda.current_device()
        if t.device.index is not None and t.device.index != cur:
            # kernels are launched on the current device's current stream: a tensor of another GPU would be
            # dereferenced on the wrong device (one process per GPU is the supported layout)
            raise MisplatError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: "
                               "call torch.cuda.set_device() (one process per GPU) first")
