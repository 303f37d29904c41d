"""ctypes front-end for oracle/craster.c (TEST INFRASTRUCTURE / cpu_baseline ONLY).

Never imported by ``collab_splats_amd``.  PARITY UNPINNED -- see craster.c / torch_oracle.py.
Works on numpy arrays, one camera per call (the reference always renders one camera,
/root/reference/collab_splats/models/rade_gs_model.py:94-95).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force: bool = False) -> None:
    """Compile craster.c (gcc, OpenMP).  Building the checker is not using it."""
    outs = [os.path.join(_BUILD, f"libcraster_{s}.so") for s in ("f32", "f64")]
    src = os.path.join(_HERE, "craster.c")
    if not force and all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs):
        return
    subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)


class _Params(C.Structure):
    pass


def _params_struct(real):
    class P(C.Structure):
        _fields_ = ([("width", C.c_int), ("height", C.c_int), ("tile_size", C.c_int)]
                    + [(n, real) for n in ("fx", "fy", "cx", "cy", "eps2d", "near_plane", "far_plane",
                                           "radius_clip", "radius_sigma", "alpha_max", "alpha_min",
                                           "t_stop", "median_t", "jacobian_margin", "plane_eps")]
                    + [("opacity_aware_radius", C.c_int), ("antialiased", C.c_int)])
    return P


class CRaster:
    """One precision of the C restatement.  ``dtype`` is np.float32 or np.float64."""

    def __init__(self, dtype=np.float32, threads: Optional[int] = None):
        build()
        self.dtype = np.dtype(dtype)
        name = "libcraster_f32.so" if self.dtype == np.float32 else "libcraster_f64.so"
        self.lib = C.CDLL(os.path.join(_BUILD, name))
        self.real = C.c_float if self.dtype == np.float32 else C.c_double
        self.P = _params_struct(self.real)
        assert self.lib.cr_sizeof_real() == self.dtype.itemsize
        self.lib.cr_tile_count.restype = C.c_int64
        if threads is not None:
            self.lib.cr_set_threads(int(threads))
        self.threads = int(self.lib.cr_num_threads())

    # -- helpers
    def _a(self, x, dt=None):
        return np.ascontiguousarray(x, dtype=dt or self.dtype)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def params(self, K, width, height, tile_size=16, eps2d=0.3, near_plane=0.01, far_plane=1e10,
               radius_clip=0.0, radius_sigma=3.33, alpha_max=0.999, alpha_min=1.0 / 255.0,
               t_stop=1e-4, median_t=0.5, jacobian_margin=0.3, plane_eps=1e-6,
               opacity_aware_radius=True, antialiased=False):
        K = np.asarray(K, dtype=np.float64)
        return self.P(int(width), int(height), int(tile_size), K[0, 0], K[1, 1], K[0, 2], K[1, 2],
                      eps2d, near_plane, far_plane, radius_clip, radius_sigma, alpha_max, alpha_min,
                      t_stop, median_t, jacobian_margin, plane_eps, int(opacity_aware_radius),
                      int(antialiased))

    # -- stages
    def project_fwd(self, means, quats, scales, opacities, viewmat, P):
        N = means.shape[0]
        dt = self.dtype
        out = dict(radii=np.zeros((N, 2), np.int32), means2d=np.zeros((N, 2), dt), depths=np.zeros(N, dt),
                   conics=np.zeros((N, 3), dt), compensations=np.zeros(N, dt), ray_ts=np.zeros(N, dt),
                   ray_planes=np.zeros((N, 2), dt), normals=np.zeros((N, 3), dt))
        ops = self._a(opacities) if opacities is not None else None
        self._keep = (self._a(means), self._a(quats), self._a(scales), ops, self._a(viewmat).reshape(16))
        m, q, s, o, V = self._keep
        self.lib.cr_project_fwd(C.c_int(N), self._p(m), self._p(q), self._p(s), self._p(o), self._p(V),
                                C.byref(P), *[self._p(out[k]) for k in
                                              ("radii", "means2d", "depths", "conics", "compensations",
                                               "ray_ts", "ray_planes", "normals")])
        return out

    def project_bwd(self, means, quats, scales, viewmat, P, radii, v_means2d, v_depths, v_conics,
                    v_comps, v_ray_ts, v_ray_planes, v_normals):
        N = means.shape[0]
        dt = self.dtype
        vm, vq, vs = np.zeros((N, 3), dt), np.zeros((N, 4), dt), np.zeros((N, 3), dt)
        args = [self._a(x) for x in (means, quats, scales)] + [self._a(viewmat).reshape(16)]
        grads = [self._a(x) for x in (v_means2d, v_depths, v_conics, v_comps, v_ray_ts, v_ray_planes, v_normals)]
        rad = self._a(radii, np.int32)
        self.lib.cr_project_bwd(C.c_int(N), *[self._p(a) for a in args], C.byref(P), self._p(rad),
                                *[self._p(g) for g in grads], self._p(vm), self._p(vq), self._p(vs))
        return vm, vq, vs

    def sh_fwd(self, degree, dirs, coeffs):
        N, Kc = coeffs.shape[0], coeffs.shape[1]
        out = np.zeros((N, 3), self.dtype)
        d, c = self._a(dirs), self._a(coeffs)
        self.lib.cr_sh_fwd(C.c_int(N), C.c_int(Kc), C.c_int(degree), self._p(d), self._p(c), self._p(out))
        return out

    def sh_bwd(self, degree, dirs, coeffs, v_colors):
        N, Kc = coeffs.shape[0], coeffs.shape[1]
        vc, vd = np.zeros((N, Kc, 3), self.dtype), np.zeros((N, 3), self.dtype)
        d, c, v = self._a(dirs), self._a(coeffs), self._a(v_colors)
        self.lib.cr_sh_bwd(C.c_int(N), C.c_int(Kc), C.c_int(degree), self._p(d), self._p(c), self._p(v),
                           self._p(vc), self._p(vd))
        return vc, vd

    def bin_sort(self, means2d, radii, depths, P, cam_tile_base=0):
        N = means2d.shape[0]
        m2, rad, dep = self._a(means2d), self._a(radii, np.int32), self._a(depths)
        tpg = np.zeros(N, np.int32)
        I = int(self.lib.cr_tile_count(C.c_int(N), self._p(m2), self._p(rad), C.byref(P), self._p(tpg)))
        cum = np.concatenate([[0], np.cumsum(tpg, dtype=np.int64)[:-1]]).astype(np.int64) if N else np.zeros(0, np.int64)
        tw = (P.width + P.tile_size - 1) // P.tile_size
        th = (P.height + P.tile_size - 1) // P.tile_size
        isect = np.zeros(I, np.uint64)
        flat = np.zeros(I, np.int32)
        slot = np.zeros(I, np.int32)
        offs = np.zeros(tw * th, np.int32)
        self.lib.cr_emit_sort(C.c_int(N), self._p(m2), self._p(rad), self._p(dep), self._p(cum),
                              C.c_int64(I), C.byref(P), C.c_int(cam_tile_base), self._p(isect),
                              self._p(flat), self._p(slot), self._p(offs))
        return dict(tiles_per_gauss=tpg, isect_ids=isect, flatten_ids=flat, isect_slot=slot,
                    isect_offsets=offs.reshape(th, tw), n_isects=I, tile_width=tw, tile_height=th)

    def blend_fwd(self, P, means2d, conics, opac, colors, ray_ts, ray_planes, normals, flatten_ids, offsets):
        H, W, D = P.height, P.width, colors.shape[1]
        dt = self.dtype
        out = dict(render=np.zeros((H, W, D), dt), alpha=np.zeros((H, W, 1), dt),
                   exp_depth=np.zeros((H, W, 1), dt), med_depth=np.zeros((H, W, 1), dt),
                   normal=np.zeros((H, W, 3), dt), last_ids=np.zeros((H, W), np.int32),
                   median_ids=np.zeros((H, W), np.int32))
        ins = [self._a(x) for x in (means2d, conics, opac, colors, ray_ts, ray_planes, normals)]
        fl, of = self._a(flatten_ids, np.int32), self._a(offsets, np.int32).reshape(-1)
        self.lib.cr_blend_fwd(C.c_int(D), *[self._p(a) for a in ins], self._p(fl), self._p(of),
                              C.c_int64(fl.shape[0]), C.byref(P),
                              *[self._p(out[k]) for k in ("render", "alpha", "exp_depth", "med_depth",
                                                          "normal", "last_ids", "median_ids")])
        return out

    def blend_margin(self, st) -> np.ndarray:
        """[H, W] relative distance of every pixel to the nearest branch threshold (alpha_min, t_stop,
        median_t) over the Gaussians it traverses -- see cr_blend_margin.  ``st`` = result of ``forward``."""
        P, pr, bs = st["P"], st["proj"], st["bins"]
        m = np.zeros((P.height, P.width), self.dtype)
        ins = [self._a(x) for x in (pr["means2d"], pr["conics"], st["opac"])]
        fl, of = self._a(bs["flatten_ids"], np.int32), self._a(bs["isect_offsets"], np.int32).reshape(-1)
        self.lib.cr_blend_margin(*[self._p(a) for a in ins], self._p(fl), self._p(of), C.c_int64(fl.shape[0]),
                                 C.byref(P), self._p(m))
        return m

    def blend_stats(self, st, shapes=((16, 16), (16, 8), (16, 4), (8, 8), (8, 4), (4, 4))) -> Dict:
        """Work statistics per block shape (design tooling; see cr_blend_stats)."""
        P, pr, bs = st["P"], st["proj"], st["bins"]
        ins = [self._a(x) for x in (pr["means2d"], pr["conics"], st["opac"])]
        fl, of = self._a(bs["flatten_ids"], np.int32), self._a(bs["isect_offsets"], np.int32).reshape(-1)
        li = self._a(st["fwd"]["last_ids"], np.int32)
        sh = np.asarray(shapes, np.int32).reshape(-1)
        out = np.zeros(len(shapes) * 4, np.int64)
        self.lib.cr_blend_stats(*[self._p(a) for a in ins], self._p(fl), self._p(of), C.c_int64(fl.shape[0]),
                                C.byref(P), self._p(li), C.c_int(len(shapes)), self._p(sh), self._p(out))
        out = out.reshape(len(shapes), 4)
        return {tuple(s): dict(traversed=int(o[0]), culled=int(o[1]), hit=int(o[2]), pairs=int(o[3]))
                for s, o in zip(shapes, out)}

    def quad_stats(self, st, sw=8, sh=4, th=32) -> Dict:
        """Loop trips of a four-sub-blocks-per-wave backward (design tooling; see cr_quad_stats)."""
        P, pr, bs = st["P"], st["proj"], st["bins"]
        ins = [self._a(x) for x in (pr["means2d"], pr["conics"], st["opac"])]
        fl, of = self._a(bs["flatten_ids"], np.int32), self._a(bs["isect_offsets"], np.int32).reshape(-1)
        li = self._a(st["fwd"]["last_ids"], np.int32)
        out = np.zeros(6, np.int64)
        self.lib.cr_quad_stats(*[self._p(a) for a in ins], self._p(fl), self._p(of), C.c_int64(fl.shape[0]),
                               C.byref(P), self._p(li), C.c_int(sw), C.c_int(sh), C.c_int(th), self._p(out))
        return dict(band_units=int(out[0]), sub_units=int(out[1]), trips_every_batch=int(out[2]),
                    trips=int(out[3]), rounds=int(out[4]), batches=int(out[5]))

    def half_stats(self, st) -> Dict:
        """Share of the 16x8 band backward's contributing trips that reach one 16x4 half only (design tooling; cr_half_stats)."""
        P, pr, bs = st["P"], st["proj"], st["bins"]
        ins = [self._a(x) for x in (pr["means2d"], pr["conics"], st["opac"])]
        fl, of = self._a(bs["flatten_ids"], np.int32), self._a(bs["isect_offsets"], np.int32).reshape(-1)
        li = self._a(st["fwd"]["last_ids"], np.int32)
        out = np.zeros(6, np.int64)
        self.lib.cr_half_stats(*[self._p(a) for a in ins], self._p(fl), self._p(of), C.c_int64(fl.shape[0]),
                               C.byref(P), self._p(li), self._p(out))
        return dict(staged=int(out[0]), contributing=int(out[1]), one_half_exact=int(out[2]), one_half_provable=int(out[3]),
                    pairs=int(out[4]), one_half_box_all_staged=int(out[5]))

    def subblock_stats(self, st) -> Dict:
        """Sub-block reach of the backward under four decompositions of a tile (design tooling; cr_subblock_stats)."""
        P, pr, bs = st["P"], st["proj"], st["bins"]
        ins = [self._a(x) for x in (pr["means2d"], pr["conics"], st["opac"])]
        fl, of = self._a(bs["flatten_ids"], np.int32), self._a(bs["isect_offsets"], np.int32).reshape(-1)
        li = self._a(st["fwd"]["last_ids"], np.int32)
        out = np.zeros(32, np.int64)
        self.lib.cr_subblock_stats(*[self._p(a) for a in ins], self._p(fl), self._p(of), C.c_int64(fl.shape[0]),
                                   C.byref(P), self._p(li), self._p(out))
        names = ("band16x8_rows", "band16x8_cols", "tile_quadrants8x8", "tile_rows16x4")
        return {n: dict(units=int(out[8 * k]), sub_provable=int(out[8 * k + 1]), sub_exact=int(out[8 * k + 2]),
                        staged=int(out[8 * k + 3])) for k, n in enumerate(names)}

    def blend_bwd(self, P, means2d, conics, opac, colors, ray_ts, ray_planes, normals, flatten_ids,
                  offsets, fwd, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal):
        N, D = means2d.shape[0], colors.shape[1]
        dt = self.dtype
        g = dict(v_means2d=np.zeros((N, 2), dt), v_means2d_abs=np.zeros((N, 2), dt),
                 v_conics=np.zeros((N, 3), dt), v_opac=np.zeros(N, dt), v_colors=np.zeros((N, D), dt),
                 v_ray_ts=np.zeros(N, dt), v_ray_planes=np.zeros((N, 2), dt), v_normals=np.zeros((N, 3), dt))
        ins = [self._a(x) for x in (means2d, conics, opac, colors, ray_ts, ray_planes, normals)]
        fl, of = self._a(flatten_ids, np.int32), self._a(offsets, np.int32).reshape(-1)
        al = self._a(fwd["alpha"])
        li, mi = self._a(fwd["last_ids"], np.int32), self._a(fwd["median_ids"], np.int32)
        ups = [self._a(x) for x in (v_render, v_alpha, v_exp_depth, v_med_depth, v_normal)]
        self.lib.cr_blend_bwd(C.c_int(D), *[self._p(a) for a in ins], self._p(fl), self._p(of),
                              C.c_int64(fl.shape[0]), C.byref(P), self._p(al), self._p(li), self._p(mi),
                              *[self._p(u) for u in ups],
                              *[self._p(g[k]) for k in ("v_means2d", "v_means2d_abs", "v_conics", "v_opac",
                                                        "v_colors", "v_ray_ts", "v_ray_planes", "v_normals")])
        return g

    # -- whole pipeline, one camera -------------------------------------------------
    def forward(self, means, quats, scales, opacities, colors, viewmat, K, width, height,
                sh_degree=None, render_mode="RGB", rasterize_mode="classic", **spec) -> Dict:
        """Returns a dict of forward outputs + everything ``backward`` needs."""
        aa = rasterize_mode == "antialiased"
        P = self.params(K, width, height, antialiased=aa, **spec)
        means, quats, scales = self._a(means), self._a(quats), self._a(scales)
        opacities, colors = self._a(opacities), self._a(colors)
        V = self._a(viewmat)
        pr = self.project_fwd(means, quats, scales, opacities, V, P)
        st = dict(P=P, proj=pr, sh_degree=sh_degree, render_mode=render_mode, aa=aa,
                  inputs=(means, quats, scales, opacities, colors, V))
        if sh_degree is not None:
            cam = -(V[:3, :3].T @ V[:3, 3])
            dirs = means - cam[None]
            raw = self.sh_fwd(sh_degree, dirs, colors)
            cols = np.maximum(raw + 0.5, 0.0).astype(self.dtype)
            st.update(dirs=dirs, sh_raw=raw)
        else:
            cols = colors
        opac = (opacities * pr["compensations"]).astype(self.dtype) if aa else opacities
        if render_mode in ("RGB+D", "RGB+ED"):
            cols = np.concatenate([cols, pr["depths"][:, None]], axis=1)
        elif render_mode in ("D", "ED"):
            cols = pr["depths"][:, None].copy()
        bs = self.bin_sort(pr["means2d"], pr["radii"], pr["depths"], P)
        fw = self.blend_fwd(P, pr["means2d"], pr["conics"], opac, cols, pr["ray_ts"], pr["ray_planes"],
                            pr["normals"], bs["flatten_ids"], bs["isect_offsets"])
        st.update(bins=bs, fwd=fw, cols=cols, opac=opac)
        render = fw["render"]
        if render_mode in ("ED", "RGB+ED"):
            render = render.copy()
            render[..., -1:] = render[..., -1:] / np.maximum(fw["alpha"], 1e-10)
        st["render"] = render
        return st

    def backward(self, st, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal):
        """Gradients wrt (means, quats, scales, opacities, colors) for upstream image grads."""
        P, pr, bs, fw = st["P"], st["proj"], st["bins"], st["fwd"]
        means, quats, scales, opacities, colors, V = st["inputs"]
        dt = self.dtype
        v_render = self._a(v_render).copy()
        v_alpha = self._a(v_alpha).copy()
        if st["render_mode"] in ("ED", "RGB+ED"):
            a = np.maximum(fw["alpha"], 1e-10)
            raw = fw["render"][..., -1:]
            vch = v_render[..., -1:].copy()
            v_render[..., -1:] = vch / a
            v_alpha += np.where(fw["alpha"] > 1e-10, -vch * raw / (a * a), 0.0).astype(dt)
        g = self.blend_bwd(P, pr["means2d"], pr["conics"], st["opac"], st["cols"], pr["ray_ts"],
                           pr["ray_planes"], pr["normals"], bs["flatten_ids"], bs["isect_offsets"], fw,
                           v_render, v_alpha, v_exp_depth, v_med_depth, v_normal)
        v_cols = g["v_colors"]
        v_depths = np.zeros_like(pr["depths"])
        if st["render_mode"] in ("RGB+D", "RGB+ED"):
            v_depths = v_cols[:, -1].copy()
            v_cols = v_cols[:, :-1]
        elif st["render_mode"] in ("D", "ED"):
            v_depths = v_cols[:, 0].copy()
            v_cols = None
        if st["aa"]:
            v_opacities = g["v_opac"] * pr["compensations"]
            v_comps = g["v_opac"] * opacities
        else:
            v_opacities = g["v_opac"]
            v_comps = np.zeros_like(pr["compensations"])
        vm, vq, vs = self.project_bwd(means, quats, scales, V, P, pr["radii"], g["v_means2d"], v_depths,
                                      g["v_conics"], v_comps, g["v_ray_ts"], g["v_ray_planes"], g["v_normals"])
        if st["sh_degree"] is not None and v_cols is not None:
            v_raw = np.where(st["sh_raw"] + 0.5 > 0, v_cols, 0.0).astype(dt)
            v_coeffs, v_dirs = self.sh_bwd(st["sh_degree"], st["dirs"], colors, v_raw)
            vm = vm + v_dirs
            v_colors_in = v_coeffs
        else:
            v_colors_in = v_cols if v_cols is not None else np.zeros_like(colors)
        return dict(v_means=vm, v_quats=vq, v_scales=vs, v_opacities=v_opacities.astype(dt),
                    v_colors=v_colors_in, v_means2d=g["v_means2d"], v_means2d_abs=g["v_means2d_abs"],
                    blend=g)
