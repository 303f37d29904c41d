#!/usr/bin/env python3
"""bench.py -- Msplats/s of the RaDe-GS rasterizer hot path (fwd+bwd) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
          --master-port P bench.py --gpus N --steps K --warmup W; `python bench.py --gpus N` alone starts the
          N ranks itself as child processes)

A "step" is one pass of the hot path over one view of the synthetic scene, exactly as the reference's caller issues it
(rade_gs_model.py:439-465): torch.exp / torch.sigmoid of the raw parameters (:443-444) -> rasterization(...,
return_depth_normal=True) -> backward of all five outputs to the six parameter tensors -- and, as in training
(one camera per step, :94-95), a DIFFERENT camera every step: the eight views of BASELINE configs[3] are cycled.
Workload (BASELINE.json configs[2], the one the metric is quoted on): 1 M random Gaussians, 1920x1080, SH degree 3, RGB+ED,
antialiased.  `value` is that drop-in step.  The same line carries `variants`: the 2 x 2 of {torch activations,
scales_are_log / opacities_are_logit inside the projection kernels (an extension of this build)} x {cycling views,
one fixed view}, each measured with the same protocol right after the headline (`value_ext` = extension + cycling);
--ext-activations / --fixed-view make one of the other corners the headline.  With N GPUs every rank renders its own
views of its own replica (configs[3]: independent views, no collective, weak scaling).  --shared-grads: all ranks hold
the SAME Gaussians, render different views and all-reduce the 236 B/Gaussian gradients over RCCL; with --dn-loss the step
is the model mirror's get_outputs -> get_loss_dict (Splatfacto's main_loss = 0.8 L1 + 0.2 (1 - SSIM) + the depth-normal
consistency loss) -> backward (configs[4] per GPU; use --gaussians 5000000; --no-ssim: the L1 + depth-normal step that
rounds 1-2 reported).
Inputs are resident in HBM before the timed region.

Protocol: W untimed steps, then EXACTLY K steps between two barrier+synchronize fences,
wall clock, max over ranks -> `value`; the same K steps are also bracketed by device events (`device_ms_median`).
Per-kernel times for `roofline` come from a separate short instrumented pass AFTER the timed region (HIP events on
the launch stream around the dominant kernels), so the timed region carries no instrumentation.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel + whole step, against the 8 TB/s specification and
against a float4 copy measured on this box) and `cpu_baseline` (the C port under oracle/, bounded samples, rank 0 at
N=1 only).
"""
from __future__ import annotations

import argparse
import functools
import gc
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24, help="timed steps (default: three passes over the eight views)")
    ap.add_argument("--warmup", type=int, default=16,
                    help="untimed steps (default: two passes over the eight views -- a graph is captured when an argument block "
                         "is seen the second time, a view's launch order exists after its first visit)")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--render-mode", default="RGB+ED")
    ap.add_argument("--rasterize-mode", default="antialiased")
    ap.add_argument("--shared-grads", action="store_true", help="shared Gaussians: all-reduce the gradients over RCCL")
    ap.add_argument("--ext-activations", action="store_true",
                    help="headline = the scales_are_log / opacities_are_logit extension (exp / sigmoid inside the projection "
                         "kernels); default: torch.exp / torch.sigmoid in front of rasterization(), as the reference calls it")
    ap.add_argument("--fixed-view", action="store_true",
                    help="headline = the same camera every step; default: the eight views of configs[3] cycled, one per step")
    ap.add_argument("--no-variants", action="store_true", help="skip the other three corners of the 2 x 2 (`variants`)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two counters-only rocprofv3 passes over a short child run, "
                         "after the timed region; part of the default line like cpu_baseline): quote profiles/pmc_traffic.json")
    ap.add_argument("--buckets", action="store_true",
                    help="attach a parallel.GradientBuckets sink even on one GPU (no collective): measures what the "
                         "data-parallel backward costs on top of the plain one")
    ap.add_argument("--dn-loss", action="store_true",
                    help="step = RadegsModel.get_outputs -> main_loss (L1 + SSIM) + depth-normal consistency loss -> backward "
                         "(configs[4])")
    ap.add_argument("--no-ssim", action="store_true", help="--dn-loss without the SSIM half of main_loss (ssim_lambda = 0)")
    ap.add_argument("--graphed", action="store_true",
                    help="capture the whole step (activations, forward, backward) into ONE hipGraph with a fixed "
                         "intersection capacity (collab_splats_amd.graphs.GraphedStep) and time its replays: the "
                         "host-bound small configurations; not the default protocol")
    ap.add_argument("--features", type=int, default=0, metavar="F",
                    help="the features model's call (rade_features_model.py:427-476): F feature channels behind the SH colours "
                         "(13 in the reference), SH + clamp + fuse + rasterize as ONE entry, fwd+bwd to coefficients AND features; "
                         "`variants` then also times the reference's own composition (spherical_harmonics, clamp_min, cat, "
                         "rasterization(sh_degree=None)) through the same library")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--key-trace", action="store_true",
                    help="print (stderr) the argument-block fields that differ between two visits of a view: why a graph did not replay")
    ap.add_argument("--rehearse-sparse", action="store_true",
                    help="with --buckets on one GPU: run every step of the sparse shared-Gaussian reduce except the collective itself")
    ap.add_argument("--oversubscribe", action="store_true", help="--gpus N on fewer than N devices (a rehearsal of the plumbing)")
    ap.add_argument("--parity-view", type=int, default=3, help="the rotated view the gradient check runs on besides view 0")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="Gaussians in the CPU-baseline sample")
    ap.add_argument("--details", default=None, metavar="FILE",
                    help="also write the line with everything behind it (per-view counts, per-tensor errors, graph / arena / path "
                         "counters, live counters) as indented JSON: the printed line itself stays under 4 KB so that the driver's "
                         "record carries it whole")
    return ap.parse_args()


def algorithmic_bytes(kernel: str, N: int, I: int, P: int, n_feat: int = 0, n_channels: int = 3) -> float:
    """SURVEY.md section 8(d) per-unit figures (fp32, SH3, D=3).  The features model (``n_feat`` > 0 feature channels behind
    the SH colours, ``n_channels`` = D' composited channels): "for D != 3 replace 12 by 4 D in the colour terms" -- per pixel
    4 (D' - 3) more bytes in every image-shaped stream, per intersection the 16-byte groups of the extra channels (16 nxq),
    per Gaussian 4 (D' - 3) more bytes in the 2-D gradient row and 4 n_feat per pass over the feature parameters."""
    x = 4.0 * max(n_channels - 3, 0) if n_feat else 0.0               # extra bytes per colour-shaped item
    gx = 16.0 * ((max(n_channels - 4, 0) + 3) // 4) if n_feat else 0.0  # extra gathered bytes per (tile, Gaussian)
    if kernel == "blend_bwd":
        return (88.0 + 2 * x) * P + (64.0 + gx) * I + (60.0 + x) * N     # saved outputs + upstream grads, re-gather, 2-D grads
    if kernel == "blend_fwd":
        return (64.0 + gx) * I + (48.0 + x) * P                 # record gather, pixel outputs incl. saved indices
    if kernel == "slab_reduce":
        return 64.0 * I + 64.0 * N                 # one gradient row per intersection in, one per Gaussian out
    if kernel == "step":
        # whole fwd+bwd call (SURVEY.md 8(d) total A = 972 N + 164 I + 136 P); features: read in the forward and the backward,
        # gradient written once (3 x 4 n_feat), the colour terms of the three 72 / 60-byte per-Gaussian rows widened by x
        return (972.0 + 12.0 * n_feat + 3 * x) * N + (164.0 + 2 * gx) * I + (136.0 + 3 * x) * P
    raise KeyError(kernel)


def git_rev() -> str:
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    try:                                              # no .git on the GPU box: the revision the library was built from
        return open(os.path.join(ROOT, "collab_splats_amd", "_build_rev.txt")).read().strip()
    except Exception:
        return "unknown"


def _c_port_run(cr, sc, scales, op, W, H, args, ups):
    st = cr.forward(sc["means"].numpy(), sc["quats"].numpy(), scales, op, sc["sh"].numpy(), sc["viewmats"][0].numpy(),
                    sc["Ks"][0].numpy(), W, H, sh_degree=3, render_mode=args.render_mode, rasterize_mode=args.rasterize_mode)
    return st, cr.backward(st, *ups)


def _c_port_time(n, W, H, args, seed, threads, budget_s, max_reps=20):
    """fwd+bwd of the C port on n Gaussians of the same generator; returns (Msplats/s, reps, n_isects, threads, state)."""
    import numpy as np
    from oracle.craster import CRaster
    from collab_splats_amd.synthetic import random_scene
    sc = random_scene(n, W, H, seed=seed)
    cr = CRaster(np.float32, threads=threads)
    scales, op = torch.exp(sc["log_scales"]).numpy(), torch.sigmoid(sc["opacity_logits"]).numpy()
    g = torch.Generator().manual_seed(7)
    cd = 4 if args.render_mode == "RGB+ED" else 3
    ups = [torch.rand(s, generator=g).numpy() for s in ((H, W, cd), (H, W, 1), (H, W, 1), (H, W, 1), (H, W, 3))]
    st, gr = _c_port_run(cr, sc, scales, op, W, H, args, ups)          # warm-up (page-in, thread pool)
    t0, reps = time.perf_counter(), 0
    while True:
        st, gr = _c_port_run(cr, sc, scales, op, W, H, args, ups)
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= max_reps:
            break
    dt = (time.perf_counter() - t0) / reps
    return n / dt / 1e6, reps, st["bins"]["n_isects"], cr.threads, (sc, scales, op, ups, gr)


def cpu_baseline(args, seed: int):
    """C port (oracle/craster.c, OpenMP) on bounded samples of the same generator: the main
    figure on `--cpu-sample` Gaussians of the benchmark image with all host threads; `threads1` = the same code on one
    thread (the reference's own CI pins one thread, scripts/test.sh:5-8); `config1` = BASELINE configs[0]
    (10 k Gaussians, 256x256); `torch_oracle` = the dense fp64 autograd oracle on a tiny scene (plumbing sanity)."""
    W, H = args.width, args.height
    n = min(args.cpu_sample, args.gaussians)
    v, reps, isects, threads, keep = _c_port_time(n, W, H, args, seed, None, 8.0)
    base = {"value": round(v, 4), "unit": "Msplats/s", "cores": threads, "kind": "port",
            "sample": f"{n} Gaussians (same generator, seed {seed}), {W}x{H}, {args.render_mode} fwd+bwd, "
                      f"{reps} timed iteration(s) after 1 warm-up, {isects} intersections, C/OpenMP fp32",
            "sample_short": f"{n} Gaussians {W}x{H} {args.render_mode} fwd+bwd, {reps} iterations, C/OpenMP fp32"}
    n1 = min(10_000, n)
    vc, repsc, isectsc, tc, _ = _c_port_time(n1, 256, 256, args, seed, None, 3.0, max_reps=20)
    base["config1"] = {"value": round(vc, 4), "cores": tc,
                       "sample": f"BASELINE configs[0]: {n1} Gaussians, 256x256, {repsc} iteration(s), {isectsc} intersections"}
    v1, reps1, isects1, _, _ = _c_port_time(n1, 256, 256, args, seed, 1, 4.0, max_reps=10)
    base["threads1"] = {"value": round(v1, 4), "cores": 1,
                        "sample": f"{n1} Gaussians, 256x256, {reps1} iteration(s), {isects1} intersections, OMP_NUM_THREADS=1 equivalent"}
    try:
        from oracle import torch_oracle as O
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import small_scene
        sm = small_scene(n=300, W=48, H=40, seed=0)
        ins = [sm[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")]
        t0 = time.perf_counter()
        out = O.rasterization(*ins, sm["viewmat"][None], sm["K"][None], 48, 40, sh_degree=3, render_mode="RGB+ED",
                              rasterize_mode="antialiased")
        sum(o.sum() for o in out[:5]).backward()
        base["torch_oracle"] = {"value": round(300 / (time.perf_counter() - t0) / 1e6, 6), "cores": torch.get_num_threads(),
                                "sample": "300 Gaussians, 48x40, dense fp64 autograd, one fwd+bwd"}
    except Exception as e:                                    # plumbing sanity only
        base["torch_oracle"] = {"error": str(e)[:100]}
    return base, keep


def rasterize_step(params, viewmats, Ks, W, H, args, ext: bool, composed: bool = False):
    """THE call of a step -- the timed loop and the parity leg both go through here.  ``ext`` False: the reference's call
    (torch.exp / torch.sigmoid in front, rade_gs_model.py:443-444); True: the same step with the activations inside the
    projection kernels (scales_are_log / opacities_are_logit, an extension of this build)."""
    from collab_splats_amd.rendering import rasterization
    kw = dict(near_plane=0.01, far_plane=1e10, sh_degree=3, packed=False, render_mode=args.render_mode, sparse_grad=False,
              absgrad=False, rasterize_mode=args.rasterize_mode, return_depth_normal=True)
    if "features" in params and composed:
        # the reference's own lines, through this library's drop-ins: rade_features_model.py:427-441 + :450-476
        from collab_splats_amd import spherical_harmonics
        cam_c = -(viewmats[0, :3, :3].T @ viewmats[0, :3, 3])
        colors = spherical_harmonics(3, params["means"] - cam_c, params["sh"])
        colors = torch.clamp_min(colors + 0.5, 0.0)
        fused = torch.cat((colors, params["features"]), dim=-1)
        kw["sh_degree"] = None
        return rasterization(params["means"], params["quats"], torch.exp(params["log_scales"]),
                             torch.sigmoid(params["opacity_logits"]), fused, viewmats, Ks, W, H, **kw)
    if "features" in params:
        kw["features"] = params["features"]
    if ext:
        return rasterization(params["means"], params["quats"], params["log_scales"], params["opacity_logits"], params["sh"],
                             viewmats, Ks, W, H, scales_are_log=True, opacities_are_logit=True, **kw)
    return rasterization(params["means"], params["quats"], torch.exp(params["log_scales"]),
                         torch.sigmoid(params["opacity_logits"]), params["sh"], viewmats, Ks, W, H, **kw)


def parity(args, params, viewmats, Ks, W, H, ext, ups_dev, st, gr, act, what: str, cr, deterministic: bool = False):
    """The second half of the metric ("grad max-rel-err vs reference"): the gradients of the step the benchmark times --
    the same `rasterize_step` call on `params` (the benchmark's own leaves when the full workload is checked) -- against
    the C port (the checker; the upstream CUDA reference is absent: parity unpinned).
    Per RAW parameter (the C port's gradients of the activated scales / opacities are chained through exp / sigmoid, `act` =
    the activated values it was given):
      max_rel_err        tensor-inf-norm relative error  max|g - g_ref| / max|g_ref|  (first metric) over ALL rows
      elementwise        max |d| / (|g_ref| + 1e-3 max|g_ref|)  (second metric) over the rows no flip touches
      max_rel_err_clean  the first metric over the rows whose screen box covers no pixel that differs between the two
                         implementations (a threshold flip: alpha = 1/255, T' = 1e-4, T = 0.5 -- tests/helpers.FlipProof)
      rows_over / rows_flip   rows above 1e-4, and how many of those a proven flip explains (the rest would be a failure)
    The five images and the two index maps are compared first: their differing pixels are the flips on record."""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    from helpers import FlipProof                                  # (post-timing leg only: test infrastructure, like oracle/)
    from collab_splats_amd import ops
    for p in params.values():
        p.grad = None
    old_det = ops.DETERMINISTIC_BACKWARD
    ops.set_deterministic(deterministic)
    try:
        out = rasterize_step(params, viewmats, Ks, W, H, args, ext)
        torch.autograd.backward(list(out[:5]), ups_dev)
        torch.cuda.synchronize()
    finally:
        ops.set_deterministic(old_det)
    scales, op = act
    want = {"means": gr["v_means"], "quats": gr["v_quats"], "log_scales": gr["v_scales"] * scales,
            "opacity_logits": gr["v_opacities"] * op * (1.0 - op), "sh": gr["v_colors"]}
    proof = FlipProof(cr.blend_margin(st), st["proj"]["means2d"], st["proj"]["radii"])
    fw, meta = st["fwd"], out[5]
    flips = np.zeros(proof.low.shape, bool)
    unexplained_px = 0
    images = {}
    for name, got, ref in (("render", out[0], st["render"]), ("alpha", out[1], fw["alpha"]), ("exp_depth", out[2], fw["exp_depth"]),
                           ("med_depth", out[3], fw["med_depth"]), ("normal", out[4], fw["normal"])):
        a, b = got[0].detach().cpu().numpy().astype(np.float64), np.asarray(ref, np.float64)
        d = np.abs(a - b) / max(np.abs(b).max(), 1e-30)
        bad = (d > 1e-4).reshape(d.shape[0], d.shape[1], -1).any(-1)
        flips |= bad
        unexplained_px += int((bad & ~proof.low).sum())
        clean = d.reshape(d.shape[0], d.shape[1], -1).max(-1)[~bad]
        images[name] = {"max_rel_err": float(f"{d.max():.3e}"), "pixels_over": int(bad.sum()),
                        "max_rel_err_clean": float(f"{(clean.max() if clean.size else 0.0):.3e}")}
    for key, got, ref in (("last_ids", meta["last_ids"], fw["last_ids"]), ("median_ids", meta["median_ids"], fw["median_ids"])):
        diff = got[0].cpu().numpy() != np.asarray(ref)
        flips |= diff
        unexplained_px += int((diff & ~proof.low).sum())
        images[key] = {"pixels_differ": int(diff.sum())}
    proof.flipped |= flips
    proof.pixel_checks += 1
    n_rows = want["means"].shape[0]
    touched = proof._box_counts(proof._sat(proof.flipped), np.arange(n_rows)) > 0          # rows whose box covers a flip
    per, worst, worst_clean, unexplained_rows = {}, 0.0, 0.0, 0
    for k, ref in want.items():
        g = params[k].grad.cpu().numpy().astype(np.float64).reshape(n_rows, -1)
        r = np.asarray(ref, np.float64).reshape(n_rows, -1)
        scale = max(np.abs(r).max(), 1e-30)
        d = np.abs(g - r)
        row = d.max(1) / scale
        over = row > 1e-4
        clean = row[~touched]
        elem = (d / (np.abs(r) + 1e-3 * scale))[~touched]
        per[k] = {"max_rel_err": float(f"{row.max():.3e}"), "max_rel_err_clean": float(f"{(clean.max() if clean.size else 0.0):.3e}"),
                  "elementwise": float(f"{(elem.max() if elem.size else 0.0):.3e}"), "rows_over": int(over.sum()),
                  "rows_flip": int((over & touched).sum())}
        worst = max(worst, row.max())
        worst_clean = max(worst_clean, clean.max() if clean.size else 0.0)
        unexplained_rows += int((over & ~touched).sum())
    return {"value": float(f"{worst:.3e}"), "value_clean": float(f"{worst_clean:.3e}"), "rows_over_unexplained": unexplained_rows,
            "per_tensor": per, "images": images, "flip_pixels": int(flips.sum()),
            "pixels_unexplained": unexplained_px, "rows_covering_a_flip": int(touched.sum()),
            "gradient_mode": "deterministic (slab + fixed-order reduce)" if deterministic else "atomic (the timed mode)",
            "on": what, "call": "the timed step's own call (" + ("extension" if ext else "torch") + " activations), raw-parameter gradients",
            "against": "oracle/craster.c (fp32 C port; upstream gsplat-rade absent: parity unpinned)", "target": 1e-4}


def live_pmc(args):
    """HBM bytes per launch of the two compositing kernels and their vector-ALU share, measured in THIS run: counters-only
    rocprofv3 passes (FETCH_SIZE + GRBM_GUI_ACTIVE | WRITE_SIZE | SQ_INSTS_VALU + SQ_ACTIVE_INST_VALU, separate passes, no trace
    domain, the program directly behind `--`) over a short child run of this file with the same workload flags, after the timed
    region.  traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024 (FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md's HBM section
    prescribes).  None when rocprofv3 is not there, when this process is itself being profiled, or when a pass fails."""
    import collections, csv, glob, shutil, signal, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None
    root = tempfile.mkdtemp(prefix="misplat_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "4", "--no-cpu-baseline", "--no-variants",
             "--no-live-pmc", "--gaussians", str(args.gaussians), "--width", str(args.width), "--height", str(args.height),
             "--render-mode", args.render_mode, "--rasterize-mode", args.rasterize_mode]
    child += ["--fixed-view"] if args.fixed_view else []
    child += ["--ext-activations"] if args.ext_activations else []
    env = dict(os.environ, TMPDIR="/tmp")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for i, counters in enumerate((["FETCH_SIZE", "GRBM_GUI_ACTIVE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"])):
            out = os.path.join(root, f"p{i}")
            proc = subprocess.Popen([exe, "--pmc", *counters, "--output-format", "csv", "-d", out, "--", *child], cwd="/tmp", env=env,
                                    stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=240)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)                  # (the group this call started, nothing else)
                proc.wait()
                return None
            if rc != 0:
                return None
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    for key in ("blend_fwd", "blend_bwd"):
                        if key + "_kernel" in r["Kernel_Name"]:
                            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    finally:
        shutil.rmtree(root, ignore_errors=True)
    res = {}
    for k, d in agg.items():
        m = {c: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for c, v in d.items()}     # (the steady half of the launches)
        if not {"FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"} <= set(m):
            return None
        simd_cycles = 1024 * m["GRBM_GUI_ACTIVE"] / 8.0
        res[k] = {"traffic": int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024),
                  "valu_issue_frac": round(m["SQ_INSTS_VALU"] * 4.0 / simd_cycles, 4),
                  "valu_busy": round(m["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, 4),
                  "counters": {c: float(f"{v:.6g}") for c, v in sorted(m.items())}}
    return res or None


def copy_roof(dev):
    """GB/s of a float4 streaming copy (read + write bytes) on this box, 512 MiB buffers: best of 5 for each kernel variant
    (plain / non-temporal / four loads in flight / the same with one contiguous piece per workgroup); returns (best, per-variant list)."""
    import ctypes as C
    from collab_splats_amd import _lib
    lib = _lib.load()
    n4 = (512 << 20) // 16
    src = torch.empty(n4 * 4, device=dev, dtype=torch.float32).normal_()
    dst = torch.empty_like(src)
    per = []
    for variant in range(4):
        best = 0.0
        for i in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.misplat_stream_copy(_lib.ptr(src), _lib.ptr(dst), C.c_int64(n4), C.c_int32(variant), _lib.stream_ptr()),
                       "misplat_stream_copy")
            e1.record()
            e1.synchronize()
            if i >= 2:
                best = max(best, 2 * n4 * 16 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        per.append(round(best, 1))
    return max(per), per


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` without a torchrun environment: start the N ranks as CHILD processes (nothing in
    this process has touched the GPU yet) and relay their output; rank 0 of the children prints the JSON line."""
    n_dev = torch.cuda.device_count()               # does not initialise the GPU
    if n_dev < args.gpus and not args.oversubscribe:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible "
              f"(--oversubscribe rehearses N ranks on fewer devices)", file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


GUIDE_COPY_GBS = 6290.0     # MI355X_MICROARCH.md: measured float4 copy


def unit_tensors(meta) -> dict:
    """What the post-timing pass needs of a step's ``meta`` to count the units the step really visited -- plain tensors only
    (a reference to ``meta`` itself would keep the step's autograd graph alive, which a whole-step capture cannot tolerate)."""
    bins = meta["_bins"] if meta._has("_bins") else {}
    part = bins.get("partial")
    return {"last_ids": meta["last_ids"], "isect_offsets": meta["isect_offsets"], "touched": bins.get("touched"),
            "grec": bins.get("grec"), "front_n": part["front_n"] if part else None}


def visited_units(meta, W, H, N, I):
    """The units one step actually visited (SURVEY.md section 8(d) prices every intersection and every Gaussian; front-only
    ordering, early termination, on-demand colours and the flagged-row backward legitimately never touch most of them on a
    dense scene):  I_trav -- list entries the compositing backward stages (per band: first entry .. deepest last_id);
    I_sort -- entries the per-tile sort ordered (the heads, with front-only ordering; else all);  N_touched -- rows that
    received a 2-D gradient (``misplat_params.touched``);  N_colour -- rows whose SH colour was evaluated (colour slot set)."""
    u = {"I": int(I), "I_trav": traversed_entries(meta, W, H), "I_sort": int(I), "N_touched": int(N), "N_colour": int(N)}
    if meta.get("touched") is not None:
        u["N_touched"] = int((meta["touched"] != 0).sum().item())
    if meta.get("grec") is not None:
        g = meta["grec"]
        u["N_colour"] = int((g[:, 12].view(torch.int32) != 0x7fc0dead).sum().item())      # (kColourUnset, csrc/blend.hip)
    if meta.get("front_n") is not None:
        fn = meta["front_n"].long()
        offs = meta["isect_offsets"].reshape(-1).long()
        cnt = torch.diff(torch.cat([offs, offs.new_tensor([I])]))
        u["I_sort"] = int(torch.where(fn >= 0, torch.minimum(fn, cnt), cnt).sum().item())
    if u["I_trav"] is None:
        u["I_trav"] = int(I)
    return u


def visited_bytes(kernel: str, N: int, P: int, u: dict) -> float:
    """SURVEY 8(d)'s per-unit figures times the units VISITED (see visited_units).  blend_bwd: 88 P + 64 I_trav + 60 N_touched.
    step: projection reads 44 B and writes 60 B for every row; 192 + 12 B of SH coefficients / colour per row whose colour is
    evaluated; 12 I binned; 24 B per sorted entry; 64 B per staged entry in the forward and again in the backward; 136 P of
    images; 60 + 368 B per row that carries a gradient (2-D gradient written, then parameters + saved geometry / colour + the
    2-D gradient read); 236 N of dense parameter gradients written."""
    if kernel == "blend_bwd":
        return 88.0 * P + 64.0 * u["I_trav"] + 60.0 * u["N_touched"]
    if kernel == "blend_fwd":
        return 64.0 * u["I_trav"] + 48.0 * P
    if kernel == "step":
        return (340.0 * N + 204.0 * u["N_colour"] + 428.0 * u["N_touched"] + 12.0 * u["I"] + 24.0 * u["I_sort"]
                + 128.0 * u["I_trav"] + 136.0 * P)
    raise KeyError(kernel)


def traversed_entries(meta, W, H):
    """List entries the compositing BACKWARD stages: per band (16 x 8 pixels) everything from the tile's first entry to the
    band's deepest `last_ids` -- what `64 * I` of SURVEY 8(d) becomes once the early termination is counted."""
    if W % 16 or H % 8:
        return None
    last = meta["last_ids"][0]
    offs = meta["isect_offsets"][0]
    bands = last.view(H // 8, 8, W // 16, 16).amax(dim=(1, 3)).long()                     # [H/8, W/16]
    rows = torch.arange(H // 8, device=last.device) // 2
    beg = offs[rows].long()                                                               # tile row of every band row
    return int((bands - beg + 1).clamp(min=0).sum().item())


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    from collab_splats_amd import parallel
    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local = local % torch.cuda.device_count()      # (a 1-GPU rehearsal of N > 1 puts every rank on cuda:0)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from collab_splats_amd import ops, radegs
    if args.key_trace:
        ops.KEY_TRACE = []
    if args.rehearse_sparse:
        parallel.REHEARSE = True
    from collab_splats_amd.synthetic import random_scene, view_matrix

    N, W, H = args.gaussians, args.width, args.height
    shared = args.shared_grads or args.dn_loss
    seed = 42 if shared else 42 + rank              # shared Gaussians vs independent scenes
    sc = random_scene(N, W, H, seed=seed)
    # the eight views of configs[3], resident on the device (the "dataset"); view 0 is the identity view of the generator
    views = [view_matrix(v).to(dev) for v in range(8)]
    Ks = sc["Ks"].to(dev)
    cd = {"RGB": 3, "RGB+ED": 4, "RGB+D": 4}[args.render_mode]
    g = torch.Generator().manual_seed(7)
    info = {"allreduce_ms": [], "it": 0, "isects": []}
    bucket = None
    mode = {"ext": bool(args.ext_activations), "fixed": bool(args.fixed_view)}

    def view_index():
        return rank % 8 if mode["fixed"] else (rank + info["it"]) % 8

    if args.dn_loss:
        # configs[4] per GPU: the model mirror (a1 + a2 + a3 + a4 + a5) with the depth-normal consistency loss active
        # (its activations are inside the kernels; main_loss = 0.8 L1 + 0.2 (1 - SSIM) as Splatfacto has it)
        cfg = radegs.RadegsModelConfig(rasterize_mode=args.rasterize_mode, regularization_from_iter=0,
                                       ssim_lambda=0.0 if args.no_ssim else 0.2)
        model = radegs.RadegsModel(cfg, sc["means"], sc["log_scales"], sc["quats"], sc["opacity_logits"], sc["sh"][:, 0],
                                   sc["sh"][:, 1:]).to(dev)
        model.train()
        model.step = 20000
        cams = []
        for v in range(8):
            c2w = torch.linalg.inv(view_matrix(v)[0])[:3, :4].clone()
            c2w[:, 1:3] *= -1.0                     # OpenCV world->camera back to the OpenGL camera-to-world the model takes
            cams.append(radegs.PinholeCamera.make(c2w, float(Ks[0, 0, 0]), float(Ks[0, 1, 1]), W, H))
        target = torch.rand(H, W, 3, generator=g).to(dev)
        leaves = [model.gauss_params[k] for k in parallel.GRAD_KEYS]
        if (shared and world > 1) or args.buckets:
            bucket = parallel.GradientBuckets(leaves)

        def step():
            if bucket is None:
                for p in leaves:
                    p.grad = None
            else:
                bucket.attach()
            out = model.get_outputs(cams[view_index()])
            loss = model.get_loss_dict(out, {"image": target})
            functools.reduce(torch.add, loss.values()).backward()      # (nerfstudio's trainer sums the dictionary this way)
            if bucket is not None:
                info["allreduce_ms"].append(bucket.allreduce())
            info["n_isects"], info["n_visible"] = model.info["n_isects"], model.info["radii"]
            info["meta"] = unit_tensors(model.info)
            info["isects"].append(int(model.info["n_isects"]))
            info["it"] += 1
    else:
        params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
        if args.features > 0:
            params["features"] = torch.rand(N, args.features, generator=g).to(dev).requires_grad_(True)
            cd += args.features
        ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, cd), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
        leaves = list(params.values())
        if (shared and world > 1) or args.buckets:
            bucket = parallel.GradientBuckets(leaves, geometry=[0, 1, 2, 3], colour=[4])

        def step():
            if bucket is None:
                for p in leaves:
                    p.grad = None
            else:
                bucket.attach()
            out = rasterize_step(params, views[view_index()], Ks, W, H, args, mode["ext"], composed=mode.get("composed", False))
            torch.autograd.backward(list(out[:5]), ups)
            if bucket is not None:
                info["allreduce_ms"].append(bucket.allreduce())
            # (only plain tensors are kept between steps: a reference to `out` / its meta would keep this step's autograd
            # graph alive, which a whole-step capture -- --graphed -- cannot tolerate)
            m = out[5]
            info["n_isects"], info["n_visible"] = m["n_isects"], m["radii"]
            info["meta"] = unit_tensors(m)
            if not torch.is_tensor(m["n_isects"]):
                info["isects"].append(int(m["n_isects"]))
            info["it"] += 1

    def fence():
        if dist.is_initialized():                  # (world > 1, or a group of one under MISPLAT_FORCE_COLLECTIVES)
            dist.barrier()
        torch.cuda.synchronize()

    def timed(run, k, w):
        """w untimed + exactly k timed steps between two fences: (wall seconds max over ranks, device-event median ms)."""
        for _ in range(w):
            run()
        info["allreduce_ms"].clear()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
        gc.collect()                               # (a full collection that is due lands here, not as a 30 - 50 ms pause in one of K short steps)
        # what the host did between the fences, for the details file: the enqueue time of every step and the interpreter's
        # collections (generation, seconds) -- a multi-millisecond pause in one of K short steps shows up here, not in the kernels
        stamps, pauses, began = [], [], [0.0]

        def on_gc(phase, gc_info):
            if phase == "start":
                began[0] = time.perf_counter()
            else:
                pauses.append((gc_info["generation"], time.perf_counter() - began[0]))

        gc.callbacks.append(on_gc)
        fence()
        info["graph0"] = ops.graph_cache_stats()                 # (host-side counters: no device work)
        t0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record()
            run()
            e1.record()
            stamps.append(time.perf_counter())
        fence()
        dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
        gc.callbacks.remove(on_gc)
        info["graph1"] = ops.graph_cache_stats()
        dev_ms = sorted(a.elapsed_time(b) for a, b in evs)
        host = sorted((b - a) * 1e3 for a, b in zip([t0] + stamps[:-1], stamps))
        info["host_trace"] = {"enqueue_ms_median": round(host[len(host) // 2], 4), "enqueue_ms_max": round(host[-1], 4),
                              "device_ms_max": round(dev_ms[-1], 4),
                              "gc": {str(g): [sum(1 for x in pauses if x[0] == g), round(sum(x[1] for x in pauses if x[0] == g) * 1e3, 3)]
                                     for g in sorted({x[0] for x in pauses})}}
        return dt, dev_ms[len(dev_ms) // 2]

    roof, roof_variants = copy_roof(dev) if rank == 0 else (0.0, [])
    eager_step, graphed = step, None
    if args.graphed:
        if args.dn_loss or shared or args.buckets:
            raise SystemExit("bench.py: --graphed covers the plain rasterization step only")
        from collab_splats_amd import graphs
        if mode["fixed"]:                          # a captured graph replays ONE argument block: one camera
            step()                                 # one eager step: learn the intersection count
            torch.cuda.synchronize()
            capacity = int(int(info["n_isects"]) * 1.5) + 4096
            graphed = graphs.GraphedStep(step, capacity=capacity)
            graphed_all = [graphed]
            step = graphed.replay
        else:
            # the headline's eight cycling views: ONE whole-step graph per resident camera, replayed in turn
            counts = []
            for v in range(8):
                info["it"] = v
                step()
                torch.cuda.synchronize()
                counts.append(int(info["n_isects"]))
            capacity = int(max(counts) * 1.5) + 4096
            one_step = step

            def step_of(v):
                info["it"] = v                     # (view_index() = this view, at capture time -- replays do not run Python)
                one_step()

            graphed_views = graphs.GraphedViews(step_of, 8, capacity)
            graphed_all = graphed_views.steps
            graphed = graphed_all[0]
            turn = {"i": 0}

            def step():
                turn["i"] += 1
                return graphed_views.replay((turn["i"] - 1) % 8)
    stats0 = dict(ops.PATH_STATS)
    g_start = ops.graph_cache_stats()
    dt, dev_med = timed(step, args.steps, args.warmup)
    host_trace_head = info.pop("host_trace", None)
    took = {k: ops.PATH_STATS[k] - stats0.get(k, 0) for k in ops.PATH_STATS}
    if ops.KEY_TRACE and rank == 0:                # MISPLAT_KEY_TRACE=1: which argument fields moved between two visits of a view
        for line in ops.key_trace_report(2 * (1 if mode["fixed"] else 8)):
            print("key-trace", line, file=sys.stderr)
    # graph replays inside the TIMED region of the headline (after the warm-up): a step goes through the cache twice (one
    # merged forward call, one backward call)
    g0, g1 = info["graph0"], info["graph1"]
    graph_timed = {"hits": g1["hits"] - g0["hits"], "captures": g1["captures"] - g0["captures"], "calls": 2 * args.steps}
    graph_timed["hit_rate"] = round(graph_timed["hits"] / max(graph_timed["calls"], 1), 4)
    # The cache captures an argument block when it sees it the SECOND time (a caller whose blocks never recur must not pay a
    # capture per call: csrc/raster.hip run_cached) -- a view's first visit runs plainly, its second captures, every later
    # one can replay.  graph_hit_rate = replays / calls that could replay, over ALL warm-up + timed steps of the headline
    # (a short warm-up only moves the plain and capturing visits into the timed region; `graph_cache_timed` is the raw
    # count of the timed region alone).  What is missing from 1.0 are address misses: the same view presenting a different
    # block (an allocator-issued pointer that moved).
    n_views_head = 1 if mode["fixed"] else 8
    replayable = 2 * sum(max((args.steps + args.warmup - v + n_views_head - 1) // n_views_head - 2, 0) for v in range(n_views_head))
    graph_all = {"hits": g1["hits"] - g_start["hits"], "captures": g1["captures"] - g_start["captures"],
                 "calls": 2 * (args.steps + args.warmup), "replayable_calls": replayable}
    graph_all["hit_rate_of_replayable"] = round(graph_all["hits"] / replayable, 4) if replayable else None
    headline_mode = dict(mode)
    if graphed is not None:
        for gph in graphed_all:
            gph.check()                            # the fixed capacity held for every replay
        step = eager_step
        step()                                     # (eager again: info[...] must not point into the graph's pool)
    allreduce_ms = sorted(ev[0].elapsed_time(ev[1]) for ev in info["allreduce_ms"] if ev is not None)

    # ---- the other corners of {torch, extension activations} x {cycling, fixed view}: same protocol, after the headline
    variants = {}

    def label(m):
        return ("ext" if m["ext"] else "torch") + "_activations+" + ("fixed_view" if m["fixed"] else "cycling_views")

    ms_head = dt / args.steps * 1e3
    variants[label(headline_mode)] = {"ms_per_step": round(ms_head, 4), "device_ms_median": round(dev_med, 4),
                                      "value": round(world * N / (dt / args.steps) / 1e6, 3)}
    if not args.no_variants and not args.dn_loss and graphed is None:
        for ext in (False, True):
            for fixed in (False, True):
                m = {"ext": ext, "fixed": fixed}
                if m == headline_mode:
                    continue
                mode.update(m)
                dtv, medv = timed(step, args.steps, args.warmup)
                variants[label(m)] = {"ms_per_step": round(dtv / args.steps * 1e3, 4), "device_ms_median": round(medv, 4),
                                      "value": round(world * N / (dtv / args.steps) / 1e6, 3)}
        mode.update(headline_mode)
    if args.features > 0 and not args.no_variants and graphed is None:
        # the reference's own composition (spherical_harmonics -> clamp_min -> cat -> rasterization(sh_degree=None)), same views
        mode.update(headline_mode, composed=True)
        dtv, medv = timed(step, args.steps, args.warmup)
        variants["composed_as_the_reference_writes_it+" + ("fixed_view" if headline_mode["fixed"] else "cycling_views")] = {
            "ms_per_step": round(dtv / args.steps * 1e3, 4), "device_ms_median": round(medv, 4),
            "value": round(world * N / (dtv / args.steps) / 1e6, 3)}
        mode.update(headline_mode, composed=False)

    # ---- instrumented pass (outside the timed region): HIP events on the launch stream around the compositing kernels
    # (the same path as the timed region: the one-entry C calls record the events themselves, directly before and
    # after the kernel, and launch plainly instead of replaying a graph while they carry events)
    ops.KERNEL_EVENTS = {}
    info["isects"].clear()
    n_inst = 10
    units = []
    stats_i = dict(ops.PATH_STATS)
    for i in range(n_inst):
        step()
        if rank == 0 and i >= 2:
            units.append(visited_units(info["meta"], W, H, N, int(info["n_isects"])))
    torch.cuda.synchronize()
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    took_i = {k: ops.PATH_STATS[k] - stats_i.get(k, 0) for k in ops.PATH_STATS}

    if rank == 0:
        isects = info["isects"][2:] or [int(info["n_isects"])]
        I = int(sum(isects) / len(isects))         # mean over the instrumented launches (the views differ)
        n_vis = int((info["n_visible"] > 0).any(-1).sum().item())
        P_px = W * H
        ktimes = {}
        for k, v in events.items():                # ms; the first two instrumented steps are warm-up; MEAN duration per launch
            ts = [a.elapsed_time(b) for a, b in v[2:]] or [0.0]
            ktimes[k] = sum(ts) / len(ts)
        dom = max((k for k in ktimes if k in ("blend_bwd", "blend_fwd", "slab_reduce")), key=ktimes.get)
        u = {k: int(sum(x[k] for x in units) / len(units)) for k in units[0]} if units else {"I": I, "I_trav": I, "I_sort": I, "N_touched": N, "N_colour": N}
        # SURVEY 8(d) prices every intersection and every Gaussian ("nominal"); a dense scene's step legitimately never visits
        # most of them (front-only ordering, early termination, on-demand colours, the flagged-row backward): when any of those
        # paths ran, `achieved` / `frac` / `step_frac` are computed from the units the step VISITED and the nominal figures are
        # kept beside them -- a fraction above the box's own copy roof can only come from counting work that was not done
        sparse_paths = bool(took_i.get("forward_lazy_colour") or took_i.get("forward_front_only") or took_i.get("backward_background_fill"))
        plain_rgb = args.features == 0 and dom in ("blend_bwd", "blend_fwd")
        nominal = algorithmic_bytes(dom, N, I, P_px, args.features, cd)
        visited = visited_bytes(dom, N, P_px, u) if plain_rgb else nominal
        abytes = visited if (sparse_paths and plain_rgb) else nominal
        achieved = abytes / (ktimes[dom] * 1e-3) / 1e9
        step_nominal = algorithmic_bytes("step", N, I, P_px, args.features, cd)
        step_visited = visited_bytes("step", N, P_px, u) if args.features == 0 else step_nominal
        step_bytes = step_visited if (sparse_paths and args.features == 0) else step_nominal
        step_gbs = step_bytes / (dev_med * 1e-3) / 1e9
        pmc = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and N == 1_000_000 and (W, H) == (1920, 1080) and not args.dn_loss:
            try:
                pmc = json.load(open(tpath))
            except Exception:
                pmc = {}
        vif = pmc.get("valu_issue_frac")
        view_txt = "1 fixed view" if headline_mode["fixed"] else "8 cycling views"
        if args.dn_loss:
            wl = (f"{N} shared Gaussians, {view_txt} {W}x{H}, model step: get_outputs -> "
                  f"{'L1' if args.no_ssim else 'L1+SSIM'} + depth-normal loss -> backward (configs[4] per GPU)")
        elif args.features > 0:
            wl = (f"features model call: {N} Gaussians, {view_txt} {W}x{H}, SH3 + {args.features} feature channels = {cd} channels "
                  f"({args.render_mode}), fwd+bwd, one entry")
        else:
            wl = (f"{N} random Gaussians, {view_txt} {W}x{H}, SH3, {args.render_mode} {args.rasterize_mode}, 5 outputs, fwd+bwd to 6 "
                  f"parameter tensors, " + ("activations in the kernels" if headline_mode["ext"] else "torch activations (the reference's call)"))
        forced = dist.is_initialized() and world == 1
        if not shared and bucket is None:
            par = "independent views, no collective"
        elif world == 1 and not forced:
            par = "one GPU, no collective" + ("; GradientBuckets sink" if bucket is not None else "")
        else:
            be = dist.get_backend() if dist.is_initialized() else "none"
            par = (f"shared Gaussians, {'RCCL' if be == 'nccl' else be + ' (rehearsal, not RCCL)'} reduce of 236 B/Gaussian"
                   + (" (a group of ONE rank with forced collectives)" if forced else ""))
        def var(label):
            return variants.get(label, {})
        fixed_l = ("ext" if headline_mode["ext"] else "torch") + "_activations+fixed_view"
        line = {
            "metric": "Msplats/s fwd+bwd @1080p (1M Gaussians); grad max-rel-err vs reference",
            "value": round(world * N / (dt / args.steps) / 1e6, 3), "unit": "Msplats/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_head, 4),
            "device_ms_median": round(dev_med, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the literal configs[2] figure (ONE view) beside the headline's eight cycling views, and the activations-in-kernel extension
            "value_fixed_view": var(fixed_l).get("value"), "ms_fixed_view": var(fixed_l).get("ms_per_step"),
            "value_ext": var("ext_activations+cycling_views").get("value"),
            "variants": {k: [v["ms_per_step"], v["value"]] for k, v in variants.items()},
            "config": {"workload": wl, "views": 1 if headline_mode["fixed"] else 8, "n_isects": I, "n_visible": n_vis,
                       "parallelism": par, "git_rev": git_rev(),
                       "graph_hit_rate": graph_timed["hit_rate"] if args.dn_loss else graph_all["hit_rate_of_replayable"],
                       "graph_timed": [graph_timed["hits"], graph_timed["captures"], graph_timed["calls"]],
                       "path": {k: took.get(k, 0) for k in ("forward", "forward_lazy_colour", "forward_view_order", "capacity_redo",
                                                           "backward_one_call", "backward_background_fill", "forward_front_only",
                                                           "forward_probe", "forward_plan_hit", "backward_plan_hit")},
                       "host": "whole-step hipGraphs" if graphed is not None else "eager"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": pmc.get(dom), "traffic_source": "file",
                         "algorithmic_bytes": abytes, "bytes_basis": "visited" if abytes is visited and sparse_paths else "nominal",
                         "achieved_nominal": round(nominal / (ktimes[dom] * 1e-3) / 1e9, 2),
                         "frac_nominal": round(nominal / (ktimes[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "units": u,
                         "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
                         "valu_busy": vif.get(dom) if isinstance(vif, dict) else None,
                         "copy_roof_GBs": round(roof, 1),
                         "step_GBs": round(step_gbs, 1), "step_frac": round(step_gbs / HBM_PEAK_GBS, 5),
                         "step_frac_nominal": round(step_nominal / (dev_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
        }
        details = {"variants": variants, "graph_cache": ops.graph_cache_stats(), "graph_cache_timed": graph_timed,
                   "graph_cache_headline": graph_all, "arena": dict(__import__("collab_splats_amd.arena", fromlist=["STATS"]).STATS),
                   "path": dict(took), "n_isects_per_view": isects[:8], "copy_roof_variants_GBs": roof_variants,
                   "units_per_step": units, "parallel_stats": dict(parallel.STATS), "host_trace": host_trace_head}
        # (compact: the slowest step's enqueue time on the host -- a pause of the interpreter or the OS in one of K short steps)
        line["config"]["host_enqueue_ms"] = [host_trace_head["enqueue_ms_median"], host_trace_head["enqueue_ms_max"]] if host_trace_head else None
        if allreduce_ms:
            line["allreduce_ms"] = round(allreduce_ms[len(allreduce_ms) // 2], 4)
        if world == 1 and not args.no_cpu_baseline and not args.no_live_pmc and not args.dn_loss and args.features == 0 and graphed is None:
            live = live_pmc(args)
            if live and dom in live:
                rf = line["roofline"]
                rf["traffic"], rf["valu_busy"], rf["traffic_source"] = live[dom]["traffic"], live[dom]["valu_busy"], "live"
                details["pmc_live"] = live
        if world == 1 and not args.no_cpu_baseline and not args.dn_loss and args.features == 0:
            import numpy as np
            from oracle.craster import CRaster
            base, keep = cpu_baseline(args, seed)
            details["cpu_baseline"] = base
            line["cpu_baseline"] = {"value": base["value"], "unit": base["unit"], "cores": base["cores"], "kind": base["kind"],
                                    "sample": base["sample_short"], "config1_value": base["config1"]["value"],
                                    "threads1_value": base["threads1"]["value"]}
            cr = CRaster(np.float32)
            full = (os.cpu_count() or 1) >= 32 and args.cpu_sample < N <= 1_000_000
            mode.update(headline_mode)
            if full:
                # enough host cores: the gradients of the FULL workload, on the benchmark's own leaves (one C-port iteration)
                sc_p, params_p, what = sc, params, f"the full workload ({N} Gaussians), view 0"
            else:
                n_s = min(args.cpu_sample, N)
                sc_p = random_scene(n_s, W, H, seed=seed)
                params_p = {k: sc_p[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
                what = f"the cpu_baseline sample ({n_s} Gaussians), view 0"
            # the C port is given the activated values as the DEVICE computes them (the kernels' exp / sigmoid are torch's
            # device expressions): bit-identical inputs on both sides, whichever form of the call is checked
            act = (torch.exp(params_p["log_scales"]).detach().cpu().numpy(),
                   torch.sigmoid(params_p["opacity_logits"]).detach().cpu().numpy())
            gcpu = torch.Generator().manual_seed(7)
            ups_np = [torch.rand(s, generator=gcpu) for s in ((1, H, W, cd), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
            ups_d = [u_.to(dev) for u_ in ups_np]
            # view 0 (the generator's identity view) AND the heaviest of the rotated views the headline cycles through
            per_view = {}
            for v in (0, args.parity_view):
                sc_v = dict(sc_p, viewmats=view_matrix(v))
                st, gr = _c_port_run(cr, sc_v, act[0], act[1], W, H, args, [u_[0].numpy() for u_ in ups_np])
                per_view[f"view{v}"] = parity(args, params_p, views[v], Ks, W, H, headline_mode["ext"], ups_d, st, gr, act,
                                              what.replace("view 0", f"view {v}"), cr)
            timed_views = list(per_view.values())
            details["grad_max_rel_err"] = per_view
            # value: all rows of both views; value_clean: the rows whose screen box covers no pixel where the two implementations
            # took different sides of a threshold (alpha 1/255, T' 1e-4, T 0.5); every row above the target is counted and must be
            # covered by such a pixel -- rows_over_unexplained must be 0
            line["grad_max_rel_err"] = {
                "value": float(f"{max(pv['value'] for pv in timed_views):.3e}"),
                "value_clean": float(f"{max(pv['value_clean'] for pv in timed_views):.3e}"),
                "rows_over_unexplained": sum(pv["rows_over_unexplained"] for pv in timed_views),
                "pixels_unexplained": sum(pv["pixels_unexplained"] for pv in timed_views),
                "target": 1e-4, "on": ("full workload" if full else "cpu sample") + f", views 0 and {args.parity_view}",
                "against": "oracle/craster.c (C port; parity unpinned)",
                "views": {k: {"value": pv["value"], "value_clean": pv["value_clean"], "flip_pixels": pv["flip_pixels"],
                              "rows_over": sum(t["rows_over"] for t in pv["per_tensor"].values())} for k, pv in per_view.items()}}
        print(json.dumps(line), flush=True)
        if args.details:
            os.makedirs(os.path.dirname(os.path.abspath(args.details)), exist_ok=True)
            with open(args.details, "w") as f:
                json.dump(dict(line, details=details), f, indent=1)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
