"""fwd+bwd of the features-model rasterizer call (rade_features_model.py:441-476): 16 fused channels (RGB + 13 distilled
features) + ED = 17 composited channels, 1 M Gaussians / 1080p, the reference's call form (torch activations).

    python scripts/features_step.py [--steps K] [--both] [--json]

--both also times the 4-channel-pass fallback.  Prints ms/step, the compositing kernels' durations (HIP events around
them) and a roofline entry with SURVEY 8(d)'s byte counts at D' = 17 (4 D' substituted for the 12 / 16 colour bytes)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collab_splats_amd import ops, rendering
from collab_splats_amd.synthetic import random_scene

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--both", action="store_true")
ap.add_argument("--json", action="store_true")
args = ap.parse_args()
N, W, H, D = 1_000_000, 1920, 1080, 16
sc = random_scene(N, W, H, seed=42)
dev = "cuda"
g = torch.Generator().manual_seed(3)
feats = torch.rand(N, D, generator=g).to(dev).requires_grad_(True)
P = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits")}
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, D + 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
info = {}


def step():
    for p in list(P.values()) + [feats]:
        p.grad = None
    out = rendering.rasterization(P["means"], P["quats"], torch.exp(P["log_scales"]), torch.sigmoid(P["opacity_logits"]), feats, V, K,
                                  W, H, sh_degree=None, render_mode="RGB+ED", rasterize_mode="antialiased",
                                  return_depth_normal=True)
    torch.autograd.backward(list(out[:5]), ups)
    info["I"] = int(out[5]["n_isects"])


for mode in ((True, False) if args.both else (True,)):
    rendering.ENABLE_ND_ONE_PASS = mode
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    ops.KERNEL_EVENTS = {}
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    k = {n: sum(a.elapsed_time(b) for a, b in v[1:]) / max(len(v) - 1, 1) for n, v in ev.items()}
    Dp, I, Px = D + 1, info["I"], W * H
    bwd_bytes = Px * ((36 + 4 * Dp) + (28 + 4 * Dp)) + I * (48 + 4 * Dp) + N * (48 + 4 * Dp)
    fwd_bytes = I * (48 + 4 * Dp) + Px * (36 + 4 * Dp)
    line = {"workload": f"features model call, {N} Gaussians, {W}x{H}, D = {D} fused channels + ED, one_pass={mode}",
            "ms_per_step": round(ms, 4), "Msplats_per_s": round(N / ms / 1e3, 1), "n_isects": I,
            "kernel_ms": {n: round(v, 4) for n, v in k.items()},
            "roofline": {"bound": "hbm", "kernel": "blend_bwd_x_atomic", "algorithmic_bytes": bwd_bytes,
                         "achieved_GBs": round(bwd_bytes / (k.get("blend_bwd", float("nan")) * 1e-3) / 1e9, 1), "peak_GBs": 8000.0,
                         "frac": round(bwd_bytes / (k.get("blend_bwd", float("nan")) * 1e-3) / 1e9 / 8000.0, 5),
                         "fwd_algorithmic_bytes": fwd_bytes,
                         "fwd_frac": round(fwd_bytes / (k.get("blend_fwd", float("nan")) * 1e-3) / 1e9 / 8000.0, 5),
                         "note": "SURVEY 8(d) byte counts with 4 D' (D' = 17) in place of the colour bytes; the kernels are VALU bound"}}
    print(json.dumps(line) if args.json else f"D=16+ED one_pass={mode}: {ms:.3f} ms/step  kernels {line['kernel_ms']}  frac {line['roofline']['frac']}", flush=True)
