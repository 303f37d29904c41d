"""Per-view distribution of the compositing work (1 M / 1080p, the eight views of configs[3]): intersections, staged
Gaussians per (tile, band) unit as the forward measures them (unit_work), and the compositing kernels' durations.
    python scripts/unit_work_stats.py [N]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collab_splats_amd import ops
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene, view_matrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W, H = 1920, 1080
dev = torch.device("cuda", 0)
sc = random_scene(N, W, H, seed=42)
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
Ks = sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
for v in list(range(8)) + list(range(8)):
    V = view_matrix(v).to(dev)
    for p in params.values():
        p.grad = None
    ops.KERNEL_EVENTS = {}
    out = rasterization(params["means"], params["quats"], params["log_scales"], params["opacity_logits"], params["sh"], V, Ks, W, H,
                        sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True,
                        scales_are_log=True, opacities_are_logit=True)
    torch.autograd.backward(list(out[:5]), ups)
    torch.cuda.synchronize()
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    work = out[0].grad_fn.sched.work.float()
    fwd = ev["blend_fwd"][0][0].elapsed_time(ev["blend_fwd"][0][1]) if "blend_fwd" in ev else float("nan")
    bwd = ev["blend_bwd"][0][0].elapsed_time(ev["blend_bwd"][0][1]) if "blend_bwd" in ev else float("nan")
    meta = out[5]
    cnt = torch.diff(torch.cat([meta["isect_offsets"].reshape(-1), torch.tensor([meta["n_isects"]], device=dev, dtype=torch.int32)])).float()
    alive = (out[1][0, ..., 0] < 0.9999).float().mean().item()
    q = torch.quantile(work, torch.tensor([0.5, 0.9, 0.99], device=dev))
    print(f"view {v}: isects {meta['n_isects']/1e6:.2f} M, bucket mean {cnt.mean():.0f} max {cnt.max():.0f}; unit work sum {work.sum()/1e6:.2f} M "
          f"mean {work.mean():.0f} p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} max {work.max():.0f}; "
          f"pixels with alpha < 0.9999: {alive:.3f}; fwd {fwd:.3f} ms bwd {bwd:.3f} ms; "
          f"fwd ns per staged entry {fwd*1e6/work.sum():.2f}", flush=True)

# ---- what predicts a unit's work before the forward runs?  (rank correlation with the same step's bucket length, with the
# previous view's work, and with the same view's work one step earlier)
import numpy as np


def spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)), np.argsort(np.argsort(b))
    return float(np.corrcoef(ra, rb)[0, 1])


prev = None
for v in list(range(8)):
    V = view_matrix(v).to(dev)
    out = rasterization(params["means"], params["quats"], params["log_scales"], params["opacity_logits"], params["sh"], V, Ks, W, H,
                        sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True,
                        scales_are_log=True, opacities_are_logit=True)
    torch.cuda.synchronize()
    work = out[0].grad_fn.sched.work.cpu().numpy().astype(np.float64)
    meta = out[5]
    cnt = torch.diff(torch.cat([meta["isect_offsets"].reshape(-1), torch.tensor([meta["n_isects"]], device=dev, dtype=torch.int32)])).cpu().numpy()
    blen = np.repeat(cnt, 2).astype(np.float64)                      # two bands per tile
    print(f"view {v}: spearman(work, bucket length) {spearman(work, blen):.3f}"
          + (f", spearman(work, previous view's work) {spearman(work, prev):.3f}" if prev is not None else ""), flush=True)
    prev = work
