#!/usr/bin/env python3
"""Minimal training loop on the MI355X: RadegsModel (host mirror of the reference's rade-gs model) +
one (fused) Adam optimizer per parameter group (learning rates of collab_splats/configs/rade_gs_method.py:44-71)
+ DefaultStrategy densification, fitting renders of a hidden synthetic scene from a ring of cameras.

    python examples/train_synthetic.py [--steps 300] [--gaussians 20000] [--width 320] [--height 200]
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collab_splats_amd import FusedAdam, fused_adam_step_all, radegs  # noqa: E402
from collab_splats_amd.synthetic import random_scene  # noqa: E402


def ring_camera(i: int, n: int, W: int, H: int) -> radegs.PinholeCamera:
    """OpenGL camera-to-world on a small arc around (0, 0, 7), looking at it."""
    ang = (i / max(n - 1, 1) - 0.5) * math.radians(30.0)
    eye = torch.tensor([7.0 * math.sin(ang), 0.0, 7.0 - 7.0 * math.cos(ang)])
    fwd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 7.0]) - eye, dim=0)    # OpenCV +z
    right = torch.nn.functional.normalize(torch.cross(torch.tensor([0.0, -1.0, 0.0]), fwd, dim=0), dim=0)
    down = torch.cross(fwd, right, dim=0)
    c2w = torch.stack([right, -down, -fwd, eye], dim=1)                               # OpenGL: y up, z back
    return radegs.PinholeCamera.make(c2w, 0.9 * W, 0.9 * W, W, H)


def build_model(sc, cfg, dev, noise=None, generator=None):
    p = {k: sc[k].clone() for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
    if noise:
        for k, s in noise.items():
            p[k] = p[k] + s * torch.randn(p[k].shape, generator=generator)
    return radegs.RadegsModel(cfg, p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"][:, 0].contiguous(),
                              p["sh"][:, 1:].contiguous()).to(dev)


LRS = {"means": 1.6e-4 * 10, "features_dc": 2.5e-3, "features_rest": 2.5e-3 / 20, "opacities": 5e-2, "scales": 5e-3,
       "quats": 1e-3}


def train(steps=300, n=20000, W=320, H=200, n_views=8, refine_every=50, seed=0, verbose=True):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    sc = random_scene(n, W, H, seed=42 + seed)
    cfg = radegs.RadegsModelConfig(rasterize_mode="antialiased", sh_degree_interval=1, regularization_from_iter=steps // 2)
    truth = build_model(sc, cfg, dev).eval()
    cams = [ring_camera(i, n_views, W, H) for i in range(n_views)]
    with torch.no_grad():
        targets = [truth.get_outputs(c)["rgb"].clone() for c in cams]
    model = build_model(sc, cfg, dev, noise={"means": 0.05, "log_scales": 0.4, "quats": 0.3, "opacity_logits": 1.0, "sh": 0.3},
                        generator=g).train()
    model.step = 10                                    # all SH degrees active
    model.strategy.refine_start_iter, model.strategy.refine_every = 0, refine_every
    model.strategy.grow_grad2d, model.strategy.refine_stop_iter = 2e-4, steps
    # one FusedAdam per parameter group (the interface the strategy edits), all stepped by ONE kernel launch
    model.optimizers = {k: FusedAdam([v], lr=LRS[k], eps=1e-15) for k, v in model.gauss_params.items()}
    log = []
    t0 = time.perf_counter()
    for it in range(steps):
        model.step = 10 + it
        cam, tgt = cams[it % n_views], targets[it % n_views]
        for o in model.optimizers.values():
            o.zero_grad(set_to_none=True)
        out = model.get_outputs(cam)
        losses = model.get_loss_dict(out, {"image": tgt})
        loss = sum(losses.values())
        loss.backward()
        fused_adam_step_all(model.optimizers)
        counts = model.strategy.step_post_backward(model.gauss_params, model.optimizers, model.strategy_state, it + 1, model.info)
        log.append((losses["main_loss"].item(), model.means.shape[0], counts))
        if verbose and (it % 50 == 0 or it == steps - 1):
            extra = "".join(f"  {k} {v.item():.5f}" for k, v in losses.items())
            print(f"step {it:4d}{extra}  gaussians {model.means.shape[0]}  dup/split/prune {counts}")
    torch.cuda.synchronize()
    if verbose:
        print(f"{steps} steps in {time.perf_counter() - t0:.2f} s")
    return log


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--gaussians", type=int, default=20000)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=200)
    a = ap.parse_args()
    train(a.steps, a.gaussians, a.width, a.height)
