"""Blend kernel time for a uniform scene vs the same Gaussians squeezed into the upper half of the image
(strip-wise XCD block map vs depth-complexity variation)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collab_splats_amd import ops
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
N, W, H = 1_000_000, 1920, 1080
dev = "cuda"
for name in ("uniform", "upper-half", "centre-blob"):
    sc = random_scene(N, W, H, seed=42)
    m = sc["means"].clone()
    if name == "upper-half":
        m[:, 1] = -m[:, 1].abs()
    if name == "centre-blob":
        m[:, :2] = m[:, :2] * torch.rand(N, 1) ** 1.5
    ins = [t.to(dev).requires_grad_(True) for t in (m, sc["quats"], torch.exp(sc["log_scales"]), torch.sigmoid(sc["opacity_logits"]), sc["sh"])]
    V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
    ops.KERNEL_EVENTS = {}
    for it in range(8):
        for t in ins: t.grad = None
        out = rasterization(*ins, V, K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True)
        torch.autograd.backward(list(out[:5]), [torch.ones_like(o) for o in out[:5]])
    torch.cuda.synchronize()
    t = {k: sum(a.elapsed_time(b) for a, b in v[3:]) / len(v[3:]) for k, v in ops.KERNEL_EVENTS.items()}
    print(f"{name:12s} I={out[5]['n_isects']:9d} blend_fwd {t['blend_fwd']:.3f} ms  blend_bwd {t['blend_bwd']:.3f} ms")
ops.KERNEL_EVENTS = None
