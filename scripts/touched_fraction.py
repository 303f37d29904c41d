"""How many visible Gaussians receive a gradient at all?  (rows the compositing backward never touches could be skipped
by the per-Gaussian backward kernels)    python scripts/touched_fraction.py [N ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
dev = torch.device("cuda:0")
W, H = 1920, 1080
for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 5_000_000]:
    sc = random_scene(N, W, H, seed=42)
    p = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
    out = rasterization(p["means"], p["quats"], torch.exp(p["log_scales"]), torch.sigmoid(p["opacity_logits"]), p["sh"],
                        sc["viewmats"].to(dev), sc["Ks"].to(dev), W, H, sh_degree=3, render_mode="RGB+ED",
                        rasterize_mode="antialiased", return_depth_normal=True)
    g = torch.Generator().manual_seed(7)
    ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
    torch.autograd.backward(list(out[:5]), ups)
    vis = int((out[5]["radii"] > 0).any(-1).sum())
    touched = int((p["opacity_logits"].grad != 0).sum())
    print(f"N={N}: visible {vis} ({vis/N:.2f}), with a gradient {touched} ({touched/max(vis,1):.2f} of the visible), "
          f"intersections {int(out[5]['n_isects'])}", flush=True)
    del p, out, ups
