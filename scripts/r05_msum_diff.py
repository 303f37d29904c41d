"""GPU: gradients of one steady-state call under the shipped library (or MISPLAT_LIB) -> an .npz; with two files, their differences."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    for k in a.files:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        s = max(np.abs(x).max(), 1e-30)
        d = np.abs(x - y) / s
        print(f"{k:16s} max|ref| {s:.4g}  max rel diff {d.max():.3e}  mean {d.mean():.3e}  p99.99 {np.quantile(d, 0.9999):.3e}")
    sys.exit(0)
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene, view_matrix
N, W, H = 1_000_000, 1920, 1080
dev = torch.device("cuda:0")
sc = random_scene(N, W, H, seed=42)
names = ("means", "quats", "log_scales", "opacity_logits", "sh")
leaves = [sc[k].to(dev).requires_grad_(True) for k in names]
V, K = view_matrix(3).to(dev), sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
for _ in range(4):
    for l in leaves:
        l.grad = None
    out = rasterization(*leaves, V, K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True,
                        scales_are_log=True, opacities_are_logit=True)
    torch.autograd.backward(list(out[:5]), ups)
torch.cuda.synchronize()
np.savez(sys.argv[1], means2d=out[5]["means2d"].grad.cpu().numpy(), **{k: l.grad.cpu().numpy() for k, l in zip(names, leaves)})
