"""Pure host cost of one fwd+bwd step: a scene so small that the GPU work is negligible."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
N, W, H = 2000, 64, 64
sc = random_scene(N, W, H, seed=42); dev = 'cuda'
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
def step():
    for p in params.values(): p.grad = None
    out = rasterization(params["means"], params["quats"], torch.exp(params["log_scales"]), torch.sigmoid(params["opacity_logits"]), params["sh"], V, K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True)
    t1 = time.perf_counter()
    torch.autograd.backward(list(out[:5]), ups)
    return t1
for _ in range(20): step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter(); tf = 0.0
for _ in range(n):
    ts = time.perf_counter(); t1 = step(); tf += t1 - ts
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"tiny scene: {dt / n * 1e3:.3f} ms/step wall (forward part {tf / n * 1e3:.3f} ms) = host-side cost of a step")
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
