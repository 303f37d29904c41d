"""Why is bench.py slower than a bare loop at 10 k Gaussians?  One difference at a time."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
N, W, H = 10000, 256, 256
dev = torch.device("cuda:0")
sc = random_scene(N, W, H, seed=42)
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
info = {}
def step(keep=False):
    for p in params.values():
        p.grad = None
    out = rasterization(params["means"], params["quats"], torch.exp(params["log_scales"]), torch.sigmoid(params["opacity_logits"]),
                        params["sh"], V, K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True)
    torch.autograd.backward(list(out[:5]), ups)
    if keep:
        info["n_isects"], info["n_visible"] = out[5]["n_isects"], out[5]["radii"]
def run(tag, n=200, events=False, keep=False):
    for _ in range(20): step(keep)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)] if events else None
    t0 = time.perf_counter()
    for i in range(n):
        if events: evs[i][0].record()
        step(keep)
        if events: evs[i][1].record()
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)
run("plain")
run("keep meta radii alive (bench info dict)", keep=True)
run("per-step timing events", events=True)
run("both", events=True, keep=True)
big = torch.empty(512 << 20, device=dev, dtype=torch.uint8); big2 = torch.empty_like(big); del big, big2
run("after a 1 GiB alloc/free", events=False, keep=False)
run("plain again")
