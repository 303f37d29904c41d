"""Isolated check of the hipGraph replay path (GPU box): same results as plain launches; eviction while in flight."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import ops
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
dev = torch.device("cuda:0")
W, H = 160, 96

def run(N, steps, graphs):
    ops.GRAPHS = graphs
    sc = random_scene(N, W, H, seed=31)
    p = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
    V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
    acc = None
    for i in range(steps):
        for t in p.values():
            t.grad = None
        out = rasterization(p["means"], p["quats"], torch.exp(p["log_scales"]), torch.sigmoid(p["opacity_logits"]), p["sh"],
                            V, K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True)
        sum(o.sum() for o in out[:5]).backward()
        with torch.no_grad():
            p["means"] += 1e-3                      # the scene (and the intersection count) drifts
    torch.cuda.synchronize()
    return [o.detach().clone() for o in out[:5]] + [t.grad.clone() for t in p.values()]

stage = sys.argv[1] if len(sys.argv) > 1 else "replay"
if stage == "replay":
    a = run(1500, 12, False)
    b = run(1500, 12, True)
    print("stats", ops.graph_cache_stats())
    for x, y in zip(a, b):
        d = (x - y).abs().max().item() / max(x.abs().max().item(), 1e-30)
        assert d < 1e-4, d
    print("replay == plain launches: OK")
else:
    for rep in range(2):
        for N in range(1000, 1000 + 40 * 37, 37):
            run(N, 3, True)
    torch.cuda.synchronize()
    print("stats", ops.graph_cache_stats())
    print("eviction under load: OK")
