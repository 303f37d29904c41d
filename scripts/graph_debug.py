import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import ops
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
dev = torch.device("cuda:0")
N, W, H = 100000, 1920, 1080
sc = random_scene(N, W, H, seed=42)
p = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
orig_wait = ops._wait_count
def wait(host):
    n = orig_wait(host)
    print("  count", n, hex(n), flush=True)
    return n
ops._wait_count = wait
try:
    for i in range(8):
        for t in p.values():
            t.grad = None
        out = rasterization(p["means"], p["quats"], torch.exp(p["log_scales"]), torch.sigmoid(p["opacity_logits"]), p["sh"],
                            V, K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True)
        torch.autograd.backward(list(out[:5]), ups)
        print("step", i, "n", out[5]["n_isects"], ops.graph_cache_stats(), flush=True)
except Exception as e:
    print("EXC", e)
torch.cuda.synchronize()
print("done")
