import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import arena, ops
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
N, W, H = 10000, 256, 256
dev = torch.device("cuda:0")
sc = random_scene(N, W, H, seed=42)
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
for it in range(8):
    for p in params.values():
        p.grad = None
    out = rasterization(params["means"], params["quats"], params["log_scales"], params["opacity_logits"], params["sh"], V, K, W, H,
                        sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True, scales_are_log=True,
                        opacities_are_logit=True)
    torch.autograd.backward(list(out[:5]), ups)
    del out
    torch.cuda.synchronize()
    rings = {k[0]: [(s._count(), s.floor, s.plan_refs, len(s.plans)) for s in r.slots] for k, r in arena._RINGS.items()}
    print(it, {k: ops.PATH_STATS[k] for k in ("forward", "forward_plan_hit", "backward_plan_hit", "forward_probe", "forward_arena_slot")}, rings, flush=True)
