"""fwd+bwd of the features-model rasterizer call (rade_features_model.py:450-476): 16 fused channels, RGB+ED."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collab_splats_amd import rendering
from collab_splats_amd.synthetic import random_scene
N, W, H, D = 1_000_000, 1920, 1080, 16
sc = random_scene(N, W, H, seed=42); dev = 'cuda'
g = torch.Generator().manual_seed(3)
feats = torch.rand(N, D, generator=g).to(dev).requires_grad_(True)
P = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits")}
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, D + 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
def step():
    for p in list(P.values()) + [feats]: p.grad = None
    out = rendering.rasterization(P["means"], P["quats"], torch.exp(P["log_scales"]), torch.sigmoid(P["opacity_logits"]), feats, V, K, W, H,
                                  sh_degree=None, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True)
    torch.autograd.backward(list(out[:5]), ups)
for mode in (True, False):
    rendering.ENABLE_ND_ONE_PASS = mode
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"D=16+ED one_pass={mode}: {(time.perf_counter()-t0)/10*1e3:.3f} ms/step")
