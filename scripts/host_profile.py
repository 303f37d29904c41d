"""Where does the HOST spend a step at a small size? (cProfile of the bench step, GPU box)"""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
sc = random_scene(N, W, H, seed=42)
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
def step():
    for p in params.values():
        p.grad = None
    out = rasterization(params["means"], params["quats"], params["log_scales"], params["opacity_logits"],
                        params["sh"], V, K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True,
                        scales_are_log=True, opacities_are_logit=True)
    torch.autograd.backward(list(out[:5]), ups)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(45)
# the backward runs on the autograd engine's device thread, which cProfile does not see: time it by hand
import collab_splats_amd.ops as ops
orig = ops._RasterFused.backward
acc = [0.0, 0]
def timed_bwd(ctx, *a):
    t = time.perf_counter()
    r = orig(ctx, *a)
    acc[0] += time.perf_counter() - t; acc[1] += 1
    return r
ops._RasterFused.backward = staticmethod(timed_bwd)
for _ in range(200): step()
torch.cuda.synchronize()
print("python time inside _RasterFused.backward: %.1f us per call (%d calls)" % (acc[0] / max(acc[1], 1) * 1e6, acc[1]))
