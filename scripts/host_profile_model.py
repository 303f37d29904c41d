"""Where does the HOST spend a MODEL step (get_outputs -> get_loss_dict -> backward)?  cProfile on the GPU box, small scene
so that the host is what is measured."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import radegs, parallel
from collab_splats_amd.synthetic import random_scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 480
H = int(sys.argv[3]) if len(sys.argv) > 3 else 270
dev = torch.device("cuda:0")
sc = random_scene(N, W, H, seed=42)
cfg = radegs.RadegsModelConfig(rasterize_mode="antialiased", regularization_from_iter=0)
model = radegs.RadegsModel(cfg, sc["means"], sc["log_scales"], sc["quats"], sc["opacity_logits"], sc["sh"][:, 0], sc["sh"][:, 1:]).to(dev)
model.train(); model.step = 20000
V, Ks = sc["viewmats"], sc["Ks"]
c2w = torch.linalg.inv(V[0])[:3, :4].clone(); c2w[:, 1:3] *= -1.0
cam = radegs.PinholeCamera.make(c2w, float(Ks[0, 0, 0]), float(Ks[0, 1, 1]), W, H)
target = torch.rand(H, W, 3).to(dev)
leaves = [model.gauss_params[k] for k in parallel.GRAD_KEYS]
def step():
    for p in leaves:
        p.grad = None
    out = model.get_outputs(cam)
    loss = model.get_loss_dict(out, {"image": target})
    sum(loss.values()).backward()
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
