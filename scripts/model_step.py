"""Times one full model step of the host mirror (a1 + a2 + a3 + a4 + a5): get_outputs -> losses -> backward."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collab_splats_amd import radegs
from collab_splats_amd.synthetic import random_scene
N, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 1920, 1080
sc = random_scene(N, W, H, seed=42); dev = 'cuda'
cfg = radegs.RadegsModelConfig(rasterize_mode="antialiased", regularization_from_iter=0, sh_degree_interval=1)
model = radegs.RadegsModel(cfg, sc["means"], sc["log_scales"], sc["quats"], sc["opacity_logits"], sc["sh"][:, 0], sc["sh"][:, 1:]).to(dev)
model.train(); model.step = 20000
c2w = torch.tensor([[1.0, 0, 0, 0], [0, -1.0, 0, 0], [0, 0, -1.0, 0]])
cam = radegs.PinholeCamera.make(c2w, 0.9 * W, 0.9 * W, W, H)
img = torch.rand(H, W, 3, device=dev)
def step():
    model.zero_grad(set_to_none=True)
    out = model.get_outputs(cam)
    loss = sum(model.get_loss_dict(out, {"image": img}).values())
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 20
for _ in range(K): step()
torch.cuda.synchronize()
print(f"model step (get_outputs + L1 + depth-normal loss + backward): {(time.perf_counter()-t0)/K*1e3:.3f} ms, N={N}, I={model.info['n_isects']}")
