"""Eval-time throughput: V views of one scene, one by one vs batched through render_views."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collab_splats_amd import radegs
from collab_splats_amd.synthetic import random_scene, view_matrix
N, W, H, V = 1_000_000, 1920, 1080, 8
dev = torch.device("cuda:0")
sc = random_scene(N, W, H, seed=42)
model = radegs.RadegsModel(radegs.RadegsModelConfig(rasterize_mode="antialiased"), sc["means"], sc["log_scales"], sc["quats"],
                           sc["opacity_logits"], sc["sh"][:, 0], sc["sh"][:, 1:]).to(dev).eval()
model.step = 10_000
flip = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0]))
cams = [radegs.PinholeCamera.make((torch.linalg.inv(view_matrix(i)[0]) @ flip)[:3, :4], 0.9 * W, 0.9 * W, W, H) for i in range(V)]
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
one = t(lambda: [model.get_outputs_for_camera(c) for c in cams])
for b in (1, 2, 4, 8):
    print(f"batch {b}: {t(lambda: model.render_views(cams, batch_size=b)):.2f} ms for {V} views ({V * N / t(lambda: model.render_views(cams, batch_size=b)) / 1e3:.0f} Msplats/s fwd)")
print(f"one by one (get_outputs_for_camera): {one:.2f} ms for {V} views")
