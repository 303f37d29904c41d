"""Design tooling: how much of the compositing work is useful, per block shape (CPU, oracle only).

    python scripts/work_stats.py [N] [W] [H] [view]

Runs the C restatement's forward on the bench scene and counts, for block shapes inside the 16x16 tile,
the (block, Gaussian) units a block-per-wave design has to traverse and the pixel pairs that contribute.
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.craster import CRaster
from collab_splats_amd.synthetic import random_scene, view_matrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
view = int(sys.argv[4]) if len(sys.argv) > 4 else -1
sc = random_scene(N, W, H, seed=42)
V = sc["viewmats"][0] if view < 0 else view_matrix(view)[0]
cr = CRaster(np.float32)
t0 = time.time()
st = cr.forward(sc["means"].numpy(), sc["quats"].numpy(), torch.exp(sc["log_scales"]).numpy(),
                torch.sigmoid(sc["opacity_logits"]).numpy(), sc["sh"].numpy(), V.numpy(), sc["Ks"][0].numpy(), W, H,
                sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased")
I = st["bins"]["n_isects"]
print(f"forward {time.time()-t0:.1f}s  N={N} I={I} P={W*H}", flush=True)
t0 = time.time()
res = cr.blend_stats(st)
print(f"stats {time.time()-t0:.1f}s")
pairs = next(iter(res.values()))["pairs"]
print(f"contributing pairs {pairs/1e6:.1f} M = {pairs/(W*H):.1f} per pixel")
for (bw, bh), r in res.items():
    px = bw * bh
    print(f"{bw:2d}x{bh:<2d}: traversed {r['traversed']/1e6:7.2f} M  culled(exact) {r['culled']/1e6:7.2f} M  hit {r['hit']/1e6:7.2f} M"
          f"  pixel slots after cull {r['culled']*px/1e6:8.1f} M  lane use {pairs/max(r['culled']*px,1):.3f}")

if os.environ.get("QUAD", "1") == "1":
    for (sw, sh) in ((8, 4), (16, 2), (4, 8)):
        for th in (16, 32, 64):
            q = cr.quad_stats(st, sw, sh, th)
            print(f"quad {sw}x{sh} th={th}: band units {q['band_units']/1e6:.2f} M  sub units {q['sub_units']/1e6:.2f} M  trips/batch-sync "
                  f"{q['trips_every_batch']/1e6:.2f} M  trips(th) {q['trips']/1e6:.2f} M  rounds {q['rounds']/1e6:.3f} M  batches {q['batches']/1e6:.3f} M", flush=True)
