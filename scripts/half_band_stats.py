"""Design tooling (CPU, oracle only): over the eight views of the headline, the share of the band backward's contributing
(band, Gaussian) trips whose contributing pixels lie in ONE 16x4 half of the 16x8 band -- exactly, and as far as a
staging-time, wave-uniform test can prove it (verdict round 4, item 2).

    python scripts/half_band_stats.py [N] [W] [H] [views...]  -> profiles/r05_half_band_share.json
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.craster import CRaster
from collab_splats_amd.synthetic import random_scene, view_matrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
views = [int(v) for v in sys.argv[4:]] or list(range(8))
sc = random_scene(N, W, H, seed=42)
cr = CRaster(np.float32)
rows, tot = [], {}
for v in views:
    V = view_matrix(v)[0]
    t0 = time.time()
    st = cr.forward(sc["means"].numpy(), sc["quats"].numpy(), torch.exp(sc["log_scales"]).numpy(),
                    torch.sigmoid(sc["opacity_logits"]).numpy(), sc["sh"].numpy(), V.numpy(), sc["Ks"][0].numpy(), W, H,
                    sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased")
    r = cr.half_stats(st)
    r["view"] = v; r["n_isects"] = int(st["bins"]["n_isects"]); r["seconds"] = round(time.time() - t0, 1)
    r["share_exact"] = round(r["one_half_exact"] / max(r["contributing"], 1), 4)
    r["share_provable"] = round(r["one_half_provable"] / max(r["contributing"], 1), 4)
    r["lane_use"] = round(r["pairs"] / max(r["contributing"] * 128, 1), 4)
    print(json.dumps(r), flush=True)
    rows.append(r)
    for k in ("staged", "contributing", "one_half_exact", "one_half_provable", "pairs", "one_half_box_all_staged"):
        tot[k] = tot.get(k, 0) + r[k]
tot["share_exact"] = round(tot["one_half_exact"] / tot["contributing"], 4)
tot["share_provable"] = round(tot["one_half_provable"] / tot["contributing"], 4)
tot["fwd_share_box_only"] = round(tot["one_half_box_all_staged"] / tot["staged"], 4)
tot["lane_use"] = round(tot["pairs"] / (tot["contributing"] * 128), 4)
out = dict(workload=f"{N} Gaussians, {W}x{H}, views {views}", total=tot, views=rows)
print(json.dumps(tot))
if N == 1_000_000 and len(views) == 8:
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_half_band_share.json"), "w"), indent=1)
