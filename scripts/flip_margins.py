"""Diagnostics (GPU box): threshold margins of the pixels where the HIP path and the C restatement differ."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.craster import CRaster
from collab_splats_amd import rasterization
from collab_splats_amd.synthetic import random_scene, view_matrix

dev = torch.device("cuda:0")
for (N, W, H, view) in ((100000, 1920, 1080, None), (100000, 1920, 1080, 5), (20000, 640, 360, None), (400000, 1920, 1080, 2)):
    sc = random_scene(N, W, H, seed=42)
    V = sc["viewmats"] if view is None else view_matrix(view)
    scales, op = torch.exp(sc["log_scales"]), torch.sigmoid(sc["opacity_logits"])
    out = rasterization(sc["means"].to(dev), sc["quats"].to(dev), scales.to(dev), op.to(dev), sc["sh"].to(dev), V.to(dev),
                        sc["Ks"].to(dev), W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased",
                        return_depth_normal=True)
    cr = CRaster(np.float32)
    st = cr.forward(sc["means"].numpy(), sc["quats"].numpy(), scales.numpy(), op.numpy(), sc["sh"].numpy(), V[0].numpy(),
                    sc["Ks"][0].numpy(), W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased")
    m = cr.blend_margin(st)
    fw = st["fwd"]
    bad = np.zeros((H, W), bool)
    for got, ref in ((out[0], st["render"]), (out[1], fw["alpha"]), (out[2], fw["exp_depth"]), (out[3], fw["med_depth"]), (out[4], fw["normal"])):
        g = got[0].cpu().numpy().astype(np.float64)
        d = np.abs(g - ref) / max(np.abs(ref).max(), 1e-30)
        bad |= (d > 1e-4).reshape(H, W, -1).any(-1)
    bad |= out[5]["last_ids"][0].cpu().numpy() != fw["last_ids"]
    bad |= out[5]["median_ids"][0].cpu().numpy() != fw["median_ids"]
    mb = np.sort(m[bad])
    print(f"N={N} {W}x{H} view={view}: {bad.sum()} differing pixels; margins/2^-24: max {mb.max()/2**-24 if mb.size else 0:.2f} "
          f"p90 {np.quantile(mb, 0.9)/2**-24 if mb.size else 0:.2f}; pixels with margin < 2e-6: {(m < 2e-6).sum()} of {m.size}", flush=True)
