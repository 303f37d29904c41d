"""Soak run (GPU box): the eager training-shaped loop for a few hundred steps -- cycling views, parameters updated in place every
step, the number of Gaussians changed every 100 steps (what densification does to the shapes) -- and what the machinery did:
capacity misses, graph replays, arena slots, allocator growth.  `python scripts/soak.py [N] [steps]` -> one JSON line."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import arena, ops
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene, view_matrix     # (the bench's scene and its eight cameras)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
W, H = 1920, 1080
FEATS = int(os.environ.get("SOAK_FEATURES", "0"))                   # > 0: the features model's call (SH + F feature channels)
dev = torch.device("cuda:0")
sc = random_scene(N, W, H, seed=42)
names = ("means", "quats", "log_scales", "opacity_logits", "sh")
params = {k: sc[k].to(dev).requires_grad_(True) for k in names}
if FEATS:
    params["features"] = torch.rand(N, FEATS, generator=torch.Generator().manual_seed(3)).to(dev).requires_grad_(True)
views = [view_matrix(v).to(dev) for v in range(8)]
K = sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4 + FEATS), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
marks = {}
front = []                                                           # (tiles sorted in front only, tiles flagged, tiles) of sampled steps
STEP = float(os.environ.get("SOAK_STEP", "1e-4"))
t_mark = time.perf_counter()
for it in range(STEPS):
    if it and it % 100 == 0:                                         # "densification": the shapes change
        n_old = params["means"].shape[0]
        keep = torch.randperm(n_old, device=dev)[: int(n_old * (0.9 if (it // 100) % 2 else 1.0))]
        extra = keep[: n_old // 20]
        idx = torch.cat((keep, extra))
        params = {k: p.detach()[idx].clone().requires_grad_(True) for k, p in params.items()}
    for p in params.values():
        p.grad = None
    out = rasterization(params["means"], params["quats"], torch.exp(params["log_scales"]), torch.sigmoid(params["opacity_logits"]),
                        params["sh"], views[it % 8], K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased",
                        return_depth_normal=True, **({"features": params["features"]} if FEATS else {}))
    torch.autograd.backward(list(out[:5]), ups)
    part = out[5]["_bins"]["partial"] if it % 8 == 3 else None       # front-only ordering (dense scenes): how stale were the pivots?
    if part is not None:
        front.append((int((part["front_n"] >= 0).sum()), int((part["tile_flag"] != 0).sum()), int(part["front_n"].numel())))
    with torch.no_grad():                                            # the optimiser's in-place update
        for k, p in params.items():
            if it % 25 == 0:
                assert torch.isfinite(p.grad).all(), (it, k)
            p.add_(torch.sign(p.grad), alpha=-STEP)                  # (a bounded step: 400 of them move a value by 0.04 at most)
    del out
    if it in (49, 99, 199, 299, STEPS - 1):
        torch.cuda.synchronize()
        now = time.perf_counter()
        marks[it + 1] = {"n_gauss": int(params["means"].shape[0]), "reserved_GB": round(torch.cuda.memory_reserved() / 2 ** 30, 3),
                         "allocated_GB": round(torch.cuda.memory_allocated() / 2 ** 30, 3), "graph": ops.graph_cache_stats(),
                         "arena": dict(arena.STATS), "rings": len(arena._RINGS), "order_tables": len(ops._ORDER_TABLES),
                         "capacity_redo": ops.PATH_STATS["capacity_redo"], "probes": ops.PATH_STATS["forward_probe"],
                         "wall_s": round(now - t_mark, 3)}
for p in params.values():
    assert torch.isfinite(p).all()
print(json.dumps({"steps": STEPS, "step_size": STEP, "marks": marks, "front_only_samples": front[-12:],
                  "front_only_flagged_mean": (sum(f[1] for f in front[4:]) / max(len(front[4:]), 1)) if front else None}))
if ops.KEY_TRACE:                                                    # (ops.KEY_TRACE = [] before the run: what moved between two visits of a view)
    for line in ops.key_trace_report(16, start=64)[:12]:
        print("key-trace", line, file=sys.stderr)
