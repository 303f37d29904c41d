"""Soak run of the model mirror (GPU box): examples/train_synthetic.py's loop -- RadegsModel.get_outputs -> get_loss_dict ->
backward -> fused Adam -> DefaultStrategy refinement -- at a size that matters, and what the machinery under it did.
`python scripts/soak_model.py [N] [steps] [refine_every]` -> one JSON line."""
import importlib.util, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import arena, ops
spec = importlib.util.spec_from_file_location("train_synthetic", os.path.join(ROOT, "examples", "train_synthetic.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
EVERY = int(sys.argv[3]) if len(sys.argv) > 3 else 100
t0 = time.perf_counter()
log = mod.train(steps=STEPS, n=N, W=1920, H=1080, n_views=8, refine_every=EVERY, verbose=False)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
losses = [l for l, _, _ in log]
print(json.dumps({"steps": STEPS, "wall_s": round(wall, 2), "loss_first8": round(sum(losses[:8]) / 8, 4), "loss_last8": round(sum(losses[-8:]) / 8, 4),
                  "sizes": sorted({n for _, n, _ in log}), "graph": ops.graph_cache_stats(), "arena": dict(arena.STATS), "rings": len(arena._RINGS),
                  "path": {k: v for k, v in ops.PATH_STATS.items() if v}, "reserved_GB": round(torch.cuda.memory_reserved() / 2 ** 30, 2),
                  "allocated_GB": round(torch.cuda.memory_allocated() / 2 ** 30, 2)}))
