import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collab_splats_amd import FusedAdam, fused_adam_step_all
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = "cuda"
shapes = dict(means=(N, 3), features_dc=(N, 3), features_rest=(N, 15, 3), opacities=(N, 1), scales=(N, 3), quats=(N, 4))
lrs = dict(means=1.6e-4, features_dc=0.0025, features_rest=0.0025 / 20, opacities=0.05, scales=0.005, quats=0.001)
def make(cls):
    ps = {k: torch.randn(s, device=dev).requires_grad_(True) for k, s in shapes.items()}
    for p in ps.values(): p.grad = torch.randn_like(p)
    return ps, {k: cls([ps[k]], lr=lrs[k], eps=1e-15) for k in shapes}
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
_, ours = make(FusedAdam)
_, ref = make(torch.optim.Adam)
_, ref_fused = make(lambda p, **k: torch.optim.Adam(p, fused=True, **k))
t1 = timeit(lambda: fused_adam_step_all(ours))
t2 = timeit(lambda: [o.step() for o in ref.values()])
t3 = timeit(lambda: [o.step() for o in ref_fused.values()])
bytes_ = sum(torch.Size(s).numel() for s in shapes.values()) * 4 * 7
print(f"N={N}: misplat fused Adam {t1:.3f} ms ({bytes_ / t1 / 1e6:.0f} GB/s algorithmic), torch Adam (foreach) {t2:.3f} ms, torch Adam(fused=True) {t3:.3f} ms")
