"""Experiment (GPU box): what does longest-first launch order buy the compositing kernels?
Runs the bench step with (a) default order, (b) fwd+bwd ordered by the forward's measured per-unit work."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import ops
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene

dev = torch.device("cuda:0")
N, W, H = 1_000_000, 1920, 1080
sc = random_scene(N, W, H, seed=42)
params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
V, K = sc["viewmats"].to(dev), sc["Ks"].to(dev)
g = torch.Generator().manual_seed(7)
ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, 4), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]

def step():
    for p in params.values():
        p.grad = None
    out = rasterization(params["means"], params["quats"], torch.exp(params["log_scales"]), torch.sigmoid(params["opacity_logits"]),
                        params["sh"], V, K, W, H, sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased",
                        return_depth_normal=True)
    torch.autograd.backward(list(out[:5]), ups)

def timed(tag, n=20):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ops.KERNEL_EVENTS = {}
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ev, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    kt = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in ev.items()}
    print(f"{tag}: step {dt*1e3:.3f} ms  fwd {kt['blend_fwd']*1e3:.0f} us  bwd {kt['blend_bwd']*1e3:.0f} us", flush=True)

timed("default order")
units = ((W + 15) // 16) * ((H + 15) // 16) * 2
work = torch.zeros(units, device=dev, dtype=torch.int32)
ops.UNIT_WORK = work
step()
torch.cuda.synchronize()
ops.UNIT_WORK = None
w = work.clone()
print("work: mean %.0f max %d p90 %.0f zero %d" % (w.float().mean(), w.max(), w.float().quantile(0.9), (w == 0).sum()))
per = (units + 7) // 8
def strip_lpt(w):
    perm = torch.empty(per * 8, device=dev, dtype=torch.int32)
    for x in range(8):
        lo, hi = x * per, min((x + 1) * per, units)
        idx = torch.argsort(w[lo:hi].float(), descending=True) + lo
        full = torch.full((per,), units, device=dev, dtype=torch.int64)      # pad: unit >= total -> block exits
        full[: hi - lo] = idx
        perm[x::8] = full.int()                                              # block b -> xcd b & 7, rank b >> 3
    return perm.contiguous()
perm = strip_lpt(w)
ops.UNIT_PERM_BWD = perm
timed("bwd longest-first (fwd-measured work)")
ops.UNIT_PERM_FWD = perm
timed("fwd + bwd longest-first")
glob = torch.argsort(w.float(), descending=True).int()
pad = torch.full((per * 8 - units,), units, device=dev, dtype=torch.int32)
ops.UNIT_PERM_FWD = ops.UNIT_PERM_BWD = torch.cat([glob, pad]).contiguous()
timed("global longest-first (no XCD strips)")
