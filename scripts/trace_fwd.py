"""Per-wave timeline of blend_fwd from an instrumented build (MISPLAT_TRACE): unit durations, slot occupancy, gaps."""
import os, sys, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from collab_splats_amd import _lib
from collab_splats_amd.rendering import rasterization
from collab_splats_amd.synthetic import random_scene
N, W, H = 1_000_000, 1920, 1080
sc = random_scene(N, W, H, seed=42); dev = "cuda"
args = [sc["means"].to(dev), sc["quats"].to(dev), torch.exp(sc["log_scales"]).to(dev), torch.sigmoid(sc["opacity_logits"]).to(dev),
        sc["sh"].to(dev), sc["viewmats"].to(dev), sc["Ks"].to(dev), W, H]
kw = dict(sh_degree=3, render_mode="RGB+ED", rasterize_mode="antialiased", return_depth_normal=True)
with torch.no_grad():
    for _ in range(3): out = rasterization(*args, **kw)
torch.cuda.synchronize()
lib = _lib.load()
nb = 16320
buf = (C.c_ulonglong * (3 * nb))()
assert lib.misplat_trace_read(buf, C.c_int(nb)) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 3).astype(np.int64)
t0, t1, hw = a[:, 0], a[:, 1], a[:, 2]
ok = t1 > t0
t0, t1, hw = t0[ok], t1[ok], hw[ok]
blk = np.arange(nb)[ok]
xcc = (hw >> 32) & 0xF
print("blocks traced", ok.sum(), "tick = 10 ns")
print("block&7 -> XCC_ID agreement:", (xcc == (blk & 7)).mean(), " XCC histogram:", np.bincount(xcc, minlength=8))
start, end = t0.min(), t1.max()
dur = (t1 - t0) * 0.01
print(f"kernel span {(end-start)*0.01:.1f} us; unit duration mean {dur.mean():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f} us")
print(f"sum of unit time / span = {dur.sum()/((end-start)*0.01):.0f} waves resident on average (6144 slots)")
edges = np.linspace(start, end, 21)
for i in range(20):
    lo, hi = edges[i], edges[i + 1]
    occ = (np.minimum(t1, hi) - np.maximum(t0, lo)).clip(min=0).sum() / (hi - lo)
    print(f"  t {i*5:3d}-{i*5+5:3d}%: {occ:7.0f} waves resident")
# per-XCC finish time
for x in range(8):
    m = xcc == x
    print(f"  XCC {x}: units {m.sum():5d} work {dur[m].sum()/1e3:6.2f} wave-ms, first start {(t0[m].min()-start)*0.01:6.1f} us, last end {(t1[m].max()-start)*0.01:6.1f} us")
# start-time order vs block order
print("launch order: corr(block id, start time) =", np.corrcoef(blk, t0)[0,1])
