import os, sys, runpy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import ops
orig_a = ops._raster_phase_a
calls = [0]
def phase_a(P, *args, **kw):
    calls[0] += 1
    s0 = ops.graph_cache_stats()
    out = orig_a(P, *args, **kw)
    s1 = ops.graph_cache_stats()
    torch.cuda.synchronize()
    st = out[-1]
    cnt = st["keep"][5]
    vals = cnt.view(torch.int64).tolist()
    print("A call", calls[0], "hit" if s1["hits"] > s0["hits"] else "capture", "counters ptr", hex(cnt.data_ptr()), vals, "host", int(st["host"][0]), flush=True)
    return out
ops._raster_phase_a = phase_a
sys.argv = ["bench.py", "--no-cpu-baseline", "--gaussians", "100000", "--steps", "6", "--warmup", "6"]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except BaseException as e:
    print("EXC", type(e).__name__, str(e)[:300])
torch.cuda.synchronize()
