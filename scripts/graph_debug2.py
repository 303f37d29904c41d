import os, sys, runpy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import ops
orig_b = ops._raster_phase_b
calls = [0]
def phase_b(P, state, cd):
    calls[0] += 1
    try:
        return orig_b(P, state, cd)
    except Exception as e:
        torch.cuda.synchronize()
        cnt = state["keep"][5]
        print("PHASE B FAILED at call", calls[0], e, "host", int(state["host"][0]), "device counters", cnt.view(torch.int64).tolist(),
              "stats", ops.graph_cache_stats(), flush=True)
        raise
ops._raster_phase_b = phase_b
sys.argv = ["bench.py", "--no-cpu-baseline", "--gaussians", "100000"]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except BaseException as e:
    print("EXC", type(e).__name__, str(e)[:300])
torch.cuda.synchronize()
print("calls", calls[0], ops.graph_cache_stats())
