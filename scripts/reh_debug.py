import os, sys, time, collections, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collab_splats_amd import parallel, ops, radegs
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t
            acc[name][0] += dt; acc[name][1] += 1
            if dt > 5e-3:
                print(f"SPIKE {name} {dt*1e3:.1f} ms at call {acc[name][1]}", file=sys.stderr)
    setattr(obj, name, g)
for n in ("allreduce", "_union_rows", "_launch", "_finish", "_settle_count", "rasterizer_done", "attach", "_rows_move"):
    wrap(parallel.GradientBuckets, n)
wrap(radegs.RadegsModel, "get_outputs"); wrap(radegs.RadegsModel, "get_loss_dict")
wrap(ops, "_wait_count"); wrap(ops, "_phase_b_launch"); wrap(ops, "_raster_phase_a"); wrap(ops, "_phase_b_prepare")
from collab_splats_amd import rendering, ops_epilogue
wrap(rendering, "rasterization"); wrap(ops_epilogue, "get_outputs_epilogue"); wrap(radegs, "camera_parameters")
import torch
orig_bw = torch.Tensor.backward
def bw(self, *a, **k):
    t = time.perf_counter()
    r = orig_bw(self, *a, **k)
    acc["backward()"][0] += time.perf_counter() - t; acc["backward()"][1] += 1
    return r
torch.Tensor.backward = bw
import ctypes as C
from collab_splats_amd import _lib
lib = _lib.load()
for n in ("misplat_touched_bits", "misplat_union_count", "misplat_union_ids", "misplat_rows_pack", "misplat_rows_unpack", "misplat_raster_bwd", "misplat_raster_fwd"):
    wrap(lib, n)
for n in ("cumsum", "zeros"):
    wrap(torch, n)
import traceback
_orig_empty = torch.empty
slow = []
def _empty(*a, **k):
    t = time.perf_counter()
    r = _orig_empty(*a, **k)
    dt = time.perf_counter() - t
    acc["empty"][0] += dt; acc["empty"][1] += 1
    if dt > 2e-4:
        slow.append((round(dt * 1e6), r.numel() * r.element_size(), str(r.device), "".join(traceback.format_stack(limit=4)[:-1])[-500:]))
    return r
torch.empty = _empty
import atexit
atexit.register(lambda: [print("SLOW", x[0], "us", x[1], "bytes", x[2], "\n", x[3], file=sys.stderr) for x in slow[-6:]] + [print("n slow", len(slow), file=sys.stderr)])
wrap(torch.Tensor, "copy_")
wrap(torch.cuda.Event, "record")
sys.argv = ["bench.py"] + sys.argv[1:]
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    print("TOTAL per step of the top-level pieces (us):", {k: round(acc[k][0] / max(acc[k][1], 1) * 1e6) for k in ("get_outputs", "get_loss_dict", "backward()", "allreduce", "attach")}, file=sys.stderr)
    for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:20s} {t / max(n, 1) * 1e6:9.1f} us x {n}", file=sys.stderr)
