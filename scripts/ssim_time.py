"""Times misplat_ssim_fwd / misplat_ssim_bwd at 1080p (HIP events around 20 calls each) and checks them against the torch
conv2d expressions in fp32 on the device.   python scripts/ssim_time.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collab_splats_amd import _lib, ops
from collab_splats_amd.ops import ptr, stream_ptr

H, W = 1080, 1920
dev = "cuda"
g = torch.Generator().manual_seed(1)
gt = torch.rand(H, W, 3, generator=g).to(dev)
rgb = (gt + 0.1 * torch.randn(H, W, 3, generator=g).to(dev)).clamp(0, 1).contiguous()
lib = _lib.load()
n = int(lib.misplat_ssim_scratch_floats(C.c_int32(H), C.c_int32(W)))
scratch = torch.empty(n, device=dev)
l1 = (gt - rgb).abs().mean()
main = torch.empty((), device=dev)
gmain = torch.ones((), device=dev)
v = torch.empty_like(rgb)


def fwd():
    ops.check(lib.misplat_ssim_fwd(C.c_int32(H), C.c_int32(W), ptr(rgb), ptr(gt), ptr(scratch), ptr(l1), C.c_float(0.2), None,
                                   ptr(main), stream_ptr()), "fwd")


def bwd():
    ops.check(lib.misplat_ssim_bwd(C.c_int32(H), C.c_int32(W), ptr(rgb), ptr(gt), ptr(scratch), ptr(gmain), C.c_float(0.2), ptr(v),
                                   stream_ptr()), "bwd")


for f, name in ((fwd, "ssim_fwd (+ final)"), (bwd, "ssim_bwd")):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        f()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 20 * 1e3:.1f} us", flush=True)
