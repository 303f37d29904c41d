#!/usr/bin/env python3
"""bench.py -- Msplats/s of the RaDe-GS rasterizer hot path (fwd+bwd) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
          --master-port P bench.py --gpus N --steps K --warmup W)

A "step" is one pass of the hot path over one view of the synthetic scene: activations
(exp/sigmoid, rade_gs_model.py:443-444) -> rasterization(..., return_depth_normal=True) -> backward
of all five outputs to the six parameter tensors.  Workload (BASELINE.json configs[2], the one the
metric is quoted on): 1 M random Gaussians, 1920x1080, SH degree 3, RGB+ED, antialiased.  With N
GPUs every rank renders its own view of its own replica (configs[3]: independent views, no
collective, weak scaling); --shared-grads adds the RCCL all-reduce of the 236 B/Gaussian
gradients (configs[4] pattern).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (timed live with events on the launch stream) and `cpu_baseline` (the C port
under oracle/, bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--render-mode", default="RGB+ED")
    ap.add_argument("--rasterize-mode", default="antialiased")
    ap.add_argument("--shared-grads", action="store_true", help="all-reduce the Gaussian gradients over RCCL")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="Gaussians in the CPU-baseline sample")
    return ap.parse_args()


def algorithmic_bytes(kernel: str, N: int, I: int, P: int) -> float:
    """SURVEY.md section 8(d) per-unit figures (fp32, SH3, D=3)."""
    if kernel == "blend_bwd":
        return 88.0 * P + 64.0 * I + 60.0 * N     # saved outputs + upstream grads, re-gather, 2-D grads
    if kernel == "blend_fwd":
        return 64.0 * I + 48.0 * P                 # record gather, pixel outputs incl. saved indices
    if kernel == "slab_reduce":
        return 64.0 * I + 64.0 * N                 # one gradient row per intersection in, one per Gaussian out
    raise KeyError(kernel)


def cpu_baseline(args, seed: int):
    """C port (oracle/craster.c, OpenMP) on a bounded sample of the same workload."""
    import numpy as np
    from oracle.craster import CRaster
    from collab_splats_amd.synthetic import random_scene
    n = min(args.cpu_sample, args.gaussians)
    sc = random_scene(n, args.width, args.height, seed=seed)
    cr = CRaster(np.float32)
    scales = torch.exp(sc["log_scales"]).numpy()
    op = torch.sigmoid(sc["opacity_logits"]).numpy()
    g = torch.Generator().manual_seed(7)
    H, W = args.height, args.width
    cd = 4 if args.render_mode == "RGB+ED" else 3
    ups = [torch.rand(s, generator=g).numpy() for s in ((H, W, cd), (H, W, 1), (H, W, 1), (H, W, 1), (H, W, 3))]

    ref = {}

    def one():
        st = cr.forward(sc["means"].numpy(), sc["quats"].numpy(), scales, op, sc["sh"].numpy(),
                        sc["viewmats"][0].numpy(), sc["Ks"][0].numpy(), W, H, sh_degree=3,
                        render_mode=args.render_mode, rasterize_mode=args.rasterize_mode)
        ref["grads"] = cr.backward(st, *ups)
        ref["render"] = st["render"]
        return st["bins"]["n_isects"]

    one()  # warm-up (page-in, thread pool)
    t0 = time.perf_counter()
    reps = 0
    while True:
        isects = one()
        reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps
    base = {"value": round(n / dt / 1e6, 4), "unit": "Msplats/s", "cores": cr.threads, "kind": "port",
            "sample": f"{n} Gaussians (same generator, seed {seed}), {W}x{H}, {args.render_mode} fwd+bwd, "
                      f"{reps} timed iteration(s) after 1 warm-up, {isects} intersections, C/OpenMP fp32"}
    return base, parity_on_sample(args, sc, scales, op, ups, ref)


def parity_on_sample(args, sc, scales, op, ups, ref):
    """The second half of the metric ("grad max-rel-err vs reference"): HIP gradients against the C port
    (the checker; the upstream CUDA reference is absent -- parity unpinned) on the
    cpu_baseline sample.  tensor-inf-norm relative error  max|g - g_ref| / max|g_ref|  per parameter."""
    import numpy as np
    from collab_splats_amd.rendering import rasterization
    dev = torch.device("cuda", torch.cuda.current_device())
    leaves = [sc["means"].to(dev).requires_grad_(True), sc["quats"].to(dev).requires_grad_(True),
              torch.from_numpy(scales).to(dev).requires_grad_(True), torch.from_numpy(op).to(dev).requires_grad_(True),
              sc["sh"].to(dev).requires_grad_(True)]
    out = rasterization(*leaves, sc["viewmats"].to(dev), sc["Ks"].to(dev), args.width, args.height, sh_degree=3,
                        render_mode=args.render_mode, rasterize_mode=args.rasterize_mode, return_depth_normal=True)
    torch.autograd.backward(list(out[:5]), [torch.from_numpy(u)[None].to(dev) for u in ups])

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

    names = ("v_means", "v_quats", "v_scales", "v_opacities", "v_colors")
    errs = {k[2:]: rel(l.grad.cpu().numpy(), ref["grads"][k]) for k, l in zip(names, leaves)}
    return {"value": max(errs.values()), "per_tensor": {k: float(f"{v:.3e}") for k, v in errs.items()},
            "against": "oracle/craster.c (fp32 C port; upstream gsplat-rade absent: parity unpinned)",
            "target": 1e-4}


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` without a torchrun environment: start the N ranks as CHILD processes (nothing in
    this process has touched the GPU yet) and relay their output; rank 0 of the children prints the JSON line."""
    import subprocess
    n_dev = torch.cuda.device_count()               # does not initialise the GPU
    if n_dev < args.gpus and os.environ.get("MISPLAT_OVERSUBSCRIBE", "0") != "1":
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible "
              f"(MISPLAT_OVERSUBSCRIBE=1 rehearses N ranks on fewer devices)", file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    from collab_splats_amd import parallel
    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local = local % torch.cuda.device_count()      # (a 1-GPU rehearsal of N > 1 puts every rank on cuda:0)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from collab_splats_amd import ops
    from collab_splats_amd.rendering import rasterization
    from collab_splats_amd.synthetic import random_scene, view_matrix

    N, W, H = args.gaussians, args.width, args.height
    seed = 42 if args.shared_grads else 42 + rank           # shared Gaussians vs independent scenes
    sc = random_scene(N, W, H, seed=seed)
    params = {k: sc[k].to(dev).requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
    viewmats = (view_matrix(rank) if world > 1 else sc["viewmats"]).to(dev)
    Ks = sc["Ks"].to(dev)
    cd = {"RGB": 3, "RGB+ED": 4, "RGB+D": 4}[args.render_mode]
    g = torch.Generator().manual_seed(7)
    ups = [torch.rand(s, generator=g).to(dev) for s in ((1, H, W, cd), (1, H, W, 1), (1, H, W, 1), (1, H, W, 1), (1, H, W, 3))]
    plist = list(params.values())
    info = {}

    def step():
        for p in plist:
            p.grad = None
        out = rasterization(params["means"], params["quats"], torch.exp(params["log_scales"]),
                            torch.sigmoid(params["opacity_logits"]), params["sh"], viewmats, Ks, W, H,
                            near_plane=0.01, far_plane=1e10, sh_degree=3, packed=False,
                            render_mode=args.render_mode, sparse_grad=False, absgrad=False,
                            rasterize_mode=args.rasterize_mode, return_depth_normal=True)
        torch.autograd.backward(list(out[:5]), ups)
        if args.shared_grads and world > 1:
            parallel.allreduce_gradients(plist)
        info["n_isects"] = out[5]["n_isects"]
        info["n_visible"] = out[5]["radii"]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ops.KERNEL_EVENTS = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    dt = parallel.max_over_ranks(dt, dev)

    if rank == 0:
        I = int(info["n_isects"])
        n_vis = int((info["n_visible"] > 0).any(-1).sum().item())
        ktimes = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in events.items() if v}  # ms
        dom = max((k for k in ktimes if k in ("blend_bwd", "blend_fwd", "slab_reduce")), key=ktimes.get)
        abytes = algorithmic_bytes(dom, N, I, W * H)
        achieved = abytes / (ktimes[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and N == 1_000_000 and (W, H) == (1920, 1080):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        ms = dt / args.steps * 1e3
        line = {
            "metric": "Msplats/s fwd+bwd @1080p (1M Gaussians); grad max-rel-err vs reference",
            "value": round(world * N / (dt / args.steps) / 1e6, 3), "unit": "Msplats/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{N} random Gaussians, 1 view {W}x{H} per GPU, SH degree 3, "
                                   f"{args.render_mode} ({args.rasterize_mode}), colour+alpha+expected/median "
                                   f"depth+normal, fwd+bwd to all six parameter tensors",
                       "views": world, "n_isects": I, "n_visible": n_vis,
                       "parallelism": ("independent views, no collective" if not args.shared_grads
                                       else "shared Gaussians, RCCL all-reduce of 236 B/Gaussian grads")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
                         "algorithmic_bytes": abytes,
                         "note": "blend kernels are VALU (v_exp/FMA) bound, not HBM bound: see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["grad_max_rel_err"] = cpu_baseline(args, seed)
            line["grad_max_rel_err"]["value"] = float(f"{line['grad_max_rel_err']['value']:.3e}")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
