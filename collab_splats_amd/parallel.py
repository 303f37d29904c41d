"""Multi-GPU: one process per GPU, camera views sharded across ranks.

The reference has no distributed code at all; north_star adds exactly one
pattern: every rank holds the same N Gaussians, renders different views, and the six parameter
gradients (59 floats = 236 B per Gaussian at SH degree 3) are summed with ONE all-reduce over
RCCL/xGMI.  Independent views (BASELINE config 4) need no collective.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist

GRAD_KEYS = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")


def init_distributed(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; 'nccl' IS RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("MISPLAT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_views(n_views: int, rank: int, world: int) -> List[int]:
    """Views owned by ``rank``: contiguous blocks, remainder to the low ranks."""
    base, rem = divmod(n_views, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def flatten_grads(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """One contiguous fp32 bucket (a single large collective suits point-to-point xGMI links
    better than six small ones)."""
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])


def unflatten_into_grads(flat: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    o = 0
    for p in params:
        n = p.numel()
        p.grad = flat[o:o + n].view_as(p).clone() if p.grad is None else p.grad.copy_(flat[o:o + n].view_as(p))
        o += n


def _reduce_device(t: torch.Tensor) -> torch.Tensor:
    """gloo (CPU rehearsal of the multi-rank path) cannot reduce device tensors: stage through host."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu()
    return t


def allreduce_gradients(params: Sequence[torch.Tensor], average: bool = False) -> torch.Tensor:
    """Sum (or mean) the gradients of the shared Gaussians over all ranks, in place."""
    flat = flatten_grads(params)
    if dist.is_initialized() and dist.get_world_size() > 1:
        buf = _reduce_device(flat)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if buf is not flat:
            flat.copy_(buf)
        if average:
            flat /= dist.get_world_size()
    unflatten_into_grads(flat, params)
    return flat


def max_over_ranks(value: float, device: torch.device) -> float:
    """bench.py's max-over-ranks of the timed region."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
