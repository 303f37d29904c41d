"""CPU, world_size 2 over gloo: the N>1 path of SURVEY.md section 8(e) -- views sharded across
ranks, ONE sum all-reduce of the flattened six-tensor gradient bucket.  (The rasterizer itself has
no CPU path, so per-rank gradients are synthetic functions of the rank's views.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from collab_splats_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shapes(n):
    return {"means": (n, 3), "scales": (n, 3), "quats": (n, 4), "opacities": (n, 1), "features_dc": (n, 3),
            "features_rest": (n, 15, 3)}


def _view_grad(view, shape):
    g = torch.Generator().manual_seed(1000 + view)
    return torch.randn(shape, generator=g)


def _worker(rank, world, port, n, n_views, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    params = [torch.zeros(s, requires_grad=True) for s in _shapes(n).values()]
    mine = parallel.shard_views(n_views, rank, world)
    for p in params:
        p.grad = sum((_view_grad(v, p.shape) for v in mine), torch.zeros(p.shape))
    flat = parallel.allreduce_gradients(params)
    assert flat.numel() == 59 * n                                      # 236 B per Gaussian
    expect = [sum((_view_grad(v, p.shape) for v in range(n_views)), torch.zeros(p.shape)) for p in params]
    ok = all(torch.allclose(p.grad, e, atol=1e-5) for p, e in zip(params, expect))
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok, mine, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_view_sharded_gradient_allreduce_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 257, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert sorted(res[0][2] + res[1][2]) == list(range(5))
    assert all(r[3] == 2.0 for r in res)


def test_allreduce_is_identity_without_process_group():
    params = [torch.zeros(4, 3, requires_grad=True), torch.zeros(4, 1, requires_grad=True)]
    params[0].grad = torch.ones(4, 3)
    flat = parallel.allreduce_gradients(params)
    assert flat.numel() == 16 and torch.equal(params[0].grad, torch.ones(4, 3)) and not params[1].grad.any()
