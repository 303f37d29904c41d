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
    """Views sharded over the ranks; every rank plays the rasterizer's backward against a GradientBuckets sink: the colour
    gradients are written IN PLACE through ``sink()`` and their bucket is launched from ``rasterizer_done()``, the geometry
    gradients arrive through autograd (fresh tensors) and travel from ``allreduce()``."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    params = [torch.zeros(s, requires_grad=True) for s in _shapes(n).values()]
    mine = parallel.shard_views(n_views, rank, world)
    bk = parallel.GradientBuckets(params)
    assert bk.colour == [4, 5] and bk.geometry == [0, 1, 2, 3]
    assert bk.flat.numel() >= 59 * n and all(v.data_ptr() % 16 == 0 for v in bk.views)     # 236 B per Gaussian, aligned slices
    bk.attach()
    local = [sum((_view_grad(v, p.shape) for v in mine), torch.zeros(p.shape)) for p in params]
    for i in bk.colour:                                                # "the kernels" write the colour bucket in place
        out = bk.sink(params[i])
        assert out is not None and out.data_ptr() == bk.views[i].data_ptr()
        out.copy_(local[i])
        params[i].grad = out
    bk.rasterizer_done()
    early = len(bk._work)                                              # the colour bucket is already travelling
    try:                                                               # ... so a second node on the same parameters must fail loudly
        bk.sink(params[4])
        early = -1
    except Exception as e:
        assert "views_per_backward" in str(e)
    for i in bk.geometry:                                              # autograd's own tensors: copied in by allreduce()
        params[i].grad = local[i].clone()
    bk.allreduce()
    expect = [sum((_view_grad(v, p.shape) for v in range(n_views)), torch.zeros(p.shape)) for p in params]
    ok = all(torch.allclose(p.grad, e, atol=1e-5) for p, e in zip(params, expect)) and early == 1
    ok &= all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, bk.views))
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


def test_gradient_buckets_without_process_group_and_two_views_in_one_backward():
    """World size 1: no collective, ``p.grad`` ends up as the slices.  A slice is handed out ONCE per ``attach()``: the
    second rasterization node of one ``backward()`` (two views per step) gets None -- an ordinary tensor that autograd adds
    in place -- so the sum of both views arrives, not twice the second one."""
    params = [torch.zeros(4, 3, requires_grad=True), torch.zeros(4, 1, requires_grad=True),
              torch.zeros(4, 15, 3, requires_grad=True)]
    bk = parallel.GradientBuckets(params, geometry=[0, 1], colour=[2])
    bk.attach(views_per_backward=2)
    g1, g2 = torch.full((4, 15, 3), 1.0), torch.full((4, 15, 3), 10.0)
    first = bk.sink(params[2])
    assert first is not None and first.data_ptr() == bk.views[2].data_ptr()
    first.copy_(g1)
    params[2].grad = first                                             # AccumulateGrad adopts the first node's output
    bk.rasterizer_done()
    assert bk.sink(params[2]) is None                                  # second node: no slice
    params[2].grad += g2                                               # ... autograd accumulates in place
    bk.rasterizer_done()
    params[0].grad = torch.ones(4, 3)
    bk.allreduce()
    assert torch.equal(params[2].grad, g1 + g2) and params[2].grad.data_ptr() == bk.views[2].data_ptr()
    assert torch.equal(params[0].grad, torch.ones(4, 3)) and not params[1].grad.any()
    assert bk.sink(torch.zeros(4, 3)) is None                          # not a parameter


def _bucket_worker(rank, world, port, n, q):
    """GradientBuckets over gloo: two buckets, gradients end up as views of ONE flat buffer; then a seeded refinement
    step driven by the all-reduced statistics must leave both replicas bit-identical (lock-step densification)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_distributed(backend="gloo")
    from collab_splats_amd.strategy import DefaultStrategy
    g0 = torch.Generator().manual_seed(3)                            # identical replicas on every rank
    shapes = _shapes(n)
    params = {k: torch.nn.Parameter(torch.randn(s, generator=g0) * (0.01 if k == "scales" else 1.0)) for k, s in shapes.items()}
    with torch.no_grad():
        params["scales"].copy_(torch.log(torch.rand(n, 3, generator=g0) * 0.03 + 0.002))
    leaves = [params[k] for k in parallel.GRAD_KEYS]
    bk = parallel.GradientBuckets(leaves)
    bk.attach()
    for i, p in enumerate(leaves):                                   # rank-dependent local gradients
        p.grad = _view_grad(10 * rank + i, p.shape)
    bk.allreduce()
    expect = [sum((_view_grad(10 * r + i, p.shape) for r in range(world)), torch.zeros(p.shape)) for i, p in enumerate(leaves)]
    ok = all(torch.allclose(p.grad, e, atol=1e-5) for p, e in zip(leaves, expect))
    ok &= all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(leaves, bk.views))      # zero-copy: grads ARE the buffer
    ok &= bk.flat.numel() == 59 * n                                       # (n = 64: no padding between the slices)
    # lock-step refinement: per-rank screen-space gradients are summed, every rank refines with the same seeded noise
    optimizers = {k: torch.optim.Adam([v], lr=1e-3) for k, v in params.items()}
    for k, v in params.items():
        optimizers[k].step()
    local2d = _view_grad(100 + rank, (1, n, 2)) * 3.0
    dist.all_reduce(local2d, op=dist.ReduceOp.SUM)
    m2d = torch.zeros(1, n, 2, requires_grad=True)
    m2d.grad = local2d
    info = {"means2d": m2d, "radii": torch.ones(1, n, 2, dtype=torch.int32), "width": 2, "height": 2, "n_cameras": 1}
    s = DefaultStrategy(refine_start_iter=0, refine_every=10, grow_grad2d=0.5, seed=11)
    st = s.initialize_state()
    counts = s.step_post_backward(params, optimizers, st, 10, info)
    digest = torch.cat([v.detach().reshape(-1) for v in params.values()]
                       + [optimizers[k].state[params[k]]["exp_avg"].reshape(-1) for k in params])
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    q.put((rank, bool(ok), bool(same), tuple(counts), int(params["means"].shape[0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradient_buckets_and_lockstep_densification_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "bucketed all-reduce wrong or not zero-copy"
    assert all(r[2] for r in res), "replicas diverged after the seeded refinement step"
    assert res[0][3] == res[1][3] and sum(res[0][3][:2]) > 0        # something was actually duplicated / split
    assert res[0][4] == res[1][4] != 64


def _sparse_worker(rank, world, port, n, case, mode, q):
    """The rasterizer's backward played against a GradientBuckets sink, with row flags: every rank touches a subset of the
    rows (``case``: disjoint / overlapping / nearly all), writes gradients into those rows of ALL six slices in place and
    leaves the others zero -- as the kernels do.  The sparse reduce (bitmaps all-gathered and OR-ed, the union's rows packed,
    reduced, scattered back) must equal the dense all-reduce of the same buffers BIT FOR BIT."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_distributed(backend="gloo")
    parallel.ALLREDUCE = mode
    g = torch.Generator().manual_seed(50 + rank)
    rows = {"disjoint": torch.arange(n)[rank::world][: n // 8],
            "overlap": torch.randperm(n, generator=torch.Generator().manual_seed(7))[: n // 5][rank * 3:],
            "most": torch.randperm(n, generator=g)[: (9 * n) // 10]}[case]
    touched = torch.zeros(n, dtype=torch.uint8)
    touched[rows] = 1
    results = []
    for sparse_env in ("0", "auto"):
        parallel.SPARSE = sparse_env
        for k in parallel.STATS:
            parallel.STATS[k] = 0
        params = [torch.zeros(s, requires_grad=True) for s in _shapes(n).values()]
        bk = parallel.GradientBuckets(params)
        bk.attach()
        for i, p in enumerate(params):                                # "the kernels": in place, touched rows only
            out = bk.sink(p)
            out.zero_()
            out[rows] = _view_grad(100 * rank + i, p.shape)[rows]
            p.grad = out
        bk.rasterizer_done(touched)
        early = len(bk._work)
        bk.allreduce()
        results.append(([p.grad.clone() for p in params], dict(parallel.STATS), early))
    (dense, st_d, _), (sparse, st_s, early) = results
    same = all(torch.equal(a, b) for a, b in zip(dense, sparse))
    untouched_zero = True
    union = torch.zeros(n, dtype=torch.uint8)
    union[rows] = 1
    dist.all_reduce(union, op=dist.ReduceOp.MAX)
    for t in sparse:
        untouched_zero &= not bool(t[union == 0].any())
    q.put((rank, same, untouched_zero, st_d, st_s, early, int(union.sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("case,mode", [("disjoint", "auto"), ("overlap", "auto"), ("most", "auto"), ("overlap", "rs_ag")])
def test_sparse_row_reduce_equals_dense_bit_for_bit_gloo_ws2(case, mode):
    n = 1003                                                             # (not a multiple of 8: the bitmap's last byte is partial)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, 2, port, n, case, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, same, untouched_zero, st_d, st_s, early, n_union in res:
        assert same, f"rank {rank}: sparse and dense sums differ ({case}, {mode})"
        assert untouched_zero and early == 1                              # the colour bucket left from rasterizer_done()
        assert st_d["dense"] == 2 and st_d["sparse"] == 0
        if case == "most":                                                # union above 35 % of the rows: the dense buffer is cheaper
            assert st_s["sparse"] == 0 and st_s["dense"] == 2
        else:
            assert st_s["sparse"] == 2 and st_s["dense"] == 0 and st_s["rows_reduced"] == 2 * n_union < 2 * 0.35 * n


def test_colour_gradient_added_after_the_early_launch_is_refused(monkeypatch):
    """the colour bucket's all-reduce starts from inside backward(); if anything else adds a colour gradient later
    in the same backward (an SH regulariser), the sum that travelled does not contain it.  The colour slices have a version
    counter of their own: ``allreduce()`` raises instead of returning a silently wrong gradient; ``early_colour=False`` is
    the way to run such a step."""
    from collab_splats_amd._lib import MisplatError
    monkeypatch.setattr(parallel, "_world", lambda: 2)                    # (no process group: _launch is stubbed below)
    monkeypatch.setattr(parallel, "_collective", lambda: True)
    params = [torch.zeros(6, 3, requires_grad=True), torch.zeros(6, 15, 3, requires_grad=True)]
    launched = []
    for early_colour in (True, False):
        bk = parallel.GradientBuckets(params, geometry=[0], colour=[1])
        monkeypatch.setattr(bk, "_launch", lambda upto, sparse=True, bk=bk: (launched.append(upto), setattr(bk, "_reduced", upto)))
        bk.attach(early_colour=early_colour)
        out = bk.sink(params[1])
        out.fill_(1.0)
        params[1].grad = out
        bk.rasterizer_done()
        assert bool(bk._reduced) == early_colour
        params[1].grad += 2.0                                             # autograd accumulates a second colour gradient in place
        params[0].grad = torch.ones(6, 3)
        if early_colour:
            with pytest.raises(MisplatError, match="early_colour=False"):
                bk.allreduce()
        else:
            bk.allreduce()
            assert torch.equal(params[1].grad, torch.full((6, 15, 3), 3.0))


def _regulariser_worker(rank, world, port, n, q):
    """a regulariser node that runs AFTER the rasterizer's in one backward().  The rasterizer's gradient has become
    ``p.grad`` (its slice of the buffer, written in place, touched rows only); the regulariser's gradient -- every row, and
    different on every rank -- is then accumulated IN PLACE at the same address.  The geometry bucket must notice (its slices
    have a version counter of their own) and travel densely: the sparse row reduce would leave the rows outside the flags
    as each rank has them, and the replicas would diverge."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_distributed(backend="gloo")
    parallel.SPARSE = "auto"
    for k in parallel.STATS:
        parallel.STATS[k] = 0
    rows = torch.arange(n)[rank::world][: n // 8]
    touched = torch.zeros(n, dtype=torch.uint8)
    touched[rows] = 1
    params = [torch.zeros(s, requires_grad=True) for s in _shapes(n).values()]
    bk = parallel.GradientBuckets(params)
    bk.attach()
    for i, p in enumerate(params):
        out = bk.sink(p)
        out.zero_()
        out[rows] = _view_grad(100 * rank + i, p.shape)[rows]
        p.grad = out
    bk.rasterizer_done(touched)
    reg = _view_grad(900 + rank, params[1].shape)                     # scale regulariser: all rows, rank-dependent
    params[1].grad += reg                                             # (what AccumulateGrad does with a stolen gradient)
    bk.allreduce()
    expect = torch.zeros(params[1].shape)
    for r in range(world):
        rr = torch.arange(n)[r::world][: n // 8]
        expect[rr] += _view_grad(100 * r + 1, params[1].shape)[rr]
        expect += _view_grad(900 + r, params[1].shape)
    digest = torch.cat([p.grad.reshape(-1) for p in params])
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    q.put((rank, bool(torch.allclose(params[1].grad, expect, atol=1e-5)), all(torch.equal(gathered[0], t) for t in gathered),
           dict(parallel.STATS)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_regulariser_behind_the_rasterizer_sends_the_geometry_bucket_densely_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_regulariser_worker, args=(r, 2, port, 1003, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, right, same, st in res:
        assert right, f"rank {rank}: the regulariser's rows were not reduced"
        assert same, "replicas diverged"
        assert st["geometry_touched_late"] == 1 and st["sparse"] == 1 and st["dense"] == 1, st   # colour sparse, geometry dense
