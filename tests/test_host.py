"""CPU: host-side logic of the model-path mirror (camera parameters, normals, strategy, synthetic
scene generator, view sharding) against the reference-generated goldens."""
import os

import numpy as np
import pytest
import torch

from collab_splats_amd import parallel, radegs, synthetic
from collab_splats_amd.strategy import DefaultStrategy

GOLD = os.path.join(os.path.dirname(__file__), "golden", "camera_goldens.npz")


@pytest.mark.parametrize("i", [0, 1, 2])
def test_camera_parameters_match_reference(i):
    g = np.load(GOLD)
    W, H = [int(v) for v in g[f"cam{i}_WH"]]
    K = g[f"cam{i}_K"]
    cam = radegs.PinholeCamera.make(torch.from_numpy(g[f"cam{i}_c2w"]), K[0, 0], K[1, 1], W, H, cx=K[0, 2], cy=K[1, 2])
    cp = radegs.camera_parameters(cam)
    assert cp["viewmats"].shape == (1, 4, 4) and cp["Ks"].shape == (1, 3, 3)
    assert np.abs(cp["viewmats"][0].numpy() - g[f"cam{i}_viewmat"]).max() < 1e-5
    assert np.abs(cp["camera_center"].numpy() - g[f"cam{i}_center"]).max() < 1e-5
    fovx, fovy = g[f"cam{i}_fov"]
    assert cp["Ks"][0, 0, 0].item() == pytest.approx(W / (2 * np.tan(fovx / 2)), rel=1e-6)
    assert cp["Ks"][0, 1, 1].item() == pytest.approx(H / (2 * np.tan(fovy / 2)), rel=1e-6)
    assert cp["Ks"][0, 0, 2].item() == W / 2 and cp["Ks"][0, 1, 2].item() == H / 2     # rade_gs_model.py:327-334
    assert (cp["image_width"], cp["image_height"]) == (W, H)


def test_build_rotation_and_normals_match_reference():
    g = np.load(GOLD)
    q = torch.from_numpy(g["rot_q"])
    assert np.abs(radegs.build_rotation(q).numpy() - g["rot_R"]).max() < 1e-6
    n = q.shape[0]
    scales = torch.log(torch.tensor([[0.3, 0.01, 0.2]]).repeat(n, 1))
    m = radegs.RadegsModel(radegs.RadegsModelConfig(), torch.zeros(n, 3), scales, q, torch.zeros(n, 1),
                           torch.zeros(n, 3), torch.zeros(n, 15, 3))
    assert np.abs(m.normals.detach().numpy() - g["rot_R"][:, :, 1]).max() < 1e-5   # column of the thin axis
    assert sorted(k for k, _ in m.gauss_params.items()) == sorted(parallel.GRAD_KEYS)


def test_get_outputs_rejects_non_camera(capsys):
    m = radegs.RadegsModel(radegs.RadegsModelConfig(), torch.zeros(1, 3), torch.zeros(1, 3), torch.ones(1, 4),
                           torch.zeros(1, 1), torch.zeros(1, 3), torch.zeros(1, 15, 3))
    assert m.get_outputs("not a camera") == {}                       # rade_gs_model.py:90-92
    assert "not a camera" in capsys.readouterr().out


def _toy_training_state(n=12, seed=0):
    g = torch.Generator().manual_seed(seed)
    params = torch.nn.ParameterDict({
        "means": torch.nn.Parameter(torch.randn(n, 3, generator=g)),
        "scales": torch.nn.Parameter(torch.log(torch.full((n, 3), 0.005))),
        "quats": torch.nn.Parameter(torch.randn(n, 4, generator=g)),
        "opacities": torch.nn.Parameter(torch.full((n, 1), 2.0)),
        "features_dc": torch.nn.Parameter(torch.rand(n, 3, generator=g)),
        "features_rest": torch.nn.Parameter(torch.rand(n, 15, 3, generator=g)),
    })
    optimizers = {k: torch.optim.Adam([v], lr=1e-3) for k, v in params.items()}
    for k, v in params.items():                       # one step so that every optimizer has moments
        v.grad = torch.ones_like(v)
        optimizers[k].step()
    return params, optimizers


def test_strategy_surface_and_statistics():
    s = DefaultStrategy(absgrad=True)
    st = s.initialize_state()
    params, _ = _toy_training_state(4)
    x = torch.zeros(1, 4, 2, requires_grad=True)
    m2d = x * 2.0
    radii = torch.ones(1, 4, 2, dtype=torch.int32)
    radii[0, 3] = 0                                    # culled Gaussian: no statistics
    info = {"means2d": m2d, "radii": radii, "width": 8, "height": 4, "n_cameras": 1}
    s.step_pre_backward(params, {}, st, 0, info)
    m2d.sum().backward()
    assert m2d.grad is not None                        # retained on the non-leaf (rade_gs_model.py:191-198)
    m2d.absgrad = torch.ones(1, 4, 2)
    assert s.step_post_backward(params, {}, st, 1, info) == (0, 0, 0)
    expect = float(np.hypot(8 / 2.0, 4 / 2.0))         # |grad| in normalised-device units
    assert torch.allclose(st["grad2d"], torch.tensor([expect, expect, expect, 0.0]))
    assert st["count"].tolist() == [1, 1, 1, 0]
    with pytest.raises(AssertionError):
        s.step_pre_backward(params, {}, st, 0, {})
    with pytest.raises(NotImplementedError):
        s.step_post_backward(params, {}, st, 1, info, packed=True)
    assert s.step_post_backward(params, {}, st, s.refine_stop_iter, info) == (0, 0, 0)


def test_strategy_densification_duplicate_split_prune_reset():
    """Adaptive density control (3DGS section 5) on a toy state: who is duplicated / split / pruned, and that
    parameters, optimizer moments and statistics stay aligned."""
    n = 12
    params, optimizers = _toy_training_state(n)
    s = DefaultStrategy(refine_start_iter=0, refine_every=10, reset_every=3000, grow_grad2d=0.5, grow_scale3d=0.01,
                        prune_opa=0.005)
    st = s.initialize_state(scene_scale=1.0)
    with torch.no_grad():
        params["scales"][4:8] = float(np.log(0.05))           # 4..7 are "large"
        params["opacities"][10:12] = -9.0                     # 10, 11 are transparent
    grad = torch.zeros(1, n, 2)
    grad[0, [0, 1, 4, 5], 0] = 1.0                            # 0,1 (small) and 4,5 (large) have high 2-D gradients
    m2d = torch.zeros(1, n, 2, requires_grad=True)
    m2d.grad = grad
    info = {"means2d": m2d, "radii": torch.ones(1, n, 2, dtype=torch.int32), "width": 2, "height": 2, "n_cameras": 1}
    old_means = params["means"].detach().clone()
    n_dup, n_split, n_prune = s.step_post_backward(params, optimizers, st, 10, info)
    assert (n_dup, n_split, n_prune) == (2, 2, 2)
    n_new = n + 2 + 2 - 2                                     # +2 copies, +2 (split: 2 in, 4 out), -2 pruned
    for k, v in params.items():
        assert v.shape[0] == n_new, k
        opt = optimizers[k]
        assert opt.param_groups[0]["params"][0] is v          # the optimizer follows the new tensor
        assert opt.state[v]["exp_avg"].shape == v.shape and opt.state[v]["exp_avg_sq"].shape == v.shape
    assert st["grad2d"].shape == (n_new,) and not st["grad2d"].any() and not st["count"].any()
    # survivors keep their moments, new Gaussians start from zero moments
    ea = optimizers["means"].state[params["means"]]["exp_avg"]
    assert ea[:6].abs().min() > 0 and not ea[-4:].any()
    # the split children: scales / 1.6, means scattered around the parent within a few sigma
    assert torch.allclose(torch.exp(params["scales"][-4:]), torch.full((4, 3), 0.05 / 1.6))
    parents = old_means[[4, 5]].repeat(2, 1)
    assert (params["means"][-4:] - parents).abs().max() < 0.05 * 6
    assert torch.sigmoid(params["opacities"]).min() > 0.005   # transparent ones are gone
    # opacity reset at reset_every: logits clamped to logit(2 * prune_opa), moments forgotten
    s2 = DefaultStrategy(refine_start_iter=10 ** 6, reset_every=20, prune_opa=0.005)
    m2d2 = torch.zeros(1, n_new, 2, requires_grad=True)
    m2d2.grad = torch.zeros(1, n_new, 2)
    info2 = {"means2d": m2d2, "radii": torch.ones(1, n_new, 2, dtype=torch.int32), "width": 2, "height": 2, "n_cameras": 1}
    s2.step_post_backward(params, optimizers, s2.initialize_state(), 20, info2)
    cap = np.log(0.01 / 0.99)
    assert params["opacities"].max().item() <= cap + 1e-6
    assert not optimizers["opacities"].state[params["opacities"]]["exp_avg"].any()
    # a training step still works after the surgery
    for k, v in params.items():
        v.grad = torch.ones_like(v)
        optimizers[k].step()


def test_synthetic_scene_is_deterministic_and_in_spec():
    a, b = synthetic.random_scene(1000, 1920, 1080, seed=42), synthetic.random_scene(1000, 1920, 1080, seed=42)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert a["Ks"][0, 0, 0].item() == pytest.approx(0.9 * 1920)
    z = a["means"][:, 2]
    assert z.min() >= 2 and z.max() <= 12 and a["sh"].shape == (1000, 16, 3)
    V = synthetic.view_matrix(3)[0]
    assert torch.allclose(V[:3, :3] @ V[:3, :3].T, torch.eye(3), atol=1e-6)
    assert torch.allclose(V[:3, :3] @ torch.tensor([0, 0, 7.0]) + V[:3, 3], torch.tensor([0, 0, 7.0]), atol=1e-5)


def test_shard_views_partitions_exactly():
    for n, w in ((8, 8), (8, 3), (5, 2), (1, 4), (0, 2)):
        got = [parallel.shard_views(n, r, w) for r in range(w)]
        assert sorted(sum(got, [])) == list(range(n))
        assert max(len(g) for g in got) - min(len(g) for g in got) <= 1


def test_fused_adam_refuses_cpu_tensors():
    """No CPU fallback: the fused optimiser fails loudly on CPU parameters."""
    import torch
    from collab_splats_amd import FusedAdam, MisplatError
    p = torch.zeros(4, requires_grad=True)
    p.grad = torch.ones(4)
    opt = FusedAdam([p], lr=1e-3)
    with pytest.raises(MisplatError):
        opt.step()
    with pytest.raises(ValueError):
        FusedAdam([p], lr=-1.0)


def test_get_loss_dict_value_vs_reference_goldens():
    """a5 (rade_gs_model.py:289-307) through the product function on CPU: the reference's own error maps in, the
    reference's own loss value out; the term is absent before ``regularization_from_iter``."""
    import numpy as np
    import os
    from collab_splats_amd import radegs
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "camera_goldens.npz"))
    one = torch.zeros(1, 3)
    model = radegs.RadegsModel(radegs.RadegsModelConfig(), one, one, torch.ones(1, 4), torch.zeros(1), one,
                               torch.zeros(1, 15, 3))
    for i in range(3):
        err = torch.from_numpy(g[f"dn{i}_err"])
        outputs = {"rgb": torch.zeros(*err.shape[1:], 3), "depth_normal_error_map": err[0].unsqueeze(-1),
                   "middepth_normal_error_map": err[1].unsqueeze(-1)}
        model.step = 100
        assert model.get_loss_dict(outputs, None) == {}
        model.step = 15000
        loss = model.get_loss_dict(outputs, None)
        assert abs(float(loss["depth_normal_loss"]) - float(g[f"dn{i}_loss"])) < 1e-7
        model.config.use_depth_normal_loss = False
        assert model.get_loss_dict(outputs, None) == {}
        model.config.use_depth_normal_loss = True


def test_image_loss_on_cpu_tensors_fails_loudly_and_scale_reg_follows_splatfacto():
    """``main_loss`` (L1 + SSIM, Splatfacto's, [UNVERIFIED-UPSTREAM]) has no CPU path: CPU images with ``ssim_lambda`` > 0
    raise; with ``ssim_lambda`` = 0 the plain L1 mean is a host expression.  ``scale_reg`` (off by default): 0.1 * mean(
    max(max(s) / min(s), 10) - 10) of the activated scales on every tenth step, 0 otherwise."""
    import math
    from collab_splats_amd import radegs
    from collab_splats_amd._lib import MisplatError
    log_s = torch.log(torch.tensor([[1.0, 1.0, 1.0], [20.0, 1.0, 2.0], [0.5, 0.01, 0.1]]))
    model = radegs.RadegsModel(radegs.RadegsModelConfig(use_depth_normal_loss=False), torch.zeros(3, 3), log_s,
                               torch.ones(3, 4), torch.zeros(3), torch.zeros(3, 3), torch.zeros(3, 15, 3))
    rgb, gt = torch.rand(12, 14, 3), torch.rand(12, 14, 3)
    with pytest.raises(MisplatError, match="no CPU fallback"):
        model.get_loss_dict({"rgb": rgb}, {"image": gt})
    model.config.ssim_lambda = 0.0
    loss = model.get_loss_dict({"rgb": rgb}, {"image": gt})
    assert set(loss) == {"main_loss", "scale_reg"} and float(loss["scale_reg"]) == 0.0
    assert abs(float(loss["main_loss"]) - float((gt - rgb).abs().mean())) < 1e-7
    model.config.use_scale_regularization = True
    model.step = 20
    want = 0.1 * ((1.0 - 1.0) * 0 + (max(20.0, 10.0) - 10.0) + (max(50.0, 10.0) - 10.0)) / 3.0
    assert math.isclose(float(model.get_loss_dict({"rgb": rgb}, {"image": gt})["scale_reg"].detach()), want, rel_tol=1e-5)
    model.step = 21
    assert float(model.get_loss_dict({"rgb": rgb}, {"image": gt})["scale_reg"]) == 0.0


def _np_refine(P, grad2d_avg, noise, step, cfg):
    """Second, independent restatement (numpy fp64) of one 3DGS refinement step -- clone small high-gradient
    Gaussians, split large ones into two samples with scales / 1.6, prune transparent (and, after the first opacity
    reset, oversized) ones -- written from Kerbl et al. 2023 section 5, NOT from strategy.py.  [UNVERIFIED-UPSTREAM:
    gsplat's own controller is absent; this pins strategy.py against a second reading of the paper only.]"""
    means, scales, quats, opac = (np.asarray(P[k], np.float64) for k in ("means", "scales", "quats", "opacities"))
    extra = {k: np.asarray(v, np.float64) for k, v in P.items() if k not in ("means", "scales", "quats", "opacities")}
    s_lin = np.exp(scales)
    high = grad2d_avg > cfg["grow_grad2d"]
    small = s_lin.max(1) <= cfg["grow_scale3d"] * cfg["scene_scale"]
    clone, splt = high & small, high & ~small
    n = len(means)
    # (1) clones are appended unchanged
    idx_c = np.nonzero(clone)[0]
    order = list(range(n)) + list(idx_c)
    means, scales, quats, opac = means[order], scales[order], quats[order], opac[order]
    extra = {k: v[order] for k, v in extra.items()}
    splt = np.concatenate([splt, np.zeros(len(idx_c), bool)])
    # (2) every split parent is replaced by two samples  mu + R diag(s) z, scales / 1.6
    idx_s, idx_r = np.nonzero(splt)[0], np.nonzero(~splt)[0]
    q = quats[idx_s] / np.linalg.norm(quats[idx_s], axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                  2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    sl = np.exp(scales[idx_s])
    kids = [means[idx_s] + np.einsum("nij,nj->ni", R, sl * noise[b]) for b in range(2)]
    means = np.concatenate([means[idx_r]] + kids)
    scales = np.concatenate([scales[idx_r], np.log(sl / 1.6), np.log(sl / 1.6)])
    quats = np.concatenate([quats[idx_r], quats[idx_s], quats[idx_s]])
    opac = np.concatenate([opac[idx_r], opac[idx_s], opac[idx_s]])
    extra = {k: np.concatenate([v[idx_r], v[idx_s], v[idx_s]]) for k, v in extra.items()}
    # (3) prune
    dead = 1.0 / (1.0 + np.exp(-opac.reshape(len(opac), -1)[:, 0])) < cfg["prune_opa"]
    if step > cfg["reset_every"]:
        dead |= np.exp(scales).max(1) > cfg["prune_scale3d"] * cfg["scene_scale"]
    keep = ~dead
    out = dict(means=means[keep], scales=scales[keep], quats=quats[keep], opacities=opac[keep])
    out.update({k: v[keep] for k, v in extra.items()})
    return out, (int(clone.sum()), int(len(idx_s)), int(dead.sum()))


@pytest.mark.parametrize("step", [10, 3010])
def test_strategy_matches_an_independent_numpy_restatement_unverified_upstream(step):
    """[UNVERIFIED-UPSTREAM] ``DefaultStrategy.step_post_backward`` against ``_np_refine`` on a 12-Gaussian toy state:
    same survivors in the same order, same children, same counts.  gsplat's controller is absent, so this pins the
    product against a second independent reading of the published algorithm, not against upstream."""
    n = 12
    params, optimizers = _toy_training_state(n)
    cfg = dict(grow_grad2d=0.5, grow_scale3d=0.01, prune_opa=0.005, prune_scale3d=0.1, reset_every=3000, scene_scale=1.0)
    s = DefaultStrategy(refine_start_iter=0, refine_every=10, reset_every=3000, grow_grad2d=0.5, grow_scale3d=0.01,
                        prune_opa=0.005, seed=7)
    st = s.initialize_state(scene_scale=1.0)
    with torch.no_grad():
        params["scales"][4:8] = float(np.log(0.05))
        params["scales"][8] = float(np.log(0.2))              # oversized: pruned only after the first opacity reset
        params["opacities"][10:12] = -9.0
    grad = torch.zeros(1, n, 2)
    grad[0, [0, 1, 4, 5], 0] = 1.0
    m2d = torch.zeros(1, n, 2, requires_grad=True)
    m2d.grad = grad
    info = {"means2d": m2d, "radii": torch.ones(1, n, 2, dtype=torch.int32), "width": 2, "height": 2, "n_cameras": 1}
    before = {k: v.detach().double().numpy().copy() for k, v in params.items()}
    noise = torch.randn(2, 2, 3, generator=s._generator(step)).double().numpy()        # 2 split parents
    avg = np.hypot(grad[0, :, 0].numpy() * 2 / 2.0, 0.0)                                # one observation each
    want, counts = _np_refine(before, avg, noise, step, cfg)
    got_counts = s.step_post_backward(params, optimizers, st, step, info)
    assert tuple(got_counts) == counts
    for k in params:
        assert params[k].shape == want[k].shape, k
        assert np.abs(params[k].detach().double().numpy() - want[k]).max() < 1e-6, k


def test_empty_crop_returns_empty_outputs_without_touching_the_gpu():
    """rade_gs_model.py:96-105: in evaluation an empty crop box short-circuits to ``get_empty_outputs`` (no render)."""
    one = torch.zeros(3, 3)
    model = radegs.RadegsModel(radegs.RadegsModelConfig(), one, one, torch.ones(3, 4), torch.zeros(3), one,
                               torch.zeros(3, 15, 3))
    model.eval()

    class Box:
        def within(self, pts):
            return torch.zeros(pts.shape[0], 1, dtype=torch.bool)

    model.crop_box = Box()
    cam = radegs.PinholeCamera.make(torch.eye(4)[:3], 20.0, 20.0, 8, 6)
    out = model.get_outputs(cam)
    assert set(out) == {"rgb", "depth", "accumulation", "background"}
    assert out["rgb"].shape == (6, 8, 3) and out["depth"].shape == (6, 8, 1) and float(out["depth"].min()) == 10.0
    assert not out["accumulation"].any()


@pytest.mark.parametrize("i", [0, 1, 2])
def test_tsdf_frame_extrinsic_equals_the_reference_viewmat(i):
    """f4 hand-off (mesh.py:1591-1630): the extrinsic given to Open3D, inv(c2w @ diag(1,-1,-1,1)), is the same OpenGL ->
    OpenCV world-to-camera matrix the reference's own ``convert_to_colmap_camera`` produced for this pose -- so it is
    pinned by the reference-generated camera goldens; the intrinsics are the camera's own (not the fov-rebuilt ones)."""
    g = np.load(GOLD)
    W, H = [int(v) for v in g[f"cam{i}_WH"]]
    K = g[f"cam{i}_K"]
    cam = radegs.PinholeCamera.make(torch.from_numpy(g[f"cam{i}_c2w"]), K[0, 0], K[1, 1], W, H, cx=K[0, 2], cy=K[1, 2])
    ext, intr = radegs.tsdf_frame(cam)
    assert ext.shape == (4, 4) and ext.dtype == np.float64
    assert np.abs(ext - g[f"cam{i}_viewmat"]).max() < 1e-5
    assert intr == dict(width=W, height=H, fx=float(K[0, 0]), fy=float(K[1, 1]), cx=float(K[0, 2]), cy=float(K[1, 2]))


def test_capacity_quantisation_and_static_capacity_context():
    """Host logic of the speculative forward: capacities are rounded UP to 8 steps per octave (buffers, addresses and
    graph keys stay put while the count drifts), never below the request; ``static_capacity`` nests and restores."""
    from collab_splats_amd import ops
    for x in (1, 4095, 4096, 4097, 100_000, 6_404_069, 2 ** 30 + 5):
        q = ops._quantise_cap(x)
        assert q >= max(x, 4096)
        assert q <= max(x, 4096) * 1.13                                    # at most one step (1/8 octave) above
        assert ops._quantise_cap(q) == q                                   # a quantised value is a fixed point
    assert ops._quantise_cap(1_000_000) == ops._quantise_cap(1_010_000)    # a 1 % drift keeps the capacity
    assert ops._STATIC_CAP is None
    with ops.static_capacity(1000):
        assert ops._STATIC_CAP == 1000
        with ops.static_capacity(None):
            assert ops._STATIC_CAP is None
        assert ops._STATIC_CAP == 1000
    assert ops._STATIC_CAP is None


def test_on_demand_colour_switch_conditions(monkeypatch):
    """``_lazy_colour_ok``: only SH colours with 16 coefficients in training, 3 or 4 composited
    channels; "auto" needs a capacity hint of a dense scene (typical bucket >= MISPLAT_LAZY_SH_MIN_BUCKET)."""
    from collab_splats_amd import _lib, ops
    P = _lib.make_params(1000, 1, 1920, 1080)
    dev = torch.device("cuda", 0)                                          # (only used as a dictionary key here)
    key = ops._cap_key(P, dev)
    monkeypatch.setattr(ops, "LAZY_SH", "1")
    assert ops._lazy_colour_ok(P, dev, 3, 16, 3, True, 4)
    assert not ops._lazy_colour_ok(P, dev, -1, 16, 3, True, 4)             # not SH
    assert not ops._lazy_colour_ok(P, dev, 3, 9, 3, True, 4)               # not 16 coefficients
    assert not ops._lazy_colour_ok(P, dev, 3, 16, 3, False, 4)             # inference: the colour kernel
    assert not ops._lazy_colour_ok(P, dev, 3, 16, 3, True, 1)              # depth-only render
    monkeypatch.setattr(ops, "LAZY_SH", "0")
    assert not ops._lazy_colour_ok(P, dev, 3, 16, 3, True, 4)
    monkeypatch.setattr(ops, "LAZY_SH", "auto")
    monkeypatch.setitem(ops._CAP_HINT, key, 100 * 8160)                    # sparse: 100 entries per tile
    assert not ops._lazy_colour_ok(P, dev, 3, 16, 3, True, 4)
    monkeypatch.setitem(ops._CAP_HINT, key, 800 * 8160)                    # dense
    assert ops._lazy_colour_ok(P, dev, 3, 16, 3, True, 4)
    ops._CAP_HINT.pop(key, None)
    assert not ops._lazy_colour_ok(P, dev, 3, 16, 3, True, 4)              # first call of a shape: no hint yet


def test_camera_upload_on_cpu_is_a_plain_copy():
    """``_upload`` stages camera data through pinned memory for a GPU only; for a CPU device it is the identity copy."""
    vals = torch.arange(28, dtype=torch.float32)
    out = radegs._upload(vals, torch.device("cpu"))
    assert torch.equal(out, vals) and not radegs._PIN_RING


def test_flip_proof_rows_need_a_pixel_that_actually_differs():
    """tests/helpers.FlipProof: a gradient row out of tolerance is only explained by a pixel inside its screen box that
    DIFFERS between the two implementations (recorded by check_pixels / check_ids), not by any low-margin pixel."""
    import helpers
    margin = np.ones((20, 30))
    margin[5, 5] = margin[10, 10] = 1e-9                            # two pixels on a threshold
    proof = helpers.FlipProof(margin, np.array([[5.0, 5.0], [10.0, 10.0], [25.0, 15.0]]), np.array([[2, 2], [2, 2], [2, 2]]))
    with pytest.raises(AssertionError, match="before the gradient rows"):
        proof.check_rows([0], "g")                                  # images / index maps first
    differs = np.zeros((20, 30), bool)
    differs[5, 5] = True
    proof.check_pixels(differs, "render")                           # (5, 5) differs and sits on a threshold: accepted
    proof.check_rows([0], "g")                                      # Gaussian 0 covers it
    with pytest.raises(AssertionError, match="covers no pixel that differs"):
        proof.check_rows([1], "g")                                  # Gaussian 1 covers only a low-margin pixel where nothing flipped
    with pytest.raises(AssertionError, match="covers no pixel that differs"):
        proof.check_rows([2], "g")
    assert proof.rows_loose_only == 1
    off = np.zeros((20, 30), bool)
    off[1, 1] = True
    with pytest.raises(AssertionError, match="not a flip"):
        proof.check_pixels(off, "alpha")                            # a differing pixel far from every threshold
    ids = np.zeros((20, 30), np.int32)
    ids2 = ids.copy()
    ids2[10, 10] = 7
    proof.check_ids(ids2, ids, ids, ids)                            # unequal last_ids at a threshold pixel: now on record
    proof.check_rows([1], "g")


def test_graphed_step_result_check_finds_autograd_history():
    from collab_splats_amd.graphs import _tensors_with_history
    a = torch.ones(2, requires_grad=True)
    b = a * 2
    assert _tensors_with_history(None) == [] and _tensors_with_history([a, a.detach(), 3, "x"]) == []
    assert _tensors_with_history({"img": [b.detach(), b], "meta": {"k": (b,)}}) == ["result['img'][1]"]
    assert _tensors_with_history((a.detach(), {"loss": b.sum()})) == ["result[1]['loss']"]


def test_graphed_step_result_check_walks_plain_objects_and_knows_torchs_warning_text():
    """A returned dataclass / namespace that holds a tensor with history is found structurally; the secondary net -- the
    text of PyTorch's stream-mismatch warning -- still exists in the installed torch (it lives in libtorch_cpu)."""
    import dataclasses, glob, mmap, types
    from collab_splats_amd import graphs

    @dataclasses.dataclass
    class Holder:
        meta: object

    a = torch.ones(2, requires_grad=True)
    b = a * 2
    assert graphs._tensors_with_history(Holder({"means2d": b})) == ["result.meta['means2d']"]
    assert graphs._tensors_with_history(types.SimpleNamespace(x=[b.detach()], y=Holder(None))) == []
    libs = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so"))
    assert libs, "libtorch_cpu.so not found next to the torch package"
    with open(libs[0], "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        assert mm.find(graphs.STALE_GRAPH_WARNING.encode()) >= 0, "PyTorch changed the text of its AccumulateGrad stream warning"


def test_resolution_schedule_rescales_the_camera_and_restores_it_even_on_error(monkeypatch):
    """rade_gs_model.py:132-136, 223: training renders at 1 / 2^k of the camera's resolution, k = max(num_downscales - step //
    resolution_schedule, 0) [UNVERIFIED-UPSTREAM: Splatfacto's schedule]; the camera comes back exactly as it was, also when the
    call fails in between (the reference's straight-line undo does not)."""
    cfg = radegs.RadegsModelConfig(num_downscales=2, resolution_schedule=100)
    m = radegs.RadegsModel(cfg, torch.zeros(1, 3), torch.zeros(1, 3), torch.ones(1, 4), torch.zeros(1, 1), torch.zeros(1, 3),
                           torch.zeros(1, 15, 3))
    m.train()
    for step, fac in ((0, 4), (99, 4), (100, 2), (199, 2), (200, 1), (5000, 1)):
        m.step = step
        assert m._get_downscale_factor() == fac
    m.eval()
    m.step = 0
    assert m._get_downscale_factor() == 1                                  # evaluation renders at full resolution
    m.train()
    cam = radegs.PinholeCamera.make(torch.eye(4)[:3], 500.0, 480.0, 640, 360)
    seen = {}

    def fake_params(camera):
        seen.update(W=int(camera.width.item()), H=int(camera.height.item()), fx=camera.fx, cx=camera.cx)
        raise RuntimeError("stop here")

    monkeypatch.setattr(m, "_get_camera_parameters", fake_params)
    with pytest.raises(RuntimeError, match="stop here"):
        m.get_outputs(cam)
    assert seen == dict(W=160, H=90, fx=125.0, cx=80.0)                   # the call saw the camera at 1/4 resolution
    assert (int(cam.width.item()), int(cam.height.item()), cam.fx, cam.fy, cam.cx, cam.cy) == (640, 360, 500.0, 480.0, 320.0, 180.0)


def test_capacity_of_the_speculative_forward_stays_put(monkeypatch):
    """ops._choose_cap: a capacity is part of every phase-B argument block (the graph key) and places the per-tile lists inside
    the arena slot, so it is chosen generously once and kept while the count of the views moves by tens of percent; it is chosen
    again when the count comes within 10 % of it, and given back when the scene has shrunk to an eighth."""
    from collab_splats_amd import ops
    monkeypatch.setattr(ops, "_CAP_CHOSEN", {})
    key = ("cap-test",)
    first = ops._choose_cap(key, 4_000_000)
    assert first >= 8_000_000 and first == ops._quantise_cap(8_000_000)
    for hint in (3_800_000, 5_400_000, 6_404_069, 7_000_000):            # the bench's eight views: 3.8 - 6.4 M
        assert ops._choose_cap(key, hint) == first
    grown = ops._choose_cap(key, 7_700_000)                                # within 10 %: chosen again, twice the count
    assert grown > first and grown >= 15_000_000
    assert ops._choose_cap(key, 7_800_000) == grown
    assert ops._choose_cap(key, 2_000_000) == grown                        # a quarter of it: kept
    small = ops._choose_cap(key, 900_000)                                  # an eighth: given back
    assert small < grown and small >= 1_800_000
    assert ops._choose_cap(("other",), 2 ** 31) == 2 ** 31 - 1             # (int32 indexing: the check of the count itself is elsewhere)


def test_key_trace_report_names_the_fields_that_moved(monkeypatch):
    """ops.key_trace_report (MISPLAT_KEY_TRACE): which fields of the argument blocks -- the graph cache's keys -- differ between
    a call and the one ``period`` calls earlier."""
    import ctypes as C
    from collab_splats_amd import _lib, ops
    P = _lib.make_params(1000, 1, 64, 48)
    a0, a1 = _lib.RasterArgs(), _lib.RasterArgs()
    a0.means, a1.means = 0x1000, 0x1000
    a0.opacities, a1.opacities = 0x2000, 0x2400
    a0.cap_isects, a1.cap_isects = 4096, 8192
    b0 = _lib.RasterBwdArgs()
    monkeypatch.setattr(ops, "KEY_TRACE", [("fwd3", bytes(P) + bytes(a0)), ("bwd", bytes(P) + bytes(b0)),
                                           ("fwd3", bytes(P) + bytes(a1)), ("bwd", bytes(P) + bytes(b0)),
                                           ("fwd2", bytes(P) + bytes(a1))])
    rep = ops.key_trace_report(2)
    assert rep[0][0] == 2 and rep[0][1] == "fwd3" and set(rep[0][2]) == {"opacities", "cap_isects"}
    assert rep[1] == (4, "entry", "fwd3", "fwd2") and len(rep) == 2          # (call 3 equals call 1: not reported)


def test_row_capacity_of_the_sparse_reduce():
    from collab_splats_amd import parallel
    assert parallel._row_capacity(0) == 1024 and parallel._row_capacity(100_000) == 150_000


def test_meta_completes_its_lazy_keys_for_every_access_style(monkeypatch):
    """``meta["flatten_ids"]`` / ``["isect_ids"]`` are produced on demand (after a front-only forward only the heads
    of the buckets are sorted).  gsplat's contract is that the keys are THERE: ``get``, ``in``, ``keys`` / ``items`` /
    ``values``, iteration, ``len``, ``dict(meta)`` and ``copy()`` complete them just like indexing does."""
    from collab_splats_amd import ops, rendering
    calls = []
    monkeypatch.setattr(ops, "isect_ids", lambda bins: calls.append("isect") or "ISECT")
    monkeypatch.setattr(ops, "complete_bins", lambda bins: calls.append("flat") or "FLAT")

    def fresh():
        calls.clear()
        return rendering._Meta({"_bins": {"partial": object()}, "radii": 1, "width": 4})

    m = fresh()
    assert not m._has("flatten_ids") and "flatten_ids" in m and "isect_ids" in m and "nope" not in m and calls == []
    assert len(m) == 5 and calls == []                            # (the two keys count before they exist)
    assert m.get("flatten_ids") == "FLAT" and calls == ["flat"] and m.get("nope", 7) == 7
    assert m["isect_ids"] == "ISECT" and calls == ["flat", "isect"] and len(m) == 5
    for access in (lambda m: list(m.keys()), lambda m: [k for k, _ in m.items()], lambda m: list(m), lambda m: list(dict(m)),
                   lambda m: list(m.copy().keys()), lambda m: list({**m})):
        m = fresh()
        assert {"flatten_ids", "isect_ids", "radii", "width", "_bins"} == set(access(m)) and sorted(calls) == ["flat", "isect"]
    m = fresh()
    assert "FLAT" in list(m.values())
    plain = rendering._Meta({"radii": 1})                         # (no bins: nothing to complete, an ordinary dict)
    assert "flatten_ids" not in plain and plain.get("flatten_ids") is None and list(plain) == ["radii"] and len(plain) == 1
    with pytest.raises(KeyError):
        plain["flatten_ids"]


def test_every_environment_switch_reaches_its_module_variable():
    """INTEGRATION.md section 5 documents twelve MISPLAT_* switches (round 4 had forty); each is read once, at import, into a
    module variable -- checked here in a child process with every one of them set to a non-default value.  Whatever else
    was a variable in round 4 is a module constant now: nothing in the package reads another MISPLAT_* name."""
    import json
    import re
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MISPLAT_DETERMINISTIC="1", MISPLAT_GRAPH="0", MISPLAT_GRAPH_ENTRIES="17", MISPLAT_LAZY_SH="0",
               MISPLAT_FRONT_ONLY="1", MISPLAT_ARENA="0", MISPLAT_ARENA_TRIM_MB="3", MISPLAT_SPARSE_REDUCE="0",
               MISPLAT_ALLREDUCE="rs_ag", MISPLAT_FORCE_COLLECTIVES="1", MISPLAT_LIB="/nonexistent/libmisplat.so")
    code = ("import json, sys; sys.path.insert(0, %r); from collab_splats_amd import ops, arena, parallel, _lib\n"
            "try:\n    _lib.load(); lib = 'loaded'\nexcept _lib.MisplatError as e:\n    lib = str(e)\n"
            "print(json.dumps(dict(det=ops.DETERMINISTIC_BACKWARD, graph=ops.GRAPHS, entries=ops.GRAPH_CACHE_ENTRIES, lazy=ops.LAZY_SH,"
            " front=ops.FRONT_ONLY, arena=arena.ENABLED, trim=arena.TRIM_BYTES, sparse=parallel.SPARSE, allreduce=parallel.ALLREDUCE,"
            " force=parallel.FORCE_COLLECTIVES, lib=lib)))") % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["det"] is True and got["graph"] is False and got["entries"] == 17 and got["lazy"] == "0" and got["front"] == "1"
    assert got["arena"] is False and got["trim"] == 3 << 20 and got["sparse"] == "0" and got["allreduce"] == "rs_ag" and got["force"] is True
    assert "/nonexistent/libmisplat.so" in got["lib"] and "no CPU fallback" in got["lib"]
    documented = {"MISPLAT_DETERMINISTIC", "MISPLAT_GRAPH", "MISPLAT_GRAPH_ENTRIES", "MISPLAT_LAZY_SH", "MISPLAT_FRONT_ONLY", "MISPLAT_ARENA",
                  "MISPLAT_ARENA_TRIM_MB", "MISPLAT_SPARSE_REDUCE", "MISPLAT_ALLREDUCE", "MISPLAT_DIST_BACKEND", "MISPLAT_FORCE_COLLECTIVES",
                  "MISPLAT_LIB"}
    read = set()
    pkg = os.path.join(ROOT, "collab_splats_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            read |= set(re.findall(r"environ(?:\.get)?\(\s*\"(MISPLAT_[A-Z_]+)\"", open(os.path.join(pkg, f)).read()))
    assert read == documented, (read - documented, documented - read)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert all(f"`{name}`" in text for name in documented)
