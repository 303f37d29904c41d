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


def test_strategy_surface():
    s = DefaultStrategy(absgrad=True)
    st = s.initialize_state()
    x = torch.zeros(1, 4, 2, requires_grad=True)
    m2d = x * 2.0
    info = {"means2d": m2d, "radii": torch.ones(1, 4, 2, dtype=torch.int32), "width": 8, "height": 4, "n_cameras": 1}
    s.step_pre_backward({}, {}, st, 0, info)
    m2d.sum().backward()
    assert m2d.grad is not None                                       # retained on the non-leaf
    m2d.absgrad = torch.ones(1, 4, 2)
    s.step_post_backward({}, {}, st, 0, info)
    assert torch.allclose(st["grad2d"], torch.full((4,), float(np.hypot(4.0, 2.0))))
    with pytest.raises(AssertionError):
        s.step_pre_backward({}, {}, st, 0, {})


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
