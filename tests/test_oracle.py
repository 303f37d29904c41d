"""CPU: the two rasterizer oracles against each other, against analytic known answers and
against finite differences (the rasterizer has no reference fixtures: parity unpinned,
SURVEY.md section 8(c)); and the camera / depth->normal oracle against vectors produced by the
reference's own code (tests/golden/camera_goldens.npz)."""
import os

import numpy as np
import pytest
import torch

from helpers import rel_err, small_scene, upstream
from oracle import camera_oracle as co
from oracle import torch_oracle as O
from oracle.craster import CRaster

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("mode,rm,deg", [("antialiased", "RGB+ED", 3), ("classic", "RGB", 1)])
def test_c_port_matches_autograd_oracle_fp64(mode, rm, deg):
    sc = small_scene()
    ins = [sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")]
    r, a, ed, md, n, meta = O.rasterization(*ins, sc["viewmat"][None], sc["K"][None], sc["W"], sc["H"],
                                            sh_degree=deg, render_mode=rm, rasterize_mode=mode)
    ups = upstream([t.shape for t in (r, a, ed, md, n)])
    torch.autograd.backward([r, a, ed, md, n], ups)
    cr = CRaster(np.float64)
    st = cr.forward(*[sc[k].numpy() for k in ("means", "quats", "scales", "opacities", "sh")],
                    sc["viewmat"].numpy(), sc["K"].numpy(), sc["W"], sc["H"], sh_degree=deg, render_mode=rm,
                    rasterize_mode=mode)
    # integer stages: exact
    assert np.array_equal(st["proj"]["radii"], meta["radii"][0].numpy())
    assert np.array_equal(st["bins"]["isect_ids"], meta["isect_ids"])
    assert np.array_equal(st["bins"]["flatten_ids"], meta["flatten_ids"])
    assert np.array_equal(st["bins"]["isect_offsets"], meta["isect_offsets"][0])
    # last / median contributor: the C port indexes the tile-sorted list, the dense oracle the
    # depth-sorted visible list -- compare the Gaussian ids they point at
    order = meta["order"] if "order" in meta else None
    for key in ("last_ids", "median_ids"):
        c_idx = st["fwd"][key]
        c_gid = np.where(c_idx >= 0, st["bins"]["flatten_ids"][np.maximum(c_idx, 0)], -1)
        t_idx = meta[key][0].numpy()
        t_gid = np.where(t_idx >= 0, meta["order_ids"][0][np.maximum(t_idx, 0)], -1)
        assert np.array_equal(c_gid, t_gid), key
    assert st["bins"]["n_isects"] > 300
    for got, ref in ((st["render"], r), (st["fwd"]["alpha"], a), (st["fwd"]["exp_depth"], ed),
                     (st["fwd"]["med_depth"], md), (st["fwd"]["normal"], n)):
        assert rel_err(got, ref[0]) < 1e-12
    gr = cr.backward(st, *[u[0].numpy() for u in ups])
    for name, t in zip(("v_means", "v_quats", "v_scales", "v_opacities", "v_colors"), ins):
        assert rel_err(gr[name], t.grad) < 1e-11, name


def test_c_port_fp32_within_tolerance_of_fp64_oracle():
    sc = small_scene(seed=3)
    ins = [sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")]
    outs = O.rasterization(*ins, sc["viewmat"][None], sc["K"][None], sc["W"], sc["H"], sh_degree=3,
                           render_mode="RGB+ED", rasterize_mode="antialiased")
    ups = upstream([t.shape for t in outs[:5]])
    torch.autograd.backward(list(outs[:5]), ups)
    cr = CRaster(np.float32)
    st = cr.forward(*[sc[k].numpy() for k in ("means", "quats", "scales", "opacities", "sh")],
                    sc["viewmat"].numpy(), sc["K"].numpy(), sc["W"], sc["H"], sh_degree=3, render_mode="RGB+ED",
                    rasterize_mode="antialiased")
    gr = cr.backward(st, *[u[0].numpy() for u in ups])
    assert rel_err(st["render"], outs[0][0]) < 1e-4          # fp32 tolerance of north_star
    for name, t in zip(("v_means", "v_quats", "v_scales", "v_opacities", "v_colors"), ins):
        assert rel_err(gr[name], t.grad) < 1e-4, name


def test_autograd_oracle_matches_finite_differences():
    sc = small_scene(n=40, W=24, H=20, seed=11)
    ins = [sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")]
    g = torch.Generator().manual_seed(2)

    def loss_of(vals):
        r, a, ed, md, n, _ = O.rasterization(*vals, sc["viewmat"][None], sc["K"][None], sc["W"], sc["H"],
                                             sh_degree=2, render_mode="RGB+ED", rasterize_mode="antialiased")
        return (r * wr).sum() + (a * wa).sum() + (ed * wd).sum() + (n * wn).sum()   # no median: piecewise const select

    r, a, ed, md, n, _ = O.rasterization(*ins, sc["viewmat"][None], sc["K"][None], sc["W"], sc["H"], sh_degree=2,
                                         render_mode="RGB+ED", rasterize_mode="antialiased")
    wr, wa, wd, wn = [torch.rand(t.shape, generator=g, dtype=torch.float64) for t in (r, a, ed, n)]
    loss_of(ins).backward()
    dirs = [torch.randn(t.shape, generator=g, dtype=torch.float64) for t in ins]
    analytic = sum((t.grad * d).sum() for t, d in zip(ins, dirs)).item()
    eps = 1e-7
    with torch.no_grad():
        lp = loss_of([t + eps * d for t, d in zip(ins, dirs)]).item()
        lm = loss_of([t - eps * d for t, d in zip(ins, dirs)]).item()
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - analytic) / max(abs(analytic), 1e-12) < 1e-5, (fd, analytic)


# ---------------------------------------------------------------- analytic known answers
def _single(mean, scale, quat, opacity, color, W=17, H=17, f=20.0, cx=8.5, cy=8.5, mode="classic", dtype=torch.float64):
    t = lambda x: torch.tensor(x, dtype=dtype)
    K = t([[f, 0, cx], [0, f, cy], [0, 0, 1]])[None]
    return O.rasterization(t([mean]), t([quat]), t([scale]), t([opacity]), t([color]), torch.eye(4, dtype=dtype)[None],
                           K, W, H, render_mode="RGB+ED", rasterize_mode=mode)


def test_kat_single_isotropic_gaussian_on_axis():
    z0, s, o, f = 4.0, 0.2, 0.8, 20.0
    col = [0.2, 0.5, 0.9]
    r, a, ed, md, n, meta = _single([0, 0, z0], [s, s, s], [1, 0, 0, 0], o, col)
    c = (8, 8)                                           # pixel centre (8.5, 8.5) == projected mean
    assert a[0, c[0], c[1], 0].item() == pytest.approx(o, rel=1e-12)          # sigma = 0, classic
    assert r[0, c[0], c[1], :3].tolist() == pytest.approx([o * x for x in col], rel=1e-12)
    assert ed[0, c[0], c[1], 0].item() == pytest.approx(o * z0, rel=1e-12)     # raw sum w*z
    assert md[0, c[0], c[1], 0].item() == pytest.approx(z0, rel=1e-12)
    assert r[0, c[0], c[1], 3].item() == pytest.approx(z0, rel=1e-12)          # ED channel, normalised
    assert n[0, c[0], c[1]].tolist() == pytest.approx([0, 0, -o], abs=1e-12)   # faces the camera
    var = (f * s / z0) ** 2 + 0.3                        # cov2d = (f s / z)^2 I + eps2d I
    d = 2.0                                              # two pixels right
    assert a[0, 8, 10, 0].item() == pytest.approx(o * np.exp(-0.5 * d * d / var), rel=1e-10)
    # antialiased: opacity scaled by sqrt(det0/det)
    r2, a2, *_ = _single([0, 0, z0], [s, s, s], [1, 0, 0, 0], o, col, mode="antialiased")
    comp = np.sqrt(((f * s / z0) ** 2) ** 2 / var ** 2)
    assert a2[0, 8, 8, 0].item() == pytest.approx(o * comp, rel=1e-10)


def test_kat_fronto_parallel_disc_depth_plane():
    """RaDe-GS linearises the RAY DISTANCE t in pixel offsets: for a
    fronto-parallel disc t_lin(p) is the first-order Taylor expansion of z0*l(p), so the z-depth
    z = t_lin / l(p) equals z0 up to O(offset^2) and has the closed form checked here."""
    z0, f = 3.0, 20.0
    mean = [0.1, -0.05, z0]
    r, a, ed, md, n, _ = _single(mean, [0.4, 0.4, 1e-3], [1, 0, 0, 0], 0.9, [1, 1, 1], f=f)
    vis = a[0, ..., 0] > 0
    assert vis.sum() > 50
    ys, xs = torch.meshgrid(torch.arange(17.0, dtype=torch.float64), torch.arange(17.0, dtype=torch.float64), indexing="ij")
    u, v = mean[0] / z0, mean[1] / z0
    pu, pv = (xs + 0.5 - 8.5) / f, (ys + 0.5 - 8.5) / f
    ell_mu, ell_p = np.sqrt(u * u + v * v + 1), torch.sqrt(pu * pu + pv * pv + 1)
    t_lin = z0 * (ell_mu + (u * (pu - u) + v * (pv - v)) / ell_mu)
    expect = t_lin / ell_p
    assert torch.allclose(md[0, ..., 0][vis], expect[vis], rtol=0, atol=2e-5)   # disc is 1e-3 thick, not a plane
    assert torch.allclose((ed / a.clamp_min(1e-10))[0, ..., 0][vis], expect[vis], atol=2e-5)
    off2 = ((pu - u) ** 2 + (pv - v) ** 2)[vis]
    assert ((md[0, ..., 0][vis] - z0).abs() <= 0.6 * z0 * off2 + 1e-4).all()      # second-order small
    nn = n[0][vis] / a[0][vis]
    assert torch.allclose(nn, torch.tensor([0.0, 0.0, -1.0], dtype=nn.dtype).expand_as(nn), atol=1e-5)


def test_kat_tilted_disc_depth_is_first_order_exact():
    z0, th = 3.0, 0.5
    q = [np.cos(th / 2), 0, np.sin(th / 2), 0]          # rotate the disc about y
    r, a, ed, md, n, _ = _single([0, 0, z0], [0.5, 0.5, 1e-3], q, 0.9, [1, 1, 1], f=30.0)
    nrm = np.array([np.sin(th), 0, np.cos(th)])          # disc normal (R e_z)
    for (py, px) in ((8, 8), (8, 9), (8, 7), (9, 8)):
        h = np.array([(px + 0.5 - 8.5) / 30.0, (py + 0.5 - 8.5) / 30.0, 1.0])
        z_exact = z0 * nrm[2] / (nrm @ h)                # ray / plane intersection, z-depth
        err = abs(md[0, py, px, 0].item() - z_exact)
        off = np.hypot(px - 8, py - 8) / 30.0
        assert err <= 3.0 * z0 * off ** 2 + 1e-9, (py, px, err)
    got = (n[0, 8, 8] / a[0, 8, 8]).numpy()
    assert np.allclose(got, -nrm, atol=1e-6) or np.allclose(got, nrm * -1, atol=1e-6)
    assert got[2] < 0


def test_kat_two_stacked_gaussians_weights_and_median_switch():
    t = lambda x: torch.tensor(x, dtype=torch.float64)
    K = t([[20.0, 0, 8.5], [0, 20.0, 8.5], [0, 0, 1]])[None]
    for o1, expect_med in ((0.7, 2.0), (0.3, 5.0)):
        means = t([[0, 0, 2.0], [0, 0, 5.0]])
        out = O.rasterization(means, t([[1, 0, 0, 0]] * 2), t([[0.2] * 3, [0.5] * 3]), t([o1, 0.6]),
                              t([[1.0, 0, 0], [0, 1.0, 0]]), torch.eye(4, dtype=torch.float64)[None], K, 17, 17)
        r, a, ed, md, n, _ = out
        w1, w2 = o1, 0.6 * (1 - o1)
        assert r[0, 8, 8].tolist() == pytest.approx([w1, w2, 0.0], rel=1e-12)
        assert a[0, 8, 8, 0].item() == pytest.approx(1 - (1 - o1) * (1 - 0.6), rel=1e-12)
        assert ed[0, 8, 8, 0].item() == pytest.approx(w1 * 2.0 + w2 * 5.0, rel=1e-12)
        assert md[0, 8, 8, 0].item() == pytest.approx(expect_med, rel=1e-12)   # T crosses 0.5 at the 1st / 2nd


def test_kat_tile_rects_and_culls():
    m2 = np.array([[16.0, 16.0], [8.0, 8.0], [8.0, 8.0], [40.0, 8.0], [5.0, 5.0]], dtype=np.float32)
    rad = np.array([[1, 1], [8, 8], [9, 9], [3, 30], [0, 0]], dtype=np.int32)
    b = O.bin_and_sort(m2, rad, np.arange(1, 6, dtype=np.float32), width=48, height=32)
    assert b["tiles_per_gauss"].tolist() == [4, 1, 4, 2, 0]      # straddle corner; exact fit; spill; clamp to 2 rows
    assert b["n_isects"] == 11 and b["isect_offsets"].shape == (2, 3)
    assert np.all(np.diff(b["isect_ids"].astype(np.int64)) >= 0)
    # culls: behind the camera, beyond far, off-screen, tiny opacity
    t = lambda x: torch.tensor(x, dtype=torch.float64)
    means = t([[0, 0, -1.0], [0, 0, 0.005], [50.0, 0, 2.0], [0, 0, 2.0], [0, 0, 3.0]])
    pr = O.project(means, t([[1, 0, 0, 0]] * 5), t([[0.1] * 3] * 5), torch.eye(4, dtype=torch.float64),
                   t([[20.0, 0, 8], [0, 20.0, 8], [0, 0, 1]]), 16, 16, opacities=t([0.9, 0.9, 0.9, 0.001, 0.9]))
    assert pr["radii"].sum(-1).tolist()[:4] == [0, 0, 0, 0] and pr["radii"][4].min() > 0


def test_c_port_empty_and_ragged_inputs():
    cr = CRaster(np.float32)
    z = lambda *s: np.zeros(s, np.float32)
    st = cr.forward(z(0, 3), z(0, 4), z(0, 3), z(0), z(0, 16, 3), np.eye(4, dtype=np.float32),
                    np.array([[20, 0, 8.5], [0, 20, 5.5], [0, 0, 1]], np.float32), 17, 11, sh_degree=3)
    assert st["bins"]["n_isects"] == 0 and st["fwd"]["alpha"].shape == (11, 17, 1) and not st["fwd"]["alpha"].any()
    assert (st["fwd"]["last_ids"] == -1).all()


# ---------------------------------------------------------------- camera / depth->normal goldens (reference-generated)
@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "camera_goldens.npz"))


@pytest.mark.parametrize("i", [0, 1, 2])
def test_camera_oracle_matches_reference_vectors(gold, i):
    W, H = [int(v) for v in gold[f"cam{i}_WH"]]
    V, Ks, c, fov = co.camera_params(torch.from_numpy(gold[f"cam{i}_c2w"]), torch.from_numpy(gold[f"cam{i}_K"]), W, H)
    assert np.abs(V.numpy() - gold[f"cam{i}_viewmat"]).max() < 1e-6
    assert np.abs(c.numpy() - gold[f"cam{i}_center"]).max() < 1e-6
    assert np.abs(np.array(fov) - gold[f"cam{i}_fov"]).max() < 1e-6
    assert float(Ks[0, 2]) == W / 2 and float(Ks[1, 2]) == H / 2           # principal point discarded
    d1 = torch.from_numpy(gold[f"dn{i}_d1"]).requires_grad_(True)
    d2 = torch.from_numpy(gold[f"dn{i}_d2"]).requires_grad_(True)
    nr = torch.from_numpy(gold[f"dn{i}_nrm"]).requires_grad_(True)
    n2 = co.depth_double_to_normal(d1, d2, float(Ks[0, 0]), float(Ks[1, 1]))
    err = co.normal_error_map(nr, n2)
    loss = co.depth_normal_loss(err)
    loss.backward()
    assert np.abs(n2.detach().numpy() - gold[f"dn{i}_normals2"]).max() < 2e-5
    assert np.abs(err.detach().numpy() - gold[f"dn{i}_err"]).max() < 2e-5
    assert abs(loss.item() - float(gold[f"dn{i}_loss"])) < 1e-6
    assert rel_err(d1.grad, gold[f"dn{i}_v_d1"]) < 1e-4
    assert rel_err(d2.grad, gold[f"dn{i}_v_d2"]) < 1e-4
    assert rel_err(nr.grad, gold[f"dn{i}_v_nrm"]) < 1e-5


def test_reference_conventions_in_goldens(gold):
    # fronto-parallel plane -> (0,0,-1) in the interior, 0 on the border (camera_utils.py:269-276)
    n2 = gold["plane_normals2"]
    assert np.allclose(n2[:, 1:-1, 1:-1], np.array([0, 0, -1.0]), atol=1e-6)
    assert not n2[:, 0].any() and not n2[:, :, 0].any() and not n2[:, -1].any() and not n2[:, :, -1].any()
    # identity-rotation pose: viewmat of SURVEY.md Appendix C
    assert np.allclose(gold["cam0_viewmat"], [[1, 0, 0, -0.1], [0, -1, 0, -0.2], [0, 0, -1, 0.3], [0, 0, 0, 1]], atol=1e-6)
    R = co.build_rotation(torch.from_numpy(gold["rot_q"]))
    assert np.abs(R.numpy() - gold["rot_R"]).max() < 1e-6
    assert np.abs(O.quat_to_rotmat(torch.from_numpy(gold["rot_q"])).numpy() - gold["rot_R"]).max() < 1e-6  # wxyz
