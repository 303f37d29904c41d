"""Generates tests/golden/camera_goldens.npz by RUNNING THE REFERENCE's own code
(/root/reference/collab_splats/utils/camera_utils.py) in the build container.

Container-only tooling: the reference cannot travel, so only the resulting
input/output vectors are committed.  The file is loaded by path (the package __init__ needs gsplat /
nerfstudio, which are absent), with a stub for the type name ``nerfstudio.cameras.cameras.Cameras``
(annotation only: camera_utils.py:21, 74, 176, 192) and a capture-only ``Tensor.cuda`` no-op
(hard .cuda() calls at camera_utils.py:57, 64, 86, 224, 237; this container has no GPU).

    python tests/golden/make_camera_goldens.py
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/collab_splats/utils/camera_utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "camera_goldens.npz")


def load_reference():
    for name in ("nerfstudio", "nerfstudio.cameras", "nerfstudio.cameras.cameras"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nerfstudio.cameras.cameras"].Cameras = type("Cameras", (), {})
    torch.Tensor.cuda = lambda self, *a, **k: self
    spec = importlib.util.spec_from_file_location("ref_camera_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class DuckCamera:
    def __init__(self, c2w, K, W, H):
        self.camera_to_worlds = c2w[None]
        self._K = K[None]
        self.width = torch.tensor([[W]])
        self.height = torch.tensor([[H]])

    def get_intrinsics_matrices(self):
        return self._K


def rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).float()


def main():
    ref = load_reference()
    g = torch.Generator().manual_seed(0)
    out = {}
    poses = [(rot(0, 0, 0), [0.1, -0.2, 0.3], 40.0, 40.0, 16, 12),
             (rot(0.3, -0.5, 0.2), [1.0, 2.0, -0.5], 100.0, 80.0, 121, 67),
             (rot(-1.1, 0.7, 2.0), [-3.0, 0.25, 4.0], 300.5, 310.25, 64, 48)]
    for i, (R, t, fx, fy, W, H) in enumerate(poses):
        c2w = torch.cat([R, torch.tensor(t)[:, None]], dim=1)
        K = torch.tensor([[fx, 0, W / 2 + 1.5], [0, fy, H / 2 - 0.75], [0, 0, 1]])   # off-centre pp: discarded
        cam = DuckCamera(c2w, K, W, H)
        cc = ref.convert_to_colmap_camera(cam)
        out[f"cam{i}_c2w"], out[f"cam{i}_K"] = c2w.numpy(), K.numpy()
        out[f"cam{i}_WH"] = np.array([W, H])
        out[f"cam{i}_viewmat"] = cc.world_view_transform.transpose(0, 1).numpy()
        out[f"cam{i}_center"] = cc.camera_center.numpy()
        out[f"cam{i}_fov"] = np.array([cc.fovx, cc.fovy])
        # depth -> normal -> error map -> loss, forward and autograd backward
        yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
        base = 3.0 + 0.02 * xx - 0.013 * yy                                   # tilted plane
        d1 = (base + 0.05 * torch.rand(H, W, generator=g)).reshape(1, H, W, 1).requires_grad_(True)
        d2 = (base + 0.3 * torch.sin(xx / 5.0) * torch.cos(yy / 7.0)).reshape(1, H, W, 1).requires_grad_(True)
        nrm = torch.nn.functional.normalize(torch.randn(1, H, W, 3, generator=g), dim=-1) * \
            torch.rand(1, H, W, 1, generator=g)
        nrm.requires_grad_(True)
        n2 = ref.depth_double_to_normal(cam, d1, d2)                           # [2,H,W,3]
        err = 1 - (nrm.unsqueeze(0) * n2).sum(dim=-1).squeeze(0)               # rade_gs_model.py:212-214
        loss = 0.05 * ((1 - 0.6) * err[0, ...].unsqueeze(-1).mean() + 0.6 * err[1, ...].unsqueeze(-1).mean())
        loss.backward()
        out[f"dn{i}_d1"], out[f"dn{i}_d2"] = d1.detach().numpy()[0, ..., 0], d2.detach().numpy()[0, ..., 0]
        out[f"dn{i}_nrm"] = nrm.detach().numpy()[0]
        out[f"dn{i}_normals2"], out[f"dn{i}_err"] = n2.detach().numpy(), err.detach().numpy()
        out[f"dn{i}_loss"] = np.array(loss.item())
        out[f"dn{i}_v_d1"], out[f"dn{i}_v_d2"] = d1.grad.numpy()[0, ..., 0], d2.grad.numpy()[0, ..., 0]
        out[f"dn{i}_v_nrm"] = nrm.grad.numpy()[0]
    # analytic depth maps: fronto-parallel plane -> (0,0,-1)
    cam = DuckCamera(torch.cat([rot(0, 0, 0), torch.zeros(3, 1)], 1),
                     torch.tensor([[50., 0, 10], [0, 50., 8], [0, 0, 1]]), 20, 16)
    flat = torch.full((1, 16, 20, 1), 2.5)
    out["plane_normals2"] = ref.depth_double_to_normal(cam, flat, flat * 2).numpy()
    q = torch.randn(32, 4, generator=g)
    q[0] = torch.tensor([1e-4, 0.0, 0.0, 0.0])
    out["rot_q"], out["rot_R"] = q.numpy(), ref.build_rotation(q).numpy()
    out["proj_fov"] = np.array([0.9, 0.7])
    out["proj_matrix"] = ref.get_projection_matrix(znear=0.01, zfar=1e3, fovx=0.9, fovy=0.7).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays")


if __name__ == "__main__":
    main()
